// scan.hpp -- device-wide prefix scans over u32 arrays (reduce-then-scan, 4096-item tiles).
#pragma once
#include "common.hpp"

namespace nolzss {

// out[i] = sum(in[0..i)).  out may alias in.  If d_total != nullptr the grand total is
// written there (device memory).  Temporaries come from the arena and are released on return.
void scan_exclusive_add_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total,
                            Arena &arena, hipStream_t stream);

// out[i] = max(in[0..i]).  out may alias in.
void scan_inclusive_max_u32(const uint32_t *in, uint32_t *out, size_t n, Arena &arena,
                            hipStream_t stream);

}  // namespace nolzss
