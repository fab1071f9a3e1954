// rc.hip -- reverse-complement DNA mode (detail::nolzss_multiple_dna_w_rc).
#include "pipeline.hpp"

namespace nolzss {

uint32_t run_rc_pipeline(Context &, const uint8_t *, size_t, size_t, void **) {
    throw std::runtime_error("reverse-complement mode: device pipeline not built yet");
}

}  // namespace nolzss
