// divup.hpp -- ceil(a / b) for sizes (shared by the device headers and the host-only files).
#pragma once
#include <cstddef>

namespace nolzss {
inline size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }
}  // namespace nolzss
