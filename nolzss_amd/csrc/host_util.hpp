// host_util.hpp -- host-only helpers of the C ABI layer (no HIP): files in host memory, host threads, byte vectors
// that are not zero-filled.  Shared by c_abi.hip, batch.hip, fasta_api.hip and the host-only fasta_reader.cpp.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "divup.hpp"

namespace nolzss {
namespace api {

// host memory for results: huge pages for large blocks (batch.hip); released by free_block
void *alloc_factor_block(size_t bytes);
void free_block(void *p);

struct FileBytes {
    std::unique_ptr<uint8_t, void (*)(void *)> block{nullptr, &free_block};
    size_t bytes = 0;
    const uint8_t *data() const { return block.get(); }
    size_t size() const { return bytes; }
    bool empty() const { return bytes == 0; }
};

FileBytes read_file(const char *path);

// Bytes that are all written right after the allocation: no value-initialisation (resize() of a std::vector<uint8_t>
// zeroes -- and page-faults -- half a gigabyte on one thread for a reference + target pair of 2^27 bases each).
template <typename T> struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
using HostBytes = std::vector<uint8_t, DefaultInitAllocator<uint8_t>>;

// fn(lo, hi) over [0, n) in contiguous pieces on up to 16 host threads (one piece on the caller's thread for short
// inputs): validation, case folding and reverse complement of sequences as long as the device factorizes in tens of
// milliseconds were 250 ms on one core for that pair
template <typename Fn> void host_parallel(size_t n, Fn fn) {
    constexpr size_t kMinPiece = size_t(4) << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t pieces = std::min<size_t>({(size_t)(hw ? hw : 1u), (size_t)16, n / kMinPiece});
    if (pieces <= 1) {
        fn((size_t)0, n);
        return;
    }
    const size_t per = (n + pieces - 1) / pieces;
    std::vector<std::thread> pool;
    std::exception_ptr err;
    std::mutex mu;
    for (size_t k = 1; k < pieces; ++k)
        pool.emplace_back([&, k] {
            try {
                fn(std::min(n, k * per), std::min(n, (k + 1) * per));
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                err = std::current_exception();
            }
        });
    try {
        fn((size_t)0, std::min(n, per));
    } catch (...) {
        std::lock_guard<std::mutex> g(mu);
        err = std::current_exception();
    }
    for (auto &t : pool) t.join();
    if (err) std::rethrow_exception(err);
}

}  // namespace api
}  // namespace nolzss
