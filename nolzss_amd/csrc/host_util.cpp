// host_util.cpp -- files in host memory and their release (host only, no HIP).
#include "host_util.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <malloc.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace nolzss {
namespace api {

// A file in host memory.  Large files are read by several threads (pread of 16 MiB pieces: a single
// read() of a cached 1 GiB file takes twice as long as its factorization) into a block on transparent
// huge pages that nothing zero-fills first.
// Giving memory back costs too: free() of a 2 GiB block of 4 KiB pages spends 0.2 s in munmap (a third of the
// time from a 2.17 GB FASTA file to its 512 factor counts).  Blocks of 256 MiB and more are released by a
// detached thread; the caller does not wait for the page tables.
void free_block(void *p) {
    if (!p) return;
    if (malloc_usable_size(p) >= (size_t(256) << 20)) {
        try {
            std::thread([p] { std::free(p); }).detach();
            return;
        } catch (...) {  // no thread to be had: release it here
        }
    }
    std::free(p);
}

FileBytes read_file(const char *path) {
    if (!path) throw std::invalid_argument("path is null");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) throw std::runtime_error(std::string("Cannot open input file: ") + path);
    struct Close {
        int fd;
        ~Close() { ::close(fd); }
    } closer{fd};
    struct stat st;
    FileBytes out;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) {
        // not a regular file (or empty): take what comes
        std::vector<uint8_t> all;
        uint8_t buf[1 << 16];
        for (;;) {
            const ssize_t got = ::read(fd, buf, sizeof buf);
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) break;
            all.insert(all.end(), buf, buf + got);
        }
        if (!all.empty()) {
            out.block.reset(static_cast<uint8_t *>(std::malloc(all.size())));
            if (!out.block) throw std::bad_alloc();
            std::memcpy(out.block.get(), all.data(), all.size());
            out.bytes = all.size();
        }
        return out;
    }
    const size_t size = (size_t)st.st_size;
    out.block.reset(static_cast<uint8_t *>(alloc_factor_block(size)));
    if (!out.block) throw std::bad_alloc();
    constexpr size_t kPiece = size_t(16) << 20;
    const size_t pieces = div_up(size, kPiece);
    unsigned hw = std::thread::hardware_concurrency();
    const size_t threads = std::min<size_t>({pieces, hw ? hw : 1u, 8u});
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto worker = [&] {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= pieces || failed.load()) break;
            size_t at = k * kPiece;
            const size_t stop = std::min(size, at + kPiece);
            while (at < stop) {
                const ssize_t got = ::pread(fd, out.block.get() + at, stop - at, (off_t)at);
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) {  // shorter than fstat said, or an I/O error
                    failed.store(true);
                    return;
                }
                at += (size_t)got;
            }
        }
    };
    if (threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
    if (failed.load()) throw std::ios_base::failure(std::string("Cannot read input file: ") + path);
    out.bytes = size;
    return out;
}

}  // namespace api
}  // namespace nolzss
