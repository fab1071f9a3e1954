// factor_file.cpp -- the v2 binary factor file: records, metadata, 48-byte footer (host only, no HIP).
// reference: FactorFileFooter, /root/reference/src/cpp/factorizer.hpp:64-77; writers factorizer.cpp:447-456, 621-629
#include "../../include/nolzss_hip.h"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>

namespace nolzss {
namespace api {

#pragma pack(push, 1)
struct FileFooter {  // FactorFileFooter, /root/reference/src/cpp/factorizer.hpp:64-77
    char magic[8];
    uint64_t num_factors, num_sequences, num_sentinels, footer_size, total_length;
};
#pragma pack(pop)
static_assert(sizeof(FileFooter) == 48, "v2 footer is 48 bytes");

// records, then `extra` metadata bytes, then the footer (footer_size counts extra + 48)
void write_v2_file(const char *out_path, const nolzss_factor *f, size_t z, uint64_t num_sequences,
                   uint64_t num_sentinels, uint64_t total_length, const std::string &extra) {
    if (!out_path) throw std::invalid_argument("output path is null");
    std::ofstream os(out_path, std::ios::binary);
    if (!os) throw std::runtime_error(std::string("Cannot create output file: ") + out_path);
    if (z) os.write(reinterpret_cast<const char *>(f), (std::streamsize)(sizeof(nolzss_factor) * z));
    if (!extra.empty()) os.write(extra.data(), (std::streamsize)extra.size());
    FileFooter ft;
    std::memcpy(ft.magic, "noLZSSv2", 8);
    ft.num_factors = z;
    ft.num_sequences = num_sequences;
    ft.num_sentinels = num_sentinels;
    ft.footer_size = sizeof(FileFooter) + extra.size();
    ft.total_length = total_length;
    os.write(reinterpret_cast<const char *>(&ft), sizeof ft);
    if (!os) throw std::runtime_error(std::string("Error writing output file: ") + out_path);
}

}  // namespace api
}  // namespace nolzss
