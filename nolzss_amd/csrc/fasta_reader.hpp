// fasta_reader.hpp -- the two host-side FASTA parsers of the path (fasta_reader.cpp, no HIP).
#pragma once
#include "host_util.hpp"

#include <memory>
#include <string>
#include <vector>

namespace nolzss {
namespace api {

// The records of a FASTA file: views into the buffer the file was read into (the bases are compacted
// in place, in front of the read position; nothing is copied or allocated per record).
struct SeqView {
    const char *ptr = nullptr;
    size_t len = 0;
    const char *data() const { return ptr; }
    size_t size() const { return len; }
};
struct FastaParse {
    std::vector<SeqView> sequences;
    std::vector<std::string> ids;
    std::vector<std::shared_ptr<char>> buffers;  // what the views point into
};

// restates parse_fasta_sequences_and_ids, /root/reference/src/cpp/fasta_processor.cpp:28-128
FastaParse parse_fasta(const char *path, bool strict);

struct NucleotideFasta {
    FileBytes data;  // the records' bases are compacted in place in here
    std::vector<std::string> ids;
    std::vector<size_t> off, len;
};

// restates _parse_fasta_content and the nucleotide check of read_nucleotide_fasta (genomics/fasta.py:28-76, :110-115);
// false: the file holds non-ASCII bytes and is left to the Python reader
bool parse_nucleotide_fasta(const char *path, NucleotideFasta &res);

// longest-processing-time-first bin packing: the shard plan every rank computes for itself
std::vector<size_t> lpt_owner(const std::vector<size_t> &lens, size_t bins);

}  // namespace api
}  // namespace nolzss
