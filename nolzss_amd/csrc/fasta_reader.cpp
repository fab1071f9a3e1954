// fasta_reader.cpp -- the two FASTA parsers of the path, host only (no HIP): the C++ rules of the concatenated
// multi-sequence entry points and the Python rules of genomics.read_nucleotide_fasta.
#include "fasta_reader.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace nolzss {
namespace api {

inline bool is_canonical_dna(char c) {
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
}

// restates parse_fasta_sequences_and_ids, /root/reference/src/cpp/fasta_processor.cpp:28-128
// (same records, ids, warnings and errors).  The reference reads line by line and appends base by
// base (0.2 GB/s); the device side takes 3-7 Gbases/s, so the file is read in one piece and every
// line goes through a 256-entry table: upper-case base, white space to skip, or anything else.
FastaParse parse_fasta(const char *path, bool strict) {
    if (!path) throw std::invalid_argument("path is null");
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) throw std::runtime_error(std::string("Cannot open FASTA file: ") + path);
    std::shared_ptr<char> data;  // (no zero fill in front of the read; large files on huge pages)
    size_t data_size = 0;
    {
        file.seekg(0, std::ios::end);
        const std::streamoff len = file.tellg();
        file.seekg(0, std::ios::beg);
        if (len > 0) {
            char *raw = static_cast<char *>(alloc_factor_block((size_t)len));
            if (!raw) throw std::bad_alloc();
            data.reset(raw, [](char *q) { std::free(q); });
            file.read(raw, len);
            data_size = (size_t)file.gcount();
        } else {  // not seekable: take what comes
            const std::string all((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
            char *raw = static_cast<char *>(std::malloc(all.size() + 1));
            if (!raw) throw std::bad_alloc();
            data.reset(raw, [](char *q) { std::free(q); });
            std::memcpy(raw, all.data(), all.size());
            data_size = all.size();
        }
    }
    constexpr uint8_t kSpace = 0, kOther = 0xff;
    uint8_t kind[256];
    for (int c = 0; c < 256; ++c) kind[c] = kOther;
    for (unsigned char c : {' ', '\t', '\n', '\v', '\f', '\r'}) kind[c] = kSpace;  // std::isspace, "C" locale
    for (unsigned char c : {'A', 'C', 'G', 'T'}) kind[c] = kind[c - 'A' + 'a'] = c;
    uint8_t plain[256];  // 0 for an upper-case base: a line of those is copied as it is
    for (int c = 0; c < 256; ++c) plain[c] = 1;
    for (unsigned char c : {'A', 'C', 'G', 'T'}) plain[c] = 0;

    FastaParse res;
    res.buffers.push_back(data);
    std::string cur_id;
    // the bases of the current record are compacted to [rec, rec + cur_len): never beyond the read
    // position, since a byte of the file yields at most one base
    char *rec = data.get();
    size_t cur_len = 0;
    size_t empty_count = 0, removed = 0;
    auto finish = [&] {
        if (cur_id.empty()) return;  // (bases in front of the first header go on into the first record, as in the reference)
        if (cur_len) {
            res.sequences.push_back(SeqView{rec, cur_len});
            res.ids.push_back(cur_id);
        } else {
            fprintf(stderr, "Warning: Skipping empty sequence with ID: %s\n", cur_id.c_str());
            ++empty_count;
        }
        rec += cur_len;
        cur_len = 0;
    };
    const char *p = data.get(), *const end = p + data_size;
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        const char *line = p;
        size_t len = (size_t)((nl ? nl : end) - p);
        p = nl ? nl + 1 : end;
        while (len && kind[(unsigned char)line[len - 1]] == kSpace) --len;  // trailing white space
        if (!len) continue;
        if (line[0] == '>') {
            finish();
            size_t start = 1;
            while (start < len && kind[(unsigned char)line[start]] == kSpace) ++start;
            size_t stop = start;
            while (stop < len && kind[(unsigned char)line[stop]] != kSpace) ++stop;
            if (start >= len) throw std::runtime_error("Empty sequence header in FASTA file");
            cur_id.assign(line + start, stop - start);
        } else {
            char *out = rec + cur_len;  // <= line
            uint8_t mixed = 0;
            for (size_t i = 0; i < len; ++i) mixed |= plain[(unsigned char)line[i]];
            if (!mixed) {  // the usual line
                if (out != line) std::memmove(out, line, len);
                cur_len += len;
                continue;
            }
            size_t k = 0;
            for (size_t i = 0; i < len; ++i) {
                const uint8_t t = kind[(unsigned char)line[i]];
                if (t == kOther) {
                    if (strict)
                        throw std::runtime_error("Invalid nucleotide '" + std::string(1, line[i]) +
                                                 "' found in sequence with ID: " + cur_id);
                    ++removed;
                } else if (t != kSpace) {
                    out[k++] = (char)t;
                }
            }
            cur_len += k;
        }
    }
    finish();
    if (empty_count) fprintf(stderr, "Warning: Skipped %zu empty sequence(s) in FASTA file\n", empty_count);
    if (!strict && removed)
        fprintf(stderr, "Warning: Removed %zu ambiguous nucleotide(s) from FASTA input\n", removed);
    if (res.sequences.empty()) throw std::runtime_error("No valid sequences found in FASTA file");
    return res;
}

[[noreturn]] void fasta_error(const std::string &msg) { throw std::runtime_error(msg); }

// restates _parse_fasta_content and the nucleotide check of read_nucleotide_fasta,
// /root/reference/src/noLZSS/genomics/fasta.py:28-76 and :110-115, for files of ASCII bytes (a file with
// other bytes is handed back to the Python reader: false).  Python's rules, kept: lines end at \n, \r\n,
// \r, \v, \f, \x1c, \x1d, \x1e (universal newlines + str.splitlines); white space (str.strip, re \s) is
// \t \n \v \f \r \x1c-\x1f and the blank; the id is the first word of the header; bases are upper-cased;
// a repeated id keeps its first place in the order and takes the LAST record's bases (dict semantics).
namespace nucfasta {
enum : uint8_t { kBase = 0, kLower = 1, kSpace = 2, kBreak = 3, kOther = 4 };
struct Kinds {
    uint8_t kind[256];
    Kinds() {
        for (int c = 0; c < 256; ++c) kind[c] = kOther;
        for (unsigned char c : {'A', 'C', 'G', 'T'}) kind[c] = kBase;
        for (int c = 'a'; c <= 'z'; ++c) kind[c] = kLower;
        for (int c : {(int)'\t', (int)' ', 0x1f}) kind[c] = kSpace;
        for (int c : {(int)'\r', (int)'\v', (int)'\f', 0x1c, 0x1d, 0x1e}) kind[c] = kBreak;  // (\n is what the scan splits at)
    }
};
const Kinds &kinds() {
    static const Kinds k;
    return k;
}
// one piece of the file: starts at the beginning of the file or at a line that starts with '>'
struct Piece {
    const uint8_t *begin = nullptr, *end = nullptr;
    std::vector<std::string> ids;   // records in order of appearance (repeats included)
    std::vector<size_t> off, len;
    bool failed = false;
    std::string error;              // "... at line " is completed with the line number in the file
    size_t error_line = 0;          // line inside the piece
    size_t lines = 0;               // lines of the piece (when it was read to its end)
};

void parse_piece(uint8_t *base, Piece &P, bool file_end_after) {
    const uint8_t *kind = kinds().kind;
    auto is_space = [&](uint8_t c) { return kind[c] == kSpace || kind[c] == kBreak || c == '\n'; };
    bool have_id = false;
    std::string cur_id;
    uint8_t *rec = const_cast<uint8_t *>(P.begin);  // the bases of the current record go to [rec, rec + cur_len), never beyond the read position
    size_t cur_len = 0, line_num = 0;
    auto store = [&] {
        P.ids.push_back(cur_id);
        P.off.push_back((size_t)(rec - base));
        P.len.push_back(cur_len);
        rec += cur_len;
        cur_len = 0;
    };
    auto fail = [&](const char *what) {
        P.failed = true;
        P.error = what;
        P.error_line = line_num;
    };
    auto one_line = [&](const uint8_t *line, size_t len) {  // a line without any line break inside
        ++line_num;
        while (len && is_space(line[len - 1])) --len;
        while (len && is_space(line[0])) ++line, --len;
        if (!len) return;
        if (line[0] == '>') {
            if (have_id) store();
            size_t start = 1;
            while (start < len && is_space(line[start])) ++start;
            if (start >= len) return fail("Empty sequence header at line ");
            size_t stop = start;
            while (stop < len && !is_space(line[stop])) ++stop;
            cur_id.assign(reinterpret_cast<const char *>(line) + start, stop - start);
            have_id = true;
            return;
        }
        if (!have_id) return fail("Sequence data before header at line ");
        uint8_t *out = rec + cur_len;  // <= line
        uint8_t mixed = 0;
        for (size_t i = 0; i < len; ++i) mixed |= kind[line[i]];
        if (!mixed) {  // the usual line: upper-case bases only
            if (out != line) std::memmove(out, line, len);
            cur_len += len;
            return;
        }
        size_t k = 0;
        for (size_t i = 0; i < len; ++i) {
            const uint8_t c = line[i], t = kind[c];
            if (t == kSpace) continue;
            out[k++] = t == kLower ? (uint8_t)(c - 32) : c;  // (anything that is not a base fails the check later)
        }
        cur_len += k;
    };
    const uint8_t *p = P.begin, *const end = P.end;
    while (p < end && !P.failed) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(end - p)));
        const uint8_t *line = p;
        size_t len = (size_t)((nl ? nl : end) - p);
        p = nl ? nl + 1 : end;
        // (a piece that is not the last one ends right behind a \n: its last line has one)
        const bool terminated = nl != nullptr || !file_end_after;
        if (terminated && len && line[len - 1] == '\r') --len;  // \r\n is one line end
        uint8_t brk = 0;
        for (size_t i = 0; i < len; ++i) brk |= (uint8_t)(kind[line[i]] == kBreak);
        if (!brk) {
            one_line(line, len);
            continue;
        }
        size_t at = 0;  // (rare) other line ends inside: \r, \v, \f, \x1c-\x1e
        for (size_t i = 0; i <= len && !P.failed; ++i)
            if (i == len || kind[line[i]] == kBreak) {
                // (a break that is the last byte of the FILE ends the last line; in front of a \n it is followed by an empty line)
                if (i < len || i > at || terminated) one_line(line + at, i - at);
                at = i + 1;
            }
    }
    if (have_id && !P.failed) store();
    P.lines = line_num;
}
}  // namespace nucfasta

bool parse_nucleotide_fasta(const char *path, NucleotideFasta &res) {
    using namespace nucfasta;
    if (!path) throw std::invalid_argument("path is null");
    res.data = read_file(path);
    uint8_t *const base = const_cast<uint8_t *>(res.data.data());
    const size_t size = res.data.size();
    const uint8_t *kind = kinds().kind;

    // pieces for the host threads: cut in front of lines that start with '>' (such a line is a header
    // whatever came before it, so every piece can be read on its own)
    unsigned hw = std::thread::hardware_concurrency();
    const size_t max_threads = std::min<size_t>(hw ? hw : 1u, 16u);
    const size_t want = std::max<size_t>(1, std::min<size_t>(max_threads, size / (size_t(8) << 20)));
    std::vector<Piece> pieces;
    {
        size_t at = 0;
        for (size_t k = 1; k <= want && at < size; ++k) {
            size_t stop = size;
            if (k < want) {
                size_t from = std::max(at + 1, size / want * k);
                stop = size;
                while (from < size) {
                    const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(base + from, '\n', size - from));
                    if (!nl || (size_t)(nl - base) + 1 >= size) break;
                    if (nl[1] == '>') {
                        stop = (size_t)(nl - base) + 1;
                        break;
                    }
                    from = (size_t)(nl - base) + 1;
                }
            }
            Piece P;
            P.begin = base + at;
            P.end = base + stop;
            pieces.push_back(std::move(P));
            at = stop;
        }
        if (pieces.empty()) pieces.emplace_back();
    }
    std::atomic<bool> non_ascii{false};
    auto work = [&](size_t k) {
        Piece &P = pieces[k];
        uint8_t any = 0;
        for (const uint8_t *q = P.begin; q < P.end; ++q) any |= *q;
        if (any & 0x80) {
            non_ascii.store(true);
            return;
        }
        parse_piece(base, P, k + 1 == pieces.size());
    };
    if (pieces.size() == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < pieces.size(); ++k) pool.emplace_back(work, k);
        for (auto &t : pool) t.join();
    }
    if (non_ascii.load()) return false;
    {
        size_t lines_before = 0;  // (the pieces in front were read to their ends; their bytes have been compacted since)
        for (size_t k = 0; k < pieces.size(); ++k) {  // the first error in file order, with its line number in the file
            if (pieces[k].failed) fasta_error(pieces[k].error + std::to_string(lines_before + pieces[k].error_line));
            lines_before += pieces[k].lines;
        }
    }
    std::map<std::string, size_t> index;  // id -> position in res.ids (a repeated id keeps its place, takes the last record)
    for (const Piece &P : pieces)
        for (size_t j = 0; j < P.ids.size(); ++j) {
            auto it = index.find(P.ids[j]);
            if (it == index.end()) {
                index.emplace(P.ids[j], res.ids.size());
                res.ids.push_back(P.ids[j]);
                res.off.push_back(P.off[j]);
                res.len.push_back(P.len[j]);
            } else {
                res.off[it->second] = P.off[j];
                res.len[it->second] = P.len[j];
            }
        }
    if (res.ids.empty()) fasta_error("No valid sequences found in FASTA file");
    // ^[ACGT]+$ (fasta.py:112), records in order; the scans run on the host threads
    std::vector<uint8_t> bad(res.ids.size(), 0);
    {
        std::atomic<size_t> next{0};
        auto check = [&] {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= res.ids.size()) break;
                const uint8_t *q = base + res.off[j];
                uint8_t b = 0;
                for (size_t i = 0; i < res.len[j]; ++i) b |= kind[q[i]];
                bad[j] = (b || !res.len[j]) ? 1 : 0;
            }
        };
        const size_t threads = std::min<size_t>(max_threads, res.ids.size());
        if (threads <= 1 || size < (size_t(8) << 20)) {
            check();
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 0; t < threads; ++t) pool.emplace_back(check);
            for (auto &t : pool) t.join();
        }
    }
    for (size_t j = 0; j < res.ids.size(); ++j) {
        if (!bad[j]) continue;
        const uint8_t *q = base + res.off[j];
        bool present[128] = {false};
        for (size_t i = 0; i < res.len[j]; ++i) present[q[i] & 127] = true;
        std::string set;
        for (int c = 0; c < 128; ++c)
            if (present[c] && kind[c] != kBase) {
                if (!set.empty()) set += ", ";
                set += "'";
                set += (char)c;
                set += "'";
            }
        fasta_error("Sequence '" + res.ids[j] + "' contains invalid nucleotides: " + (set.empty() ? "set()" : "{" + set + "}"));
    }
    return true;
}

// Longest-processing-time-first bin packing, the plan every rank of a sharded job computes for itself:
// records by (length descending, index), bins by (load, index).
std::vector<size_t> lpt_owner(const std::vector<size_t> &lens, size_t bins) {
    std::vector<size_t> order(lens.size()), owner(lens.size(), 0), load(bins, 0);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
    for (size_t j : order) {
        const size_t b = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        owner[j] = b;
        load[b] += lens[j];
    }
    return owner;
}

}  // namespace api
}  // namespace nolzss
