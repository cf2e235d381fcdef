// oswald_amd/host/oswald_host.h -- host side of the search path, above the C ABI.
//
// C++ mirror of the reference's host functions for this path (same names,
// argument meaning and on-disk formats), written from their behaviour:
//   preprocess_db ................ reference host/src/sequences.c:4-220
//   load_query_sequences ......... reference host/src/sequences.c:223-391
//   assemble_multiple_chunks_db .. reference host/src/sequences.c:393-623
//   load_database_headers ........ reference host/src/sequences.c:1096-1127
//   sort_scores (top-r part) ..... reference host/src/utils.c:3-86
//   substitution matrices ........ reference host/src/submat.c
// The device side is reached only through include/oswald_hip.h.
#ifndef OSWALD_HOST_H
#define OSWALD_HOST_H

#include <cstdint>
#include <string>
#include <vector>

namespace oswald {

constexpr int kDummy = 23;            // PREPROCESSED_DUMMY_ELEMENT, reference sequences.h:17
constexpr int kFpgaBlockWidth = 28;   // reference arguments.h:24 (group lengths are padded to it)
constexpr unsigned kMaxSequenceLength = 65520; // 28 * 2340: longest sequence whose padded group length still fits uint16
constexpr int kFpgaVectorLength = 16; // reference arguments.h:23
constexpr const char *kVersion = "1.0";

// 'A'..'Z' -> 0..22 with J, O, U -> 23; every other byte goes through the same
// arithmetic as the reference (sequences.c:165-175), so files are identical
// for identical input bytes.
uint8_t encode_residue(uint8_t c);

struct FastaRecord {
    std::string title;     // the '>' line without its newline
    std::string residues;  // raw letters, line breaks removed
};

// Throws std::runtime_error on I/O errors or records longer than 65535.
std::vector<FastaRecord> read_fasta(const std::string &path);

// Stable ascending order by length (ties keep file order).
std::vector<size_t> length_order(const std::vector<FastaRecord> &recs);

struct PreprocessStats { uint64_t sequences = 0, residues = 0; int max_title_length = 0; };
// Writes <out>.desc, <out>.info, <out>.seq.
PreprocessStats preprocess_db(const std::string &input_filename, const std::string &out_filename, int n_procs);

struct Queries {
    std::vector<uint8_t> a;           // all queries back to back, preprocessed codes
    std::vector<uint16_t> m;          // lengths, ascending
    std::vector<uint32_t> a_disp;     // nq + 1 offsets
    std::vector<std::string> titles;  // with the leading '>'
    uint64_t Q = 0;
};
Queries load_query_sequences(const std::string &queries_filename);

struct Chunk {
    std::vector<uint8_t> b;       // interleaved groups, b[disp[g] + j*W + lane]
    std::vector<uint16_t> n;      // padded group lengths
    std::vector<uint16_t> nbb;    // n / 28 (kept for interface parity, unused by the GPU path)
    std::vector<uint32_t> disp;   // byte offset of each group in b
    uint64_t accum = 0;           // groups before this chunk
};

struct Database {
    uint64_t sequences_count = 0, D = 0, vect_sequences_count = 0, vD = 0, max_chunk_vD = 0;
    uint16_t sequences_db_max_length = 0;
    int max_title_length = 0;
    std::vector<Chunk> chunks;
};
Database assemble_multiple_chunks_db(const std::string &sequences_filename, int vector_length, uint64_t max_buffer_size,
                                     unsigned num_devices);

std::vector<std::string> load_database_headers(const std::string &sequences_filename, uint64_t sequences_count);

// The r best entries of scores[0..n) in the order the reference's sort_scores
// leaves them: descending score, ties by descending index.
void top_scores(const int32_t *scores, uint64_t n, uint64_t r, std::vector<int32_t> &out_scores, std::vector<uint64_t> &out_index);

// 24 x 32 int8 table, or nullptr; name as on the command line ("blosum62", ...).
const int8_t *submat_by_name(const std::string &name);

}  // namespace oswald

#endif
