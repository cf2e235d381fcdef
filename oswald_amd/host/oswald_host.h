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

#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
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
// Writes <out>.desc, <out>.info, <out>.seq (the reference's three files, byte for byte) and, next to them,
// <out>.g16: the group cache (see GroupCacheHeader) that lets `-O search` skip the interleave.
PreprocessStats preprocess_db(const std::string &input_filename, const std::string &out_filename, int n_procs);

// <db>.g16 -- the database already in the layout assemble_multiple_chunks_db() builds in memory: groups of
// 16 length-sorted sequences interleaved column by column and padded with the dummy residue to the group
// length (reference host/src/sequences.c:457-498).  The layout does not depend on the chunk plan (a chunk is a
// run of whole groups), so a search maps the file and hands slices of it to the device, whatever -k / -f say.
// <db>.seq stays the canonical database; the cache is used only if it belongs to it: count, D, group lengths and the
// CRC-32 of the length table must match, and then either <db>.seq still has the size AND modification time it had when
// the cache was written (the cheap test, no residue is read), or -- the file was touched or rewritten -- the CRC-32C of
// ALL its residues matches.  Anything else (including a .seq that cannot be read) is a mismatch: the cache is ignored
// with a warning and the groups are interleaved from <db>.seq.
struct GroupCacheHeader {
    char magic[8];          // "OSWG16\0\0"
    uint32_t version;       // 2
    uint32_t vector_length; // 16
    uint64_t sequences_count, D, groups, vD;
    uint64_t seq_file_bytes;
    uint32_t lengths_crc32; // of the uint16 length table of <db>.seq
    uint32_t residues_crc32c;// CRC-32C of the whole residue area of <db>.seq
    int64_t seq_mtime_ns;   // modification time of <db>.seq when the cache was written
    uint64_t reserved;
};                          // followed by uint16 n[groups], zero padding to a multiple of 64 B, then uint8 b[vD]
static_assert(sizeof(GroupCacheHeader) == 80, "on-disk header");
// Writes <db>.g16 from <db>.info / <db>.seq (also usable on a database preprocessed by the reference).
void write_group_cache(const std::string &sequences_filename);

struct Queries {
    std::vector<uint8_t> a;           // all queries back to back, preprocessed codes
    std::vector<uint16_t> m;          // lengths, ascending
    std::vector<uint32_t> a_disp;     // nq + 1 offsets
    std::vector<std::string> titles;  // with the leading '>'
    uint64_t Q = 0;
};
Queries load_query_sequences(const std::string &queries_filename);

struct MappedFile;                // read-only mapping of <db>.g16, shared by the chunks that point into it

struct Chunk {
    const uint8_t *b = nullptr;   // interleaved groups, b[disp[g] + j*W + lane]: points into `owned` or into the mapped cache
    uint64_t b_size = 0;
    std::vector<uint8_t> owned;
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
    std::shared_ptr<MappedFile> cache; // set when the chunks point into <db>.g16
};
Database assemble_multiple_chunks_db(const std::string &sequences_filename, int vector_length, uint64_t max_buffer_size,
                                     unsigned num_devices);
// The memory the chunks' residues live in, as (start, bytes) ranges a caller may page-lock for DMA: the mapped group cache
// as ONE range (the chunks are slices of it), else every chunk's own buffer.  The reference allocates these buffers
// 64-byte aligned "for DMA" (posix_memalign(AOCL_ALIGNMENT, ...), host/src/sequences.c:470-476).
std::vector<std::pair<const void *, size_t>> residue_ranges(const Database &db);

// Host compute path (`-m 2`, and the host share of `-m 1`; host_search.cpp): exact scores of every query against the
// groups [g0, g1) of a chunk, written to scores[qi * row_stride + col0 + (g - g0) * 16 + lane].  A mode of its own,
// never a fallback of the accelerator path.  Stands where the reference has its host SIMD kernels
// (host/src/HybridSearch.c:1540-1880, :790-1140).
// cpu_vector_length: the command line's -v -- 16 selects the SSE4.1 kernel (the reference's default host path), 32 the AVX2 one.
void host_search_groups(const Queries &q, const Chunk &c, uint64_t g0, uint64_t g1, int vector_length, const int8_t *submat, int open_gap,
                        int extend_gap, int threads, int32_t *scores, uint64_t row_stride, uint64_t col0, int cpu_vector_length = 32,
                        const std::atomic<bool> *cancel = nullptr, std::atomic<uint64_t> *cells_done = nullptr, int block_width = 256);
// block_width: the command line's -b -- the 8-bit stage works through a query in blocks of that many rows (0: unblocked).
// cancel: when it becomes true the groups not yet started are skipped and a group in progress is left at its next query
// (scores not computed stay untouched); cells_done accumulates query length x n[g] x 16 of every (group, query) finished
// (what the hybrid mode's calibration measures the host's speed on).

std::vector<std::string> load_database_headers(const std::string &sequences_filename, uint64_t sequences_count);
// The titles of the given sequences only (any order, duplicates allowed): one pass over the mapped file instead of
// one std::string per database sequence.  Lines the file does not have come back empty, like above.
std::vector<std::string> load_database_headers_at(const std::string &sequences_filename, const std::vector<uint64_t> &indices);

// Hardware threads this process may really keep busy: those of its affinity mask, and no more than its cgroup's CPU
// bandwidth allows (cgroup v2 cpu.max / v1 cpu.cfs_quota_us: a container of 16 CPUs' worth on a 256-thread host shows 256
// threads, and a team of 128 is throttled for the rest of every 100-ms period once it has burnt the quota -- round 5 found
// the hybrid mode's 10-ms accelerator test taking 90 ms that way).  cgroup_root: where the cgroup file
// system is mounted (tests point it elsewhere).  The reference takes -c as given (arguments.h:27).
unsigned usable_cpus(const char *cgroup_root = "/sys/fs/cgroup");

// The r best entries of scores[0..n) in the order the reference's sort_scores
// leaves them: descending score, ties by descending index.
void top_scores(const int32_t *scores, uint64_t n, uint64_t r, std::vector<int32_t> &out_scores, std::vector<uint64_t> &out_index);

// 24 x 32 int8 table, or nullptr; name as on the command line ("blosum62", ...).
const int8_t *submat_by_name(const std::string &name);

}  // namespace oswald

#endif
