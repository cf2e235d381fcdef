// oswald_amd/host/oswald_host.cpp -- formats, loaders and chunk assembly of the
// search path (see oswald_host.h for the reference functions mirrored).
#include "oswald_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <stdexcept>
#include <nmmintrin.h>

#include <fcntl.h>
#include <sched.h>
#include <thread>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace oswald {

#include "submat_tables.inc"

const int8_t *submat_by_name(const std::string &name)
{
    if (name == "blosum45") return k_blosum45;
    if (name == "blosum50") return k_blosum50;
    if (name == "blosum62") return k_blosum62;
    if (name == "blosum80") return k_blosum80;
    if (name == "blosum90") return k_blosum90;
    if (name == "pam30") return k_pam30;
    if (name == "pam70") return k_pam70;
    if (name == "pam250") return k_pam250;
    return nullptr;
}

uint8_t encode_residue(uint8_t c)
{
    // the reference computes on (signed) char: J, O, U become 'Z'+1, then 'A' plus
    // the number of removed letters below the symbol is subtracted
    signed char s = (signed char)c;
    if (s == 'J' || s == 'O' || s == 'U') s = 'Z' + 1;
    signed char diff = 'A';
    if (s > 'J') diff++;
    if (s > 'O') diff++;
    if (s > 'U') diff++;
    return (uint8_t)(signed char)(s - diff);
}

std::vector<FastaRecord> read_fasta(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("OSWALD: An error occurred while opening input sequence file.");
    std::vector<FastaRecord> recs;
    std::string line;
    bool have = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '>') {
            recs.emplace_back();
            recs.back().title = line;
            have = true;
        } else {
            if (!have) {
                if (line.empty()) continue;
                throw std::runtime_error("OSWALD: input is not FASTA (first record has no '>' line).");
            }
            recs.back().residues += line;
            // 65520 = 28 * 2340: the group length (longest sequence rounded up to a multiple of 28,
            // sequences.c:457-463) is stored as uint16 and would wrap for 65521..65535 (it does in the reference)
            if (recs.back().residues.size() > kMaxSequenceLength)
                throw std::runtime_error("OSWALD: sequence longer than " + std::to_string(kMaxSequenceLength) + " residues: " + recs.back().title);
        }
    }
    return recs;
}

std::vector<size_t> length_order(const std::vector<FastaRecord> &recs)
{
    std::vector<size_t> order(recs.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return recs[x].residues.size() < recs[y].residues.size(); });
    return order;
}

PreprocessStats preprocess_db(const std::string &input_filename, const std::string &out_filename, int /*n_procs*/)
{
    const std::vector<FastaRecord> recs = read_fasta(input_filename);
    const std::vector<size_t> order = length_order(recs);
    PreprocessStats st;
    st.sequences = recs.size();
    // titles, one per line, in sorted order
    {
        std::ofstream f(out_filename + ".desc", std::ios::binary);
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence header file.");
        for (size_t i : order) {
            f << recs[i].title << '\n';
            // the reference counts the title line with its newline, plus one
            st.max_title_length = std::max<int>(st.max_title_length, (int)recs[i].title.size() + 2);
        }
    }
    std::vector<uint16_t> lengths(recs.size());
    for (size_t k = 0; k < order.size(); ++k) {
        lengths[k] = (uint16_t)recs[order[k]].residues.size();
        st.residues += lengths[k];
    }
    {
        FILE *f = fopen((out_filename + ".info").c_str(), "w");
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening info file.");
        fprintf(f, "%ld %ld %d", (long)st.sequences, (long)st.residues, st.max_title_length);
        fclose(f);
    }
    {
        std::ofstream f(out_filename + ".seq", std::ios::binary);
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence file.");
        f.write((const char *)lengths.data(), (std::streamsize)(lengths.size() * sizeof(uint16_t)));
        std::string buf;
        for (size_t i : order) {
            buf = recs[i].residues;
            for (char &c : buf) c = (char)encode_residue((uint8_t)c);
            f.write(buf.data(), (std::streamsize)buf.size());
        }
    }
    write_group_cache(out_filename);
    return st;
}

Queries load_query_sequences(const std::string &queries_filename)
{
    const std::vector<FastaRecord> recs = read_fasta(queries_filename);
    const std::vector<size_t> order = length_order(recs); // "Query no." follows this order
    Queries q;
    q.a_disp.push_back(0);
    for (size_t i : order) {
        q.m.push_back((uint16_t)recs[i].residues.size());
        q.titles.push_back(recs[i].title);
        for (char c : recs[i].residues) q.a.push_back(encode_residue((uint8_t)c));
        q.a_disp.push_back((uint32_t)q.a.size());
    }
    q.Q = q.a.size();
    return q;
}

struct MappedFile {
    const uint8_t *p = nullptr;
    size_t bytes = 0;
    ~MappedFile() { if (p) munmap((void *)p, bytes); }
    static std::shared_ptr<MappedFile> open(const std::string &path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); return nullptr; }
        // MAP_POPULATE: the pages are read in and mapped HERE, while the database is being loaded -- the reference reads its
        // database from disk before its clock starts (sequences.c:407-439).  Mapped lazily, the first upload of every
        // chunk would take the page faults inside the timed region (12 ms instead of 2.6 ms for a 128 MiB chunk).
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return nullptr;
        auto r = std::make_shared<MappedFile>();
        r->p = (const uint8_t *)m;
        r->bytes = (size_t)st.st_size;
        return r;
    }
};

namespace {

uint32_t crc32_of(const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

struct DbInfo { uint64_t count = 0, D = 0; int max_title_length = 0; };

DbInfo read_info(const std::string &sequences_filename)
{
    FILE *f = fopen((sequences_filename + ".info").c_str(), "r");
    if (!f) throw std::runtime_error("OSWALD: An error occurred while opening info file.");
    long a = 0, b = 0;
    int c = 0;
    if (fscanf(f, "%ld %ld %d", &a, &b, &c) != 3) { fclose(f); throw std::runtime_error("OSWALD: malformed info file."); }
    fclose(f);
    if (a < 0 || b < 0 || c < 0) throw std::runtime_error("OSWALD: malformed info file (negative count).");
    DbInfo r;
    r.count = (uint64_t)a;
    r.D = (uint64_t)b;
    r.max_title_length = c;
    return r;
}

int64_t mtime_ns_of(const struct stat &st) { return (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec; }

// The length table of <db>.seq, checked against <db>.info and the file BEFORE anything is sized from them: the files
// are input (possibly truncated, possibly not written by this tool).  <db>.seq must be exactly count x uint16 + D
// bytes, the lengths must add up to D, be sorted ascending (what preprocessing guarantees and the group layout relies
// on: a group is as long as its LAST sequence) and stay within the format's limit.
std::vector<uint16_t> read_length_table(const std::string &sequences_filename, const DbInfo &info, struct stat *seq_stat)
{
    const std::string path = sequences_filename + ".seq";
    struct stat st;
    if (stat(path.c_str(), &st) != 0) throw std::runtime_error("OSWALD: An error occurred while opening sequence file.");
    if (info.count > (1ull << 40) || info.D > (1ull << 46) || (uint64_t)st.st_size != info.count * sizeof(uint16_t) + info.D)
        throw std::runtime_error((uint64_t)st.st_size < info.count * sizeof(uint16_t) + info.D ? "OSWALD: sequence file is shorter than its info file says."
                                                                                               : "OSWALD: sequence file does not have the size its info file says.");
    std::vector<uint16_t> len(info.count);
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence file.");
    f.read((char *)len.data(), (std::streamsize)(len.size() * sizeof(uint16_t)));
    if (!f) throw std::runtime_error("OSWALD: sequence file is shorter than its info file says.");
    uint64_t sum = 0;
    for (size_t i = 0; i < len.size(); ++i) {
        sum += len[i];
        if (i > 0 && len[i] < len[i - 1]) throw std::runtime_error("OSWALD: sequence file is not sorted by length (not a preprocessed database?).");
    }
    if (sum != info.D) throw std::runtime_error("OSWALD: the sequence lengths do not add up to the residue count of the info file.");
    if (!len.empty() && len.back() > kMaxSequenceLength) // a .seq written by the reference may hold such a sequence; its padded length wraps there
        throw std::runtime_error("OSWALD: database holds a sequence of " + std::to_string(len.back()) + " residues (limit " + std::to_string(kMaxSequenceLength) + ")");
    if (seq_stat) *seq_stat = st;
    return len;
}

// padded group lengths: the longest (= last) sequence of the group, rounded up to a multiple of 28
std::vector<uint16_t> group_lengths(const std::vector<uint16_t> &len, int W)
{
    const uint64_t N = len.size(), G = (N + W - 1) / W;
    std::vector<uint16_t> n(G);
    for (uint64_t g = 0; g < G; ++g) {
        const uint64_t last = std::min(N, (g + 1) * W) - 1;
        const uint32_t l = len[last];
        if (l > kMaxSequenceLength) // a .seq written by the reference may hold such a sequence; its padded length wraps there
            throw std::runtime_error("OSWALD: database holds a sequence of " + std::to_string(l) + " residues (limit " + std::to_string(kMaxSequenceLength) + ")");
        n[g] = (uint16_t)((l + kFpgaBlockWidth - 1) / kFpgaBlockWidth * kFpgaBlockWidth);
    }
    return n;
}

// groups [g0, g1) interleaved into dst (pre-filled with the dummy residue): dst[gdisp[g] - gdisp[g0] + col*W + lane]
void interleave_groups(const std::vector<uint16_t> &len, const uint8_t *residues, const std::vector<uint64_t> &seq_off,
                       const std::vector<uint64_t> &gdisp, int W, uint64_t g0, uint64_t g1, uint8_t *dst)
{
    const uint64_t N = len.size();
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t g = g0; g < g1; ++g) {
        uint8_t *d = dst + (gdisp[g] - gdisp[g0]);
        for (int lane = 0; lane < W; ++lane) {
            const uint64_t sidx = g * W + lane;
            if (sidx >= N) break;
            const uint8_t *src = residues + seq_off[sidx];
            for (uint32_t col = 0; col < len[sidx]; ++col) d[(uint64_t)col * W + lane] = src[col];
        }
    }
}

// CRC-32C (Castagnoli) with the SSE4.2 instruction, three interleaved streams would be faster still; ~5 GB/s is enough
// for a check that runs only when <db>.seq was touched after its cache was written.
uint32_t crc32c_of(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    uint64_t c = crc ^ 0xFFFFFFFFu;
    while (n && ((uintptr_t)p & 7)) { c = _mm_crc32_u8((uint32_t)c, *p++); --n; }
    for (; n >= 8; n -= 8, p += 8) c = _mm_crc32_u64(c, *(const uint64_t *)p);
    while (n--) c = _mm_crc32_u8((uint32_t)c, *p++);
    return (uint32_t)c ^ 0xFFFFFFFFu;
}

// CRC-32C of the residue area of <db>.seq, streamed; false if the file cannot be read to the end
bool residues_crc32c_of_file(const std::string &seq_path, uint64_t count, uint64_t D, uint32_t *out)
{
    FILE *f = fopen(seq_path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> buf(4u << 20);
    bool ok = fseek(f, (long)(count * sizeof(uint16_t)), SEEK_SET) == 0;
    uint32_t crc = 0;
    for (uint64_t left = D; ok && left > 0;) {
        const size_t want = (size_t)std::min<uint64_t>(left, buf.size());
        ok = fread(buf.data(), 1, want, f) == want;
        if (ok) crc = crc32c_of(buf.data(), want, crc);
        left -= want;
    }
    fclose(f);
    *out = crc;
    return ok;
}

size_t cache_payload_offset(uint64_t groups) { return (sizeof(GroupCacheHeader) + groups * sizeof(uint16_t) + 63) / 64 * 64; }

} // namespace

void write_group_cache(const std::string &sequences_filename)
{
    const int W = kFpgaVectorLength;
    const DbInfo info = read_info(sequences_filename);
    struct stat seq_stat;
    const std::vector<uint16_t> len = read_length_table(sequences_filename, info, &seq_stat);
    std::vector<uint8_t> s(info.D);
    const uint64_t seq_bytes = len.size() * sizeof(uint16_t) + s.size();
    {
        std::ifstream f(sequences_filename + ".seq", std::ios::binary);
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence file.");
        f.seekg((std::streamoff)(len.size() * sizeof(uint16_t)));
        f.read((char *)s.data(), (std::streamsize)s.size());
        if (!f) throw std::runtime_error("OSWALD: sequence file is shorter than its info file says.");
    }
    const std::vector<uint16_t> n = group_lengths(len, W);
    const uint64_t N = info.count, G = n.size();
    std::vector<uint64_t> seq_off(N + 1, 0), gdisp(G + 1, 0);
    for (uint64_t i = 0; i < N; ++i) seq_off[i + 1] = seq_off[i] + len[i];
    for (uint64_t g = 0; g < G; ++g) gdisp[g + 1] = gdisp[g] + (uint64_t)n[g] * W;
    GroupCacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "OSWG16\0\0", 8);
    h.version = 2;
    h.vector_length = (uint32_t)W;
    h.sequences_count = N;
    h.D = info.D;
    h.groups = G;
    h.vD = gdisp[G];
    h.seq_file_bytes = seq_bytes;
    h.lengths_crc32 = crc32_of((const uint8_t *)len.data(), len.size() * sizeof(uint16_t));
    h.residues_crc32c = crc32c_of(s.data(), s.size());
    h.seq_mtime_ns = mtime_ns_of(seq_stat);
    const std::string tmp = sequences_filename + ".g16.tmp";
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening the group cache file.");
        f.write((const char *)&h, sizeof h);
        f.write((const char *)n.data(), (std::streamsize)(n.size() * sizeof(uint16_t)));
        const std::vector<char> zeros(64, 0);
        f.write(zeros.data(), (std::streamsize)(cache_payload_offset(G) - sizeof h - n.size() * sizeof(uint16_t)));
        // in slabs of groups, so that the padded copy never exists in memory as a whole
        const uint64_t slab = 4096;
        std::vector<uint8_t> buf;
        for (uint64_t g0 = 0; g0 < G; g0 += slab) {
            const uint64_t g1 = std::min(G, g0 + slab);
            buf.assign(gdisp[g1] - gdisp[g0], (uint8_t)kDummy);
            interleave_groups(len, s.data(), seq_off, gdisp, W, g0, g1, buf.data());
            f.write((const char *)buf.data(), (std::streamsize)buf.size());
        }
        if (!f) throw std::runtime_error("OSWALD: An error occurred while writing the group cache file.");
    }
    if (rename(tmp.c_str(), (sequences_filename + ".g16").c_str()) != 0)
        throw std::runtime_error("OSWALD: An error occurred while renaming the group cache file.");
}

Database assemble_multiple_chunks_db(const std::string &sequences_filename, int W, uint64_t max_buffer_size, unsigned num_devices)
{
    Database db;
    const DbInfo info = read_info(sequences_filename);
    db.sequences_count = info.count;
    db.D = info.D;
    db.max_title_length = info.max_title_length;
    struct stat seq_stat;
    const std::vector<uint16_t> len = read_length_table(sequences_filename, info, &seq_stat);
    if (db.sequences_count == 0) return db;
    db.sequences_db_max_length = len.back();
    const uint64_t N = db.sequences_count, G = (N + W - 1) / W;
    db.vect_sequences_count = G;
    const std::vector<uint16_t> n = group_lengths(len, W);
    std::vector<uint64_t> gdisp(G + 1, 0);
    for (uint64_t g = 0; g < G; ++g) gdisp[g + 1] = gdisp[g] + (uint64_t)n[g] * W;
    db.vD = gdisp[G];

    // the group cache, if it belongs to this database (OSWALD_NO_GROUP_CACHE=1: ignore it)
    const uint8_t *cached_b = nullptr;
    if (W == kFpgaVectorLength && !getenv("OSWALD_NO_GROUP_CACHE")) {
        std::shared_ptr<MappedFile> mf = MappedFile::open(sequences_filename + ".g16");
        if (mf && mf->bytes >= sizeof(GroupCacheHeader)) {
            GroupCacheHeader h;
            memcpy(&h, mf->p, sizeof h);
            bool ok = !memcmp(h.magic, "OSWG16\0\0", 8) && h.version == 2 && h.vector_length == (uint32_t)W && h.sequences_count == N &&
                      h.D == db.D && h.groups == G && h.vD == db.vD && (uint64_t)seq_stat.st_size == h.seq_file_bytes &&
                      mf->bytes == cache_payload_offset(G) + db.vD &&
                      h.lengths_crc32 == crc32_of((const uint8_t *)len.data(), len.size() * sizeof(uint16_t)) &&
                      !memcmp(mf->p + sizeof h, n.data(), n.size() * sizeof(uint16_t));
            if (ok && h.seq_mtime_ns != mtime_ns_of(seq_stat)) {
                // <db>.seq was touched after the cache was written (same size, same lengths): its residues decide
                uint32_t crc = 0;
                ok = residues_crc32c_of_file(sequences_filename + ".seq", N, db.D, &crc) && crc == h.residues_crc32c;
            }
            if (ok) {
                db.cache = mf;
                cached_b = mf->p + cache_payload_offset(G);
            } else {
                fprintf(stderr, "OSWALD: %s.g16 does not match the database; interleaving from %s.seq (re-run -O preprocess to refresh it).\n",
                        sequences_filename.c_str(), sequences_filename.c_str());
            }
        }
    }
    std::vector<uint8_t> s;
    std::vector<uint64_t> seq_off;
    if (!cached_b) {
        s.resize(db.D);
        std::ifstream fseq(sequences_filename + ".seq", std::ios::binary);
        if (!fseq) throw std::runtime_error("OSWALD: An error occurred while opening sequence file.");
        fseq.seekg((std::streamoff)(len.size() * sizeof(uint16_t)));
        fseq.read((char *)s.data(), (std::streamsize)s.size());
        if (!fseq) throw std::runtime_error("OSWALD: sequence file is shorter than its info file says.");
        seq_off.assign(N + 1, 0);
        for (uint64_t i = 0; i < N; ++i) seq_off[i + 1] = seq_off[i] + len[i];
    }
    // chunk plan: a single device fills chunks up to max_buffer_size; several devices aim at
    // ceil(vD/ndev) (divided again by ndev until it fits) and close a chunk just after passing it
    uint64_t buffer_size = max_buffer_size;
    if (num_devices > 1) {
        buffer_size = (uint64_t)std::ceil((double)db.vD / (double)num_devices);
        while (buffer_size > max_buffer_size) buffer_size = (uint64_t)std::ceil((double)buffer_size / (double)num_devices);
    }
    uint64_t i = 0;
    while (i < G) {
        uint64_t j = 0, chunk_size = 0, accum = (uint64_t)n[i] * W;
        const uint64_t start = i;
        while (i < G && chunk_size <= buffer_size && chunk_size + accum <= max_buffer_size) {
            chunk_size += accum;
            ++j;
            ++i;
            if (i < G) accum = (uint64_t)n[i] * W;
        }
        if (j == 0) throw std::runtime_error("OSWALD: max_chunk_size is smaller than one group of sequences.");
        Chunk c;
        c.accum = start;
        c.n.assign(n.begin() + start, n.begin() + start + j);
        c.nbb.resize(j);
        c.disp.resize(j);
        for (uint64_t k = 0; k < j; ++k) {
            c.nbb[k] = (uint16_t)((c.n[k] + kFpgaBlockWidth - 1) / kFpgaBlockWidth);
            c.disp[k] = (uint32_t)(gdisp[start + k] - gdisp[start]);
        }
        const uint64_t bytes = gdisp[start + j] - gdisp[start];
        c.b_size = bytes;
        if (cached_b) {
            c.b = cached_b + gdisp[start];
        } else {
            c.owned.assign(bytes, (uint8_t)kDummy);
            interleave_groups(len, s.data(), seq_off, gdisp, W, start, start + j, c.owned.data());
        }
        db.max_chunk_vD = std::max(db.max_chunk_vD, bytes);
        db.chunks.push_back(std::move(c));
    }
    for (Chunk &c : db.chunks) if (!cached_b) c.b = c.owned.data(); // (after the moves: the vectors' buffers are stable now)
    return db;
}

unsigned usable_cpus(const char *cgroup_root)
{
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
    if (n == 0) n = 1;
    const std::string root = cgroup_root ? cgroup_root : "/sys/fs/cgroup";
    // cgroup v2: "<quota> <period>" or "max <period>", in the process's own group and in every group above it (inside a container
    // the mount point IS the container's group); cgroup v1: two files
    std::vector<std::string> dirs = {root};
    {
        std::ifstream f("/proc/self/cgroup");
        std::string line;
        while (std::getline(f, line))
            if (line.rfind("0::", 0) == 0) {
                std::string path = line.substr(3);
                while (!path.empty() && path != "/") {
                    dirs.push_back(root + path);
                    const size_t k = path.find_last_of('/');
                    path = k == std::string::npos || k == 0 ? "" : path.substr(0, k);
                }
            }
    }
    for (const std::string &d : dirs) {
        std::ifstream f(d + "/cpu.max");
        std::string q;
        double period = 0;
        if (f >> q >> period && q != "max" && period > 0) {
            const double cpus = atof(q.c_str()) / period;
            if (cpus > 0) n = std::min<unsigned>(n, std::max(1u, (unsigned)cpus));
        }
    }
    {
        std::ifstream fq(root + "/cpu/cpu.cfs_quota_us"), fp(root + "/cpu/cpu.cfs_period_us");
        double quota = 0, period = 0;
        if (fq >> quota && fp >> period && quota > 0 && period > 0) n = std::min<unsigned>(n, std::max(1u, (unsigned)(quota / period)));
    }
    return n;
}

std::vector<std::pair<const void *, size_t>> residue_ranges(const Database &db)
{
    std::vector<std::pair<const void *, size_t>> r;
    if (db.cache) r.push_back({db.cache->p, db.cache->bytes});
    else for (const Chunk &c : db.chunks) if (c.b && c.b_size) r.push_back({c.b, (size_t)c.b_size});
    return r;
}

std::vector<std::string> load_database_headers(const std::string &sequences_filename, uint64_t sequences_count)
{
    std::ifstream f(sequences_filename + ".desc", std::ios::binary);
    if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence description file.");
    std::vector<std::string> h;
    h.reserve(sequences_count);
    std::string line;
    while (h.size() < sequences_count && std::getline(f, line)) h.push_back(line);
    h.resize(sequences_count);
    return h;
}

std::vector<std::string> load_database_headers_at(const std::string &sequences_filename, const std::vector<uint64_t> &indices)
{
    std::shared_ptr<MappedFile> mf = MappedFile::open(sequences_filename + ".desc");
    if (!mf) {
        // an empty description file maps to nothing; a missing one is an error like in load_database_headers
        std::ifstream f(sequences_filename + ".desc", std::ios::binary);
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence description file.");
        return std::vector<std::string>(indices.size());
    }
    std::vector<size_t> by_line(indices.size());
    std::iota(by_line.begin(), by_line.end(), 0);
    std::sort(by_line.begin(), by_line.end(), [&](size_t x, size_t y) { return indices[x] < indices[y]; });
    std::vector<std::string> out(indices.size());
    const char *p = (const char *)mf->p, *end = p + mf->bytes;
    uint64_t line = 0;
    for (size_t k : by_line) {
        while (line < indices[k] && p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            p = nl ? nl + 1 : end;
            ++line;
        }
        if (p >= end) break;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        out[k].assign(p, nl ? nl : end);
    }
    return out;
}

void top_scores(const int32_t *scores, uint64_t n, uint64_t r, std::vector<int32_t> &out_scores, std::vector<uint64_t> &out_index)
{
    r = std::min(r, n);
    std::vector<uint64_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    auto before = [&](uint64_t x, uint64_t y) { return scores[x] != scores[y] ? scores[x] > scores[y] : x > y; };
    std::partial_sort(idx.begin(), idx.begin() + r, idx.end(), before);
    out_scores.resize(r);
    out_index.resize(r);
    for (uint64_t k = 0; k < r; ++k) { out_index[k] = idx[k]; out_scores[k] = scores[idx[k]]; }
}

}  // namespace oswald

// ---------------------------------------------------------------------------
// C doors for the Python tests (ctypes): the same functions, flat.
// ---------------------------------------------------------------------------
extern "C" {

static thread_local std::string g_host_err;
const char *oswald_host_last_error(void) { return g_host_err.c_str(); }

int oswald_host_preprocess(const char *fasta, const char *out, int threads, uint64_t *stats3)
{
    try {
        const oswald::PreprocessStats st = oswald::preprocess_db(fasta, out, threads);
        if (stats3) { stats3[0] = st.sequences; stats3[1] = st.residues; stats3[2] = (uint64_t)st.max_title_length; }
        return 0;
    } catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}

uint8_t oswald_host_encode(uint8_t c) { return oswald::encode_residue(c); }

unsigned oswald_host_usable_cpus(const char *cgroup_root) { return oswald::usable_cpus(cgroup_root); }

const int8_t *oswald_host_submat(const char *name) { return oswald::submat_by_name(name); }

static oswald::Queries g_q;
int oswald_host_load_queries(const char *fasta, uint64_t *nq, uint64_t *Q)
{
    try { g_q = oswald::load_query_sequences(fasta); *nq = g_q.m.size(); *Q = g_q.Q; return 0; }
    catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}
const uint8_t *oswald_host_queries_a(void) { return g_q.a.data(); }
const uint16_t *oswald_host_queries_m(void) { return g_q.m.data(); }
const uint32_t *oswald_host_queries_disp(void) { return g_q.a_disp.data(); }
const char *oswald_host_queries_title(uint64_t i) { return g_q.titles[i].c_str(); }

static oswald::Database g_db;
int oswald_host_assemble(const char *dbname, int W, uint64_t max_chunk, unsigned ndev, uint64_t *out8)
{
    try {
        g_db = oswald::assemble_multiple_chunks_db(dbname, W, max_chunk, ndev);
        out8[0] = g_db.sequences_count; out8[1] = g_db.D; out8[2] = g_db.sequences_db_max_length; out8[3] = (uint64_t)g_db.max_title_length;
        out8[4] = g_db.vect_sequences_count; out8[5] = g_db.vD; out8[6] = g_db.max_chunk_vD; out8[7] = g_db.chunks.size();
        return 0;
    } catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}
uint64_t oswald_host_chunk_groups(unsigned c) { return g_db.chunks[c].n.size(); }
uint64_t oswald_host_chunk_accum(unsigned c) { return g_db.chunks[c].accum; }
uint64_t oswald_host_chunk_vD(unsigned c) { return g_db.chunks[c].b_size; }
const uint8_t *oswald_host_chunk_b(unsigned c) { return g_db.chunks[c].b; }
int oswald_host_db_from_cache(void) { return g_db.cache ? 1 : 0; }
int oswald_host_write_group_cache(const char *dbname)
{
    try { oswald::write_group_cache(dbname); return 0; }
    catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}
const uint16_t *oswald_host_chunk_n(unsigned c) { return g_db.chunks[c].n.data(); }
const uint16_t *oswald_host_chunk_nbb(unsigned c) { return g_db.chunks[c].nbb.data(); }
const uint32_t *oswald_host_chunk_disp(unsigned c) { return g_db.chunks[c].disp.data(); }

// the host compute path on chunk c of the assembled database, for the queries loaded last (all groups)
int oswald_host_search_chunk_v(unsigned c, const char *submat_name, int open_gap, int extend_gap, int threads, int cpu_vector_length, int32_t *scores)
{
    try {
        const int8_t *sm = oswald::submat_by_name(submat_name);
        if (!sm) throw std::runtime_error("unknown substitution matrix");
        const oswald::Chunk &ch = g_db.chunks.at(c);
        oswald::host_search_groups(g_q, ch, 0, ch.n.size(), oswald::kFpgaVectorLength, sm, open_gap, extend_gap, threads, scores,
                                   ch.n.size() * oswald::kFpgaVectorLength, 0, cpu_vector_length);
        return 0;
    } catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}

// ... with the command line's -b (query rows per block of the 8-bit stage; 0: unblocked)
int oswald_host_search_chunk_vb(unsigned c, const char *submat_name, int open_gap, int extend_gap, int threads, int cpu_vector_length, int block_width, int32_t *scores)
{
    try {
        const int8_t *sm = oswald::submat_by_name(submat_name);
        if (!sm) throw std::runtime_error("unknown substitution matrix");
        const oswald::Chunk &ch = g_db.chunks.at(c);
        oswald::host_search_groups(g_q, ch, 0, ch.n.size(), oswald::kFpgaVectorLength, sm, open_gap, extend_gap, threads, scores,
                                   ch.n.size() * oswald::kFpgaVectorLength, 0, cpu_vector_length, nullptr, nullptr, block_width);
        return 0;
    } catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}

int oswald_host_search_chunk(unsigned c, const char *submat_name, int open_gap, int extend_gap, int threads, int32_t *scores)
{
    try {
        const int8_t *sm = oswald::submat_by_name(submat_name);
        if (!sm) throw std::runtime_error("unknown substitution matrix");
        const oswald::Chunk &ch = g_db.chunks.at(c);
        oswald::host_search_groups(g_q, ch, 0, ch.n.size(), oswald::kFpgaVectorLength, sm, open_gap, extend_gap, threads, scores,
                                   ch.n.size() * oswald::kFpgaVectorLength, 0);
        return 0;
    } catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}

static std::vector<std::string> g_headers;
int oswald_host_load_headers(const char *dbname, uint64_t count)
{
    try { g_headers = oswald::load_database_headers(dbname, count); return 0; }
    catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}
const char *oswald_host_header(uint64_t i) { return g_headers[i].c_str(); }
int oswald_host_load_headers_at(const char *dbname, const uint64_t *indices, uint64_t count)
{
    try { g_headers = oswald::load_database_headers_at(dbname, std::vector<uint64_t>(indices, indices + count)); return 0; }
    catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}

void oswald_host_top_scores(const int32_t *scores, uint64_t n, uint64_t r, int32_t *out_scores, uint64_t *out_index)
{
    std::vector<int32_t> s;
    std::vector<uint64_t> ix;
    oswald::top_scores(scores, n, r, s, ix);
    for (size_t k = 0; k < s.size(); ++k) { out_scores[k] = s[k]; out_index[k] = ix[k]; }
}

} // extern "C"
