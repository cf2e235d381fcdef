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

Database assemble_multiple_chunks_db(const std::string &sequences_filename, int W, uint64_t max_buffer_size, unsigned num_devices)
{
    Database db;
    {
        FILE *f = fopen((sequences_filename + ".info").c_str(), "r");
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening info file.");
        long a = 0, b = 0;
        int c = 0;
        if (fscanf(f, "%ld %ld %d", &a, &b, &c) != 3) { fclose(f); throw std::runtime_error("OSWALD: malformed info file."); }
        fclose(f);
        db.sequences_count = (uint64_t)a;
        db.D = (uint64_t)b;
        db.max_title_length = c;
    }
    std::vector<uint16_t> len(db.sequences_count);
    std::vector<uint8_t> s(db.D);
    {
        std::ifstream f(sequences_filename + ".seq", std::ios::binary);
        if (!f) throw std::runtime_error("OSWALD: An error occurred while opening sequence file.");
        f.read((char *)len.data(), (std::streamsize)(len.size() * sizeof(uint16_t)));
        f.read((char *)s.data(), (std::streamsize)s.size());
        if (!f) throw std::runtime_error("OSWALD: sequence file is shorter than its info file says.");
    }
    if (db.sequences_count == 0) return db;
    db.sequences_db_max_length = len.back();
    const uint64_t N = db.sequences_count, G = (N + W - 1) / W;
    db.vect_sequences_count = G;
    std::vector<uint64_t> seq_off(N + 1, 0);
    for (uint64_t i = 0; i < N; ++i) seq_off[i + 1] = seq_off[i] + len[i];
    // group length = longest (= last) sequence of the group, rounded up to a multiple of 28
    std::vector<uint16_t> n(G);
    for (uint64_t g = 0; g < G; ++g) {
        const uint64_t last = std::min(N, (g + 1) * W) - 1;
        const uint32_t l = len[last];
        if (l > kMaxSequenceLength) // a .seq written by the reference may hold such a sequence; its padded length wraps there
            throw std::runtime_error("OSWALD: database holds a sequence of " + std::to_string(l) + " residues (limit " + std::to_string(kMaxSequenceLength) + ")");
        n[g] = (uint16_t)((l + kFpgaBlockWidth - 1) / kFpgaBlockWidth * kFpgaBlockWidth);
    }
    std::vector<uint64_t> gdisp(G + 1, 0);
    for (uint64_t g = 0; g < G; ++g) gdisp[g + 1] = gdisp[g] + (uint64_t)n[g] * W;
    db.vD = gdisp[G];
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
        c.b.assign(bytes, (uint8_t)kDummy);
        for (uint64_t k = 0; k < j; ++k) {
            const uint64_t g = start + k;
            uint8_t *dst = c.b.data() + c.disp[k];
            for (int lane = 0; lane < W; ++lane) {
                const uint64_t sidx = g * W + lane;
                if (sidx >= N) break;
                const uint8_t *src = s.data() + seq_off[sidx];
                for (uint32_t col = 0; col < len[sidx]; ++col) dst[(uint64_t)col * W + lane] = src[col];
            }
        }
        db.max_chunk_vD = std::max(db.max_chunk_vD, bytes);
        db.chunks.push_back(std::move(c));
    }
    return db;
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
uint64_t oswald_host_chunk_vD(unsigned c) { return g_db.chunks[c].b.size(); }
const uint8_t *oswald_host_chunk_b(unsigned c) { return g_db.chunks[c].b.data(); }
const uint16_t *oswald_host_chunk_n(unsigned c) { return g_db.chunks[c].n.data(); }
const uint16_t *oswald_host_chunk_nbb(unsigned c) { return g_db.chunks[c].nbb.data(); }
const uint32_t *oswald_host_chunk_disp(unsigned c) { return g_db.chunks[c].disp.data(); }

static std::vector<std::string> g_headers;
int oswald_host_load_headers(const char *dbname, uint64_t count)
{
    try { g_headers = oswald::load_database_headers(dbname, count); return 0; }
    catch (const std::exception &e) { g_host_err = e.what(); return -1; }
}
const char *oswald_host_header(uint64_t i) { return g_headers[i].c_str(); }

void oswald_host_top_scores(const int32_t *scores, uint64_t n, uint64_t r, int32_t *out_scores, uint64_t *out_index)
{
    std::vector<int32_t> s;
    std::vector<uint64_t> ix;
    oswald::top_scores(scores, n, r, s, ix);
    for (size_t k = 0; k < s.size(); ++k) { out_scores[k] = s[k]; out_index[k] = ix[k]; }
}

} // extern "C"
