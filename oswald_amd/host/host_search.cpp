// oswald_amd/host/host_search.cpp -- the HOST compute path of the command line tool: `-m 2` (host only) and the
// host share of `-m 1` (hybrid).  It is a mode the caller selects, never a fallback: `-m 0` and the C ABI fail
// loudly without a GPU.  It stands where the reference has its host SIMD kernels (reference
// host/src/HybridSearch.c:1540-1880 SSE, :790-1140 AVX2), whose contract is "the exact Smith-Waterman score of every
// (query, database sequence)"; the formulation here is its own: sixteen database sequences of a group side by side
// in the sixteen int16 lanes of one AVX2 register, the database streamed column by column, a per-column score
// profile (24 vectors, two byte shuffles each), H and E of every query row in two arrays, saturating int16
// arithmetic, and an exact scalar int32 pass for the few sequences whose score reaches the int16 ceiling.  `-v 32`
// selects that AVX2 kernel; `-v 16` (the reference's default: its SSE path, BASELINE configs[0]) the same formulation
// on SSE4.1 -- the sixteen sequences in two 8 x int16 registers.
//   Round 5: like the reference's host kernels (HybridSearch.c:1618-1680: 8 -> 16 -> 32 bits) the search STARTS in
// saturating int8 -- the same formulation with sixteen (SSE4.1) or thirty-two (AVX2: two groups side by side) sequences
// per register -- and only the groups in which a lane reaches 127 are redone in int16, and of those the lanes at 32767
// in int32.  A score below the ceiling of a stage is exact: the only saturation that can go unnoticed is the negative
// one of E and F, and a negative E or F never decides a cell (H >= 0).
#include "oswald_host.h"

#include <immintrin.h>
#include <omp.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace oswald {

namespace {

// exact score of one query against lane `lane` of an interleaved group (dummy residues score 0 and are harmless)
int32_t scalar_score(const uint8_t *a, uint32_t m, const uint8_t *grp, uint32_t ncols, int W, int lane, const int8_t *submat, int goe, int ge,
                     std::vector<int32_t> &H, std::vector<int32_t> &E)
{
    H.assign(m + 1, 0);
    E.assign(m + 1, 0);
    int32_t best = 0;
    for (uint32_t j = 0; j < ncols; ++j) {
        const uint32_t r = grp[(size_t)j * W + lane] & 31u;
        int32_t diag = 0, f = 0;
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t ai = a[i] < 24 ? a[i] : 23;
            int32_t h = diag + submat[ai * 32 + r];
            h = std::max(std::max(h, E[i]), std::max(f, 0));
            diag = H[i];
            H[i] = h;
            best = std::max(best, h);
            const int32_t u = h - goe;
            E[i] = std::max(E[i] - ge, u);
            f = std::max(f - ge, u);
        }
    }
    return best;
}

struct Scratch {
    std::vector<int16_t> H, E; // 16 lanes per query row
    std::vector<int8_t> H8, E8; // 32 (AVX2) or 16 (SSE4.1) lanes per query row of a block
    std::vector<int8_t> Hb8, Fb8; // ... and per column: the row above a block (-b)
    std::vector<int32_t> h32, e32;
};

// 8-bit first stage, AVX2: TWO groups side by side (lanes 0..15: group A, 16..31: group B; a group that is through -- or
// absent: ncolsB = 0 -- reads dummy residues, which score 0 against everything and cannot raise a score).  out: the
// lanes' best scores, 127 = reached the ceiling.
__attribute__((target("avx2"))) void simd_pair8(const uint8_t *a, uint32_t m, const uint8_t *grpA, uint32_t ncolsA, const uint8_t *grpB, uint32_t ncolsB,
                                               const int8_t *submat, int goe, int ge, uint32_t bw, Scratch &s, int8_t out[32])
{
    // (penalties beyond 127 act like 127 on values that never exceed 127: the result is <= 0 either way)
    const __m256i vgoe = _mm256_set1_epi8((char)std::min(goe, 127)), vge = _mm256_set1_epi8((char)std::min(ge, 127)), zero = _mm256_setzero_si256();
    __m256i best = zero;
    __m256i lo[24], hi[24];
    for (int c = 0; c < 24; ++c) {
        lo[c] = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i *)(submat + c * 32)));
        hi[c] = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i *)(submat + c * 32 + 16)));
    }
    __m256i P[24];
    const __m256i fifteen = _mm256_set1_epi8(15), mask31 = _mm256_set1_epi8(31);
    const __m128i dummy = _mm_set1_epi8(23);
    const uint32_t ncols = std::max(ncolsA, ncolsB);
    // -b: the query is worked through in blocks of bw rows, every block over all columns (H and E of a block's rows stay in
    // the L1 cache); the row above a block -- H and F of the block before at its last row, one vector per column -- is carried
    // in Hb / Fb (the reference blocks the DATABASE sequence and carries a column, HybridSearch.c:1589-1600: the same idea on
    // the other axis, which is the one this column-streamed kernel keeps arrays along)
    if (bw == 0 || bw > m) bw = m;
    const bool blocked = bw < m;
    s.H8.assign((size_t)bw * 32, 0);
    s.E8.assign((size_t)bw * 32, 0);
    if (blocked) { s.Hb8.assign((size_t)ncols * 32, 0); s.Fb8.assign((size_t)ncols * 32, 0); }
    __m256i *H = (__m256i *)s.H8.data(), *E = (__m256i *)s.E8.data(), *Hb = (__m256i *)s.Hb8.data(), *Fb = (__m256i *)s.Fb8.data(); // unaligned accesses below
    for (uint32_t i0 = 0; i0 < m; i0 += bw) {
        const uint32_t rows = std::min(bw, m - i0);
        if (i0) { std::memset(s.H8.data(), 0, (size_t)rows * 32); std::memset(s.E8.data(), 0, (size_t)rows * 32); }
        __m256i above_prev = zero; // H(i0 - 1, j - 1)
        for (uint32_t j = 0; j < ncols; ++j) {
            const __m128i ra = j < ncolsA ? _mm_loadu_si128((const __m128i *)(grpA + (size_t)j * 16)) : dummy;
            const __m128i rb = j < ncolsB ? _mm_loadu_si128((const __m128i *)(grpB + (size_t)j * 16)) : dummy;
            const __m256i r = _mm256_and_si256(_mm256_inserti128_si256(_mm256_castsi128_si256(ra), rb, 1), mask31);
            const __m256i upper = _mm256_cmpgt_epi8(r, fifteen), r15 = _mm256_and_si256(r, fifteen);
            for (int c = 0; c < 24; ++c) P[c] = _mm256_blendv_epi8(_mm256_shuffle_epi8(lo[c], r), _mm256_shuffle_epi8(hi[c], r15), upper);
            __m256i diag = above_prev, f = zero, h = zero;
            if (blocked && i0) { above_prev = _mm256_loadu_si256(Hb + j); f = _mm256_loadu_si256(Fb + j); }
            for (uint32_t i = 0; i < rows; ++i) {
                const uint32_t ai = a[i0 + i] < 24 ? a[i0 + i] : 23;
                h = _mm256_adds_epi8(diag, P[ai]);
                const __m256i e = _mm256_loadu_si256(E + i);
                h = _mm256_max_epi8(_mm256_max_epi8(h, e), _mm256_max_epi8(f, zero));
                diag = _mm256_loadu_si256(H + i);
                _mm256_storeu_si256(H + i, h);
                best = _mm256_max_epi8(best, h);
                const __m256i u = _mm256_subs_epi8(h, vgoe);
                _mm256_storeu_si256(E + i, _mm256_max_epi8(_mm256_subs_epi8(e, vge), u));
                f = _mm256_max_epi8(_mm256_subs_epi8(f, vge), u);
            }
            if (blocked) { _mm256_storeu_si256(Hb + j, h); _mm256_storeu_si256(Fb + j, f); }
        }
    }
    _mm256_storeu_si256((__m256i *)out, best);
}

// 8-bit first stage, SSE4.1: one group per register
__attribute__((target("sse4.1"))) void simd_group8_sse41(const uint8_t *a, uint32_t m, const uint8_t *grp, uint32_t ncols, const int8_t *submat, int goe, int ge,
                                                        uint32_t bw, Scratch &s, int8_t out[16])
{
    const __m128i vgoe = _mm_set1_epi8((char)std::min(goe, 127)), vge = _mm_set1_epi8((char)std::min(ge, 127)), zero = _mm_setzero_si128();
    __m128i best = zero;
    __m128i lo[24], hi[24], P[24];
    for (int c = 0; c < 24; ++c) {
        lo[c] = _mm_loadu_si128((const __m128i *)(submat + c * 32));
        hi[c] = _mm_loadu_si128((const __m128i *)(submat + c * 32 + 16));
    }
    const __m128i fifteen = _mm_set1_epi8(15);
    if (bw == 0 || bw > m) bw = m; // (row blocks: see simd_pair8)
    const bool blocked = bw < m;
    s.H8.assign((size_t)bw * 16, 0);
    s.E8.assign((size_t)bw * 16, 0);
    if (blocked) { s.Hb8.assign((size_t)ncols * 16, 0); s.Fb8.assign((size_t)ncols * 16, 0); }
    __m128i *H = (__m128i *)s.H8.data(), *E = (__m128i *)s.E8.data(), *Hb = (__m128i *)s.Hb8.data(), *Fb = (__m128i *)s.Fb8.data();
    for (uint32_t i0 = 0; i0 < m; i0 += bw) {
        const uint32_t rows = std::min(bw, m - i0);
        if (i0) { std::memset(s.H8.data(), 0, (size_t)rows * 16); std::memset(s.E8.data(), 0, (size_t)rows * 16); }
        __m128i above_prev = zero;
        for (uint32_t j = 0; j < ncols; ++j) {
            const __m128i r = _mm_and_si128(_mm_loadu_si128((const __m128i *)(grp + (size_t)j * 16)), _mm_set1_epi8(31));
            const __m128i upper = _mm_cmpgt_epi8(r, fifteen), r15 = _mm_and_si128(r, fifteen);
            for (int c = 0; c < 24; ++c) P[c] = _mm_blendv_epi8(_mm_shuffle_epi8(lo[c], r), _mm_shuffle_epi8(hi[c], r15), upper);
            __m128i diag = above_prev, f = zero, h = zero;
            if (blocked && i0) { above_prev = _mm_loadu_si128(Hb + j); f = _mm_loadu_si128(Fb + j); }
            for (uint32_t i = 0; i < rows; ++i) {
                const uint32_t ai = a[i0 + i] < 24 ? a[i0 + i] : 23;
                const __m128i e = _mm_loadu_si128(E + i);
                h = _mm_max_epi8(_mm_max_epi8(_mm_adds_epi8(diag, P[ai]), e), _mm_max_epi8(f, zero));
                diag = _mm_loadu_si128(H + i);
                _mm_storeu_si128(H + i, h);
                best = _mm_max_epi8(best, h);
                const __m128i u = _mm_subs_epi8(h, vgoe);
                _mm_storeu_si128(E + i, _mm_max_epi8(_mm_subs_epi8(e, vge), u));
                f = _mm_max_epi8(_mm_subs_epi8(f, vge), u);
            }
            if (blocked) { _mm_storeu_si128(Hb + j, h); _mm_storeu_si128(Fb + j, f); }
        }
    }
    _mm_storeu_si128((__m128i *)out, best);
}

// sixteen lanes at once; returns the lanes' best scores (32767 = reached the ceiling)
__attribute__((target("avx2"))) void simd_group(const uint8_t *a, uint32_t m, const uint8_t *grp, uint32_t ncols, const int8_t *submat, int goe, int ge,
                                               Scratch &s, int16_t out[16])
{
    s.H.assign((size_t)m * 16, 0);
    s.E.assign((size_t)m * 16, 0);
    __m256i *H = (__m256i *)s.H.data(), *E = (__m256i *)s.E.data(); // unaligned accesses below
    const __m256i vgoe = _mm256_set1_epi16((short)std::min(goe, 32767)), vge = _mm256_set1_epi16((short)std::min(ge, 32767)), zero = _mm256_setzero_si256();
    __m256i best = zero;
    __m128i lo[24], hi[24];
    for (int c = 0; c < 24; ++c) {
        lo[c] = _mm_loadu_si128((const __m128i *)(submat + c * 32));
        hi[c] = _mm_loadu_si128((const __m128i *)(submat + c * 32 + 16));
    }
    __m256i P[24];
    const __m128i fifteen = _mm_set1_epi8(15);
    for (uint32_t j = 0; j < ncols; ++j) {
        const __m128i r = _mm_and_si128(_mm_loadu_si128((const __m128i *)(grp + (size_t)j * 16)), _mm_set1_epi8(31));
        const __m128i upper = _mm_cmpgt_epi8(r, fifteen); // codes 16..31 live in the second half of a matrix row
        for (int c = 0; c < 24; ++c) {
            const __m128i s8 = _mm_blendv_epi8(_mm_shuffle_epi8(lo[c], r), _mm_shuffle_epi8(hi[c], _mm_and_si128(r, fifteen)), upper);
            P[c] = _mm256_cvtepi8_epi16(s8);
        }
        __m256i diag = zero, f = zero;
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t ai = a[i] < 24 ? a[i] : 23;
            __m256i h = _mm256_adds_epi16(diag, P[ai]);
            const __m256i e = _mm256_loadu_si256(E + i);
            h = _mm256_max_epi16(_mm256_max_epi16(h, e), _mm256_max_epi16(f, zero));
            diag = _mm256_loadu_si256(H + i);
            _mm256_storeu_si256(H + i, h);
            best = _mm256_max_epi16(best, h);
            const __m256i u = _mm256_subs_epi16(h, vgoe);
            _mm256_storeu_si256(E + i, _mm256_max_epi16(_mm256_subs_epi16(e, vge), u));
            f = _mm256_max_epi16(_mm256_subs_epi16(f, vge), u);
        }
    }
    _mm256_storeu_si256((__m256i *)out, best);
}

// the same on SSE4.1: the sixteen lanes in two 128-bit registers (lanes 0..7 and 8..15)
__attribute__((target("sse4.1"))) void simd_group_sse41(const uint8_t *a, uint32_t m, const uint8_t *grp, uint32_t ncols, const int8_t *submat, int goe, int ge,
                                                       Scratch &s, int16_t out[16])
{
    s.H.assign((size_t)m * 16, 0);
    s.E.assign((size_t)m * 16, 0);
    __m128i *H = (__m128i *)s.H.data(), *E = (__m128i *)s.E.data(); // two per query row; unaligned accesses below
    const __m128i vgoe = _mm_set1_epi16((short)std::min(goe, 32767)), vge = _mm_set1_epi16((short)std::min(ge, 32767)), zero = _mm_setzero_si128();
    __m128i best0 = zero, best1 = zero;
    __m128i lo[24], hi[24];
    for (int c = 0; c < 24; ++c) {
        lo[c] = _mm_loadu_si128((const __m128i *)(submat + c * 32));
        hi[c] = _mm_loadu_si128((const __m128i *)(submat + c * 32 + 16));
    }
    __m128i P0[24], P1[24];
    const __m128i fifteen = _mm_set1_epi8(15);
    for (uint32_t j = 0; j < ncols; ++j) {
        const __m128i r = _mm_and_si128(_mm_loadu_si128((const __m128i *)(grp + (size_t)j * 16)), _mm_set1_epi8(31));
        const __m128i upper = _mm_cmpgt_epi8(r, fifteen);
        for (int c = 0; c < 24; ++c) {
            const __m128i s8 = _mm_blendv_epi8(_mm_shuffle_epi8(lo[c], r), _mm_shuffle_epi8(hi[c], _mm_and_si128(r, fifteen)), upper);
            P0[c] = _mm_cvtepi8_epi16(s8);
            P1[c] = _mm_cvtepi8_epi16(_mm_srli_si128(s8, 8));
        }
        __m128i diag0 = zero, diag1 = zero, f0 = zero, f1 = zero;
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t ai = a[i] < 24 ? a[i] : 23;
            const __m128i e0 = _mm_loadu_si128(E + 2 * i), e1 = _mm_loadu_si128(E + 2 * i + 1);
            __m128i h0 = _mm_max_epi16(_mm_max_epi16(_mm_adds_epi16(diag0, P0[ai]), e0), _mm_max_epi16(f0, zero));
            __m128i h1 = _mm_max_epi16(_mm_max_epi16(_mm_adds_epi16(diag1, P1[ai]), e1), _mm_max_epi16(f1, zero));
            diag0 = _mm_loadu_si128(H + 2 * i);
            diag1 = _mm_loadu_si128(H + 2 * i + 1);
            _mm_storeu_si128(H + 2 * i, h0);
            _mm_storeu_si128(H + 2 * i + 1, h1);
            best0 = _mm_max_epi16(best0, h0);
            best1 = _mm_max_epi16(best1, h1);
            const __m128i u0 = _mm_subs_epi16(h0, vgoe), u1 = _mm_subs_epi16(h1, vgoe);
            _mm_storeu_si128(E + 2 * i, _mm_max_epi16(_mm_subs_epi16(e0, vge), u0));
            _mm_storeu_si128(E + 2 * i + 1, _mm_max_epi16(_mm_subs_epi16(e1, vge), u1));
            f0 = _mm_max_epi16(_mm_subs_epi16(f0, vge), u0);
            f1 = _mm_max_epi16(_mm_subs_epi16(f1, vge), u1);
        }
    }
    _mm_storeu_si128((__m128i *)out, best0);
    _mm_storeu_si128((__m128i *)(out + 8), best1);
}

} // namespace

void host_search_groups(const Queries &q, const Chunk &c, uint64_t g0, uint64_t g1, int W, const int8_t *submat, int open_gap, int extend_gap,
                        int threads, int32_t *scores, uint64_t row_stride, uint64_t col0, int cpu_vector_length, const std::atomic<bool> *cancel,
                        std::atomic<uint64_t> *cells_done, int block_width)
{
    if (W != kFpgaVectorLength) throw std::runtime_error("OSWALD: the host path works on groups of 16 sequences.");
    const int goe = open_gap + extend_gap, ge = extend_gap;
    // -v 32: the AVX2 kernel, -v 16: the SSE4.1 kernel (the reference's two host paths); whichever the CPU lacks falls
    // to the next one down, the scalar kernel last
    const bool avx2 = cpu_vector_length != 16 && __builtin_cpu_supports("avx2");
    const bool sse41 = !avx2 && __builtin_cpu_supports("sse4.1");
    const uint64_t nq = q.m.size();
    if (threads < 1) threads = 1;
    // the 8-bit stage takes the groups two at a time on AVX2 (neighbours in the sorted database: the same length but for a few columns)
    const uint64_t ngroups = g1 > g0 ? g1 - g0 : 0, step = avx2 ? 2 : 1, nunits = (ngroups + step - 1) / step;
    const uint32_t bw = block_width > 0 ? (uint32_t)block_width : 0u; // -b: query rows per block of the 8-bit stage (0: the whole query)
    const bool int8_stage = (avx2 || sse41) && !std::getenv("OSWALD_HOST_NO_INT8"); // (test hook: start in int16, the kernel of rounds 1-4)
#pragma omp parallel num_threads(threads)
    {
        Scratch s;
#pragma omp for schedule(dynamic, 1)
        for (int64_t ui = 0; ui < (int64_t)nunits; ++ui) { // longest groups first
            if (cancel && cancel->load(std::memory_order_relaxed)) continue; // (the caller no longer needs the rest: hybrid calibration)
            // unit ui: groups gA (and, on AVX2, gB = gA - 1 if there is one)
            const uint64_t gA = g1 - 1 - (uint64_t)ui * step;
            const bool haveB = step == 2 && gA > g0;
            const uint64_t gB = haveB ? gA - 1 : gA;
            for (uint64_t qi = 0; qi < nq; ++qi) {
                if (qi && cancel && cancel->load(std::memory_order_relaxed)) break; // (a unit in progress is left at the next query)
                const uint8_t *a = q.a.data() + q.a_disp[qi];
                const uint32_t m = q.m[qi];
                int8_t lane8[32];
                bool have8 = false;
                if (int8_stage && avx2) { simd_pair8(a, m, c.b + c.disp[gA], c.n[gA], haveB ? c.b + c.disp[gB] : nullptr, haveB ? c.n[gB] : 0, submat, goe, ge, bw, s, lane8); have8 = true; }
                else if (int8_stage) { simd_group8_sse41(a, m, c.b + c.disp[gA], c.n[gA], submat, goe, ge, bw, s, lane8); have8 = true; }
                for (int part = 0; part < (haveB ? 2 : 1); ++part) {
                    const uint64_t g = part ? gB : gA;
                    const uint8_t *grp = c.b + c.disp[g];
                    const uint32_t ncols = c.n[g];
                    const int8_t *l8 = lane8 + 16 * part;
                    int32_t *dst = scores + qi * row_stride + col0 + (g - g0) * W;
                    bool redo = !have8;
                    for (int lane = 0; lane < W && !redo; ++lane) redo = l8[lane] == 127;
                    int16_t lane16[16];
                    if (redo) { // a lane at the 8-bit ceiling (or no 8-bit stage): the group in int16
                        if (avx2) simd_group(a, m, grp, ncols, submat, goe, ge, s, lane16);
                        else if (sse41) simd_group_sse41(a, m, grp, ncols, submat, goe, ge, s, lane16);
                    }
                    for (int lane = 0; lane < W; ++lane) {
                        if (have8 && l8[lane] < 127) dst[lane] = l8[lane];
                        else if ((avx2 || sse41) && lane16[lane] < 32767) dst[lane] = lane16[lane];
                        else dst[lane] = scalar_score(a, m, grp, ncols, W, lane, submat, goe, ge, s.h32, s.e32); // at the int16 ceiling: exact int32
                    }
                    // every finished (group, query) counts: a calibration that stops the clock while groups are in progress
                    // would otherwise rate the host on whole groups only (up to 2 x too low with a few threads)
                    if (cells_done) cells_done->fetch_add((uint64_t)m * ncols * W, std::memory_order_relaxed);
                }
            }
        }
    }
}

} // namespace oswald
