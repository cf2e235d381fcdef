// oswald_amd/csrc/oswald_hip.cpp -- implementation of the C ABI declared in
// include/oswald_hip.h on top of the HIP runtime and the kernels of
// sw_kernels.hip.  This is the layer that stands in for OSWALD's OpenCL
// bring-up (reference host/src/utils.c:99-191) and enqueue path (reference
// host/src/FPGAsearch.c:82-238).  There is no CPU fallback: without a GPU
// every entry point fails with OSWALD_HIP_ENODEV / OSWALD_HIP_ERUNTIME.
#include "oswald_hip.h"
#include "sw_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e__ = (expr);                                                                              \
        if (e__ != hipSuccess)                                                                                \
            return fail(e__ == hipErrorOutOfMemory ? OSWALD_HIP_ENOMEM : OSWALD_HIP_ERUNTIME, "%s: %s", #expr, \
                        hipGetErrorString(e__));                                                              \
    } while (0)

// A device buffer that only ever grows.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Chunk {
    bool live = false;
    uint32_t ngroups = 0, W = 0, nblocks = 0;
    uint32_t score_stride = 0;   // nblocks*128
    uint32_t max_ncols4 = 0;     // largest stored extent of a block
    uint64_t total_col4 = 0;     // stored 4-column groups incl. the pad group per block
    DevBuf tiled, blocks, items, scores, ovf;
    std::vector<uint32_t> ncols4_alloc; // host copy, per block
    uint32_t nitems = 0, nitems_wg = 0;  // wave items / workgroup items of the queue
    uint64_t items_version = ~0ull;     // query-set version the item list was built for
    int items_bits = 0;                 // cell width it was planned for
    uint32_t max_lg = 0;                // widest geometry in the item list
    bool searched = false;
};

struct EventPair { hipEvent_t a, b; };

struct Device {
    int id = -1;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    uint32_t grid = 0;               // persistent workgroups per launch
    DevBuf queries, qlen, a_disp, prof_off, prof, submat, bnd, counters, staging_b, staging_n, staging_disp;
    DevBuf topr_scores, topr_index, wg_times;
    uint64_t bnd_stride = 0;         // uint2 per wave slot
    uint64_t queries_version = ~0ull; // what is currently uploaded
    uint64_t scoring_version = ~0ull;
    std::vector<Chunk> chunks;
    std::vector<EventPair> ev_pool, ev_used;
    double dp_ms = 0;
    uint64_t dp_launches = 0, rerun_items = 0;
};

} // namespace

struct oswald_hip_ctx {
    std::vector<Device> dev;
    // scoring
    bool have_scoring = false;
    int8_t submat[24 * 32];
    int open_gap = 10, extend_gap = 2, cell_bits = 16;
    uint64_t scoring_version = 0;
    // queries (host copies)
    bool have_queries = false;
    std::vector<uint8_t> a;
    std::vector<uint16_t> m;
    std::vector<uint32_t> a_disp, prof_off;
    uint32_t nq = 0, total_rowblocks = 0, max_rowblocks = 0;
    uint64_t queries_version = 0;
    bool profiling = false;
};

namespace {

int check_dev(oswald_hip_ctx *ctx, int dev)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (dev < 0 || dev >= (int)ctx->dev.size()) return fail(OSWALD_HIP_ENODEV, "device index %d out of range (context has %zu)", dev, ctx->dev.size());
    return 0;
}

// Upload query set / scoring to a device if it is stale and rebuild the profile.
int sync_queries(oswald_hip_ctx *ctx, Device &d)
{
    if (!ctx->have_scoring) return fail(OSWALD_HIP_ESTATE, "oswald_hip_set_scoring has not been called");
    if (!ctx->have_queries) return fail(OSWALD_HIP_ESTATE, "oswald_hip_set_queries has not been called");
    if (d.queries_version == ctx->queries_version && d.scoring_version == ctx->scoring_version) return 0;
    const uint32_t nq = ctx->nq;
    HIP_TRY(d.queries.reserve(ctx->a.size() + 16));
    HIP_TRY(d.qlen.reserve(nq * sizeof(uint16_t) + 16));
    HIP_TRY(d.a_disp.reserve((nq + 1) * sizeof(uint32_t)));
    HIP_TRY(d.prof_off.reserve((nq + 1) * sizeof(uint32_t)));
    HIP_TRY(d.submat.reserve(24 * 32));
    HIP_TRY(d.prof.reserve((size_t)ctx->total_rowblocks * 32 * sizeof(uint2) + 4096));
    if (!ctx->a.empty()) HIP_TRY(hipMemcpyAsync(d.queries.p, ctx->a.data(), ctx->a.size(), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.qlen.p, ctx->m.data(), nq * sizeof(uint16_t), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.a_disp.p, ctx->a_disp.data(), nq * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.prof_off.p, ctx->prof_off.data(), nq * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.submat.p, ctx->submat, 24 * 32, hipMemcpyHostToDevice, d.stream));
    HIP_TRY(osw_launch_build_profile((const uint8_t *)d.queries.p, (const uint32_t *)d.a_disp.p, (const uint16_t *)d.qlen.p,
                                     (const uint32_t *)d.prof_off.p, (const int8_t *)d.submat.p, nq, ctx->max_rowblocks,
                                     (uint2 *)d.prof.p, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream)); // host vectors may change after return
    d.queries_version = ctx->queries_version;
    d.scoring_version = ctx->scoring_version;
    return 0;
}

// Work queue of a chunk for the current query set.  An item is (query, block,
// sub-block, geometry G): G = 1 is a whole 128-sequence block on one wave; a
// heavier item (many strips x many columns) is cut into G sub-blocks of 128/G
// sequences whose G strips run side by side in the wave, so that no item is
// longer than a fraction of a wave's fair share of the launch.  The heaviest
// ones become workgroup items: four sub-blocks on the four waves of a workgroup
// sharing one 4x larger profile slice (taller rounds at high G).
// Cost model in VALU issue slots: rounds x (columns + pipeline fill) x (10 per
// row + ~35 per column).  Heaviest first within each class.
int build_items(oswald_hip_ctx *ctx, Device &d, Chunk &c)
{
    if (c.items_version == ctx->queries_version && c.items_bits == ctx->cell_bits) return 0;
    const uint32_t nq = ctx->nq;
    const bool i32 = ctx->cell_bits == 32;
    const uint32_t rmax = i32 ? OSW_RMAX32 : OSW_RMAX16, ldsr = i32 ? OSW_LDS_ROWS32 : OSW_LDS_ROWS16;
    const uint32_t ldsr_wg = ldsr * (OSW_WG_THREADS / 64);
    auto item_cost = [&](uint32_t m, uint32_t lg, uint32_t ncols, uint32_t lds_rows) {
        const OswPlan pl = osw_plan(m, 1u << lg, lds_rows, rmax);
        const double rows = 4.0 * (pl.base * pl.rounds + pl.extra); // rows per lane group over all rounds
        return (double)(ncols + (1u << lg)) * (10.0 * rows + 35.0 * pl.rounds);
    };
    // widest useful geometry per query: strips of >= 8 rows for wave items (per-column
    // overhead), >= 4 for workgroup items, and no lane group entirely past the query
    auto lg_limit = [&](uint32_t m, uint32_t lds_rows, uint32_t min_rows) {
        uint32_t lg = 0;
        while (lg < 6) {
            const uint32_t G2 = 2u << lg;
            const OswPlan pl = osw_plan(m, G2, lds_rows, rmax);
            if (G2 * osw_plan_maxrows(pl) > lds_rows || 4 * pl.base < min_rows || G2 * min_rows > pl.m4) break;
            ++lg;
        }
        return lg;
    };
    // Default geometry of a query: as many lane groups as it has 32-row strips, so that the
    // groups hand their bottom rows to each other in registers and (almost) nothing spills to
    // HBM.  Up to 128 rows that fits a wave's private LDS slice (wave item, one round); longer
    // queries run as workgroup items with the 4x larger shared slice (G <= 16, 512 rows per round).
    struct Mode { bool wg; uint32_t lg; };
    std::vector<Mode> def(nq);
    std::vector<uint32_t> lgmax(nq), lgmax_wg(nq);
    double total = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        lgmax[q] = lg_limit(ctx->m[q], ldsr, 8);
        lgmax_wg[q] = i32 ? 0 : lg_limit(ctx->m[q], ldsr_wg, 4);
        const uint32_t m4 = std::max(4u, (ctx->m[q] + 3u) & ~3u);
        uint32_t lg = 0;
        while ((rmax << lg) < m4 && lg < 4) ++lg;         // smallest G with G*rmax >= m4, at most 16
        if ((rmax << lg) <= ldsr || lgmax_wg[q] < 2) {
            def[q] = {false, std::min(lg, lgmax[q])};
            while ((rmax << def[q].lg) > ldsr && def[q].lg > 0) --def[q].lg; // int32 / tiny LDS: stay inside the slice
        } else {
            def[q] = {true, std::max(2u, std::min(lg, lgmax_wg[q]))};
        }
        for (uint32_t b = 0; b < c.nblocks; ++b)
            total += (double)(1u << def[q].lg) * item_cost(ctx->m[q], def[q].lg, c.ncols4_alloc[b] * 4, def[q].wg ? ldsr_wg : ldsr);
    }
    const double nwaves = (double)d.grid * (OSW_WG_THREADS / 64);
    const double target_div = getenv("OSWALD_HIP_TARGET_DIV") ? atof(getenv("OSWALD_HIP_TARGET_DIV")) : 3.0;
    const double target = std::max(total / nwaves / target_div, 4.0e4);
    // test hooks: OSWALD_HIP_FORCE_LG=k runs every item at geometry G = 2^k,
    // OSWALD_HIP_FORCE_WG=1 / 0 forces / forbids workgroup items
    int force_lg = -1, force_wg = -1;
    uint32_t wg_min_cols = 384, wg_wide_cols = 2048;
    if (const char *e = getenv("OSWALD_HIP_WG_MINCOLS")) wg_min_cols = (uint32_t)atoi(e);
    if (const char *e = getenv("OSWALD_HIP_WG_WIDECOLS")) wg_wide_cols = (uint32_t)atoi(e);
    if (const char *e = getenv("OSWALD_HIP_FORCE_LG")) force_lg = atoi(e);
    if (const char *e = getenv("OSWALD_HIP_FORCE_WG")) force_wg = atoi(e);
    if (force_lg > 6) force_lg = 6;
    if (i32) force_wg = 0; // the int32 kernel has no workgroup phase
    struct It { double cost; uint32_t x, b; };
    std::vector<It> its, its_wg;
    its.reserve((size_t)nq * c.nblocks + 1024);
    c.max_lg = 0;
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t b = 0; b < c.nblocks; ++b) {
            const uint32_t ncols = c.ncols4_alloc[b] * 4, m = ctx->m[q];
            bool wg = def[q].wg;
            uint32_t lg = def[q].lg;
            if (wg && ncols < wg_min_cols) {
                // short block: the pipeline fill of a wide geometry (G columns per round) would cost more
                // than the spill it saves; run as wave items at G = 4
                wg = false;
                lg = std::min(2u, lgmax[q]);
            } else if (wg && ncols < wg_wide_cols && lg > 3) {
                lg = 3; // medium block: 8 groups (256 rows per round)
            }
            double cost = item_cost(m, lg, ncols, wg ? ldsr_wg : ldsr);
            if (cost > target) {
                // too long for one wave's share: widen the geometry (shorter critical path, somewhat less
                // efficient).  Among the geometries that fit the target take the one with the least total
                // work; if none fits, the one with the shortest critical path.
                double best_fit_work = -1, best_cost = cost;
                bool fwg = wg, cwg = wg;
                uint32_t flg = lg, clg = lg;
                auto consider = [&](bool w, uint32_t k) {
                    const double ck = item_cost(m, k, ncols, w ? ldsr_wg : ldsr), work = ck * (double)(1u << k);
                    if (ck <= target && (best_fit_work < 0 || work < best_fit_work)) { best_fit_work = work; fwg = w; flg = k; }
                    if (ck < best_cost) { best_cost = ck; cwg = w; clg = k; }
                };
                if (!wg) for (uint32_t k = lg + 1; k <= lgmax[q]; ++k) consider(false, k);
                if (!i32) for (uint32_t k = std::max(2u, wg ? lg + 1 : 2u); k <= lgmax_wg[q]; ++k) consider(true, k);
                if (best_fit_work >= 0) { wg = fwg; lg = flg; } else { wg = cwg; lg = clg; }
                cost = item_cost(m, lg, ncols, wg ? ldsr_wg : ldsr);
            }
            if (force_lg >= 0) { lg = (uint32_t)force_lg; wg = false; }
            if (force_wg == 1) { wg = true; if (lg < 2) lg = 2; }
            if (force_wg == 0) wg = false;
            if (force_lg >= 0 || force_wg >= 0) cost = item_cost(m, lg, ncols, wg ? ldsr_wg : ldsr);
            const uint32_t G = 1u << lg;
            c.max_lg = std::max(c.max_lg, lg);
            if (wg) for (uint32_t s = 0; s < G; s += 4) its_wg.push_back({cost, OSW_ITEM_PACK(q, s, lg, 3u), b});
            else for (uint32_t s = 0; s < G; ++s) its.push_back({cost, OSW_ITEM_PACK(q, s, lg, 3u), b});
        }
    auto by_cost = [](const It &x, const It &y) { return x.cost > y.cost; };
    std::stable_sort(its.begin(), its.end(), by_cost);
    std::stable_sort(its_wg.begin(), its_wg.end(), by_cost);
    // issue priority of the long items (see set_wave_prio in sw_kernels.hip)
    double planned = 0;
    for (const It &i : its) planned += i.cost;
    for (const It &i : its_wg) planned += i.cost * 4;
    const double fair = planned / nwaves;
    const bool no_prio = getenv("OSWALD_HIP_NO_PRIO") != nullptr;
    auto prio_of = [&](double cost) { return no_prio ? 0u : cost > fair / 2 ? 3u : cost > fair / 4 ? 2u : cost > fair / 8 ? 1u : 0u; };
    std::vector<uint2> flat;
    flat.reserve(its.size() + its_wg.size());
    for (const It &i : its_wg) flat.push_back(make_uint2(i.x | (prio_of(i.cost) << 30), i.b));
    for (const It &i : its) flat.push_back(make_uint2(i.x | (prio_of(i.cost) << 30), i.b));
    c.nitems_wg = (uint32_t)its_wg.size();
    c.nitems = (uint32_t)its.size();
    if (getenv("OSWALD_HIP_DEBUG")) {
        double sw = 0, sg = 0;
        for (const It &i : its) sw += i.cost;
        for (const It &i : its_wg) sg += i.cost * 4;
        fprintf(stderr, "[oswald_hip] plan: total %.3g slots, %.0f waves, target %.3g; wave items %zu (sum %.3g, max %.3g), "
                        "workgroup items %zu (sum %.3g, max %.3g), max lg %u\n",
                total, nwaves, target, its.size(), sw, its.empty() ? 0.0 : its[0].cost, its_wg.size(), sg,
                its_wg.empty() ? 0.0 : its_wg[0].cost, c.max_lg);
        uint32_t hist[2][8] = {{0}};
        for (const It &i : its) hist[0][OSW_ITEM_LG(i.x)]++;
        for (const It &i : its_wg) hist[1][OSW_ITEM_LG(i.x)]++;
        for (int w = 0; w < 2; ++w)
            for (int k = 0; k < 7; ++k)
                if (hist[w][k]) fprintf(stderr, "[oswald_hip]   %s lg=%d: %u items\n", w ? "workgroup" : "wave", k, hist[w][k]);
    }
    HIP_TRY(c.items.reserve(flat.size() * sizeof(uint2) + 16));
    HIP_TRY(c.ovf.reserve((size_t)nq * c.nblocks * 64 * sizeof(uint2) + 16));
    HIP_TRY(c.scores.reserve((size_t)nq * c.score_stride * sizeof(int32_t) + 16));
    if (!flat.empty()) {
        HIP_TRY(hipMemcpyAsync(c.items.p, flat.data(), flat.size() * sizeof(uint2), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
    }
    c.items_version = ctx->queries_version;
    c.items_bits = ctx->cell_bits;
    return 0;
}

void drain_events(Device &d)
{
    for (auto &e : d.ev_used) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { d.dp_ms += ms; d.dp_launches++; }
        d.ev_pool.push_back(e);
    }
    d.ev_used.clear();
}

} // namespace

extern "C" {

int oswald_hip_abi_version(void) { return OSWALD_HIP_ABI_VERSION; }

const char *oswald_hip_last_error(void) { return g_err.c_str(); }

int oswald_hip_device_count(int *count)
{
    if (!count) return fail(OSWALD_HIP_EINVAL, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(OSWALD_HIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return 0;
}

int oswald_hip_init(int ndev, const int *device_ids, oswald_hip_ctx **out)
{
    if (!out) return fail(OSWALD_HIP_EINVAL, "ctx out-pointer is null");
    *out = nullptr;
    if (ndev <= 0) return fail(OSWALD_HIP_EINVAL, "ndev must be >= 1");
    int have = 0;
    hipError_t e = hipGetDeviceCount(&have);
    if (e != hipSuccess || have <= 0)
        return fail(OSWALD_HIP_ENODEV, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    oswald_hip_ctx *ctx = new (std::nothrow) oswald_hip_ctx;
    if (!ctx) return fail(OSWALD_HIP_ENOMEM, "out of host memory");
    ctx->dev.resize(ndev);
    for (int i = 0; i < ndev; ++i) {
        Device &d = ctx->dev[i];
        d.id = device_ids ? device_ids[i] : i;
        if (d.id < 0 || d.id >= have) { delete ctx; return fail(OSWALD_HIP_ENODEV, "device %d requested, %d visible", d.id, have); }
        hipError_t r = hipSetDevice(d.id);
        if (r == hipSuccess) r = hipGetDeviceProperties(&d.prop, d.id);
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        int per_cu = 0;
        if (r == hipSuccess) r = (hipError_t)osw_occupancy_pk16(&per_cu);
        if (r != hipSuccess) { delete ctx; return fail(OSWALD_HIP_ERUNTIME, "bring-up of device %d failed: %s", d.id, hipGetErrorString(r)); }
        if (per_cu < 1) per_cu = 1;
        d.grid = (uint32_t)d.prop.multiProcessorCount * (uint32_t)per_cu;
        r = d.counters.reserve(OSW_CTR_COUNT * sizeof(uint32_t));
        if (r != hipSuccess) { delete ctx; return fail(OSWALD_HIP_ENOMEM, "device %d: %s", d.id, hipGetErrorString(r)); }
    }
    *out = ctx;
    return 0;
}

int oswald_hip_finalize(oswald_hip_ctx *ctx)
{
    if (!ctx) return 0;
    for (Device &d : ctx->dev) {
        (void)hipSetDevice(d.id);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        for (Chunk &c : d.chunks) { c.tiled.release(); c.blocks.release(); c.items.release(); c.scores.release(); c.ovf.release(); }
        for (DevBuf *b : {&d.queries, &d.qlen, &d.a_disp, &d.prof_off, &d.prof, &d.submat, &d.bnd, &d.counters, &d.staging_b,
                          &d.staging_n, &d.staging_disp, &d.topr_scores, &d.topr_index, &d.wg_times})
            b->release();
        drain_events(d);
        for (auto &e : d.ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete ctx;
    return 0;
}

int oswald_hip_info(oswald_hip_ctx *ctx, int dev, char *buf, size_t buflen)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!buf || buflen == 0) return fail(OSWALD_HIP_EINVAL, "buffer is null");
    const Device &d = ctx->dev[dev];
    snprintf(buf, buflen,
             "Device %d: %s (%s)\n"
             "  compute units:            %d\n"
             "  max clock:                %d MHz\n"
             "  global memory:            %zu MiB\n"
             "  LDS per workgroup:        %zu KiB\n"
             "  wavefront size:           %d\n"
             "  L2 cache:                 %d KiB\n"
             "  persistent workgroups:    %u x %d threads\n",
             d.id, d.prop.name, d.prop.gcnArchName, d.prop.multiProcessorCount, d.prop.clockRate / 1000,
             d.prop.totalGlobalMem >> 20, d.prop.sharedMemPerBlock >> 10, d.prop.warpSize, d.prop.l2CacheSize >> 10, d.grid,
             OSW_WG_THREADS);
    return 0;
}

int oswald_hip_set_scoring(oswald_hip_ctx *ctx, const int8_t *submat, int open_gap, int extend_gap, int cell_bits)
{
    if (!ctx || !submat) return fail(OSWALD_HIP_EINVAL, "null argument");
    if (open_gap < 0 || extend_gap < 0) return fail(OSWALD_HIP_EINVAL, "gap penalties must be >= 0");
    if (open_gap + extend_gap > 32767) return fail(OSWALD_HIP_EINVAL, "open+extend must fit int16");
    if (cell_bits == 0) cell_bits = 16;
    if (cell_bits != 16 && cell_bits != 32) return fail(OSWALD_HIP_EINVAL, "cell_bits must be 16 or 32");
    memcpy(ctx->submat, submat, 24 * 32);
    ctx->open_gap = open_gap;
    ctx->extend_gap = extend_gap;
    ctx->cell_bits = cell_bits;
    ctx->have_scoring = true;
    ctx->scoring_version++;
    return 0;
}

int oswald_hip_set_queries(oswald_hip_ctx *ctx, const uint8_t *a, uint64_t Q, const uint16_t *m, const uint32_t *a_disp, uint32_t nq)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (nq > 0 && (!m || !a_disp || (Q > 0 && !a))) return fail(OSWALD_HIP_EINVAL, "null query arrays");
    for (uint32_t q = 0; q < nq; ++q)
        if ((uint64_t)a_disp[q] + m[q] > Q) return fail(OSWALD_HIP_EINVAL, "query %u runs past the residue buffer (disp %u + len %u > %llu)", q, a_disp[q], m[q], (unsigned long long)Q);
    ctx->a.assign(a, a + Q);
    ctx->m.assign(m, m + nq);
    ctx->a_disp.assign(a_disp, a_disp + nq);
    ctx->prof_off.resize(nq);
    uint32_t off = 0, mx = 1;
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t rb = (m[q] + 3u) / 4u;
        if (rb == 0) rb = 1;
        ctx->prof_off[q] = off;
        off += rb;
        mx = std::max(mx, rb);
    }
    ctx->total_rowblocks = off;
    ctx->max_rowblocks = mx;
    ctx->nq = nq;
    ctx->have_queries = true;
    ctx->queries_version++;
    return 0;
}

int oswald_hip_chunk_upload(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                            uint32_t ngroups, uint32_t W, int *chunk)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!chunk) return fail(OSWALD_HIP_EINVAL, "chunk out-pointer is null");
    if (W != 16 && W != 32 && W != 64 && W != 128) return fail(OSWALD_HIP_EINVAL, "lane_width must be 16, 32, 64 or 128");
    if (ngroups > 0 && (!b || !n || !disp)) return fail(OSWALD_HIP_EINVAL, "null chunk arrays");
    for (uint32_t g = 0; g < ngroups; ++g)
        if ((uint64_t)disp[g] + (uint64_t)n[g] * W > vD)
            return fail(OSWALD_HIP_EINVAL, "group %u runs past the chunk (disp %u + %u*%u > %llu)", g, disp[g], n[g], W, (unsigned long long)vD);
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    int slot = -1;
    for (size_t i = 0; i < d.chunks.size(); ++i) if (!d.chunks[i].live) { slot = (int)i; break; }
    if (slot < 0) { d.chunks.emplace_back(); slot = (int)d.chunks.size() - 1; }
    Chunk &c = d.chunks[slot];
    const uint32_t gpb = OSW_BLOCK_SEQS / W;
    c.ngroups = ngroups;
    c.W = W;
    c.nblocks = (ngroups + gpb - 1) / gpb;
    c.score_stride = c.nblocks * OSW_BLOCK_SEQS;
    c.ncols4_alloc.assign(c.nblocks, 0);
    std::vector<OswBlock> blocks(c.nblocks);
    uint64_t off = 0;
    c.max_ncols4 = 0;
    for (uint32_t B = 0; B < c.nblocks; ++B) {
        uint32_t mx = 0;
        for (uint32_t g = B * gpb; g < std::min(ngroups, (B + 1) * gpb); ++g) mx = std::max<uint32_t>(mx, n[g]);
        const uint32_t nc4 = (mx + 3) / 4;
        if (off > 0xfffffff0ull) return fail(OSWALD_HIP_EINVAL, "chunk too large for 32-bit column offsets");
        blocks[B].col4_off = (uint32_t)off;
        blocks[B].ncols4_alloc = nc4;
        blocks[B].ncols4 = nc4;
        blocks[B].seq0 = B * OSW_BLOCK_SEQS;
        c.ncols4_alloc[B] = nc4;
        c.max_ncols4 = std::max(c.max_ncols4, nc4);
        off += (uint64_t)nc4 + 2; // + the two prefetch pad groups
    }
    c.total_col4 = off;
    HIP_TRY(c.tiled.reserve((off + OSW_TILED_TAIL_GROUPS) * 64 * sizeof(uint2)));
    HIP_TRY(c.blocks.reserve(c.nblocks * sizeof(OswBlock) + 16));
    HIP_TRY(d.staging_b.reserve(vD + 64));
    HIP_TRY(d.staging_n.reserve(ngroups * sizeof(uint16_t) + 16));
    HIP_TRY(d.staging_disp.reserve(ngroups * sizeof(uint32_t) + 16));
    if (ngroups > 0) {
        HIP_TRY(hipMemcpyAsync(d.staging_b.p, b, vD, hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.staging_n.p, n, ngroups * sizeof(uint16_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.staging_disp.p, disp, ngroups * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(c.blocks.p, blocks.data(), c.nblocks * sizeof(OswBlock), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(osw_launch_retile((const uint8_t *)d.staging_b.p, (const uint16_t *)d.staging_n.p, (const uint32_t *)d.staging_disp.p,
                                  ngroups, W, (OswBlock *)c.blocks.p, c.nblocks, (uint2 *)c.tiled.p, d.stream));
    }
    // strip-boundary scratch: one region per resident wave, sized for the longest block
    const uint64_t stride = ((uint64_t)c.max_ncols4 * 4 + OSW_SCRATCH_PAD_COLS) * 64; // uint2 per wave slot
    if (stride > d.bnd_stride) {
        const uint64_t slots = (uint64_t)d.grid * (OSW_WG_THREADS / 64);
        HIP_TRY(hipStreamSynchronize(d.stream));
        HIP_TRY(d.bnd.reserve(slots * stride * sizeof(uint2)));
        d.bnd_stride = stride;
    }
    HIP_TRY(hipStreamSynchronize(d.stream)); // caller's buffers are free again (reference: clFinish, FPGAsearch.c:197)
    c.items_version = ~0ull;
    c.searched = false;
    c.live = true;
    *chunk = slot;
    return 0;
}

int oswald_hip_chunk_search(oswald_hip_ctx *ctx, int dev, int chunk, int32_t *scores_out)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    Chunk &c = d.chunks[chunk];
    HIP_TRY(hipSetDevice(d.id));
    if (int r = sync_queries(ctx, d)) return r;
    if (int r = build_items(ctx, d, c)) return r;
    if (c.nitems + c.nitems_wg == 0) { c.searched = true; return 0; }

    OswSearchArgs a;
    memset(&a, 0, sizeof a);
    a.tiled = (const uint2 *)c.tiled.p;
    a.blocks = (const OswBlock *)c.blocks.p;
    a.items = (const uint2 *)c.items.p;
    a.nitems = c.nitems;
    a.nitems_wg = c.nitems_wg;
    a.two_ended_waves = getenv("OSWALD_HIP_TWO_ENDED") ? (uint32_t)atoi(getenv("OSWALD_HIP_TWO_ENDED")) : 0u;
    a.force_all = ctx->cell_bits == 32 ? 1u : 0u;
    a.prof = (const uint2 *)d.prof.p;
    a.prof_off = (const uint32_t *)d.prof_off.p;
    a.qlen = (const uint16_t *)d.qlen.p;
    a.bnd = (uint2 *)d.bnd.p;
    a.bnd_stride = d.bnd_stride;
    a.scores = (int32_t *)c.scores.p;
    a.score_stride = c.score_stride;
    a.counters = (uint32_t *)d.counters.p;
    a.ovf_items = (uint2 *)c.ovf.p;
    const uint32_t goe = (uint32_t)(ctx->open_gap + ctx->extend_gap), ge = (uint32_t)ctx->extend_gap;
    a.goe_pk = goe | (goe << 16);
    a.ge_pk = ge | (ge << 16);
    a.goe = (int32_t)goe;
    a.ge = (int32_t)ge;

    const bool dbg_times = getenv("OSWALD_HIP_DEBUG_TIMES") != nullptr;
    if (dbg_times) {
        HIP_TRY(d.wg_times.reserve((size_t)d.grid * 4 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d.wg_times.p, 0, (size_t)d.grid * 4 * sizeof(unsigned long long), d.stream));
        a.wg_times = (unsigned long long *)d.wg_times.p;
    }
    EventPair ev{};
    if (ctx->profiling) {
        if (d.ev_pool.empty()) {
            HIP_TRY(hipEventCreate(&ev.a));
            HIP_TRY(hipEventCreate(&ev.b));
        } else { ev = d.ev_pool.back(); d.ev_pool.pop_back(); }
    }
    HIP_TRY(hipMemsetAsync(d.counters.p, 0, OSW_CTR_COUNT * sizeof(uint32_t), d.stream));
    const uint32_t grid = std::min<uint32_t>(d.grid, (c.nitems + 3) / 4 + c.nitems_wg);
    if (ctx->profiling) HIP_TRY(hipEventRecord(ev.a, d.stream));
    if (ctx->cell_bits == 16) HIP_TRY(osw_launch_pk16(a, grid, d.stream));
    HIP_TRY(osw_launch_i32(a, grid, d.stream));
    if (ctx->profiling) { HIP_TRY(hipEventRecord(ev.b, d.stream)); d.ev_used.push_back(ev); }
    c.searched = true;
    if (dbg_times) {
        // diagnostics only: when did the workgroups of the DP launch start / leave phase 1 / finish
        HIP_TRY(hipStreamSynchronize(d.stream));
        std::vector<unsigned long long> t((size_t)grid * 4);
        HIP_TRY(hipMemcpy(t.data(), d.wg_times.p, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (uint32_t g = 0; g < grid; ++g) { if (t[g * 4]) t0 = std::min(t0, t[g * 4]); t1 = std::max(t1, std::max(t[g * 4 + 2], t[g * 4 + 3])); }
        const double span = (double)(t1 - t0) / 100.0; // us
        uint32_t hist_p1[10] = {0}, hist_end[10] = {0};
        double sum_end = 0;
        for (uint32_t g = 0; g < grid; ++g) {
            const double p1 = (double)(t[g * 4 + 1] - t0) / 100.0, e = (double)(std::max(t[g * 4 + 2], t[g * 4 + 3]) - t0) / 100.0;
            hist_p1[std::min(9, (int)(p1 / span * 10))]++;
            hist_end[std::min(9, (int)(e / span * 10))]++;
            sum_end += e;
        }
        fprintf(stderr, "[oswald_hip] DP launch span %.1f us over %u workgroups; mean finish at %.0f%% of span\n", span, grid, 100.0 * sum_end / grid / span);
        fprintf(stderr, "[oswald_hip]   phase-1 exits by decile:");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_p1[k]);
        fprintf(stderr, "\n[oswald_hip]   finishes by decile:    ");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_end[k]);
        fprintf(stderr, "\n");
    }
    if (scores_out) {
        const size_t row = (size_t)c.ngroups * c.W * sizeof(int32_t);
        HIP_TRY(hipMemcpy2DAsync(scores_out, row, c.scores.p, (size_t)c.score_stride * sizeof(int32_t), row, ctx->nq,
                                 hipMemcpyDeviceToHost, d.stream));
    }
    return 0;
}

int oswald_hip_chunk_release(oswald_hip_ctx *ctx, int dev, int chunk)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    HIP_TRY(hipSetDevice(d.id));
    HIP_TRY(hipStreamSynchronize(d.stream));
    d.chunks[chunk].live = false; // buffers are kept for the next upload into this slot
    return 0;
}

int oswald_hip_search_chunk_async(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n,
                                  const uint32_t *disp, uint32_t ngroups, uint32_t W, int32_t *scores_out)
{
    int h = -1;
    if (int r = oswald_hip_chunk_upload(ctx, dev, b, vD, n, disp, ngroups, W, &h)) return r;
    int r = oswald_hip_chunk_search(ctx, dev, h, scores_out);
    // the slot is recycled by the next upload; the stream keeps the work ordered
    ctx->dev[dev].chunks[h].live = false;
    return r;
}

int oswald_hip_wait(oswald_hip_ctx *ctx, int dev)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (dev >= (int)ctx->dev.size()) return fail(OSWALD_HIP_ENODEV, "device index %d out of range", dev);
    for (int i = 0; i < (int)ctx->dev.size(); ++i) {
        if (dev >= 0 && i != dev) continue;
        HIP_TRY(hipSetDevice(ctx->dev[i].id));
        HIP_TRY(hipStreamSynchronize(ctx->dev[i].stream));
    }
    return 0;
}

int oswald_hip_chunk_topr(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t nvalid, uint32_t r, int32_t *scores, uint32_t *index)
{
    if (int rc = check_dev(ctx, dev)) return rc;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size()) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    Chunk &c = d.chunks[chunk];
    if (!c.searched) return fail(OSWALD_HIP_ESTATE, "chunk %d has not been searched", chunk);
    if (!scores || !index) return fail(OSWALD_HIP_EINVAL, "null output");
    if (nvalid > c.ngroups * c.W) return fail(OSWALD_HIP_EINVAL, "nvalid %u exceeds the chunk's %u lanes", nvalid, c.ngroups * c.W);
    if (r == 0 || ctx->nq == 0) return 0;
    HIP_TRY(hipSetDevice(d.id));
    const size_t cnt = (size_t)ctx->nq * r;
    HIP_TRY(d.topr_scores.reserve(cnt * sizeof(int32_t)));
    HIP_TRY(d.topr_index.reserve(cnt * sizeof(uint32_t)));
    HIP_TRY(osw_launch_topr((const int32_t *)c.scores.p, c.score_stride, nvalid, r, ctx->nq, (int32_t *)d.topr_scores.p,
                            (uint32_t *)d.topr_index.p, d.stream));
    HIP_TRY(hipMemcpyAsync(scores, d.topr_scores.p, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipMemcpyAsync(index, d.topr_index.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    return 0;
}

int oswald_hip_set_profiling(oswald_hip_ctx *ctx, int enable)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    ctx->profiling = enable != 0;
    return 0;
}

int oswald_hip_kernel_stats(oswald_hip_ctx *ctx, int dev, double *dp_kernel_ms, uint64_t *dp_launches, uint64_t *rerun_items, int reset)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    HIP_TRY(hipStreamSynchronize(d.stream));
    drain_events(d);
    uint32_t ctr[8] = {0};
    HIP_TRY(hipMemcpy(ctr, d.counters.p, sizeof ctr, hipMemcpyDeviceToHost));
    if (dp_kernel_ms) *dp_kernel_ms = d.dp_ms;
    if (dp_launches) *dp_launches = d.dp_launches;
    if (rerun_items) *rerun_items = ctr[OSW_CTR_OVF]; // of the most recent search
    if (reset) { d.dp_ms = 0; d.dp_launches = 0; }
    return 0;
}

int oswald_hip_chunk_geometry(oswald_hip_ctx *ctx, int dev, int chunk, uint64_t *out6)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    if (!out6) return fail(OSWALD_HIP_EINVAL, "null output");
    Chunk &c = d.chunks[chunk];
    HIP_TRY(hipSetDevice(d.id));
    HIP_TRY(hipStreamSynchronize(d.stream));
    std::vector<OswBlock> blocks(c.nblocks);
    if (c.nblocks) HIP_TRY(hipMemcpy(blocks.data(), c.blocks.p, c.nblocks * sizeof(OswBlock), hipMemcpyDeviceToHost));
    uint64_t alloc = 0, live = 0;
    for (const OswBlock &b : blocks) { alloc += b.ncols4_alloc; live += b.ncols4; }
    out6[0] = c.nblocks;
    out6[1] = alloc;
    out6[2] = live;
    out6[3] = live * 64 * sizeof(uint2);
    out6[4] = c.nitems + c.nitems_wg;
    out6[5] = c.max_lg;
    return 0;
}

} // extern "C"
