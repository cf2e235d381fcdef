// oswald_amd/csrc/oswald_hip.cpp -- implementation of the C ABI declared in
// include/oswald_hip.h on top of the HIP runtime and the kernels of
// sw_kernels.hip.  This is the layer that stands in for OSWALD's OpenCL
// bring-up (reference host/src/utils.c:99-191) and enqueue path (reference
// host/src/FPGAsearch.c:82-238).  There is no CPU fallback: without a GPU
// every entry point fails with OSWALD_HIP_ENODEV / OSWALD_HIP_ERUNTIME.
#include "oswald_hip.h"
#include "sw_kernels.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
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

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) return fail(OSWALD_HIP_ECOMM, "%s: %s", #expr, ncclGetErrorString(r__)); \
    } while (0)

// Environment hooks.  They are read when a context is CONFIGURED (oswald_hip_init, oswald_hip_set_scoring,
// oswald_hip_set_queries), never on the per-search path.  The default build knows the test hooks and the
// host-side timing prints only -- none of them changes a result.  The planner sweep knobs and the kernel timing
// diagnostics (one of which, NOSPILL, yields wrong scores) exist only in the -DOSW_DIAG build
// (`make -C oswald_amd/csrc diag` -> liboswald_hip_diag.so, which tools/ load on request and nothing else does).
struct Tunables {
    int cell_bits_default = 16;        // OSWALD_HIP_CELL_BITS: cell arithmetic when the caller passes cell_bits 0
    bool no_frame = false;             // OSWALD_HIP_NO_FRAME=1: plain biased int16 cell only
    int force_lg = -1, force_wg = -1;  // OSWALD_HIP_FORCE_LG=k: every item at G = 2^k; OSWALD_HIP_FORCE_WG=0|1
    int pairs = 1;                     // OSWALD_HIP_PAIRS=0|1|2: never pair / pair when cheaper / pair every neighbour
    bool debug_plan = false;           // OSWALD_HIP_DEBUG=1: print the work-queue plan
    bool debug_phases = false;         // OSWALD_HIP_DEBUG_PHASES=1: wall time of the host-side phases
    bool no_stream_classes = false;     // OSWALD_HIP_NO_STREAM_CLASSES=1 (A/B hook): the DMA streams are ordinary streams, as before the second session of round 4
    bool plan_on_estimates = false;     // OSWALD_HIP_PLAN_EST=1 (experiment): every search is planned on the group-length extents
    bool plan_waits_for_upload = false; // OSWALD_HIP_PLAN_WAITS=1 (test hook): a search waits for its chunk's upload and plans on the live extents (the behaviour before the second session of round 4)
    bool no_pin = false;               // OSWALD_HIP_NO_PIN=1: do not pin the caller's score table for the download
    size_t fake_free_mem = 0;          // OSWALD_HIP_FAKE_FREE_MEM=bytes: oswald_hip_max_chunk_size reckons with a device that has no more free (test hook)
    // planner parameters: constants in the default build, OSWALD_HIP_* sweep knobs with -DOSW_DIAG
    double pair_margin = 0.95, col_cost = 10.0, target_div = 1.25, quad_frac = 0.5, entries_per_wg = 4.0;
    uint32_t wg_min_cols = 2048, wg_wide_cols = 2048, wg_min_cols_single = 0, wg_wide_cols_single = 0, wg_min_rounds = 0, wg_min_cols_long = 0, two_ended = 0, one_ended_wg = 1, grid_per_cu = 0;
    bool no_prio = false, one_stream = false;
    bool debug_times = false, debug_nospill = false; // -DOSW_DIAG only
    void refresh();
};
bool g_debug_slow = false; // OSWALD_HIP_DEBUG_SLOW=1: report allocations / pinning that take > 5 ms

void Tunables::refresh()
{
    auto num = [](const char *name, double dflt) { const char *e = getenv(name); return e ? atof(e) : dflt; };
    auto flag = [](const char *name) { return getenv(name) != nullptr; };
    *this = Tunables();
    cell_bits_default = (int)num("OSWALD_HIP_CELL_BITS", 16);
    no_frame = flag("OSWALD_HIP_NO_FRAME");
    force_lg = std::min(6, (int)num("OSWALD_HIP_FORCE_LG", -1));
    force_wg = (int)num("OSWALD_HIP_FORCE_WG", -1);
    pairs = (int)num("OSWALD_HIP_PAIRS", 1);
    debug_plan = flag("OSWALD_HIP_DEBUG");
    debug_phases = flag("OSWALD_HIP_DEBUG_PHASES");
    plan_waits_for_upload = flag("OSWALD_HIP_PLAN_WAITS");
    plan_on_estimates = flag("OSWALD_HIP_PLAN_EST");
    no_stream_classes = flag("OSWALD_HIP_NO_STREAM_CLASSES");
    no_pin = flag("OSWALD_HIP_NO_PIN");
    fake_free_mem = (size_t)num("OSWALD_HIP_FAKE_FREE_MEM", 0);
    g_debug_slow = flag("OSWALD_HIP_DEBUG_SLOW");
#ifdef OSW_DIAG
    pair_margin = num("OSWALD_HIP_PAIR_MARGIN", pair_margin);
    col_cost = num("OSWALD_HIP_COL_COST", col_cost);
    target_div = num("OSWALD_HIP_TARGET_DIV", target_div);
    quad_frac = num("OSWALD_HIP_QUAD_FRAC", quad_frac);
    entries_per_wg = num("OSWALD_HIP_ENTRIES_PER_WG", entries_per_wg);
    wg_min_cols = (uint32_t)num("OSWALD_HIP_WG_MINCOLS", wg_min_cols);
    wg_wide_cols = (uint32_t)num("OSWALD_HIP_WG_WIDECOLS", wg_wide_cols);
    wg_min_cols_single = (uint32_t)num("OSWALD_HIP_WG_MINCOLS_SINGLE", wg_min_cols_single);
    wg_wide_cols_single = (uint32_t)num("OSWALD_HIP_WG_WIDECOLS_SINGLE", wg_wide_cols_single);
    wg_min_rounds = (uint32_t)num("OSWALD_HIP_WG_MINROUNDS", wg_min_rounds);
    wg_min_cols_long = (uint32_t)num("OSWALD_HIP_WG_MINCOLS_LONG", wg_min_cols_long);
    two_ended = (uint32_t)num("OSWALD_HIP_TWO_ENDED", 0);
    one_ended_wg = (uint32_t)num("OSWALD_HIP_ONE_ENDED_WG", one_ended_wg);
    grid_per_cu = (uint32_t)num("OSWALD_HIP_GRID_PER_CU", 0);
    no_prio = flag("OSWALD_HIP_NO_PRIO");
    one_stream = flag("OSWALD_HIP_ONE_STREAM");
    debug_times = flag("OSWALD_HIP_DEBUG_TIMES");
    debug_nospill = flag("OSWALD_HIP_DEBUG_NOSPILL");
#endif
}

// wall time of the host-side phases of an upload / search (Tunables::debug_phases)
struct PhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit PhaseTimer(bool enabled) : on(enabled), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[oswald_hip] phase %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// ... and without any synchronisation of its own: which section of a call held the host (OSWALD_HIP_DEBUG_SLOW)
struct HoldTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit HoldTimer(bool enabled) : on(enabled), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(n - t).count();
        if (ms > 3.0) fprintf(stderr, "[oswald_hip] the host was held %.1f ms in: %s\n", ms, what);
        t = n;
    }
};

// A device buffer that only ever grows.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        // (a failed hipFree / hipMalloc leaves HIP's last error set; callers that recover from the failure must not see it
        // again in the next launch check: it is cleared here, the code is returned)
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) { (void)hipGetLastError(); return e; } p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, want);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 5.0 && g_debug_slow) fprintf(stderr, "[oswald_hip] slow hipMalloc: %zu bytes took %.1f ms\n", want, ms);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return e; }
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
    DevBuf tiled, blocks, sub_cols_buf, scores, ovf, ovf8;
    // The work queues, two sets in turn: a plan goes to the device on the copy stream AT ONCE, and since a search is planned
    // while its chunk's upload is still on its way, the search of the slot's previous chunk may still be running -- and reading
    // the set the plan before this one was copied into.  (The set before that is free: a slot is re-used only after
    // oswald_hip_chunk_release, which returns when the released chunk's upload has landed, i.e. after the search of the chunk
    // before it.)
    DevBuf items_buf[2], items_q_buf[2];
    int items_cur = 0;
    // ... and their page-locked sources, so that the copy is asynchronous and the host never waits for the copy stream (whatever
    // holds it: with a slot re-used beside a running search the plan's copy has been seen to wait for that search, 110 ms);
    // the launches of the search wait for ev_items on the device instead
    uint2 *items_pin[2] = {nullptr, nullptr};
    size_t items_pin_cap[2] = {0, 0};   // entries (both queues, one behind the other)
    hipEvent_t ev_items = nullptr;
    DevBuf &items_dev() { return items_buf[items_cur]; }
    DevBuf &items_q_dev() { return items_q_buf[items_cur]; }
    const uint16_t *sub_cols_dev() const { return (const uint16_t *)sub_cols_buf.p; }
    std::vector<uint32_t> ncols4_alloc; // host copy, per block
    // host copy of the live extents (see osw_retile16 / osw_block_extent), for the planner: PAGE-LOCKED, so that the copy
    // queued behind the re-tile is asynchronous -- into pageable memory hipMemcpyAsync blocks the caller until the copy has
    // run, i.e. until the re-tile kernel found room on the GPU: an "asynchronous" upload queued beside a running search (the
    // persistent search grid leaves no wave slot free) returned when that search was over, 114 ms later
    uint16_t *sub_cols = nullptr;
    size_t sub_cols_cap = 0;            // entries
    // The same table as the host can tell it from the group lengths alone (every sequence as long as its group: at most 27
    // columns of the reference's x28 padding + the spread inside a group too long): what the planner works with while the
    // chunk's upload is still on its way -- it only ORDERS and SIZES the items by these figures, the kernels read the live
    // extents on the device -- so that a search can be planned and queued behind an upload the host has not waited for.
    std::vector<uint16_t> sub_cols_est;
    bool items_exact = false;           // the item list was planned on the live extents
    std::vector<OswBlock> blocks_host;  // the block table as planned on the host ...
    OswBlock *blocks_pin = nullptr;     // ... and its page-locked copy, the source of the asynchronous upload
    size_t blocks_pin_cap = 0;
    DevBuf st_b, st_n, st_disp;         // the caller's arrays as they arrive on the device (the re-tile kernel's input): per slot,
                                        // so that the copies of the next upload never wait for a re-tile that has not found room yet
    hipEvent_t ev_copy = nullptr;       // recorded on the copy stream behind them
    hipEvent_t ev_down = nullptr;       // recorded on the download stream behind the copy of the chunk's score table to the caller ...
    bool down_pending = false;          // ... which the next search that writes the slot's table must wait for
    uint32_t nitems = 0, nitems_wg = 0;  // wave items / workgroup items of the queue
    uint32_t nitems_q = 0, nitems_q_wg = 0; // the same for the query-pair kernel's queue
    uint64_t items_version = ~0ull;     // query-set version the item list was built for
    int items_bits = 0;                 // cell width it was planned for
    uint32_t max_lg = 0;                // widest geometry in the item list
    uint64_t planned_spill_bytes = 0;   // strip-boundary spill traffic (written + read back) one search of the chunk causes, from the plan
    bool searched = false;
    bool upload_pending = false;        // uploaded with _async: the host has not waited for the upload since
    uint64_t up_seq = 0;                // position of the upload on the device's upload stream
    hipEvent_t ev_up = nullptr;         // recorded on the upload stream behind the chunk's upload
    hipEvent_t ev_use = nullptr;        // recorded on the search stream behind the chunk's last search ...
    bool use_pending = false;           // ... which an upload into the same slot must wait for
    // the chunk's place in the database (oswald_hip_chunk_set_index): database index of its k-th sequence =
    // index_map[k] if a map was given, else first_index + k; nvalid real sequences
    bool has_index = false;
    uint32_t first_index = 0, nvalid = 0;
    std::shared_ptr<const std::vector<uint32_t>> index_map; // host copy: source of the (asynchronous) upload below
    // the index map on the device: two buffers in turn (a slot re-used while its last search -- which reads the old map --
    // is still running gets the other one), copied on the DMA-only copy stream (a copy queued on the search stream would
    // hold the caller until the running search is over: the host copy is pageable); ev_map: the copy has landed
    DevBuf index_map_dev[2];
    int map_cur = 0;
    bool map_pending = false;
    hipEvent_t ev_map = nullptr;
};

struct EventPair { hipEvent_t a, b, c, d; bool c_used; }; // a..b: all DP launches of a chunk search; c..d: the int16 re-run of the 8-bit pass (c_used); d..b: the int32 re-run

struct Device {
    int id = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // second queue: the single-query launch runs beside the query-pair launch
    hipStream_t stream_up = nullptr; // uploads (re-tile): the next chunk comes in while the current one is searched
    hipStream_t stream_copy = nullptr; // ... and the copies of the caller's arrays: DMA only, never queued behind a kernel
    hipStream_t stream_down = nullptr; // score tables on their way to the caller, beside the next chunk's search
    uint64_t up_seq = 0;             // uploads queued so far
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipDeviceProp_t prop;
    uint32_t grid = 0;               // persistent workgroups per launch
    uint32_t grid_q8 = 0;            // ... of the 8-bit kernel (more workgroups per CU; at most 2 x grid: it runs alone and may use both halves of the spill scratch)
    DevBuf queries, qlen, a_disp, prof_off, prof, prof_alt, prof_seq, prof_seq_alt, prof_pair_i16, pair_q, pair_off, pair_len, prof_pair, submat, bnd, counters;
    DevBuf topr_scores, topr_index, topr_cand, wg_times, scores_packed, top_pages, prof_pair8;
    std::vector<uint32_t> top_pages_host; // source of the asynchronous upload of top_pages
    std::vector<std::shared_ptr<const std::vector<uint32_t>>> retired_maps; // index maps of re-used slots whose upload may still be queued
    std::vector<void *> registered;  // caller score tables pinned for an in-flight download (released at the next wait)
    uint64_t bnd_stride = 0;         // spill columns x lanes ({H,F} entries) per wave slot, behind the slot's zero and trash pages
    uint64_t queries_version = ~0ull; // what is currently uploaded
    uint64_t scoring_version = ~0ull;
    std::vector<Chunk> chunks;
    std::vector<EventPair> ev_pool, ev_used;
    // context-level top-r (oswald_hip_topr_begin ... oswald_hip_topr): the device's RUNNING list, [nq][r] tagged keys
    // ((score << 32 | database index) << 1 | 1, 0 = none); every chunk search folds its chunk's r best into it
    // (top_run[top_cur] -> top_run[top_cur ^ 1]); top_gather receives the lists of the other GPUs (RCCL all-gather)
    DevBuf top_run[2], top_gather, top_final;
    int top_cur = 0;
    bool top_any = false;            // a list has been folded in since _begin
    hipEvent_t ev_top = nullptr;     // "the running list is complete" (for a sibling entry on the same GPU)
    // RCCL: one communicator per PHYSICAL GPU of the context, held by the first context device on it (the leader);
    // further context devices on the same GPU (device_ids {0, 0}: a test configuration) hand their lists to the leader
    int leader = -1;                 // index of the first context device on this GPU
    int comm_rank = -1;              // leader: rank in the context's communicator (order of first appearance)
    ncclComm_t comm = nullptr;       // leader, when the context spans more than one GPU
    double dp_ms = 0, rerun16_ms = 0, rerun32_ms = 0;
    uint64_t dp_launches = 0, rerun_items = 0;
};

} // namespace

struct oswald_hip_ctx {
    std::vector<Device> dev;
    Tunables tun;
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
    // query batching: pairs of queries of similar length share a lane (CellPK16Q); the rest run alone
    std::vector<uint32_t> pair_q, pair_off, singles;
    std::vector<uint16_t> pair_len;
    uint32_t pair_rowblocks = 0, pair_max_rowblocks = 0;
    uint64_t queries_version = 0;
    bool profiling = false;
    uint32_t topr_r = 0;             // oswald_hip_topr_begin: every search also selects the chunk's top r (0: off)
    uint64_t topr_queries_version = 0; // the query set the lists are being collected for
    int nphys = 0;                   // distinct GPUs of the context (= ranks of the in-context communicator)
    // process-level communicator (oswald_hip_comm_init_rank): the contexts of several processes, one rank each, held
    // by context device 0; oswald_hip_topr then returns the list of ALL ranks on every rank
    ncclComm_t pcomm = nullptr;
    int pcomm_nranks = 0, pcomm_rank = -1;
    void *top_host = nullptr;        // pinned: the final list on its way to the caller
    size_t top_host_bytes = 0;
};

namespace {

// cell_bits 16 runs the column-frame int16 cell (ArithI16S) with the plain biased cell as its fallback;
// OSWALD_HIP_NO_FRAME=1 (test hook) runs the plain cell only
bool first_pass_is_frame(const oswald_hip_ctx *ctx) { return ctx->cell_bits == 16 && !ctx->tun.no_frame; }

// cell_bits 8: the SWAR 8-bit first pass (CellQ8) runs the query PAIRS; what leaves its 7-bit range is re-run by the
// plain packed-int16 kernel, what reaches that one's ceiling by the int32 kernel.  It needs every profile entry
// S + bias, with bias = -min S, and both gap penalties to be 7-bit values; otherwise the search runs on the int16
// cells alone (the reference's int8 kernels wrap in that case, HybridSearch.c:1520).
int bias8_of(const oswald_hip_ctx *ctx)
{
    int mn = 0, mx = 0;
    for (int i = 0; i < 24 * 32; ++i) { mn = std::min<int>(mn, ctx->submat[i]); mx = std::max<int>(mx, ctx->submat[i]); }
    return (mx - mn <= 127 && ctx->open_gap <= 127 && ctx->extend_gap <= 127) ? -mn : -1;
}
// ... and the cell's offset c = max(open + extend, bias) must leave room for scores (q8_cell.h: CellQ8::offset_for)
int offset8_of(const oswald_hip_ctx *ctx)
{
    const int bias = bias8_of(ctx);
    if (bias < 0) return -1;
    const int c = std::max(ctx->open_gap + ctx->extend_gap, bias);
    return c <= 64 ? c : -1;
}
bool first_pass_is_q8(const oswald_hip_ctx *ctx) { return ctx->cell_bits == 8 && offset8_of(ctx) >= 0; }

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
    HIP_TRY(hipStreamSynchronize(d.stream)); // a call that failed half-way may have left copies from the host arrays queued
    const uint32_t nq = ctx->nq;
    HIP_TRY(d.queries.reserve(ctx->a.size() + 16));
    HIP_TRY(d.qlen.reserve(nq * sizeof(uint16_t) + 16));
    HIP_TRY(d.a_disp.reserve((nq + 1) * sizeof(uint32_t)));
    HIP_TRY(d.prof_off.reserve((nq + 1) * sizeof(uint32_t)));
    HIP_TRY(d.submat.reserve(24 * 32));
    HIP_TRY(d.prof.reserve((size_t)ctx->total_rowblocks * 32 * sizeof(uint2) + 4096));
    HIP_TRY(d.prof_seq.reserve((size_t)ctx->total_rowblocks * 32 * sizeof(uint4) + 4096));
    if (!ctx->a.empty()) HIP_TRY(hipMemcpyAsync(d.queries.p, ctx->a.data(), ctx->a.size(), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.qlen.p, ctx->m.data(), nq * sizeof(uint16_t), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.a_disp.p, ctx->a_disp.data(), nq * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.prof_off.p, ctx->prof_off.data(), nq * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.submat.p, ctx->submat, 24 * 32, hipMemcpyHostToDevice, d.stream));
    // plain integer profile: the exact int32 kernel and the pair profiles read `prof`, the plain single-query int16 cell `prof_seq`
    HIP_TRY(osw_launch_build_profile((const uint8_t *)d.queries.p, (const uint32_t *)d.a_disp.p, (const uint16_t *)d.qlen.p,
                                     (const uint32_t *)d.prof_off.p, (const int8_t *)d.submat.p, nq, ctx->max_rowblocks,
                                     0, (uint2 *)d.prof.p, (uint4 *)d.prof_seq.p, d.stream));
    // the column-frame int16 cell reads S + ge
    const bool alt = first_pass_is_frame(ctx);
    if (alt) {
        HIP_TRY(d.prof_alt.reserve((size_t)ctx->total_rowblocks * 32 * sizeof(uint2) + 4096));
        HIP_TRY(d.prof_seq_alt.reserve((size_t)ctx->total_rowblocks * 32 * sizeof(uint4) + 4096));
        HIP_TRY(osw_launch_build_profile((const uint8_t *)d.queries.p, (const uint32_t *)d.a_disp.p, (const uint16_t *)d.qlen.p,
                                         (const uint32_t *)d.prof_off.p, (const int8_t *)d.submat.p, nq, ctx->max_rowblocks,
                                         ctx->extend_gap, (uint2 *)d.prof_alt.p, (uint4 *)d.prof_seq_alt.p, d.stream));
    }
    // constant "row above a first round": 64 {H,F} entries of zeros, 64 of the biased-int16 floor (1024), then the
    // column-frame cell's floor table, entry k = 1024 + k * ge (capped below the fp16 inf pattern), then the 8-bit cell's page.  Uploaded on the
    // device's stream like everything else here (ordered behind a search still in flight).
    std::vector<uint32_t> &pages = d.top_pages_host; // a member: an early return must not free the source of a queued copy
    pages.assign((128 + OSW_I16S_TABLE + 64) * 2, 0u);
    for (size_t i = 64; i < 128; ++i) pages[2 * i] = pages[2 * i + 1] = 0x04000400u;
    // ... and behind the table 64 entries of the 8-bit cell's "zero": its offset c in every byte (q8_cell.h)
    if (first_pass_is_q8(ctx))
        for (size_t i = 128 + OSW_I16S_TABLE; i < 128 + OSW_I16S_TABLE + 64; ++i) pages[2 * i] = pages[2 * i + 1] = (uint32_t)offset8_of(ctx) * 0x01010101u;
    for (size_t k = 0; k < OSW_I16S_TABLE; ++k) {
        const uint32_t v = (uint32_t)std::min<uint64_t>(1024ull + (uint64_t)k * (uint64_t)ctx->extend_gap, 0x7bffull);
        pages[2 * (128 + k)] = pages[2 * (128 + k) + 1] = v | (v << 16);
    }
    HIP_TRY(d.top_pages.reserve(pages.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(d.top_pages.p, pages.data(), pages.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
    const uint32_t np = (uint32_t)ctx->pair_len.size();
    if (np > 0) {
        HIP_TRY(d.pair_q.reserve(2 * np * sizeof(uint32_t)));
        HIP_TRY(d.pair_off.reserve(np * sizeof(uint32_t)));
        HIP_TRY(d.pair_len.reserve(np * sizeof(uint16_t) + 16));
        HIP_TRY(d.prof_pair.reserve((size_t)ctx->pair_rowblocks * 32 * sizeof(uint4) + 4096));
        HIP_TRY(hipMemcpyAsync(d.pair_q.p, ctx->pair_q.data(), 2 * np * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.pair_off.p, ctx->pair_off.data(), np * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.pair_len.p, ctx->pair_len.data(), np * sizeof(uint16_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(osw_launch_build_pair_profile((const uint2 *)(alt ? d.prof_alt.p : d.prof.p), (const uint32_t *)d.prof_off.p, (const uint16_t *)d.qlen.p,
                                              (const uint32_t *)d.pair_q.p, (const uint32_t *)d.pair_off.p, (const uint16_t *)d.pair_len.p, np,
                                              ctx->pair_max_rowblocks, alt /* column-frame pair cell: 32-bit integer sums */, (uint4 *)d.prof_pair.p, d.stream));
        if (first_pass_is_q8(ctx)) {
            HIP_TRY(d.prof_pair8.reserve((size_t)ctx->pair_rowblocks * 32 * sizeof(uint2) + 4096));
            HIP_TRY(osw_launch_build_pair_profile8((const uint2 *)d.prof.p, (const uint32_t *)d.prof_off.p, (const uint16_t *)d.qlen.p,
                                                   (const uint32_t *)d.pair_q.p, (const uint32_t *)d.pair_off.p, (const uint16_t *)d.pair_len.p, np,
                                                   ctx->pair_max_rowblocks, bias8_of(ctx), (uint2 *)d.prof_pair8.p, d.stream));
        }
        if (alt) { // plain int16 pair profile for the items the first-pass cell hands to the plain cell
            HIP_TRY(d.prof_pair_i16.reserve((size_t)ctx->pair_rowblocks * 32 * sizeof(uint4) + 4096));
            HIP_TRY(osw_launch_build_pair_profile((const uint2 *)d.prof.p, (const uint32_t *)d.prof_off.p, (const uint16_t *)d.qlen.p,
                                                  (const uint32_t *)d.pair_q.p, (const uint32_t *)d.pair_off.p, (const uint16_t *)d.pair_len.p, np,
                                                  ctx->pair_max_rowblocks, false, (uint4 *)d.prof_pair_i16.p, d.stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(d.stream)); // host vectors may change after return
    d.queries_version = ctx->queries_version;
    d.scoring_version = ctx->scoring_version;
    return 0;
}

// plan_pairs(), build_items(): the planner (a file of its own: see its header)
#include "osw_planner.inc"

// after the stream has been synchronised: the downloads are done, unpin the callers' tables
void release_registered(Device &d)
{
    for (void *p : d.registered) (void)hipHostUnregister(p);
    d.registered.clear();
    d.retired_maps.clear();
}

// Strip-boundary spill scratch: one region per resident wave and launch (two launches run side by side), every region
// a reserved page, a trash page and (columns + pad) x 32 {H,F} entries: the boundary row of the longest block at
// two lane groups (the planner runs longer blocks, or G = 1 beyond half of it, at a geometry with fewer lanes per
// group).  Sized from the longest sequence the device is asked to hold (at most 4096 columns' worth) and grown when
// a later chunk needs more; a multi-GB hipMalloc occasionally takes ~200 ms, which is why oswald_hip_reserve exists.
int ensure_scratch(Device &d, uint32_t max_cols)
{
    const uint64_t cols = std::min<uint64_t>(std::max<uint64_t>(max_cols, 1024), 4096);
    const uint64_t stride = (cols + OSW_SCRATCH_PAD_COLS) * 32u;
    if (stride <= d.bnd_stride && d.bnd.p) return 0;
    HIP_TRY(hipStreamSynchronize(d.stream)); // nothing may still be spilling into the old regions
    HIP_TRY(hipStreamSynchronize(d.stream2));
    const uint64_t slots = (uint64_t)d.grid * (OSW_WG_THREADS / 64);
    // the larger region is allocated BESIDE the old one and swapped in on success: if the allocation fails the old
    // scratch, its stride and the plans made for it all stay valid (the caller gets OSWALD_HIP_ENOMEM)
    DevBuf bigger;
    HIP_TRY(bigger.reserve(2 * slots * (stride + OSW_SCRATCH_DATA) * sizeof(uint2)));
    if (hipError_t e = hipMemset2DAsync(bigger.p, (stride + OSW_SCRATCH_DATA) * sizeof(uint2), 0, OSW_SCRATCH_DATA * sizeof(uint2), 2 * slots, d.stream)) {
        bigger.release();
        return fail(OSWALD_HIP_ERUNTIME, "hipMemset2DAsync(spill scratch): %s", hipGetErrorString(e));
    }
    for (Chunk &c : d.chunks) c.items_version = ~0ull; // the plans were made for the old region size
    d.bnd.release();
    d.bnd = bigger;
    d.bnd_stride = stride;
    return 0;
}

// An upload queued with oswald_hip_chunk_upload_async has landed once its event on the upload stream has: the host
// waits for THAT, not for the search stream -- a search of another chunk may be running meanwhile.
int finish_upload(Device &d, Chunk &c)
{
    if (!c.upload_pending) return 0;
    HIP_TRY(hipEventSynchronize(c.ev_up));
    for (Chunk &k : d.chunks) if (k.up_seq <= c.up_seq) k.upload_pending = false; // one in-order stream: everything queued before is done too
    return 0;
}

void drain_events(Device &d)
{
    if (g_debug_slow && d.ev_used.size() > 1) { // (OSWALD_HIP_DEBUG_SLOW: how the searches of a pass lie on the device's time line)
        for (size_t k = 0; k < d.ev_used.size(); ++k) {
            float dur = 0, gap = 0;
            (void)hipEventElapsedTime(&dur, d.ev_used[k].a, d.ev_used[k].b);
            if (k + 1 < d.ev_used.size()) (void)hipEventElapsedTime(&gap, d.ev_used[k].b, d.ev_used[k + 1].a);
            fprintf(stderr, "[oswald_hip] search %zu: %.3f ms on the device, %.3f ms to the next search's start\n", k, dur, gap);
        }
    }
    for (auto &e : d.ev_used) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { d.dp_ms += ms; d.dp_launches++; }
        if (e.c_used && hipEventElapsedTime(&ms, e.c, e.d) == hipSuccess) d.rerun16_ms += ms;
        if (hipEventElapsedTime(&ms, e.d, e.b) == hipSuccess) d.rerun32_ms += ms;
        d.ev_pool.push_back(e);
    }
    d.ev_used.clear();
}

int queue_topr(oswald_hip_ctx *ctx, Device &d, Chunk &c, uint32_t nvalid, uint32_t r)
{
    const size_t cnt = (size_t)ctx->nq * r;
    HIP_TRY(d.topr_scores.reserve(cnt * sizeof(int32_t)));
    HIP_TRY(d.topr_index.reserve(cnt * sizeof(uint32_t)));
    HIP_TRY(d.topr_cand.reserve((size_t)ctx->nq * osw_topr_parts(nvalid) * r * sizeof(unsigned long long)));
    HIP_TRY(osw_launch_topr((const int32_t *)c.scores.p, c.score_stride, nvalid, r, ctx->nq, (unsigned long long *)d.topr_cand.p, (int32_t *)d.topr_scores.p,
                            (uint32_t *)d.topr_index.p, d.stream));
    return 0;
}

// Context-level top-r (oswald_hip_topr_begin): select the top r of the chunk just searched on its device, as tagged
// DATABASE keys, and fold them into the device's running list -- all queued on the device's stream behind the search:
// the caller is not made to wait, nothing leaves the device, and the chunk may be released right after.
int topr_after_search(oswald_hip_ctx *ctx, Device &d, Chunk &c)
{
    if (ctx->topr_r == 0 || !c.has_index || ctx->nq == 0) return 0;
    const uint32_t r = ctx->topr_r;
    if (ctx->topr_queries_version != ctx->queries_version)
        return fail(OSWALD_HIP_ESTATE, "the query set changed since oswald_hip_topr_begin: call it again before searching");
    if (c.nvalid > c.ngroups * c.W) return fail(OSWALD_HIP_EINVAL, "chunk index: nvalid %u exceeds the chunk's %u lanes", c.nvalid, c.ngroups * c.W);
    if (c.nitems + c.nitems_wg + c.nitems_q + c.nitems_q_wg == 0 || c.nvalid == 0) return 0; // nothing was searched: nothing to add
    if (!d.top_run[0].p || !d.top_run[1].p) return fail(OSWALD_HIP_ESTATE, "oswald_hip_topr_begin has not prepared device %d", d.id);
    if (c.map_pending) { HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_map, 0)); c.map_pending = false; } // the chunk's index map has landed
    HIP_TRY(d.topr_cand.reserve((size_t)ctx->nq * osw_topr_parts(c.nvalid) * r * sizeof(unsigned long long)));
    HIP_TRY(osw_launch_topr_fold_chunk((const int32_t *)c.scores.p, c.score_stride, c.nvalid, r, ctx->nq,
                                       c.index_map ? (const uint32_t *)c.index_map_dev[c.map_cur].p : nullptr, c.first_index, (unsigned long long *)d.topr_cand.p,
                                       (const unsigned long long *)d.top_run[d.top_cur].p, (unsigned long long *)d.top_run[d.top_cur ^ 1].p, d.stream));
    d.top_cur ^= 1;
    d.top_any = true;
    return 0;
}

// THE merge of top lists (the reference's order, utils.c:3-86: descending score, equal scores by DESCENDING database
// index = descending key score << 32 | index): K candidates per query, score < 0 = empty slot, -> the r best.
void merge_candidates(uint32_t nq, size_t K, const int32_t *cs, const uint32_t *ci, uint32_t r, int32_t *out_s, uint32_t *out_i)
{
    std::vector<uint64_t> keys;
    for (uint32_t q = 0; q < nq; ++q) {
        keys.clear();
        for (size_t k = 0; k < K; ++k)
            if (cs[q * K + k] >= 0) keys.push_back(((uint64_t)(uint32_t)cs[q * K + k] << 32) | ci[q * K + k]);
        const size_t take = std::min<size_t>(r, keys.size());
        std::partial_sort(keys.begin(), keys.begin() + take, keys.end(), std::greater<uint64_t>());
        for (size_t j = 0; j < r; ++j) {
            out_s[(size_t)q * r + j] = j < take ? (int32_t)(keys[j] >> 32) : -1;
            out_i[(size_t)q * r + j] = j < take ? (uint32_t)(keys[j] & 0xffffffffu) : 0xffffffffu;
        }
    }
}

} // namespace

extern "C" {

int oswald_hip_abi_version(void) { return OSWALD_HIP_ABI_VERSION; }

const char *oswald_hip_last_error(void) { return g_err.c_str(); }

// Page-locked host memory for the caller's chunk buffers and score tables.
int oswald_hip_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr) return fail(OSWALD_HIP_EINVAL, "null out-pointer");
    *ptr = nullptr;
    if (bytes == 0) return 0;
    HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocPortable));
    return 0;
}

int oswald_hip_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return 0;
}

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
    ctx->tun.refresh();
    ctx->dev.resize(ndev);
    for (int i = 0; i < ndev; ++i) {
        Device &d = ctx->dev[i];
        d.id = device_ids ? device_ids[i] : i;
        if (d.id < 0 || d.id >= have) { delete ctx; return fail(OSWALD_HIP_ENODEV, "device %d requested, %d visible", d.id, have); }
        hipError_t r = hipSetDevice(d.id);
        if (r == hipSuccess) r = hipGetDeviceProperties(&d.prop, d.id);
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&d.stream2, hipStreamNonBlocking);
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&d.stream_up, hipStreamNonBlocking);
        // The two DMA-only streams sit in priority classes of their own -- copy: highest, download: lowest.  Not for the priority
        // (no kernel runs on either): the runtime maps the streams of a class onto a small pool of hardware queues
        // (GPU_MAX_HW_QUEUES, 4 by default, shared with whatever streams the caller's process has), and a copy on a stream that
        // shares its hardware queue with the search stream starts only when the search -- a persistent grid that holds its queue for
        // the whole launch -- has ended.  As the fourth ordinary stream of the process the copy stream did share the search stream's
        // queue (tools/xfer_overlap.hip, profiles/r04_xfer_overlap.txt: 128 MB beside a 60 ms kernel: done after 2.4 ms on a queue of
        // its own, after 60 ms on the shared one).  Two classes, not one: the download stream WAITS for searches (a table leaves
        // behind its search), and a wait at the head of a hardware queue holds whatever else shares that queue.
        int prio_least = 0, prio_greatest = 0;
        if (r == hipSuccess) r = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (ctx->tun.no_stream_classes) prio_least = prio_greatest = 0; // OSWALD_HIP_NO_STREAM_CLASSES=1 (A/B hook): ordinary streams
        if (r == hipSuccess) r = hipStreamCreateWithPriority(&d.stream_copy, hipStreamNonBlocking, prio_greatest);
        if (r == hipSuccess) r = hipStreamCreateWithPriority(&d.stream_down, hipStreamNonBlocking, prio_least);
        if (g_debug_slow) fprintf(stderr, "[oswald_hip] stream priorities: least %d, greatest %d\n", prio_least, prio_greatest);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&d.ev_fork, hipEventDisableTiming);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&d.ev_join, hipEventDisableTiming);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&d.ev_top, hipEventDisableTiming);
        int per_cu = 0;
        if (r == hipSuccess) r = (hipError_t)osw_occupancy_pk16(&per_cu);
        if (r != hipSuccess) { delete ctx; return fail(OSWALD_HIP_ERUNTIME, "bring-up of device %d failed: %s", d.id, hipGetErrorString(r)); }
        if (per_cu < 1) per_cu = 1;
        d.grid = (uint32_t)d.prop.multiProcessorCount * (uint32_t)per_cu;
        int per_cu8 = 0;
        if (osw_occupancy_q8(&per_cu8) != (int)hipSuccess || per_cu8 < 1) per_cu8 = per_cu;
        d.grid_q8 = std::min<uint32_t>(2 * d.grid, (uint32_t)d.prop.multiProcessorCount * (uint32_t)per_cu8);
        r = d.counters.reserve((OSW_CTR_BLOCKS * OSW_CTR_COUNT + 8) * sizeof(uint32_t));
        if (r != hipSuccess) { delete ctx; return fail(OSWALD_HIP_ENOMEM, "device %d: %s", d.id, hipGetErrorString(r)); }
        // Bring-up costs that would otherwise land in the first search (the reference times its searches after
        // init(), main.c:46 / FPGAsearch.c:80): the runtime's staging for copies from / to pageable memory (the first
        // copy of a process takes ~10 ms, later ones run at ~20 GB/s) and the first launch of every kernel.
        {
            std::vector<char> tmp(16u << 20, 1);
            DevBuf scratch;
            r = scratch.reserve(tmp.size());
            if (r == hipSuccess) r = hipMemcpyAsync(scratch.p, tmp.data(), tmp.size(), hipMemcpyHostToDevice, d.stream);
            if (r == hipSuccess) r = hipMemcpyAsync(tmp.data(), scratch.p, tmp.size(), hipMemcpyDeviceToHost, d.stream);
            // (the chunks arrive on the copy stream, score tables leave on the download stream: their first copies too)
            for (int rep = 0; rep < 4 && r == hipSuccess; ++rep) r = hipMemcpyAsync(scratch.p, tmp.data(), tmp.size(), hipMemcpyHostToDevice, d.stream_copy);
            if (r == hipSuccess) r = hipStreamSynchronize(d.stream_copy);
            if (r == hipSuccess) r = hipMemcpyAsync(tmp.data(), scratch.p, tmp.size(), hipMemcpyDeviceToHost, d.stream_down);
            if (r == hipSuccess) r = hipStreamSynchronize(d.stream_down);
            if (r == hipSuccess) r = hipMemsetAsync(d.counters.p, 0, (OSW_CTR_BLOCKS * OSW_CTR_COUNT + 8) * sizeof(uint32_t), d.stream);
            OswSearchArgs a;
            memset(&a, 0, sizeof a); // empty queues: every wave leaves at once
            a.counters = (uint32_t *)d.counters.p;
            a.counters_ovf = (uint32_t *)d.counters.p + OSW_CTR_BLOCKS * OSW_CTR_COUNT;
            if (r == hipSuccess) r = osw_launch_s16q(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_s16(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_pk16q(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_pk16(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_i32(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_i32r(a, 8, d.stream);
            if (r == hipSuccess) r = osw_launch_q8(a, 1, d.stream);
            if (r == hipSuccess) r = hipStreamSynchronize(d.stream);
            scratch.release();
            if (r != hipSuccess) { delete ctx; return fail(OSWALD_HIP_ERUNTIME, "warm-up of device %d failed: %s", d.id, hipGetErrorString(r)); }
        }
    }
    // one RCCL rank per distinct GPU of the context, in order of first appearance; a context on one GPU needs none
    for (int i = 0; i < ndev; ++i) {
        Device &d = ctx->dev[i];
        for (int k = 0; k < i && d.leader < 0; ++k) if (ctx->dev[k].id == d.id) d.leader = k;
        if (d.leader < 0) { d.leader = i; d.comm_rank = ctx->nphys++; }
    }
    if (ctx->nphys > 1) {
        std::vector<int> ids;
        std::vector<ncclComm_t> comms(ctx->nphys, nullptr);
        for (const Device &d : ctx->dev) if (d.comm_rank >= 0) ids.push_back(d.id);
        const ncclResult_t nr = ncclCommInitAll(comms.data(), ctx->nphys, ids.data());
        if (nr != ncclSuccess) {
            oswald_hip_finalize(ctx);
            return fail(OSWALD_HIP_ECOMM, "ncclCommInitAll over %d GPUs failed: %s (the top-r gather of a multi-GPU context runs over RCCL; there is no other path)",
                        ctx->nphys, ncclGetErrorString(nr));
        }
        for (Device &d : ctx->dev) if (d.comm_rank >= 0) d.comm = comms[d.comm_rank];
    }
    *out = ctx;
    return 0;
}

int oswald_hip_finalize(oswald_hip_ctx *ctx)
{
    if (!ctx) return 0;
    for (Device &d : ctx->dev) {
        (void)hipSetDevice(d.id);
        for (hipStream_t st : {d.stream, d.stream2, d.stream_copy, d.stream_up, d.stream_down}) if (st) (void)hipStreamSynchronize(st);
        release_registered(d);
        if (d.stream2) (void)hipStreamSynchronize(d.stream2);
        if (d.stream_copy) { (void)hipStreamSynchronize(d.stream_copy); (void)hipStreamDestroy(d.stream_copy); d.stream_copy = nullptr; }
        if (d.stream_down) { (void)hipStreamSynchronize(d.stream_down); (void)hipStreamDestroy(d.stream_down); d.stream_down = nullptr; }
        if (d.stream_up) { (void)hipStreamSynchronize(d.stream_up); (void)hipStreamDestroy(d.stream_up); d.stream_up = nullptr; }
        for (Chunk &c : d.chunks) {
            if (c.ev_up) (void)hipEventDestroy(c.ev_up);
            if (c.ev_use) (void)hipEventDestroy(c.ev_use);
            if (c.ev_copy) (void)hipEventDestroy(c.ev_copy);
            if (c.ev_down) (void)hipEventDestroy(c.ev_down);
            c.ev_down = nullptr;
            if (c.sub_cols) (void)hipHostFree(c.sub_cols);
            for (int k = 0; k < 2; ++k) { if (c.items_pin[k]) (void)hipHostFree(c.items_pin[k]); c.items_pin[k] = nullptr; c.items_pin_cap[k] = 0; }
            if (c.ev_items) (void)hipEventDestroy(c.ev_items);
            c.ev_items = nullptr;
            if (c.blocks_pin) (void)hipHostFree(c.blocks_pin);
            c.ev_up = c.ev_use = c.ev_copy = nullptr;
            c.sub_cols = nullptr;
            c.sub_cols_cap = 0;
            c.blocks_pin = nullptr;
            c.blocks_pin_cap = 0;
            c.st_b.release(); c.st_n.release(); c.st_disp.release();
        }
        if (d.comm) { (void)ncclCommDestroy(d.comm); d.comm = nullptr; }
        if (&d == &ctx->dev[0] && ctx->pcomm) { (void)ncclCommDestroy(ctx->pcomm); ctx->pcomm = nullptr; }
        for (Chunk &c : d.chunks) { c.tiled.release(); c.blocks.release(); c.sub_cols_buf.release(); for (int k = 0; k < 2; ++k) { c.items_buf[k].release(); c.items_q_buf[k].release(); } c.scores.release(); c.ovf.release(); c.ovf8.release(); c.index_map_dev[0].release(); c.index_map_dev[1].release(); if (c.ev_map) (void)hipEventDestroy(c.ev_map); c.ev_map = nullptr; }
        for (DevBuf *b : {&d.queries, &d.qlen, &d.a_disp, &d.prof_off, &d.prof, &d.prof_alt, &d.prof_seq, &d.prof_seq_alt, &d.prof_pair_i16, &d.pair_q, &d.pair_off, &d.pair_len, &d.prof_pair, &d.submat, &d.bnd, &d.counters,
                          &d.topr_scores, &d.topr_index, &d.topr_cand, &d.wg_times, &d.scores_packed, &d.top_pages, &d.prof_pair8,
                          &d.top_run[0], &d.top_run[1], &d.top_gather, &d.top_final})
            b->release();
        if (d.ev_top) (void)hipEventDestroy(d.ev_top);
        drain_events(d);
        for (auto &e : d.ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); (void)hipEventDestroy(e.c); (void)hipEventDestroy(e.d); }
        if (d.stream2) { (void)hipStreamSynchronize(d.stream2); (void)hipStreamDestroy(d.stream2); }
        if (d.ev_fork) (void)hipEventDestroy(d.ev_fork);
        if (d.ev_join) (void)hipEventDestroy(d.ev_join);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    if (ctx->top_host) (void)hipHostFree(ctx->top_host);
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
    // The reference's matrices are 24 rows x 32 columns with zeros in column 23 (the dummy residue) and in the padding columns 24..31
    // (host/src/submat.c), and its preprocessing emits the codes 0..23 only.  The single-query kernels keep 24 entries per profile
    // row-block and the re-tile kernels store a residue code >= 24 as 23: exact for every matrix whose padding columns equal column 23.
    for (int i = 0; i < 24; ++i)
        for (int j = 24; j < 32; ++j)
            if (submat[i * 32 + j] != submat[i * 32 + 23])
                return fail(OSWALD_HIP_EINVAL, "substitution matrix: column %d differs from column 23 in row %d (the padding columns 24..31 must repeat the dummy residue's column, as in the reference's matrices)", j, i);
    ctx->tun.refresh();
    if (cell_bits == 0) cell_bits = ctx->tun.cell_bits_default;
    if (cell_bits != 8 && cell_bits != 16 && cell_bits != 32) return fail(OSWALD_HIP_EINVAL, "cell_bits must be 0 (default), 8, 16 or 32");
    const bool was_q8 = ctx->have_scoring && first_pass_is_q8(ctx);
    memcpy(ctx->submat, submat, 24 * 32);
    ctx->open_gap = open_gap;
    ctx->extend_gap = extend_gap;
    bool repair = ctx->have_queries && cell_bits != ctx->cell_bits; // the pairing rule depends on the arithmetic
    ctx->cell_bits = cell_bits;
    ctx->have_scoring = true;
    ctx->scoring_version++;
    if (ctx->have_queries && was_q8 != first_pass_is_q8(ctx)) repair = true; // (eligibility depends on the matrix and the penalties)
    if (repair) { plan_pairs(ctx); ctx->queries_version++; }
    return 0;
}

int oswald_hip_set_queries(oswald_hip_ctx *ctx, const uint8_t *a, uint64_t Q, const uint16_t *m, const uint32_t *a_disp, uint32_t nq)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (nq > 0 && (!m || !a_disp || (Q > 0 && !a))) return fail(OSWALD_HIP_EINVAL, "null query arrays");
    if (nq > 65535) return fail(OSWALD_HIP_EINVAL, "at most 65535 queries per set (work items carry a 16-bit query index); search in several sets");
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
    ctx->tun.refresh();
    plan_pairs(ctx);
    ctx->have_queries = true;
    ctx->queries_version++;
    return 0;
}

static int chunk_upload_impl(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                             uint32_t ngroups, uint32_t W, int *chunk, bool async)
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
    // A free slot whose buffers are large enough already (the smallest such), else a new one: growing a buffer frees the
    // old one, and hipFree waits for the device -- i.e. for whatever search is running beside this upload.
    int slot = -1;
    for (size_t i = 0; i < d.chunks.size(); ++i) {
        const Chunk &k = d.chunks[i];
        if (k.live || k.upload_pending || k.st_b.cap < vD + 64) continue;
        const size_t need_scores = (size_t)ctx->nq * ((ngroups + OSW_BLOCK_SEQS / W - 1) / (OSW_BLOCK_SEQS / W)) * OSW_BLOCK_SEQS * sizeof(int32_t);
        if (k.scores.cap && k.scores.cap < need_scores + 16) continue; // (its score table and re-run queues would have to grow too)
        if (slot < 0 || k.st_b.cap < d.chunks[slot].st_b.cap) slot = (int)i;
    }
    if (slot < 0)
        for (size_t i = 0; i < d.chunks.size(); ++i) if (!d.chunks[i].live && !d.chunks[i].upload_pending && !d.chunks[i].st_b.p) { slot = (int)i; break; } // (never used)
    if (slot < 0) { d.chunks.emplace_back(); slot = (int)d.chunks.size() - 1; }
    Chunk &c = d.chunks[slot];
    PhaseTimer pt(ctx->tun.debug_phases);
    const uint32_t gpb = OSW_BLOCK_SEQS / W;
    c.ngroups = ngroups;
    c.W = W;
    c.nblocks = (ngroups + gpb - 1) / gpb;
    c.score_stride = c.nblocks * OSW_BLOCK_SEQS;
    c.ncols4_alloc.assign(c.nblocks, 0);
    std::vector<OswBlock> &blocks = c.blocks_host; // (a member: oswald_hip_chunk_upload_async returns before the copy has run)
    blocks.assign(c.nblocks, OswBlock{});
    uint64_t off = OSW_TILED_PAD_GROUPS; // all-dummy columns in front of the first block
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
        off += (uint64_t)nc4 + OSW_TILED_PAD_GROUPS; // + the all-dummy groups the kernels prefetch / drain through
    }
    c.total_col4 = off;
    // live extents as far as the group lengths tell them (sub_cols_est): lane l of a block holds its sequences 2l, 2l+1
    c.sub_cols_est.assign((size_t)c.nblocks * 128, 0);
    for (uint32_t B = 0; B < c.nblocks; ++B) {
        uint16_t lane_n[64];
        for (uint32_t l = 0; l < 64; ++l) {
            const uint32_t g = B * gpb + (2 * l) / W;
            lane_n[l] = g < ngroups ? n[g] : 0;
        }
        uint16_t *e = c.sub_cols_est.data() + (size_t)B * 128;
        for (uint32_t t = 0; t < 127; ++t) {
            const uint32_t lg = 31u - (uint32_t)__builtin_clz(t + 1u), sigma = t + 1u - (1u << lg), gl = 64u >> lg;
            uint16_t mx = 0;
            for (uint32_t k = 0; k < gl; ++k) mx = std::max(mx, lane_n[sigma * gl + k]);
            e[t] = mx;
        }
    }
    if (int r = ensure_scratch(d, c.max_ncols4 * 4)) return r;
    HIP_TRY(c.tiled.reserve((off + OSW_TILED_TAIL_GROUPS) * 64 * sizeof(uint2)));
    HIP_TRY(c.blocks.reserve(c.nblocks * sizeof(OswBlock) + 16));
    HIP_TRY(c.sub_cols_buf.reserve((size_t)c.nblocks * 128 * sizeof(uint16_t) + 16));
    if ((size_t)c.nblocks * 128 > c.sub_cols_cap) {
        if (c.sub_cols) HIP_TRY(hipHostFree(c.sub_cols));
        c.sub_cols = nullptr;
        c.sub_cols_cap = 0;
        const size_t want = (size_t)c.nblocks * 128 + (size_t)c.nblocks * 16 + 128;
        HIP_TRY(hipHostMalloc((void **)&c.sub_cols, want * sizeof(uint16_t), hipHostMallocPortable));
        c.sub_cols_cap = want;
    }
    HIP_TRY(c.st_b.reserve(vD + 64));
    HIP_TRY(c.st_n.reserve(ngroups * sizeof(uint16_t) + 16));
    HIP_TRY(c.st_disp.reserve(ngroups * sizeof(uint32_t) + 16));
    if (c.nblocks > c.blocks_pin_cap) {
        if (c.blocks_pin) HIP_TRY(hipHostFree(c.blocks_pin));
        c.blocks_pin = nullptr;
        c.blocks_pin_cap = 0;
        const size_t want = (size_t)c.nblocks + c.nblocks / 8 + 16;
        HIP_TRY(hipHostMalloc((void **)&c.blocks_pin, want * sizeof(OswBlock), hipHostMallocPortable));
        c.blocks_pin_cap = want;
    }
    if (c.nblocks) memcpy(c.blocks_pin, blocks.data(), c.nblocks * sizeof(OswBlock));
    if (!c.ev_up) HIP_TRY(hipEventCreateWithFlags(&c.ev_up, hipEventDisableTiming));
    if (!c.ev_use) HIP_TRY(hipEventCreateWithFlags(&c.ev_use, hipEventDisableTiming));
    if (!c.ev_copy) HIP_TRY(hipEventCreateWithFlags(&c.ev_copy, hipEventDisableTiming));
    if (!c.ev_down) HIP_TRY(hipEventCreateWithFlags(&c.ev_down, hipEventDisableTiming));
    pt.lap("upload: plan + allocations");
    // Uploads have streams of their own: chunk k+1 comes in while chunk k is searched.  The copies of the caller's arrays
    // go into the slot's own staging buffers on a stream that carries nothing but DMA -- they start at once, whatever the
    // GPU is computing.  The re-tile (a kernel) goes on the upload stream behind them; the persistent grid of a running
    // search leaves it no wave slot, so it runs when that search drains -- by then the host has long returned (every
    // host-side source / destination of this stream is page-locked: an asynchronous copy from or into pageable memory
    // would hold the caller until it has run).  The slot may still be in use by the search of the chunk it held before
    // (oswald_hip_search_chunk_async, or a release right behind a search): the upload stream waits for that search.
    hipStream_t up = d.stream_up;
    if (c.use_pending) { HIP_TRY(hipStreamWaitEvent(up, c.ev_use, 0)); c.use_pending = false; }
    if (ngroups > 0) {
        HIP_TRY(hipMemcpyAsync(c.st_b.p, b, vD, hipMemcpyHostToDevice, d.stream_copy));
        HIP_TRY(hipMemcpyAsync(c.st_n.p, n, ngroups * sizeof(uint16_t), hipMemcpyHostToDevice, d.stream_copy));
        HIP_TRY(hipMemcpyAsync(c.st_disp.p, disp, ngroups * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream_copy));
        HIP_TRY(hipEventRecord(c.ev_copy, d.stream_copy));
        HIP_TRY(hipStreamWaitEvent(up, c.ev_copy, 0));
        HIP_TRY(hipMemcpyAsync(c.blocks.p, c.blocks_pin, c.nblocks * sizeof(OswBlock), hipMemcpyHostToDevice, up));
        HIP_TRY(osw_launch_fill(c.tiled.p, OSW_DUMMY_CODE8, (off + OSW_TILED_TAIL_GROUPS) * 64 * sizeof(uint2), up)); // pads = dummy residue
        HIP_TRY(osw_launch_retile((const uint8_t *)c.st_b.p, (const uint16_t *)c.st_n.p, (const uint32_t *)c.st_disp.p,
                                  ngroups, W, (OswBlock *)c.blocks.p, c.nblocks, (uint16_t *)c.tiled.p, (uint16_t *)c.sub_cols_buf.p, up));
    }
    if (pt.on) { HIP_TRY(hipStreamSynchronize(up)); pt.lap("upload: H2D + re-tile"); }
    if (c.nblocks) HIP_TRY(hipMemcpyAsync(c.sub_cols, c.sub_cols_dev(), (size_t)c.nblocks * 128 * sizeof(uint16_t), hipMemcpyDeviceToHost, up));
    HIP_TRY(hipEventRecord(c.ev_up, up));
    c.up_seq = ++d.up_seq;
    c.items_version = ~0ull;
    c.searched = false;
    c.has_index = false;
    if (c.index_map) d.retired_maps.push_back(std::move(c.index_map)); // (its copy to the device may still be queued: freed at the next synchronisation)
    c.index_map.reset();
    c.live = true;
    c.upload_pending = true;
    *chunk = slot;
    if (!async) {
        if (int r = finish_upload(d, c)) return r; // caller's buffers are free again (reference: clFinish, FPGAsearch.c:197)
        pt.lap("upload: sync");
    }
    return 0;
}

int oswald_hip_chunk_upload(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                            uint32_t ngroups, uint32_t W, int *chunk)
{
    return chunk_upload_impl(ctx, dev, b, vD, n, disp, ngroups, W, chunk, false);
}

int oswald_hip_chunk_upload_async(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                                  uint32_t ngroups, uint32_t W, int *chunk)
{
    return chunk_upload_impl(ctx, dev, b, vD, n, disp, ngroups, W, chunk, true);
}

int oswald_hip_reserve(oswald_hip_ctx *ctx, int dev, uint32_t max_sequence_length)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (dev >= (int)ctx->dev.size()) return fail(OSWALD_HIP_ENODEV, "device index %d out of range", dev);
    for (int i = 0; i < (int)ctx->dev.size(); ++i) {
        if (dev >= 0 && i != dev) continue;
        Device &d = ctx->dev[i];
        HIP_TRY(hipSetDevice(d.id));
        if (int r = ensure_scratch(d, max_sequence_length + 28)) return r; // + the padding of the group lengths
        HIP_TRY(hipStreamSynchronize(d.stream));
    }
    return 0;
}

// Device memory one byte of chunk (one padded residue of the interleaved groups) takes, worst case, for each of the
// three chunks a device holds while one is searched and the next two come in: the staging copy of the upload (1), the
// re-tiled residues (a 128-sequence block is padded to its longest group: <= 1.25), the all-dummy columns behind every
// block (18 x 512 B per block of >= 128 x 28 B: 2.6), and per sequence -- at most one per 28 bytes, the shortest
// padded group length -- 4 B of score, 8 B of int32 re-run queue and 8 B of int16 re-run queue per query.
int oswald_hip_max_chunk_size(oswald_hip_ctx *ctx, int dev, uint32_t nq, uint32_t max_sequence_length, uint64_t *bytes)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!bytes) return fail(OSWALD_HIP_EINVAL, "null output");
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (ctx->tun.fake_free_mem) free_b = std::min<size_t>(free_b, ctx->tun.fake_free_mem);
    // the spill scratch of the longest sequence, if it is not there yet (ensure_scratch)
    const uint64_t cols = std::min<uint64_t>(std::max<uint64_t>((uint64_t)max_sequence_length + 28, 1024), 4096);
    const uint64_t stride = (cols + OSW_SCRATCH_PAD_COLS) * 32u;
    const uint64_t scratch = 2ull * d.grid * (OSW_WG_THREADS / 64) * (stride + OSW_SCRATCH_DATA) * sizeof(uint2);
    uint64_t usable = (uint64_t)(0.8 * (double)free_b);
    if (stride > d.bnd_stride || !d.bnd.p) usable = usable > scratch ? usable - scratch : 0;
    const double per_byte = 3.0 * (1.0 + 1.25 + 2.6 + 20.0 * (double)std::max(nq, 1u) / 28.0);
    const uint64_t fit = (uint64_t)((double)usable / per_byte);
    *bytes = std::min<uint64_t>(fit, 0xfff00000ull); // (column offsets inside a chunk are 32-bit)
    return 0;
}

int oswald_hip_chunk_search(oswald_hip_ctx *ctx, int dev, int chunk, int32_t *scores_out)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    Chunk &c = d.chunks[chunk];
    HIP_TRY(hipSetDevice(d.id));
    PhaseTimer pt(ctx->tun.debug_phases);
    if (ctx->topr_r && c.has_index && ctx->topr_queries_version != ctx->queries_version)
        return fail(OSWALD_HIP_ESTATE, "the query set changed since oswald_hip_topr_begin: call it again before searching");
    // An upload the host has not waited for and that has not landed by itself (its re-tile finds no wave slot while a search
    // is running): the search is planned on the extents the group lengths give and queued BEHIND the upload on the device --
    // the host neither waits for the search before this one to drain nor keeps the device waiting for its plan afterwards.
    // Otherwise the planner reads the live extents the upload brought back.
    bool landed = true;
    if (c.upload_pending) {
        const hipError_t q = hipEventQuery(c.ev_up);
        if (q == hipErrorNotReady) { landed = false; (void)hipGetLastError(); }
        else if (q != hipSuccess) return fail(OSWALD_HIP_ERUNTIME, "hipEventQuery(upload): %s", hipGetErrorString(q));
        else if (int r = finish_upload(d, c)) return r;
    }
    if (ctx->tun.plan_waits_for_upload && !landed) { if (int r = finish_upload(d, c)) return r; landed = true; }
    HoldTimer ht(g_debug_slow);
    if (int r = sync_queries(ctx, d)) return r;
    pt.lap("search: queries + profiles");
    ht.lap("search: queries + profiles");
    if (int r = build_items(ctx, d, c, landed && !ctx->tun.plan_on_estimates)) return r;
    pt.lap("search: work-queue plan");
    ht.lap("search: work-queue plan (incl. the queues' copy)");
    if (!landed) HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_up, 0)); // everything queued below finds the chunk in place
    if (c.ev_items) HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_items, 0)); // ... and its work queues (copy stream)
    ht.lap("search: stream wait for the upload");
    if (c.nitems + c.nitems_wg + c.nitems_q + c.nitems_q_wg == 0) { c.searched = true; return topr_after_search(ctx, d, c); }
    if (!d.bnd.p || d.bnd_stride == 0) return fail(OSWALD_HIP_ESTATE, "device %d has no spill scratch (an earlier allocation failed)", dev);
    if (c.down_pending) { HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_down, 0)); c.down_pending = false; } // the table of the slot's last search is still on its way out

    OswSearchArgs a;
    memset(&a, 0, sizeof a);
    a.tiled = (const uint16_t *)c.tiled.p;
    a.blocks = (const OswBlock *)c.blocks.p;
    a.sub_cols = c.sub_cols_dev();
    a.items = (const uint2 *)c.items_dev().p;
    a.nitems = c.nitems;
    a.nitems_wg = c.nitems_wg;
    a.two_ended_waves = ctx->tun.two_ended;
    a.one_ended_wg = ctx->tun.one_ended_wg;
    a.force_all = ctx->cell_bits == 32 ? 1u : 0u;
    a.debug_nospill = ctx->tun.debug_nospill ? 1u : 0u; // -DOSW_DIAG builds only (timing experiment: results are wrong); always 0 otherwise
    a.prof = (const uint2 *)d.prof.p;
    a.prof_off = (const uint32_t *)d.prof_off.p;
    a.qlen = (const uint16_t *)d.qlen.p;
    a.bnd = (uint2 *)d.bnd.p;
    a.top_pages = (const uint2 *)d.top_pages.p;
    a.bnd_stride = d.bnd_stride + OSW_SCRATCH_DATA;
    a.scores = (int32_t *)c.scores.p;
    a.score_stride = c.score_stride;
    a.counters = (uint32_t *)d.counters.p;
    a.counters_ovf = (uint32_t *)d.counters.p + OSW_CTR_BLOCKS * OSW_CTR_COUNT;
    a.ovf_items = (uint2 *)c.ovf.p;
    const uint32_t goe = (uint32_t)(ctx->open_gap + ctx->extend_gap), ge = (uint32_t)ctx->extend_gap;
    if (first_pass_is_frame(ctx)) {
        // column-frame int16 cell: gap OPEN in the goe slot; the plain cell it falls back to gets (goe, ge)
        const uint32_t go = (uint32_t)ctx->open_gap;
        a.goe_pk = go | (go << 16);
        a.ge_pk = ge | (ge << 16);
        a.goe_fb = goe | (goe << 16);
        a.ge_fb = ge | (ge << 16);
    } else {
        a.goe_pk = goe | (goe << 16);
        a.ge_pk = ge | (ge << 16);
    }
    a.goe = (int32_t)goe;
    a.ge = (int32_t)ge;

    const bool dbg_times = ctx->tun.debug_times; // -DOSW_DIAG builds only
    if (dbg_times) {
        HIP_TRY(d.wg_times.reserve((size_t)d.grid * 5 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d.wg_times.p, 0, (size_t)d.grid * 5 * sizeof(unsigned long long), d.stream));
        a.wg_times = (unsigned long long *)d.wg_times.p;
    }
    EventPair ev{};
    if (ctx->profiling) {
        if (d.ev_pool.empty()) {
            HIP_TRY(hipEventCreate(&ev.a));
            HIP_TRY(hipEventCreate(&ev.b));
            HIP_TRY(hipEventCreate(&ev.c));
            HIP_TRY(hipEventCreate(&ev.d));
        } else { ev = d.ev_pool.back(); d.ev_pool.pop_back(); }
        ev.c_used = false;
    }
    HIP_TRY(hipMemsetAsync(d.counters.p, 0, (OSW_CTR_BLOCKS * OSW_CTR_COUNT + 8) * sizeof(uint32_t), d.stream));
    const uint32_t grid_cap = ctx->tun.grid_per_cu ? std::min<uint32_t>(d.grid, (uint32_t)d.prop.multiProcessorCount * ctx->tun.grid_per_cu) : d.grid; // (-DOSW_DIAG sweep)
    const uint32_t grid = std::min<uint32_t>(grid_cap, std::max<uint32_t>(1, (c.nitems + 3) / 4 + c.nitems_wg));
    if (ctx->profiling) HIP_TRY(hipEventRecord(ev.a, d.stream));
    const bool frame = first_pass_is_frame(ctx);
    const auto launch_single = frame ? osw_launch_s16 : osw_launch_pk16;
    const auto launch_pair = frame ? osw_launch_s16q : osw_launch_pk16q;
    OswSearchArgs as = a; // single queries on the int16 cells: {S, 1} entries (`a` itself stays on the plain integer profile for the int32 kernel)
    as.prof = (const uint2 *)d.prof_seq.p;
    if (frame) { as.prof = (const uint2 *)d.prof_seq_alt.p; as.prof_fb = (const uint2 *)d.prof_seq.p; }
    if (first_pass_is_q8(ctx)) {
        // 8-bit first pass over the query pairs, then -- all on this stream, each kernel reading what the one before
        // queued -- a leftover unpaired query on the plain int16 kernel, the int16 re-run of what left the 7-bit range
        // (queue length on the device), and below the int32 re-run of what reached the int16 ceiling
        if (c.nitems_q > 0) {
            OswSearchArgs aq = a;
            aq.items = (const uint2 *)c.items_q_dev().p;
            aq.nitems = c.nitems_q;
            aq.nitems_wg = c.nitems_q_wg; // 0: wave items only
            aq.prof = (const uint2 *)d.prof_pair8.p;
            aq.prof_off = (const uint32_t *)d.pair_off.p;
            aq.qlen = (const uint16_t *)d.pair_len.p;
            aq.pair_q = (const uint32_t *)d.pair_q.p;
            aq.counters = (uint32_t *)d.counters.p + OSW_CTR_COUNT;
            aq.ovf8_items = (uint2 *)c.ovf8.p;
            aq.bias8 = (uint32_t)bias8_of(ctx);
            aq.go8 = (uint32_t)ctx->open_gap;
            aq.ge8 = (uint32_t)ctx->extend_gap;
            aq.off8 = (uint32_t)offset8_of(ctx);
            HIP_TRY(osw_launch_q8(aq, std::min<uint32_t>(d.grid_q8, (c.nitems_q + 3) / 4), d.stream));
        }
        if (c.nitems + c.nitems_wg > 0) HIP_TRY(osw_launch_pk16(as, grid, d.stream));
        if (c.nitems_q > 0) {
            OswSearchArgs ar = a; // plain single-query profile ({S, 1} entries), (open+extend, extend)
            ar.prof = (const uint2 *)d.prof_seq.p;
            ar.items = (const uint2 *)c.ovf8.p;
            ar.nitems = 0;
            ar.nitems_wg = 0;
            ar.nitems_dev = a.counters_ovf + 1; // workgroup entries (four waves on the four lanes of a flagged quad, geometry 64)
            ar.counters = (uint32_t *)d.counters.p + 2 * OSW_CTR_COUNT;
            if (ctx->profiling) { HIP_TRY(hipEventRecord(ev.c, d.stream)); ev.c_used = true; }
            HIP_TRY(osw_launch_pk16(ar, std::min<uint32_t>(d.grid, 512u), d.stream));
        }
    } else if (ctx->cell_bits != 32 && c.nitems_q + c.nitems_q_wg > 0) {
        // query pairs first (the bulk of a multi-query search), on their own queue counters
        OswSearchArgs aq = a;
        aq.items = (const uint2 *)c.items_q_dev().p;
        aq.nitems = c.nitems_q;
        aq.nitems_wg = c.nitems_q_wg;
        aq.prof = (const uint2 *)d.prof_pair.p;
        aq.prof_fb = (const uint2 *)d.prof_pair_i16.p;
        aq.prof_off = (const uint32_t *)d.pair_off.p;
        aq.qlen = (const uint16_t *)d.pair_len.p;
        aq.pair_q = (const uint32_t *)d.pair_q.p;
        aq.counters = (uint32_t *)d.counters.p + OSW_CTR_COUNT;
        const uint32_t gq = std::min<uint32_t>(grid_cap, (c.nitems_q + 3) / 4 + c.nitems_q_wg);
        if (c.nitems + c.nitems_wg > 0 && !ctx->tun.one_stream) {
            // the single-query launch goes to a second stream so that its workgroups fill the slots the
            // pair launch frees while it drains (both are persistent grids pulling from their own queues)
            HIP_TRY(hipEventRecord(d.ev_fork, d.stream));
            HIP_TRY(launch_pair(aq, gq, d.stream));
            HIP_TRY(hipStreamWaitEvent(d.stream2, d.ev_fork, 0));
            OswSearchArgs a2 = as; // its own half of the spill scratch: the two launches overlap
            a2.bnd = a.bnd + (size_t)d.grid * (OSW_WG_THREADS / 64) * a.bnd_stride;
            HIP_TRY(launch_single(a2, grid, d.stream2));
            HIP_TRY(hipEventRecord(d.ev_join, d.stream2));
            HIP_TRY(hipStreamWaitEvent(d.stream, d.ev_join, 0));
        } else {
            HIP_TRY(launch_pair(aq, gq, d.stream));
            if (c.nitems + c.nitems_wg > 0) HIP_TRY(launch_single(as, grid, d.stream));
        }
    } else if (ctx->cell_bits != 32 && c.nitems + c.nitems_wg > 0) {
        HIP_TRY(launch_single(as, grid, d.stream));
    }
    if (ctx->profiling) HIP_TRY(hipEventRecord(ev.d, d.stream));
    // cell_bits 32: the whole plan on the int32 kernel; else: the re-run of what reached the int16 cells' ceiling (queue on the device)
    if (ctx->cell_bits == 32) HIP_TRY(osw_launch_i32(a, std::min<uint32_t>(d.grid, 1024u), d.stream));
    else HIP_TRY(osw_launch_i32r(a, d.grid * (OSW_WG_THREADS / 64), d.stream));
    if (ctx->profiling) { HIP_TRY(hipEventRecord(ev.b, d.stream)); d.ev_used.push_back(ev); }
    c.searched = true;
    ht.lap("search: launches");
    if (int r = topr_after_search(ctx, d, c)) return r;
    ht.lap("search: top-r launches");
    HIP_TRY(hipEventRecord(c.ev_use, d.stream)); // an upload into this slot waits for it
    c.use_pending = true;
    if (dbg_times) {
        // diagnostics only: when did the workgroups of the DP launch start / leave phase 1 / finish
        HIP_TRY(hipStreamSynchronize(d.stream));
        std::vector<unsigned long long> t((size_t)grid * 5);
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
        {
            uint32_t ctr[8] = {0};
            HIP_TRY(hipMemcpy(ctr, (uint32_t *)d.counters.p + OSW_CTR_BLOCKS * OSW_CTR_COUNT, sizeof ctr, hipMemcpyDeviceToHost));
            // waves x span in core-clock cycles (2.4 GHz nominal) against the cycles spent in the slice reloads of workgroup items
            const double wave_cycles = (double)grid * 4.0 * span * 2400.0;
            fprintf(stderr, "[oswald_hip]   workgroup items: slice reload + barrier waits %.3g cycles = %.1f%% of all wave time\n",
                    (double)ctr[4] * 1024.0, 100.0 * (double)ctr[4] * 1024.0 / wave_cycles);
        }
        fprintf(stderr, "[oswald_hip]   phase-1 exits by decile:");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_p1[k]);
        fprintf(stderr, "\n[oswald_hip]   finishes by decile:    ");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_end[k]);
        fprintf(stderr, "\n");
        {
            // per CU: when its LAST workgroup finished, and the time-integral of its resident workgroups (4 = full)
            std::vector<std::pair<unsigned long long, std::vector<double>>> cus; // (cu id, finish times of its workgroups)
            for (uint32_t g = 0; g < grid; ++g) {
                const unsigned long long cu = t[(size_t)grid * 4 + g];
                const double e = (double)(std::max(t[g * 4 + 2], t[g * 4 + 3]) - t0) / 100.0;
                auto it = std::find_if(cus.begin(), cus.end(), [&](const auto &x) { return x.first == cu; });
                if (it == cus.end()) { cus.push_back({cu, {}}); it = cus.end() - 1; }
                it->second.push_back(e);
            }
            uint32_t hist_last[10] = {0}, hist_n[8] = {0};
            double sum_last = 0, occ = 0;
            for (auto &c2 : cus) {
                const double last = *std::max_element(c2.second.begin(), c2.second.end());
                hist_last[std::min(9, (int)(last / span * 10))]++;
                hist_n[std::min<size_t>(7, c2.second.size())]++;
                sum_last += last;
                for (double e : c2.second) occ += e;
            }
            fprintf(stderr, "[oswald_hip]   %zu CUs; last finish per CU by decile:", cus.size());
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_last[k]);
            fprintf(stderr, "; mean last finish %.0f%% of span; workgroups per CU histogram:", 100.0 * sum_last / cus.size() / span);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %u", hist_n[k]);
            fprintf(stderr, "; mean resident workgroups per CU over the span %.2f\n", occ / cus.size() / span);
        }
    }
    if (pt.on) { HIP_TRY(hipStreamSynchronize(d.stream)); pt.lap("search: kernels"); }
    if (scores_out) {
        // The caller's table is [nq][ngroups*W]; ours is pitched to whole wave blocks.  It leaves on the download stream,
        // behind the search (ev_use) and beside whatever the search stream runs next: row by row when the rows are few (plain
        // DMA, nothing that needs a wave slot), packed on the device first when they are many (a pitched copy into
        // pageable memory runs at a fraction of a GB/s).
        const uint32_t row = c.ngroups * c.W;
        const size_t bytes = (size_t)ctx->nq * row * sizeof(int32_t);
        // pin the caller's table for the copy unless it is page-locked already (oswald_hip_host_alloc): the DMA engine then
        // writes it directly (a copy into a pageable buffer it has not seen before runs at ~1 GB/s: 8.9 ms for the 8 MB of
        // C2; this way 0.5 ms)
        hipPointerAttribute_t attr;
        const bool pinned_already = hipPointerGetAttributes(&attr, scores_out) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError(); // (an unknown -- pageable -- pointer is reported as an error by some runtimes)
        if (bytes >= (1u << 20) && !ctx->tun.no_pin && !pinned_already) {
            const auto t0 = std::chrono::steady_clock::now();
            if (hipHostRegister(scores_out, bytes, hipHostRegisterPortable) == hipSuccess) d.registered.push_back(scores_out);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 5.0 && g_debug_slow) fprintf(stderr, "[oswald_hip] slow hipHostRegister: %zu bytes took %.1f ms\n", bytes, ms);
            else (void)hipGetLastError(); // e.g. already pinned by the caller: the plain copy below is still correct
            pt.lap("search: pin caller's table");
        }
        HIP_TRY(hipStreamWaitEvent(d.stream_down, c.ev_use, 0));
        if (row == c.score_stride || ctx->nq == 1) {
            HIP_TRY(hipMemcpyAsync(scores_out, c.scores.p, bytes, hipMemcpyDeviceToHost, d.stream_down));
        } else if (ctx->nq <= 64) {
            for (uint32_t q = 0; q < ctx->nq; ++q)
                HIP_TRY(hipMemcpyAsync(scores_out + (size_t)q * row, (const int32_t *)c.scores.p + (size_t)q * c.score_stride, (size_t)row * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, d.stream_down));
        } else {
            HIP_TRY(d.scores_packed.reserve(bytes));
            HIP_TRY(hipMemcpy2DAsync(d.scores_packed.p, (size_t)row * sizeof(int32_t), c.scores.p, (size_t)c.score_stride * sizeof(int32_t),
                                     (size_t)row * sizeof(int32_t), ctx->nq, hipMemcpyDeviceToDevice, d.stream_down));
            HIP_TRY(hipMemcpyAsync(scores_out, d.scores_packed.p, bytes, hipMemcpyDeviceToHost, d.stream_down));
        }
        HIP_TRY(hipEventRecord(c.ev_down, d.stream_down));
        c.down_pending = true;
        ht.lap("search: table download queued");
        if (pt.on) { HIP_TRY(hipStreamSynchronize(d.stream_down)); pt.lap("search: D2H of the score table"); }
    }
    return 0;
}

int oswald_hip_chunk_release(oswald_hip_ctx *ctx, int dev, int chunk)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    HIP_TRY(hipSetDevice(d.id));
    // the caller's b / n / disp are free once the upload has landed; a search of the chunk may still be running -- the
    // next upload into the slot waits for it on the device (ev_use), the host does not
    if (int r = finish_upload(d, d.chunks[chunk])) return r;
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
        HIP_TRY(hipStreamSynchronize(ctx->dev[i].stream_copy));
        HIP_TRY(hipStreamSynchronize(ctx->dev[i].stream_up));
        HIP_TRY(hipStreamSynchronize(ctx->dev[i].stream));
        HIP_TRY(hipStreamSynchronize(ctx->dev[i].stream_down));
        for (Chunk &c : ctx->dev[i].chunks) { c.upload_pending = false; c.use_pending = false; c.down_pending = false; }
        release_registered(ctx->dev[i]);
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
    if (r > 1024) return fail(OSWALD_HIP_EINVAL, "top-r on the device supports r <= 1024 (asked for %u)", r);
    if (int rc = queue_topr(ctx, d, c, nvalid, r)) return rc;
    HIP_TRY(hipMemcpyAsync(scores, d.topr_scores.p, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipMemcpyAsync(index, d.topr_index.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    return 0;
}

int oswald_hip_chunk_set_index(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t first_index, uint32_t nvalid, const uint32_t *index_map)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    Chunk &c = d.chunks[chunk];
    if (nvalid > c.ngroups * c.W) return fail(OSWALD_HIP_EINVAL, "nvalid %u exceeds the chunk's %u lanes", nvalid, c.ngroups * c.W);
    if (!index_map && (uint64_t)first_index + nvalid > 0xffffffffull) return fail(OSWALD_HIP_EINVAL, "database indices must fit 32 bits");
    HIP_TRY(hipSetDevice(d.id));
    if (c.index_map) { // a map given twice to the same upload: the old one may still be on its way, or being read by a search of the chunk
        HIP_TRY(hipStreamSynchronize(d.stream_copy));
        HIP_TRY(hipStreamSynchronize(d.stream));
        c.index_map.reset();
    }
    c.first_index = first_index;
    c.nvalid = nvalid;
    if (index_map && nvalid > 0) {
        // the map goes to the device (the chunk's top list is selected there, on database keys); the host copy is the
        // source of that asynchronous upload and lives as long as the chunk's index does
        c.index_map = std::make_shared<const std::vector<uint32_t>>(index_map, index_map + nvalid);
        c.map_cur ^= 1;
        if (!c.ev_map) HIP_TRY(hipEventCreateWithFlags(&c.ev_map, hipEventDisableTiming));
        HIP_TRY(c.index_map_dev[c.map_cur].reserve((size_t)nvalid * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(c.index_map_dev[c.map_cur].p, c.index_map->data(), (size_t)nvalid * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream_copy));
        HIP_TRY(hipEventRecord(c.ev_map, d.stream_copy));
        c.map_pending = true;
    }
    c.has_index = true;
    return 0;
}

int oswald_hip_topr_begin(oswald_hip_ctx *ctx, uint32_t r)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (r > 1024) return fail(OSWALD_HIP_EINVAL, "top-r on the device supports r <= 1024 (asked for %u): download the score table instead", r);
    if (r > 0 && !ctx->have_queries) return fail(OSWALD_HIP_ESTATE, "oswald_hip_set_queries has not been called");
    const size_t bytes = (size_t)ctx->nq * r * sizeof(unsigned long long);
    for (Device &d : ctx->dev) {
        HIP_TRY(hipSetDevice(d.id));
        d.top_any = false;
        d.top_cur = 0;
        if (bytes == 0) continue;
        for (int k = 0; k < 2; ++k) {
            if (bytes > d.top_run[k].cap) HIP_TRY(hipStreamSynchronize(d.stream)); // (growing frees the old list: nothing may still be reading it)
            HIP_TRY(d.top_run[k].reserve(bytes));
        }
        // the candidates of a chunk's partitions (at most 64 per score row), sized once: growing it behind a launched search
        // would wait for that search (hipFree)
        HIP_TRY(d.topr_cand.reserve((size_t)ctx->nq * 64 * r * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d.top_run[0].p, 0, bytes, d.stream)); // key 0 = none; ordered behind whatever still reads the old list
    }
    ctx->topr_r = r;
    ctx->topr_queries_version = ctx->queries_version;
    return 0;
}

// The gather of oswald_hip_topr.  Level 1: context devices that share a GPU hand their running lists to the first of
// them (plain reads on the same GPU, ordered by an event).  Level 2: the GPUs of the context -- one RCCL rank each,
// communicator made by ncclCommInitAll at bring-up -- all-gather their lists over xGMI, and GPU 0 of the context folds
// them.  Level 3: with a process-level communicator (oswald_hip_comm_init_rank) the contexts' lists are all-gathered
// between the processes and folded again, so every rank ends up with the list of the whole job.  Then ONE copy of
// nq x r (score, index) pairs to the host.  Every fold is osw_topr_fold on tagged keys: descending score, equal
// scores by descending database index (utils.c:3-86).
int oswald_hip_topr(oswald_hip_ctx *ctx, uint32_t r, int32_t *scores, uint32_t *db_index)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (r == 0 || ctx->nq == 0) return 0;
    if (!scores || !db_index) return fail(OSWALD_HIP_EINVAL, "null output");
    if (ctx->topr_r == 0) return fail(OSWALD_HIP_ESTATE, "oswald_hip_topr_begin has not been called");
    if (r > ctx->topr_r) return fail(OSWALD_HIP_EINVAL, "r = %u exceeds the %u lists were collected for (oswald_hip_topr_begin)", r, ctx->topr_r);
    if (ctx->topr_queries_version != ctx->queries_version) return fail(OSWALD_HIP_ESTATE, "the query set changed while top lists were being collected");
    const uint32_t nq = ctx->nq, R = ctx->topr_r;
    const size_t cnt = (size_t)nq * R, bytes = cnt * sizeof(unsigned long long);
    typedef unsigned long long key_t;
    // level 1: siblings -> leader (the sibling's list is on the same GPU: read in place once its stream got there)
    for (Device &d : ctx->dev) {
        if (&ctx->dev[d.leader] == &d || !d.top_any) continue;
        HIP_TRY(hipSetDevice(d.id));
        HIP_TRY(hipEventRecord(d.ev_top, d.stream));
    }
    for (Device &d : ctx->dev) {
        if (&ctx->dev[d.leader] != &d) continue;
        HIP_TRY(hipSetDevice(d.id));
        // room for the lists an all-gather brings in, allocated before anything is queued
        const size_t nlists = (size_t)std::max(ctx->nphys, &d == &ctx->dev[0] ? ctx->pcomm_nranks : 0);
        if (nlists > 1 || (ctx->pcomm && &d == &ctx->dev[0])) HIP_TRY(d.top_gather.reserve(std::max<size_t>(nlists, 1) * bytes));
        for (Device &s : ctx->dev) {
            if (&s == &d || &ctx->dev[s.leader] != &d || !s.top_any) continue;
            HIP_TRY(hipStreamWaitEvent(d.stream, s.ev_top, 0));
            HIP_TRY(osw_launch_topr_fold_lists2((const key_t *)s.top_run[s.top_cur].p, (const key_t *)d.top_run[d.top_cur].p, R, nq,
                                                (key_t *)d.top_run[d.top_cur ^ 1].p, d.stream));
            d.top_cur ^= 1;
            d.top_any = true;
        }
    }
    Device &root = ctx->dev[0];
    // level 2: the GPUs of the context
    if (ctx->nphys > 1) {
        NCCL_TRY(ncclGroupStart());
        for (Device &d : ctx->dev) {
            if (d.comm_rank < 0) continue;
            (void)hipSetDevice(d.id);
            const ncclResult_t nr = ncclAllGather(d.top_run[d.top_cur].p, d.top_gather.p, cnt, ncclUint64, d.comm, d.stream);
            if (nr != ncclSuccess) { (void)ncclGroupEnd(); return fail(OSWALD_HIP_ECOMM, "ncclAllGather (GPU %d): %s", d.id, ncclGetErrorString(nr)); }
        }
        NCCL_TRY(ncclGroupEnd());
        HIP_TRY(hipSetDevice(root.id));
        HIP_TRY(osw_launch_topr_fold_lists((const key_t *)root.top_gather.p, (uint32_t)ctx->nphys, cnt, R, nq, (key_t *)root.top_run[root.top_cur ^ 1].p, root.stream));
        root.top_cur ^= 1;
    }
    HIP_TRY(hipSetDevice(root.id));
    // level 3: the ranks of the job
    if (ctx->pcomm && ctx->pcomm_nranks > 0) {
        NCCL_TRY(ncclAllGather(root.top_run[root.top_cur].p, root.top_gather.p, cnt, ncclUint64, ctx->pcomm, root.stream));
        HIP_TRY(osw_launch_topr_fold_lists((const key_t *)root.top_gather.p, (uint32_t)ctx->pcomm_nranks, cnt, R, nq, (key_t *)root.top_run[root.top_cur ^ 1].p, root.stream));
        root.top_cur ^= 1;
    }
    // the first r of every query's R keys -> (score, index), one copy to the host
    const size_t out_cnt = (size_t)nq * r;
    HIP_TRY(root.top_final.reserve(out_cnt * 8));
    if (ctx->top_host_bytes < out_cnt * 8) {
        if (ctx->top_host) (void)hipHostFree(ctx->top_host);
        ctx->top_host = nullptr;
        ctx->top_host_bytes = 0;
        HIP_TRY(hipHostMalloc(&ctx->top_host, out_cnt * 8, hipHostMallocPortable));
        ctx->top_host_bytes = out_cnt * 8;
    }
    HIP_TRY(osw_launch_topr_untag((const key_t *)root.top_run[root.top_cur].p, nq, R, r, (int32_t *)root.top_final.p, (uint32_t *)root.top_final.p + out_cnt, root.stream));
    HIP_TRY(hipMemcpyAsync(ctx->top_host, root.top_final.p, out_cnt * 8, hipMemcpyDeviceToHost, root.stream));
    for (Device &d : ctx->dev) { // everything queued is done: the lists have been folded, downloads of score tables have landed
        HIP_TRY(hipSetDevice(d.id));
        HIP_TRY(hipStreamSynchronize(d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream_down));
        if (!d.retired_maps.empty()) HIP_TRY(hipStreamSynchronize(d.stream_copy)); // (their copies ran on the copy stream)
        for (Chunk &c : d.chunks) c.down_pending = false;
        release_registered(d);
    }
    memcpy(scores, ctx->top_host, out_cnt * sizeof(int32_t));
    memcpy(db_index, (const char *)ctx->top_host + out_cnt * sizeof(int32_t), out_cnt * sizeof(uint32_t));
    return 0;
}

int oswald_hip_comm_unique_id(void *id, size_t id_bytes)
{
    if (!id || id_bytes < sizeof(ncclUniqueId)) return fail(OSWALD_HIP_EINVAL, "the id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memset(id, 0, id_bytes);
    memcpy(id, &u, sizeof u);
    return 0;
}

int oswald_hip_comm_init_rank(oswald_hip_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank)
{
    if (!ctx || !id) return fail(OSWALD_HIP_EINVAL, "null argument");
    if (id_bytes < sizeof(ncclUniqueId)) return fail(OSWALD_HIP_EINVAL, "the id must be the %zu bytes oswald_hip_comm_unique_id wrote", sizeof(ncclUniqueId));
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(OSWALD_HIP_EINVAL, "rank %d of %d", rank, nranks);
    if (ctx->pcomm) return fail(OSWALD_HIP_ESTATE, "the context already has a process-level communicator");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    HIP_TRY(hipSetDevice(ctx->dev[0].id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(ncclCommInitRank(&comm, nranks, u, rank));
    int count = 0, me = -1;
    ncclResult_t nr = ncclCommCount(comm, &count);
    if (nr == ncclSuccess) nr = ncclCommUserRank(comm, &me);
    if (nr != ncclSuccess || count != nranks || me != rank) {
        (void)ncclCommDestroy(comm);
        return fail(OSWALD_HIP_ECOMM, "the communicator reports rank %d of %d, expected %d of %d (%s)", me, count, rank, nranks, ncclGetErrorString(nr));
    }
    ctx->pcomm = comm;
    ctx->pcomm_nranks = nranks;
    ctx->pcomm_rank = rank;
    return 0;
}

int oswald_hip_comm_info(oswald_hip_ctx *ctx, int *out4)
{
    if (!ctx || !out4) return fail(OSWALD_HIP_EINVAL, "null argument");
    int version = 0;
    NCCL_TRY(ncclGetVersion(&version));
    out4[0] = ctx->nphys;
    out4[1] = 0;
    out4[2] = -1;
    out4[3] = version;
    if (ctx->nphys > 1) { // what the in-context communicator itself reports
        int count = 0;
        NCCL_TRY(ncclCommCount(ctx->dev[0].comm, &count));
        out4[0] = count;
    }
    if (ctx->pcomm) {
        NCCL_TRY(ncclCommCount(ctx->pcomm, &out4[1]));
        NCCL_TRY(ncclCommUserRank(ctx->pcomm, &out4[2]));
    }
    return 0;
}

int oswald_hip_merge_candidates(uint32_t nq, uint64_t ncand, const int32_t *cand_scores, const uint32_t *cand_index, uint32_t r,
                                int32_t *scores, uint32_t *db_index)
{
    if (nq == 0 || r == 0) return 0;
    if ((ncand > 0 && (!cand_scores || !cand_index)) || !scores || !db_index) return fail(OSWALD_HIP_EINVAL, "null argument");
    merge_candidates(nq, (size_t)ncand, cand_scores, cand_index, r, scores, db_index);
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
    HIP_TRY(hipMemcpy(ctr, (uint32_t *)d.counters.p + OSW_CTR_BLOCKS * OSW_CTR_COUNT, sizeof ctr, hipMemcpyDeviceToHost));
    if (dp_kernel_ms) *dp_kernel_ms = d.dp_ms;
    if (dp_launches) *dp_launches = d.dp_launches;
    if (rerun_items) *rerun_items = ctr[0]; // of the most recent search
    if (reset) { d.dp_ms = 0; d.dp_launches = 0; d.rerun16_ms = 0; d.rerun32_ms = 0; }
    return 0;
}

int oswald_hip_rerun_stats(oswald_hip_ctx *ctx, int dev, double *ms2)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!ms2) return fail(OSWALD_HIP_EINVAL, "null output");
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    HIP_TRY(hipStreamSynchronize(d.stream));
    drain_events(d);
    ms2[0] = d.rerun16_ms;
    ms2[1] = d.rerun32_ms;
    return 0;
}

int oswald_hip_rerun_counts(oswald_hip_ctx *ctx, int dev, uint64_t *out2)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!out2) return fail(OSWALD_HIP_EINVAL, "null output");
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    HIP_TRY(hipStreamSynchronize(d.stream));
    uint32_t ctr[2] = {0, 0};
    HIP_TRY(hipMemcpy(ctr, (uint32_t *)d.counters.p + OSW_CTR_BLOCKS * OSW_CTR_COUNT, sizeof ctr, hipMemcpyDeviceToHost));
    out2[0] = ctr[1]; // 8-bit pass -> int16
    out2[1] = ctr[0]; // int16 -> int32
    return 0;
}

int oswald_hip_chunk_geometry(oswald_hip_ctx *ctx, int dev, int chunk, uint64_t *out8)
{
    if (int r = check_dev(ctx, dev)) return r;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    if (!out8) return fail(OSWALD_HIP_EINVAL, "null output");
    uint64_t *out6 = out8;
    Chunk &c = d.chunks[chunk];
    HIP_TRY(hipSetDevice(d.id));
    if (int r = finish_upload(d, c)) return r;
    HIP_TRY(hipStreamSynchronize(d.stream));
    std::vector<OswBlock> blocks(c.nblocks);
    if (c.nblocks) HIP_TRY(hipMemcpy(blocks.data(), c.blocks.p, c.nblocks * sizeof(OswBlock), hipMemcpyDeviceToHost));
    uint64_t alloc = 0, live = 0;
    for (const OswBlock &b : blocks) { alloc += b.ncols4_alloc; live += b.ncols4; }
    out6[0] = c.nblocks;
    out6[1] = alloc;
    out6[2] = live;
    out6[3] = live * 64 * sizeof(uint2);
    out6[4] = c.nitems + 4ull * c.nitems_wg + c.nitems_q + 4ull * c.nitems_q_wg; // wave-level work items (a phase-1 entry is four)
    out6[5] = c.max_lg;
    out8[6] = c.planned_spill_bytes;
    out8[7] = 0;
    return 0;
}

} // extern "C"
