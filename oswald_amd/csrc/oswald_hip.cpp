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
#include <stdexcept>
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

// What stands between the implementations and the C ABI (see the exported entry points at the end of the file).
template <class F>
int guarded(const char *entry, F &&body) noexcept
{
    auto report = [entry](int code, const char *what) noexcept {
        try { return fail(code, "%s: %s", entry, what); }
        catch (...) { return code; } // (the message itself could not be stored: the code still says what happened)
    };
    try { return body(); }
    catch (const std::bad_alloc &) { return report(OSWALD_HIP_ENOMEM, "out of host memory (std::bad_alloc)"); }
    catch (const std::length_error &e) { return report(OSWALD_HIP_ENOMEM, e.what()); }
    catch (const std::exception &e) { return report(OSWALD_HIP_ERUNTIME, e.what()); }
    catch (...) { return report(OSWALD_HIP_ERUNTIME, "unknown exception"); }
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
    bool no_direct_table = false;      // OSWALD_HIP_NO_DIRECT_TABLE=1 (A/B and test hook): score tables leave by DMA on the download stream even when the kernels could write them
    size_t fake_free_mem = 0;          // OSWALD_HIP_FAKE_FREE_MEM=bytes: oswald_hip_max_chunk_size reckons with a device that has no more free (test hook)
    int pair_tails = 1;                // OSWALD_HIP_PAIR_TAILS=0|1|2: the rows a pair's longer query has beyond the shorter one's are padded (0: rounds 1-4) / run as the pair item's TAIL on the single-query cell where the cost model says so (1) / always (2: test hook)
    double warm_ms = 250.0;            // OSWALD_HIP_WARM_MS=ms: how long at most oswald_hip_init keeps the devices busy behind itself (0: not at all); see Device::warm_stop
    size_t split_bytes = 32u << 20;    // OSWALD_HIP_SPLIT_BYTES=bytes: from this size on an asynchronous upload that finds its device idle is cut into head + rest (0: never; a small value: test hook)
    // planner parameters: constants in the default build, OSWALD_HIP_* sweep knobs with -DOSW_DIAG
    double pair_margin = 0.95, col_cost = 10.0, target_div = 1.25, quad_frac = 0.5, entries_per_wg = 4.0;
    uint32_t wg_min_cols = 2048, wg_wide_cols = 2048, wg_min_cols_single = 0, wg_wide_cols_single = 0, wg_min_rounds = 0, wg_min_cols_long = 0, two_ended = 0, one_ended_wg = 1, grid_per_cu = 0;
    bool no_prio = false, one_stream = false;
    bool debug_times = false, debug_nospill = false; // -DOSW_DIAG only
    void refresh();
};
bool g_debug_slow = false; // OSWALD_HIP_DEBUG_SLOW=1: report allocations / pinning that take > 5 ms
size_t g_fail_alloc_above = 0; // OSWALD_HIP_FAIL_DEVICE_ALLOC_ABOVE=bytes (test hook): a device allocation of more than that fails like a device that is full

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
    no_direct_table = flag("OSWALD_HIP_NO_DIRECT_TABLE");
    fake_free_mem = (size_t)num("OSWALD_HIP_FAKE_FREE_MEM", 0);
    split_bytes = (size_t)num("OSWALD_HIP_SPLIT_BYTES", (double)(32u << 20));
    warm_ms = std::min(num("OSWALD_HIP_WARM_MS", 250.0), 2000.0);
    pair_tails = (int)num("OSWALD_HIP_PAIR_TAILS", 1);
    g_debug_slow = flag("OSWALD_HIP_DEBUG_SLOW");
    g_fail_alloc_above = (size_t)num("OSWALD_HIP_FAIL_DEVICE_ALLOC_ABOVE", 0);
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
        if (ms > 1.0) fprintf(stderr, "[oswald_hip] the host was held %.1f ms in: %s\n", ms, what);
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
        if (p && !view) { hipError_t e = hipFree(p); if (e != hipSuccess) { (void)hipGetLastError(); return e; } }
        p = nullptr; cap = 0; view = false; // (a slice of an arena / a slot's slab that must grow becomes an allocation of its own; the slice goes back with its slab)
        size_t want = bytes + bytes / 8 + 256;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = g_fail_alloc_above && want > g_fail_alloc_above ? hipErrorOutOfMemory : hipMalloc(&p, want);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 5.0 && g_debug_slow) fprintf(stderr, "[oswald_hip] slow hipMalloc: %zu bytes took %.1f ms\n", want, ms);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
    // a buffer that is a slice of an Arena (below): owned by the arena, re-assigned with every query set
    bool view = false;
    void assign(void *ptr, size_t bytes) { release(); p = ptr; cap = bytes; view = true; }
};

// Page-locked host memory that only ever grows: the library's own staging of everything small it sends to a device.  Round 5:
// a small hipMemcpyAsync from PAGEABLE memory (a std::vector, or the caller's n[] / disp[] / index map) issued while a chunk is on
// the link held the caller for ~9 ms -- the runtime stages such a copy on the caller's thread and waits for it -- so nothing
// pageable is handed to a copy any more: small inputs are copied (memcpy) into one of these first.
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); if (e != hipSuccess) { (void)hipGetLastError(); return e; } p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocPortable);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One allocation for the many small per-query-set buffers of a device (queries, lengths, offsets, matrix, profiles in every
// form, constant pages): a slab made at bring-up and cut anew for every query set, so that the first search of a query set
// -- inside the caller's timed region, like the reference's query upload (FPGAsearch.c:85) -- makes no allocation at all
// (a hipMalloc maps device memory through the kernel driver, beside whatever upload is in flight; slices of a slab that
// exists cost nothing).  A set that needs more than the slab holds gets a larger one (one hipMalloc).
struct Arena {
    DevBuf slab;
    size_t used = 0;
    static size_t up(size_t n) { return (n + 511) & ~(size_t)511; } // (512: a slot's `tiled` slice must lie a whole number of 4-column groups from any other, see Group)
    void *take(size_t bytes) { void *r = (char *)slab.p + used; used += up(bytes); return r; }
};

struct Chunk {
    bool live = false;
    uint32_t ngroups = 0, W = 0, nblocks = 0;
    uint32_t score_stride = 0;   // nblocks*128
    uint32_t max_ncols4 = 0;     // largest stored extent of a block
    uint64_t total_col4 = 0;     // stored 4-column groups incl. the pad group per block
    DevBuf tiled, blocks, sub_cols_buf, scores, ovf, ovf8;
    DevBuf slab;                        // oswald_hip_reserve_chunks: ONE device allocation that the buffers above and st_b are slices of (a creation of the slot's buffers
                                        // inside a caller's clock is one call into the driver per slot instead of six; a buffer that must grow later gets an allocation of its own)
    // The work queues, two sets in turn, in PAGE-LOCKED HOST memory that the kernels read in place (a wave fetches one 8-byte
    // entry per work item over the link: microseconds against items of 0.1 - 10 ms).  Round 4 copied every plan to the device on
    // the copy stream; there the few hundred KB queued BEHIND the bulk copies of the chunks coming in, and the first search of a
    // pass -- the head of a chunk cut in two, whose rest is on the link -- waited 1.4 ms for its queues (round 5,
    // profiles/r05_inclusive_probe_q1.txt).  Two sets: a search is planned while its chunk's upload is still on its way, and the
    // search of the slot's previous chunk may still be running -- and pulling from the set the plan before this one was written to.
    // (The set before that is free: a slot is re-used only after oswald_hip_chunk_release, which returns when the released chunk's
    // upload has landed, i.e. after the search of the chunk before it.)
    uint2 *items_pin[2] = {nullptr, nullptr};
    size_t items_pin_cap[2] = {0, 0};   // entries (both queues, one behind the other)
    size_t items_q_off[2] = {0, 0};     // where the query-pair kernel's queue starts in the set
    int items_cur = 0;
    // ... and who read a set last: recorded on the search stream behind the launches that pull from it.  A plan that is about to
    // overwrite a set waits for its last reader on the host (with one plan per residency that reader is long gone, but
    // a resident chunk searched twice in a row and re-planned in between came back to a set a running search was still pulling
    // from); build_items also keeps an estimate plan while a search of the chunk is pending, so the wait is never met on the
    // pipelined path.
    hipEvent_t ev_set_read[2] = {nullptr, nullptr};
    bool set_read_pending[2] = {false, false};
    const uint2 *items_ptr() const { return items_pin[items_cur]; }
    const uint2 *items_q_ptr() const { return items_pin[items_cur] + items_q_off[items_cur]; }
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
    DevBuf st_b;                        // the caller's residues as they arrive on the device (the re-tile kernel's input): per slot,
                                        // so that the copies of the next upload never wait for a re-tile that has not found room yet
    PinBuf nd_pin;                      // the caller's n[] and disp[], copied at the upload call; the re-tile kernel reads them in place
    hipEvent_t ev_copy = nullptr;       // recorded on the copy stream behind them
    hipEvent_t ev_down = nullptr;       // recorded on the download stream behind the copy of the chunk's score table to the caller ...
    bool down_pending = false;          // ... which the next search that writes the slot's table must wait for
    uint32_t nitems = 0, nitems_wg = 0;  // wave items / workgroup items of the queue
    uint32_t nitems_q = 0, nitems_q_wg = 0; // the same for the query-pair kernel's queue
    uint32_t nitems_short = 0;              // ... of which SHORT pair items (they run their pair's tail, OswSearchArgs::hand)
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
    bool index_map = false;             // the chunk has a map
    // the index map on the device: two buffers in turn (a slot re-used while its last search -- which reads the old map --
    // is still running gets the other one), copied on the DMA-only copy stream from a page-locked copy of the caller's map
    // (map_pin, two in turn like their targets); ev_map[k]: the copy of buffer k has landed
    DevBuf index_map_dev[2];
    PinBuf map_pin[2];
    int map_cur = 0;
    bool map_pending = false;
    hipEvent_t ev_map[2] = {nullptr, nullptr};
    // ... and who read device buffer k last: recorded on the search stream behind the top-list fold that reads it.  A map that is about
    // to overwrite a buffer waits for that reader on the host (set_index(A), search, set_index(B), search, set_index(C)
    // without a wait in between let copy C overwrite buffer A while search A's fold had not run yet; in the pipelined use the
    // reader is long gone and the wait costs nothing)
    hipEvent_t ev_map_read[2] = {nullptr, nullptr};
    bool map_read_pending[2] = {false, false};
    // A chunk the library cut in two at its upload (oswald_hip_chunk_upload_async on an idle device): the caller's handle is
    // the HEAD's slot, `next` the slot of the REST (groups head_groups .. of the caller's arrays), which no handle names.
    // Every per-chunk entry point walks the chain.
    int next = -1;
    bool is_cont = false;               // this slot is the rest of another slot's chunk
    bool is_group = false;              // not a slot: the combined view of several resident chunks (Device::group, oswald_hip_search_resident)
};

// Slots a device keeps at most before an upload GROWS the largest free one instead of opening another: three resident chunks
// (one searched, two coming in) and the two pieces of a first chunk cut at its upload.
#define OSW_MAX_SLOTS 5

// Several RESIDENT chunks searched as ONE launch (oswald_hip_search_resident, round 6).  Every launch boundary costs a ramp, a ragged end
// and -- short launches -- clock (DESIGN 4 "Single queries": one launch instead of three is worth 7 % to a one-query search of 1 M
// sequences, 1.2 % to twenty queries).  The kernels address a chunk through its block table, and nothing in them says that the blocks of
// a table lie in ONE allocation: the group's table lists the blocks of all members with their column offsets counted from the lowest
// `tiled` address among them (32-bit offsets of 512 B: a span of 2 TB) and their sequences' columns in ONE score table; `g` is a chunk
// in everything but its residues -- combined live extents, work queues planned over all blocks, score table, re-run queues.
struct Group {
    std::vector<int> members;           // slots (chains flattened)
    std::vector<uint64_t> up_seq;       // ... and which upload each held when the group was made
    std::vector<uint32_t> col0, blk0;   // a member's first column of the group's score table / first block of its block table
    std::vector<uint16_t> sub_cols_host; // combined live extents (what g.sub_cols points at)
    Chunk g;
    bool valid = false;
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
    hipEvent_t ev_mark = nullptr;    // OSWALD_HIP_DEBUG_SLOW: "the first upload of a pass was issued" (reference point of the device's time line)
    bool mark_set = false;
    hipDeviceProp_t prop;
    uint32_t grid = 0;               // persistent workgroups per launch
    uint32_t grid_q8 = 0;            // ... of the 8-bit kernel (more workgroups per CU; at most 2 x grid: it runs alone and may use both halves of the spill scratch)
    DevBuf queries, qlen, a_disp, prof_off, prof, prof_alt, prof_seq, prof_seq_alt, prof_pair_i16, pair_q, pair_off, pair_len, prof_pair, submat, bnd, counters;
    DevBuf topr_scores, topr_index, topr_cand, wg_times, scores_packed, top_pages, prof_pair8, floor_i32, pair_rows, tail_len, tail_off;
    Arena qset;                           // what the buffers of the current query set (queries ... top_pages, prof_pair8) are slices of
    uint8_t *qstage = nullptr;            // page-locked: the small inputs of the query set as the arena holds them, read in place by the copy kernel
    size_t qstage_cap = 0;
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
    // The warm-up (round 6).  A GPU that has had nothing to do runs its first kernels at low clocks: the first search of a PROCESS took
    // 1.2 ms longer than the same search a moment later (one 375-residue query against 1 M sequences: 15.0 against 13.8 ms on the
    // device, profiles/r05_cli_q1_1m_phases.txt) -- and every run of the command-line tool is a first search.  oswald_hip_init
    // therefore ends by starting a kernel that keeps every CU busy on a stream of its own (osw_spin), BEHIND the bring-up and
    // beside whatever the caller does next before its clock starts (the tool: loading and page-locking the database); the first call
    // that gives the device real work -- queries, buffers, a chunk -- tells it to stop (a word in page-locked memory the kernel
    // polls: it is gone some 20 us later), and it stops by itself after Tunables::warm_ms.  OSWALD_HIP_WARM_MS=0: no warm-up.
    std::unique_ptr<Group> group;     // the last group searched (its tables and plan are kept while the members stay as they are)
    hipStream_t stream_warm = nullptr;
    uint32_t *warm_stop = nullptr;   // page-locked; *warm_stop = 1: leave
    bool warm_running = false;
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
    // Tails (sw_kernels.h, OswSearchArgs::hand): a SHORT item of pair i runs pair_rows[i] rows as a pair (the shorter query's, rounded up
    // to 4; = pair_len where the pair has no tail) and then the rest of the longer query -- tail_len[7 i + lg] rows from row-block
    // tail_off[7 i + lg] of that query's single-query profile on, behind a pair item of geometry 2^lg -- on the single-query cell
    bool have_tails = false;                        // some pair of the set has a tail
    std::vector<uint16_t> pair_rows;                // [pair]
    std::vector<uint16_t> tail_len;                 // [pair * OSW_TAIL_GEOMS + lg]
    std::vector<uint32_t> tail_off;
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

// the first real work of a context ends the warm-up its bring-up started (Device::warm_stop)
void stop_warm(oswald_hip_ctx *ctx)
{
    for (Device &d : ctx->dev)
        if (d.warm_running) { *(volatile uint32_t *)d.warm_stop = 1u; d.warm_running = false; }
}

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
    HoldTimer ht(g_debug_slow);
    // the searches of the query set before this one are through (they read the arena and, before it, the staging buffer)
    HIP_TRY(hipStreamSynchronize(d.stream));
    ht.lap("queries: wait for the stream");
    const uint32_t nq = ctx->nq;
    // Every buffer of the query set is a slice of the device's arena (see Arena): sizes first, one slab, then the slices.  The
    // small INPUTS come first, in one run: they are put together in a page-locked staging buffer with the same offsets and
    // brought over by ONE kernel that reads the staging buffer in place -- no copy engine, nothing pageable: round 5 found the
    // six little hipMemcpyAsync from the context's std::vectors holding the caller for 8.6 ms of a 25-ms search (the runtime
    // stages a pageable copy on the caller's thread, and the copy engine was busy with the chunk coming in).
    const bool alt_ = first_pass_is_frame(ctx), q8_ = first_pass_is_q8(ctx);
    const uint32_t np_ = (uint32_t)ctx->pair_len.size();
    const size_t prof8 = (size_t)ctx->total_rowblocks * 32 * sizeof(uint2) + 4096, prof16 = (size_t)ctx->total_rowblocks * 32 * sizeof(uint4) + 4096;
    const size_t pair16 = (size_t)ctx->pair_rowblocks * 32 * sizeof(uint4) + 4096, pair8 = (size_t)ctx->pair_rowblocks * 32 * sizeof(uint2) + 4096;
    const size_t pages_bytes = (size_t)(128 + OSW_I16S_TABLE + 64) * 2 * sizeof(uint32_t);
    struct Slice { DevBuf *buf; size_t bytes; const void *src; size_t src_bytes; };
    const Slice slices[] = {// inputs (src: what the staging buffer holds at the slice's offset; top_pages is generated in place below)
                            {&d.queries, ctx->a.size() + 16, ctx->a.data(), ctx->a.size()},
                            {&d.qlen, nq * sizeof(uint16_t) + 16, ctx->m.data(), nq * sizeof(uint16_t)},
                            {&d.a_disp, (nq + 1) * sizeof(uint32_t), ctx->a_disp.data(), nq * sizeof(uint32_t)},
                            {&d.prof_off, (nq + 1) * sizeof(uint32_t), ctx->prof_off.data(), nq * sizeof(uint32_t)},
                            {&d.submat, 24 * 32, ctx->submat, 24 * 32},
                            {&d.pair_q, np_ ? 2 * np_ * sizeof(uint32_t) : 0, ctx->pair_q.data(), 2 * np_ * sizeof(uint32_t)},
                            {&d.pair_off, np_ ? np_ * sizeof(uint32_t) : 0, ctx->pair_off.data(), np_ * sizeof(uint32_t)},
                            {&d.pair_len, np_ ? np_ * sizeof(uint16_t) + 16 : 0, ctx->pair_len.data(), np_ * sizeof(uint16_t)},
                            {&d.pair_rows, np_ ? np_ * sizeof(uint16_t) + 16 : 0, ctx->pair_rows.data(), np_ * sizeof(uint16_t)},
                            {&d.tail_len, np_ ? np_ * OSW_TAIL_GEOMS * sizeof(uint16_t) + 16 : 0, ctx->tail_len.data(), np_ * OSW_TAIL_GEOMS * sizeof(uint16_t)},
                            {&d.tail_off, np_ ? np_ * OSW_TAIL_GEOMS * sizeof(uint32_t) + 16 : 0, ctx->tail_off.data(), np_ * OSW_TAIL_GEOMS * sizeof(uint32_t)},
                            {&d.top_pages, pages_bytes, nullptr, 0},
                            // built on the device
                            {&d.prof, prof8, nullptr, 0}, {&d.prof_seq, prof16, nullptr, 0}, {&d.prof_alt, prof8, nullptr, 0}, {&d.prof_seq_alt, alt_ ? prof16 : 0, nullptr, 0}, {&d.floor_i32, (size_t)OSW_I32F_TABLE * sizeof(uint2), nullptr, 0},
                            {&d.prof_pair, np_ ? pair16 : 0, nullptr, 0}, {&d.prof_pair8, np_ && q8_ ? pair8 : 0, nullptr, 0}, {&d.prof_pair_i16, np_ && alt_ ? pair16 : 0, nullptr, 0}};
    constexpr size_t kInputs = 12;
    size_t total = 0, inputs_bytes = 0;
    for (size_t i = 0; i < sizeof slices / sizeof slices[0]; ++i) { total += Arena::up(slices[i].bytes); if (i + 1 == kInputs) inputs_bytes = total; }
    if (total > d.qset.slab.cap) {
        for (const Slice &sl : slices) sl.buf->release(); // (views into the old slab)
        HIP_TRY(d.qset.slab.reserve(total + total / 2));
    }
    if (inputs_bytes > d.qstage_cap) {
        if (d.qstage) HIP_TRY(hipHostFree(d.qstage));
        d.qstage = nullptr;
        d.qstage_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&d.qstage, inputs_bytes + inputs_bytes / 2, hipHostMallocPortable));
        d.qstage_cap = inputs_bytes + inputs_bytes / 2;
    }
    d.qset.used = 0;
    memset(d.qstage, 0, inputs_bytes);
    uint32_t *pages = nullptr;
    for (const Slice &sl : slices) {
        const size_t off = d.qset.used;
        if (sl.bytes) sl.buf->assign(d.qset.take(sl.bytes), sl.bytes); else sl.buf->release();
        if (sl.src && sl.src_bytes) memcpy(d.qstage + off, sl.src, sl.src_bytes);
        if (sl.buf == &d.top_pages) pages = (uint32_t *)(d.qstage + off);
    }
    // constant "row above a first round": 64 {H,F} entries of zeros, 64 of the biased-int16 floor (1024), then the
    // column-frame cell's floor table, entry k = 1024 + k * ge (capped below the fp16 inf pattern), then 64 entries of the 8-bit
    // cell's "zero": its offset c in every byte (q8_cell.h)
    for (size_t i = 64; i < 128; ++i) pages[2 * i] = pages[2 * i + 1] = 0x04000400u;
    if (q8_)
        for (size_t i = 128 + OSW_I16S_TABLE; i < 128 + OSW_I16S_TABLE + 64; ++i) pages[2 * i] = pages[2 * i + 1] = (uint32_t)offset8_of(ctx) * 0x01010101u;
    for (size_t k = 0; k < OSW_I16S_TABLE; ++k) {
        const uint32_t v = (uint32_t)std::min<uint64_t>(1024ull + (uint64_t)k * (uint64_t)ctx->extend_gap, 0x7bffull);
        pages[2 * (128 + k)] = pages[2 * (128 + k) + 1] = v | (v << 16);
    }
    ht.lap("queries: buffers and staging");
    HIP_TRY(osw_launch_copy16(d.qstage, d.qset.slab.p, inputs_bytes, d.stream));
    // plain integer profile: the exact int32 kernel and the pair profiles read `prof`, the plain single-query int16 cell `prof_seq`
    HIP_TRY(osw_launch_build_profile((const uint8_t *)d.queries.p, (const uint32_t *)d.a_disp.p, (const uint16_t *)d.qlen.p,
                                     (const uint32_t *)d.prof_off.p, (const int8_t *)d.submat.p, nq, ctx->max_rowblocks,
                                     0, (uint2 *)d.prof.p, (uint4 *)d.prof_seq.p, d.stream));
    // the column-frame int16 cell reads S + ge
    // S + ge: the column-frame int16 cell, and the int32 cell of the re-run and of cell_bits = 32 (its floor table: built on the device too)
    const bool alt = first_pass_is_frame(ctx);
    HIP_TRY(osw_launch_build_profile((const uint8_t *)d.queries.p, (const uint32_t *)d.a_disp.p, (const uint16_t *)d.qlen.p,
                                     (const uint32_t *)d.prof_off.p, (const int8_t *)d.submat.p, nq, ctx->max_rowblocks,
                                     ctx->extend_gap, (uint2 *)d.prof_alt.p, alt ? (uint4 *)d.prof_seq_alt.p : nullptr, d.stream));
    HIP_TRY(osw_launch_floor_i32((uint2 *)d.floor_i32.p, OSW_I32F_TABLE, (uint32_t)ctx->extend_gap, d.stream));
    const uint32_t np = (uint32_t)ctx->pair_len.size();
    if (np > 0) {
        HIP_TRY(osw_launch_build_pair_profile((const uint2 *)(alt ? d.prof_alt.p : d.prof.p), (const uint32_t *)d.prof_off.p, (const uint16_t *)d.qlen.p,
                                              (const uint32_t *)d.pair_q.p, (const uint32_t *)d.pair_off.p, (const uint16_t *)d.pair_len.p, np,
                                              ctx->pair_max_rowblocks, alt /* column-frame pair cell: 32-bit integer sums */, (uint4 *)d.prof_pair.p, d.stream));
        if (first_pass_is_q8(ctx)) {
            HIP_TRY(osw_launch_build_pair_profile8((const uint2 *)d.prof.p, (const uint32_t *)d.prof_off.p, (const uint16_t *)d.qlen.p,
                                                   (const uint32_t *)d.pair_q.p, (const uint32_t *)d.pair_off.p, (const uint16_t *)d.pair_len.p, np,
                                                   ctx->pair_max_rowblocks, bias8_of(ctx), (uint2 *)d.prof_pair8.p, d.stream));
        }
        if (alt) { // plain int16 pair profile for the items the first-pass cell hands to the plain cell
            HIP_TRY(osw_launch_build_pair_profile((const uint2 *)d.prof.p, (const uint32_t *)d.prof_off.p, (const uint16_t *)d.qlen.p,
                                                  (const uint32_t *)d.pair_q.p, (const uint32_t *)d.pair_off.p, (const uint16_t *)d.pair_len.p, np,
                                                  ctx->pair_max_rowblocks, false, (uint4 *)d.prof_pair_i16.p, d.stream));
        }
    }
    // (nothing here waits for the device: the kernels read the staging buffer, which is this device's own and is written again
    // only behind the stream synchronisation at the top; the context's vectors may change as soon as this returns)
    ht.lap("queries: staging copy and profile kernels queued");
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
            float dur = 0, gap = 0, since = 0;
            if (k == 0 && d.mark_set && hipEventElapsedTime(&since, d.ev_mark, d.ev_used[0].a) == hipSuccess)
                fprintf(stderr, "[oswald_hip] search 0 started on the device %.3f ms after the pass's first upload was issued\n", since);
            (void)hipGetLastError();
            (void)hipEventElapsedTime(&dur, d.ev_used[k].a, d.ev_used[k].b);
            if (k + 1 < d.ev_used.size()) (void)hipEventElapsedTime(&gap, d.ev_used[k].b, d.ev_used[k + 1].a);
            fprintf(stderr, "[oswald_hip] search %zu: %.3f ms on the device, %.3f ms to the next search's start\n", k, dur, gap);
        }
    }
    d.mark_set = false;
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
int topr_after_search(oswald_hip_ctx *ctx, Device &d, Chunk &c, const int32_t *scores = nullptr, uint32_t stride = 0, bool searched = false)
{
    // (scores / stride: where the chunk's scores are -- its own table, or its columns of the table of a search over several resident chunks)
    if (!scores) { scores = (const int32_t *)c.scores.p; stride = c.score_stride; searched = c.nitems + c.nitems_wg + c.nitems_q + c.nitems_q_wg != 0; }
    if (ctx->topr_r == 0 || !c.has_index || ctx->nq == 0) return 0;
    const uint32_t r = ctx->topr_r;
    if (ctx->topr_queries_version != ctx->queries_version)
        return fail(OSWALD_HIP_ESTATE, "the query set changed since oswald_hip_topr_begin: call it again before searching");
    if (c.nvalid > c.ngroups * c.W) return fail(OSWALD_HIP_EINVAL, "chunk index: nvalid %u exceeds the chunk's %u lanes", c.nvalid, c.ngroups * c.W);
    if (!searched || c.nvalid == 0) return 0; // nothing was searched: nothing to add
    if (!d.top_run[0].p || !d.top_run[1].p) return fail(OSWALD_HIP_ESTATE, "oswald_hip_topr_begin has not prepared device %d", d.id);
    if (c.map_pending) { HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_map[c.map_cur], 0)); c.map_pending = false; } // the chunk's index map has landed
    HIP_TRY(d.topr_cand.reserve((size_t)ctx->nq * osw_topr_parts(c.nvalid) * r * sizeof(unsigned long long)));
    HIP_TRY(osw_launch_topr_fold_chunk(scores, stride, c.nvalid, r, ctx->nq,
                                       c.index_map ? (const uint32_t *)c.index_map_dev[c.map_cur].p : nullptr, c.first_index, (unsigned long long *)d.topr_cand.p,
                                       (const unsigned long long *)d.top_run[d.top_cur].p, (unsigned long long *)d.top_run[d.top_cur ^ 1].p, d.stream));
    if (c.index_map) {
        const int k = c.map_cur;
        if (!c.ev_map_read[k]) HIP_TRY(hipEventCreateWithFlags(&c.ev_map_read[k], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c.ev_map_read[k], d.stream));
        c.map_read_pending[k] = true;
    }
    d.top_cur ^= 1;
    d.top_any = true;
    return 0;
}

// THE merge of top lists (the reference's order, utils.c:3-86: descending score, equal scores by DESCENDING database
// index = descending key score << 32 | index): K candidates per query, score < 0 = empty slot, -> the r best.
void merge_candidates(uint32_t nq, size_t K, const int32_t *cs, const uint32_t *ci, uint32_t r, int32_t *out_s, uint32_t *out_i)
{
    std::vector<uint64_t> keys;
    keys.reserve(K); // (one allocation; a K no memory can hold is std::length_error / std::bad_alloc here, before anything is read)
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
static int host_alloc_impl(size_t bytes, void **ptr)
{
    if (!ptr) return fail(OSWALD_HIP_EINVAL, "null out-pointer");
    *ptr = nullptr;
    if (bytes == 0) return 0;
    HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocPortable));
    return 0;
}

static int host_free_impl(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return 0;
}

// Page-locks memory the caller already has (a mapped or loaded database): uploads from it are asynchronous DMA.
static int host_register_impl(void *ptr, size_t bytes)
{
    if (!ptr || bytes == 0) return fail(OSWALD_HIP_EINVAL, "nothing to register");
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? OSWALD_HIP_ENOMEM : OSWALD_HIP_ERUNTIME, "hipHostRegister(%p, %zu bytes): %s (the memory stays pageable; uploads from it still work)", ptr, bytes,
                    hipGetErrorString(e));
    }
    return 0;
}

static int host_unregister_impl(void *ptr)
{
    if (!ptr) return 0;
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(OSWALD_HIP_ERUNTIME, "hipHostUnregister(%p): %s", ptr, hipGetErrorString(e)); }
    return 0;
}

static int device_count_impl(int *count)
{
    if (!count) return fail(OSWALD_HIP_EINVAL, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(OSWALD_HIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return 0;
}

static int init_impl(int ndev, const int *device_ids, oswald_hip_ctx **out)
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
    struct HookOff { size_t keep; HookOff() : keep(g_fail_alloc_above) { g_fail_alloc_above = 0; } ~HookOff() { g_fail_alloc_above = keep; } } hook_off; // (the test hook "device is full" spares bring-up)
    ctx->profiling = g_debug_slow; // (OSWALD_HIP_DEBUG_SLOW: the device's time line of every pass, see drain_events)
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
        r = d.qset.slab.reserve(16u << 20); // the query sets' arena: the BASELINE set needs 8 MB
        if (r == hipSuccess) r = d.counters.reserve((OSW_CTR_BLOCKS * OSW_CTR_COUNT + 8) * sizeof(uint32_t));
        if (r != hipSuccess) { delete ctx; return fail(OSWALD_HIP_ENOMEM, "device %d: %s", d.id, hipGetErrorString(r)); }
        // The device's own working memory -- the spill scratch of the strip boundaries, a region per resident wave -- is part of the
        // bring-up (round 6): it belongs to the kernels like the FPGA kernel's on-chip row buffers belong to the bitstream init() loads
        // (utils.c:99-173), it does not depend on the inputs (sized for the longest block a region ever holds: 4 096 columns at two
        // lane groups; longer blocks run at narrower lane groups), and nothing the reference creates inside its clock
        // (FPGAsearch.c:85-96: queries, chunk arrays, profiles, scores) corresponds to it.  What does -- the chunk slots -- is made
        // inside the caller's clock (oswald_hip_reserve_chunks, or the first uploads).
        if (ensure_scratch(d, 4096) != 0) { const std::string why = g_err; delete ctx; return fail(OSWALD_HIP_ENOMEM, "device %d: spill scratch: %s", d.id, why.c_str()); }
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
            if (r == hipSuccess) r = osw_launch_s16qt(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_pk16qt(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_s16(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_pk16q(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_pk16(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_i32(a, 1, d.stream);
            if (r == hipSuccess) r = osw_launch_i32r(a, 16, d.stream); // (regions: one workgroup of the pipeline)
            if (r == hipSuccess) r = osw_launch_q8(a, 1, d.stream);
            if (r == hipSuccess) r = osw_warm_aux_kernels(d.stream);
            if (r == hipSuccess) r = osw_launch_copy16(nullptr, nullptr, 0, d.stream_up);
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
    if (ctx->tun.warm_ms > 0) {
        for (Device &d : ctx->dev) {
            if (&ctx->dev[d.leader] != &d) continue; // (one per GPU)
            hipError_t r = hipSetDevice(d.id);
            if (r == hipSuccess) r = hipHostMalloc((void **)&d.warm_stop, 64, hipHostMallocPortable);
            if (r == hipSuccess) { *d.warm_stop = 0u; r = hipStreamCreateWithFlags(&d.stream_warm, hipStreamNonBlocking); }
            void *dp = nullptr;
            if (r == hipSuccess) r = hipHostGetDevicePointer(&dp, d.warm_stop, 0);
            if (r == hipSuccess) r = osw_launch_spin((uint32_t *)d.counters.p, d.grid, ctx->tun.warm_ms, (const uint32_t *)dp, d.stream_warm);
            if (r == hipSuccess) d.warm_running = true;
            else (void)hipGetLastError(); // (a warm-up that cannot be started is no reason to fail the bring-up)
        }
    }
    *out = ctx;
    return 0;
}

static void free_group(Device &d);

static int finalize_impl(oswald_hip_ctx *ctx)
{
    if (!ctx) return 0;
    stop_warm(ctx);
    for (Device &d : ctx->dev) {
        (void)hipSetDevice(d.id);
        if (d.stream_warm) { (void)hipStreamSynchronize(d.stream_warm); (void)hipStreamDestroy(d.stream_warm); d.stream_warm = nullptr; }
        if (d.warm_stop) { (void)hipHostFree(d.warm_stop); d.warm_stop = nullptr; }
        for (hipStream_t st : {d.stream, d.stream2, d.stream_copy, d.stream_up, d.stream_down}) if (st) (void)hipStreamSynchronize(st);
        free_group(d);
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
            for (int k = 0; k < 2; ++k) { if (c.ev_set_read[k]) (void)hipEventDestroy(c.ev_set_read[k]); c.ev_set_read[k] = nullptr; c.set_read_pending[k] = false; }
            if (c.blocks_pin) (void)hipHostFree(c.blocks_pin);
            c.ev_up = c.ev_use = c.ev_copy = nullptr;
            c.sub_cols = nullptr;
            c.sub_cols_cap = 0;
            c.blocks_pin = nullptr;
            c.blocks_pin_cap = 0;
            c.st_b.release(); c.nd_pin.release(); c.map_pin[0].release(); c.map_pin[1].release();
        }
        if (d.comm) { (void)ncclCommDestroy(d.comm); d.comm = nullptr; }
        if (&d == &ctx->dev[0] && ctx->pcomm) { (void)ncclCommDestroy(ctx->pcomm); ctx->pcomm = nullptr; }
        for (Chunk &c : d.chunks) { c.tiled.release(); c.blocks.release(); c.sub_cols_buf.release(); c.scores.release(); c.ovf.release(); c.ovf8.release(); c.index_map_dev[0].release(); c.index_map_dev[1].release(); c.slab.release(); for (int k = 0; k < 2; ++k) { if (c.ev_map[k]) (void)hipEventDestroy(c.ev_map[k]); c.ev_map[k] = nullptr; if (c.ev_map_read[k]) (void)hipEventDestroy(c.ev_map_read[k]); c.ev_map_read[k] = nullptr; } }
        for (DevBuf *b : {&d.queries, &d.qlen, &d.a_disp, &d.prof_off, &d.prof, &d.prof_alt, &d.prof_seq, &d.prof_seq_alt, &d.prof_pair_i16, &d.pair_q, &d.pair_off, &d.pair_len, &d.prof_pair, &d.submat, &d.bnd, &d.counters,
                          &d.topr_scores, &d.topr_index, &d.topr_cand, &d.wg_times, &d.scores_packed, &d.top_pages, &d.prof_pair8, &d.floor_i32, &d.pair_rows, &d.tail_len, &d.tail_off,
                          &d.top_run[0], &d.top_run[1], &d.top_gather, &d.top_final})
            b->release();
        d.qset.slab.release();
        if (d.qstage) (void)hipHostFree(d.qstage);
        d.qstage = nullptr;
        if (d.ev_top) (void)hipEventDestroy(d.ev_top);
        if (d.ev_mark) (void)hipEventDestroy(d.ev_mark);
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

static int info_impl(oswald_hip_ctx *ctx, int dev, char *buf, size_t buflen)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!buf || buflen == 0) return fail(OSWALD_HIP_EINVAL, "buffer is null");
    const Device &d = ctx->dev[dev];
    char bus[32] = "unknown";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, d.id) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "unknown"); }
    snprintf(buf, buflen,
             "Device %d: %s (%s)\n"
             "  PCI bus id:               %s\n"
             "  compute units:            %d\n"
             "  max clock:                %d MHz\n"
             "  global memory:            %zu MiB\n"
             "  LDS per workgroup:        %zu KiB\n"
             "  wavefront size:           %d\n"
             "  L2 cache:                 %d KiB\n"
             "  persistent workgroups:    %u x %d threads\n",
             d.id, d.prop.name, d.prop.gcnArchName, bus, d.prop.multiProcessorCount, d.prop.clockRate / 1000,
             d.prop.totalGlobalMem >> 20, d.prop.sharedMemPerBlock >> 10, d.prop.warpSize, d.prop.l2CacheSize >> 10, d.grid,
             OSW_WG_THREADS);
    return 0;
}

static int set_scoring_impl(oswald_hip_ctx *ctx, const int8_t *submat, int open_gap, int extend_gap, int cell_bits)
{
    if (!ctx || !submat) return fail(OSWALD_HIP_EINVAL, "null argument");
    if (open_gap < 0 || extend_gap < 0) return fail(OSWALD_HIP_EINVAL, "gap penalties must be >= 0");
    stop_warm(ctx);
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

static int set_queries_impl(oswald_hip_ctx *ctx, const uint8_t *a, uint64_t Q, const uint16_t *m, const uint32_t *a_disp, uint32_t nq)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (nq > 0 && (!m || !a_disp || (Q > 0 && !a))) return fail(OSWALD_HIP_EINVAL, "null query arrays");
    stop_warm(ctx);
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

// One slot's worth of an upload: groups [0, ngroups) of the arrays given, whose displacements are `disp_bias` too large (the
// rest of a chunk the library cut in two keeps the caller's displacement table: its bytes start at b, its table at the entry
// of its first group, and the device-side base pointer is moved back by the bias instead of the table being rebased).
// -> the slot's index in *slot_out.
static int upload_slot(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp, uint32_t disp_bias,
                       uint32_t ngroups, uint32_t W, int *slot_out)
{
    for (uint32_t g = 0; g < ngroups; ++g)
        if (disp[g] < disp_bias || (uint64_t)(disp[g] - disp_bias) + (uint64_t)n[g] * W > vD)
            return fail(OSWALD_HIP_EINVAL, "group %u runs past the chunk (disp %u + %u*%u > %llu)", g, disp[g] - disp_bias, n[g], W, (unsigned long long)vD);
    Device &d = ctx->dev[dev];
    // A free slot whose buffers are large enough already (the smallest such), else -- before a new one is opened -- the
    // LARGEST free slot, which then grows: oswald_hip_max_chunk_size reckons with three resident chunks per device (four slots
    // with a first chunk cut in two), and slots are never given back before finalize.  Growing a buffer frees the
    // old one, and hipFree waits for the device -- i.e. for whatever search is running beside this upload: a device gets there
    // only when its caller sends chunks of growing size.
    const size_t need_scores = (size_t)ctx->nq * ((ngroups + OSW_BLOCK_SEQS / W - 1) / (OSW_BLOCK_SEQS / W)) * OSW_BLOCK_SEQS * sizeof(int32_t);
    int slot = -1, largest_free = -1;
    for (size_t i = 0; i < d.chunks.size(); ++i) {
        const Chunk &k = d.chunks[i];
        if (k.live || k.upload_pending) continue;
        if (k.st_b.p && (largest_free < 0 || k.st_b.cap > d.chunks[largest_free].st_b.cap)) largest_free = (int)i;
        if (k.st_b.cap < vD + 64) continue;
        if (k.scores.cap && k.scores.cap < need_scores + 16) continue; // (its score table and re-run queues would have to grow too)
        if (slot < 0 || k.st_b.cap < d.chunks[slot].st_b.cap) slot = (int)i;
    }
    if (slot < 0)
        for (size_t i = 0; i < d.chunks.size(); ++i) if (!d.chunks[i].live && !d.chunks[i].upload_pending && !d.chunks[i].st_b.p) { slot = (int)i; break; } // (never used)
    if (slot < 0 && largest_free >= 0 && d.chunks.size() >= OSW_MAX_SLOTS) slot = largest_free; // grows below
    if (slot < 0) { d.chunks.emplace_back(); slot = (int)d.chunks.size() - 1; }
    Chunk &c = d.chunks[slot];
    PhaseTimer pt(ctx->tun.debug_phases);
    const uint32_t gpb = OSW_BLOCK_SEQS / W;
    c.ngroups = ngroups;
    c.W = W;
    c.nblocks = (ngroups + gpb - 1) / gpb;
    c.score_stride = c.nblocks * OSW_BLOCK_SEQS;
    c.ncols4_alloc.assign(c.nblocks, 0);
    c.next = -1;
    c.is_cont = false;
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
        // entry (G - 1) + sigma = the longest lane of sub-block sigma at geometry G: a binary heap over the 64 lanes, filled
        // from the leaves (G = 64: entries 63 .. 126) up -- entry t covers the entries 2t + 1 and 2t + 2
        uint16_t *e = c.sub_cols_est.data() + (size_t)B * 128;
        for (uint32_t l = 0; l < 64; ++l) e[63 + l] = lane_n[l];
        for (int t = 62; t >= 0; --t) e[t] = std::max(e[2 * t + 1], e[2 * t + 2]);
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
    const size_t disp_off = ((size_t)ngroups * sizeof(uint16_t) + 63) & ~(size_t)63;
    HIP_TRY(c.nd_pin.reserve(disp_off + (size_t)ngroups * sizeof(uint32_t) + 64));
    if (ngroups) { memcpy(c.nd_pin.p, n, (size_t)ngroups * sizeof(uint16_t)); memcpy((char *)c.nd_pin.p + disp_off, disp, (size_t)ngroups * sizeof(uint32_t)); }
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
    if (g_debug_slow && !d.mark_set) {
        if (!d.ev_mark) HIP_TRY(hipEventCreate(&d.ev_mark));
        HIP_TRY(hipEventRecord(d.ev_mark, d.stream_copy));
        d.mark_set = true;
    }
    HoldTimer ht(g_debug_slow);
    if (ngroups > 0) {
        HIP_TRY(hipMemcpyAsync(c.st_b.p, b, vD, hipMemcpyHostToDevice, d.stream_copy));
        ht.lap("upload: copy of the residues queued");
        HIP_TRY(hipEventRecord(c.ev_copy, d.stream_copy));
        ht.lap("upload: event behind the copy recorded");
        HIP_TRY(hipStreamWaitEvent(up, c.ev_copy, 0));
        ht.lap("upload: upload stream made to wait for the copy");
        // (the block table comes over by a kernel that reads its page-locked source in place, and the live extents go back the same
        // way: the upload stream carries kernels only.  Round 5: a small hipMemcpyAsync on this stream -- behind a re-tile that is
        // waiting for a wave slot -- held the caller for 7 - 9 ms in a process's first pass)
        HIP_TRY(osw_launch_copy16(c.blocks_pin, c.blocks.p, c.nblocks * sizeof(OswBlock), up));
        ht.lap("upload: block table queued");
        // (the all-dummy columns around every block are written by the re-tile itself: osw_write_pads)
        // (the base pointer moved back by the bias: base + disp[g] is the group's place in the staging copy)
        // (n[] and disp[] are read where they are: the slot's page-locked copy)
        HIP_TRY(osw_launch_retile((const uint8_t *)c.st_b.p - disp_bias, (const uint16_t *)c.nd_pin.p, (const uint32_t *)((const char *)c.nd_pin.p + disp_off),
                                  ngroups, W, (OswBlock *)c.blocks.p, c.nblocks, (uint16_t *)c.tiled.p, (uint16_t *)c.sub_cols_buf.p, up));
    }
    ht.lap("upload: re-tile queued");
    if (pt.on) { HIP_TRY(hipStreamSynchronize(up)); pt.lap("upload: H2D + re-tile"); }
    if (c.nblocks) HIP_TRY(osw_launch_copy16(c.sub_cols_dev(), c.sub_cols, (size_t)c.nblocks * 128 * sizeof(uint16_t), up)); // device -> the page-locked host copy, written by the kernel
    HIP_TRY(hipEventRecord(c.ev_up, up));
    ht.lap("upload: live extents' way back queued");
    c.up_seq = ++d.up_seq;
    c.items_version = ~0ull;
    c.searched = false;
    c.has_index = false;
    c.index_map = false;
    c.live = true;
    c.upload_pending = true;
    *slot_out = slot;
    return 0;
}

// No search and no upload of this device is still on its way: a chunk that arrives now is what the device will wait for.
// (Asked of the chunks' own events, not of the streams: a few microseconds of housekeeping queued on a stream -- the memset of
// oswald_hip_topr_begin -- must not count.)
static bool device_is_idle(Device &d)
{
    for (Chunk &c : d.chunks) {
        if (c.use_pending && c.ev_use) {
            if (hipEventQuery(c.ev_use) != hipSuccess) { (void)hipGetLastError(); return false; }
            c.use_pending = false;
        }
        if (c.upload_pending && c.ev_up && hipEventQuery(c.ev_up) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    return true;
}

static int upload_pieces(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                         uint32_t ngroups, uint32_t W, int *chunk, bool async)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!chunk) return fail(OSWALD_HIP_EINVAL, "chunk out-pointer is null");
    if (W != 16 && W != 32 && W != 64 && W != 128) return fail(OSWALD_HIP_EINVAL, "lane_width must be 16, 32, 64 or 128");
    if (ngroups > 0 && (!b || !n || !disp)) return fail(OSWALD_HIP_EINVAL, "null chunk arrays");
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    stop_warm(ctx);
    // The first chunk of a search is what nothing can hide: its copy (2.3 ms per 128 MiB over PCIe Gen5), its re-tile and the
    // plan of its search all happen before the device has anything to do.  An asynchronous upload of some size that finds its
    // device idle is therefore cut into a HEAD of whole 128-sequence blocks and the REST: the head is in the device after a
    // fraction of the copy, and the rest comes in, is re-tiled and planned while the head is being searched.  How much the
    // head must hold for its search to cover the rest's copy depends on the query set: a search reads the database at
    // ~(sum of the query lengths / 200) times the rate of the link (one 375-residue query: 28 GB/s of residues against
    // 57 GB/s of copy; twenty queries of 100 .. 1000: 1 GB/s) -- head = 1 / (1 + that ratio) of the bytes, at least a twelfth
    // (the rest's re-tile and plan, ~2 ms, want hiding too), at most half.  (Round 4 had this in the CLI; every caller of
    // the C ABI gets it now.)  The reference uploads and searches in turn (FPGAsearch.c:180-223).
    const uint32_t gpb = OSW_BLOCK_SEQS / W;
    uint32_t head_groups = 0;
    if (async && ctx->tun.split_bytes && vD >= ctx->tun.split_bytes && ngroups >= 3 * gpb && disp[0] == 0 && device_is_idle(d)) {
        double sum_m = 0, max_m = 0;
        if (ctx->have_queries) for (uint16_t m : ctx->m) { sum_m += m; max_m = std::max<double>(max_m, m); }
        const double frac = std::min(0.5, std::max(1.0 / 12.0, sum_m > 0 ? 1.0 / (1.0 + sum_m / 200.0) : 1.0 / 12.0));
        // ... and two searches instead of one end twice: the last items of a launch run on a GPU that is emptying, and an item --
        // a query against a block -- lasts as long as the query is long (one 5000-residue query against 100 000 sequences, 38 MB:
        // cut in two the pass took 22.4 instead of 19.4 ms for 0.6 ms of copy hidden).  The cut is made only where what it hides --
        // the rest's share of the copy at the link's ~57 GB/s -- clearly exceeds two such endings
        // (~ longest query x mean columns x 6.5 instructions x 4 cycles x 3 waves per SIMD / 128 cells, at 2.3 GHz).
        const double mean_cols = (double)vD / ((double)ngroups * W);
        const double gain_ms = (1.0 - frac) * (double)vD / 57.0e6, loss_ms = 2.0 * max_m * mean_cols * 6.5 * 4.0 * 3.0 / 128.0 / 2.3e6 * (ctx->nq > 1 ? 2.0 : 1.0);
        const bool pays = gain_ms >= 1.5 * loss_ms || ctx->tun.split_bytes < (1u << 20); // (a split size below 1 MiB is the tests' hook: always)
        uint32_t g = gpb;
        while (g + gpb < ngroups && (double)disp[g + gpb] <= frac * (double)vD) g += gpb;
        // (the reference's layout: the groups back to back in order -- anything else is uploaded in one piece)
        bool in_order = true;
        for (uint32_t k = 0; k + 1 < ngroups && in_order; ++k) in_order = (uint64_t)disp[k] + (uint64_t)n[k] * W <= disp[k + 1];
        if (pays && in_order && g + gpb <= ngroups && disp[g] > 0 && disp[g] < vD) head_groups = g;
    }
    if (head_groups == 0) {
        int slot = -1;
        if (int r = upload_slot(ctx, dev, b, vD, n, disp, 0, ngroups, W, &slot)) return r;
        *chunk = slot;
        if (!async) {
            if (int r = finish_upload(d, d.chunks[slot])) return r; // caller's buffers are free again (reference: clFinish, FPGAsearch.c:197)
        }
        return 0;
    }
    int head = -1, rest = -1;
    const uint32_t cut = disp[head_groups];
    if (int r = upload_slot(ctx, dev, b, cut, n, disp, 0, head_groups, W, &head)) return r;
    if (int r = upload_slot(ctx, dev, b + cut, vD - cut, n + head_groups, disp + head_groups, cut, ngroups - head_groups, W, &rest)) {
        // the head's copy from the caller's `b` is queued: it must have landed before the caller -- who sees a failed call -- may free
        // the buffer; then the slot is free again
        const std::string why = g_err;
        (void)finish_upload(d, d.chunks[head]);
        d.chunks[head].live = false;
        g_err = why;
        return r;
    }
    d.chunks[head].next = rest;
    d.chunks[rest].is_cont = true;
    *chunk = head;
    if (ctx->tun.debug_phases || g_debug_slow)
        fprintf(stderr, "[oswald_hip] upload cut in two: head %u groups / %u bytes (slot %d), rest %u groups / %llu bytes (slot %d)\n", head_groups, cut, head, ngroups - head_groups,
                (unsigned long long)(vD - cut), rest);
    return 0;
}

static int chunk_upload_impl(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                            uint32_t ngroups, uint32_t W, int *chunk)
{
    return upload_pieces(ctx, dev, b, vD, n, disp, ngroups, W, chunk, false);
}

static int chunk_upload_async_impl(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp,
                                  uint32_t ngroups, uint32_t W, int *chunk)
{
    return upload_pieces(ctx, dev, b, vD, n, disp, ngroups, W, chunk, true);
}

static int reserve_impl(oswald_hip_ctx *ctx, int dev, uint32_t max_sequence_length)
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

// The buffers of `slots` chunk slots for chunks of up to chunk_bytes bytes in ngroups groups, allocated now -- before the
// caller's clock starts -- instead of by the first uploads and searches: mapping a few hundred MB of device memory takes
// milliseconds (20 ms per GB on the round-5 box: 9 ms of a 27-ms one-query search at 1 M sequences went into the first
// upload's allocations).  A hint: a chunk that needs more grows its slot as before.
// what: 1 = the slots' page-locked HOST staging (work queues, live extents, block table, n[] / disp[]), 2 = their DEVICE buffers
static int reserve_slots(oswald_hip_ctx *ctx, int dev, uint64_t chunk_bytes, uint32_t ngroups, uint32_t W, uint32_t nq, uint32_t slots, int what)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (dev >= (int)ctx->dev.size()) return fail(OSWALD_HIP_ENODEV, "device index %d out of range", dev);
    if (W != 16 && W != 32 && W != 64 && W != 128) return fail(OSWALD_HIP_EINVAL, "lane_width must be 16, 32, 64 or 128");
    if (chunk_bytes > 0xfff00000ull) return fail(OSWALD_HIP_EINVAL, "a chunk holds at most %llu bytes", 0xfff00000ull);
    slots = std::min<uint32_t>(slots, OSW_MAX_SLOTS);
    const uint32_t gpb = OSW_BLOCK_SEQS / W, nblocks = (ngroups + gpb - 1) / gpb;
    if (what & 2) stop_warm(ctx); // (the host staging is made before a caller's clock starts: the warm-up goes on beside it)
    for (int i = 0; i < (int)ctx->dev.size(); ++i) {
        if (dev >= 0 && i != dev) continue;
        Device &d = ctx->dev[i];
        HIP_TRY(hipSetDevice(d.id));
        // `slots` FREE slots (slots that hold a chunk -- a caller's resident chunks -- do not count)
        uint32_t made = 0;
        for (size_t k = 0; made < slots; ++k) {
            if (k >= d.chunks.size()) d.chunks.emplace_back();
            Chunk &c = d.chunks[k];
            if (c.live || c.upload_pending) continue;
            ++made;
            if (what & 2) {
                // re-tiled residues: a byte per residue of every block padded to its longest group (sorted databases: ~1.02 x the chunk)
                // + the all-dummy column groups around every block
                const uint64_t col4 = (chunk_bytes + chunk_bytes / 16) / 512 + (uint64_t)(nblocks + 1) * OSW_TILED_PAD_GROUPS + OSW_TILED_TAIL_GROUPS;
                struct Want { DevBuf *buf; size_t bytes; };
                const Want wants[] = {{&c.tiled, (size_t)(col4 * 64 * sizeof(uint2))}, {&c.blocks, nblocks * sizeof(OswBlock) + 16}, {&c.sub_cols_buf, (size_t)nblocks * 128 * sizeof(uint16_t) + 16},
                                      {&c.st_b, (size_t)chunk_bytes + 64}, {&c.scores, nq ? (size_t)nq * nblocks * OSW_BLOCK_SEQS * sizeof(int32_t) + 16 : 0},
                                      {&c.ovf, nq ? ((size_t)nq * nblocks * 128 + (size_t)(nq / 2) * nblocks * 128) * sizeof(uint2) + 16 : 0}};
                size_t need = 0;
                for (const Want &w : wants) if (w.bytes > w.buf->cap) need += Arena::up(w.bytes + w.bytes / 8 + 256);
                if (need > 0 && !c.slab.p) {
                    // a fresh slot: one slab, the buffers are slices of it (with the head room their own allocations would have had)
                    HIP_TRY(c.slab.reserve(need));
                    size_t used = 0;
                    for (const Want &w : wants)
                        if (w.bytes > w.buf->cap) { const size_t sz = Arena::up(w.bytes + w.bytes / 8 + 256); w.buf->assign((char *)c.slab.p + used, sz); used += sz; }
                } else {
                    for (const Want &w : wants) if (w.bytes) HIP_TRY(w.buf->reserve(w.bytes)); // (a slot that has buffers: each grows by itself)
                }
            }
            if (what & 1) {
                HIP_TRY(c.nd_pin.reserve((size_t)ngroups * 6 + 192));
                if ((size_t)nblocks * 128 > c.sub_cols_cap) {
                    if (c.sub_cols) HIP_TRY(hipHostFree(c.sub_cols));
                    c.sub_cols = nullptr;
                    c.sub_cols_cap = 0;
                    const size_t want = (size_t)nblocks * 128 + (size_t)nblocks * 16 + 128;
                    HIP_TRY(hipHostMalloc((void **)&c.sub_cols, want * sizeof(uint16_t), hipHostMallocPortable));
                    c.sub_cols_cap = want;
                }
                if (nblocks > c.blocks_pin_cap) {
                    if (c.blocks_pin) HIP_TRY(hipHostFree(c.blocks_pin));
                    c.blocks_pin = nullptr;
                    c.blocks_pin_cap = 0;
                    const size_t want = (size_t)nblocks + nblocks / 8 + 16;
                    HIP_TRY(hipHostMalloc((void **)&c.blocks_pin, want * sizeof(OswBlock), hipHostMallocPortable));
                    c.blocks_pin_cap = want;
                }
                if (nq > 0) {
                    // work queues: an entry of 8 B per (entity, sub-block) -- a few per block and query
                    const size_t entries = (size_t)nblocks * (nq + 1) * 16 + 4096;
                    for (int set = 0; set < 2; ++set) {
                        if (entries > c.items_pin_cap[set]) {
                            if (c.items_pin[set]) HIP_TRY(hipHostFree(c.items_pin[set]));
                            c.items_pin[set] = nullptr;
                            c.items_pin_cap[set] = 0;
                            HIP_TRY(hipHostMalloc((void **)&c.items_pin[set], entries * sizeof(uint2), hipHostMallocPortable));
                            c.items_pin_cap[set] = entries;
                        }
                    }
                }
                if (!c.ev_up) HIP_TRY(hipEventCreateWithFlags(&c.ev_up, hipEventDisableTiming));
                if (!c.ev_use) HIP_TRY(hipEventCreateWithFlags(&c.ev_use, hipEventDisableTiming));
                if (!c.ev_copy) HIP_TRY(hipEventCreateWithFlags(&c.ev_copy, hipEventDisableTiming));
                if (!c.ev_down) HIP_TRY(hipEventCreateWithFlags(&c.ev_down, hipEventDisableTiming));
            }
        }
    }
    return 0;
}

// The buffers of `slots` chunk slots for chunks of up to chunk_bytes bytes in ngroups groups, made in one place instead of by the
// first uploads and searches as they come.  The reference does this in two places: its HOST buffers -- the score tables and the
// profiles' staging, page-aligned for DMA -- before its clock starts (posix_memalign, FPGAsearch.c:69-74, tick at :80), its six
// DEVICE buffers behind the tick (clCreateBuffer, :85-96).  oswald_hip_reserve_host is the former: the slots' page-locked staging
// (work queues, live extents, block tables, copies of n[] / disp[]: pinning host pages costs ~0.15 ms per MB, tools/alloc_probe.hip);
// oswald_hip_reserve_chunks makes both -- whatever is not there yet: a caller that mirrors the reference's clock calls _host
// before it and _chunks inside (device memory: microseconds per buffer on this runtime).  Hints: a chunk that needs more grows its slot.
static int reserve_chunks_impl(oswald_hip_ctx *ctx, int dev, uint64_t chunk_bytes, uint32_t ngroups, uint32_t W, uint32_t nq, uint32_t slots)
{
    return reserve_slots(ctx, dev, chunk_bytes, ngroups, W, nq, slots, 3);
}

static int reserve_host_impl(oswald_hip_ctx *ctx, int dev, uint32_t ngroups, uint32_t W, uint32_t nq, uint32_t slots)
{
    return reserve_slots(ctx, dev, 0, ngroups, W, nq, slots, 1);
}

// Device memory one byte of chunk (one padded residue of the interleaved groups) takes, worst case, for each of the
// chunks' worth of slots a device keeps (three resident chunks -- one searched while the next two come in -- and the pieces of a first
// chunk the library cut in two, OSW_MAX_SLOTS): the staging copy of the upload (1), the
// re-tiled residues (a 128-sequence block is padded to its longest group: <= 1.25), the all-dummy columns behind every
// block (18 x 512 B per block of >= 128 x 28 B: 2.6), and per sequence -- at most one per 28 bytes, the shortest
// padded group length -- 4 B of score, 8 B of int32 re-run queue and 8 B of int16 re-run queue per query.
static int max_chunk_size_impl(oswald_hip_ctx *ctx, int dev, uint32_t nq, uint32_t max_sequence_length, uint64_t *bytes)
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
    // (four chunks' worth: slots are kept, and after a first chunk cut in two -- head and rest fit no full chunk -- three more are opened)
    const double per_byte = 4.1 * (1.0 + 1.25 + 2.6 + 20.0 * (double)std::max(nq, 1u) / 28.0);
    const uint64_t fit = (uint64_t)((double)usable / per_byte);
    *bytes = std::min<uint64_t>(fit, 0xfff00000ull); // (column offsets inside a chunk are 32-bit)
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// A chunk search in three steps, all of them queueing only: PLAN (queries / profiles in place, work queues planned and on
// their way, the search stream ordered behind the chunk's upload and its queues), LAUNCH (the DP kernels of the first-pass
// arithmetic, the re-run tiers, the chunk's top list), DOWNLOAD (the score table to the caller, on the download stream).
// ---------------------------------------------------------------------------------------------------------------
static int search_plan(oswald_hip_ctx *ctx, Device &d, Chunk &c, PhaseTimer &pt, HoldTimer &ht)
{
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
    if (int r = sync_queries(ctx, d)) return r;
    pt.lap("search: queries + profiles");
    ht.lap("search: queries + profiles");
    if (int r = build_items(ctx, d, c, landed && !ctx->tun.plan_on_estimates)) return r;
    pt.lap("search: work-queue plan");
    ht.lap("search: work-queue plan (incl. the queues' copy)");
    if (!landed) HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_up, 0)); // everything queued below finds the chunk in place
    ht.lap("search: stream wait for the upload");
    return 0;
}

// table_dev: the caller's score table as the device addresses it (columns of this slot: already offset), or null
static int search_launch(oswald_hip_ctx *ctx, Device &d, Chunk &c, HoldTimer &ht, int32_t *table_dev, size_t table_stride)
{
    if (c.nitems + c.nitems_wg + c.nitems_q + c.nitems_q_wg == 0) { c.searched = true; return c.is_group ? 0 : topr_after_search(ctx, d, c); }
    if (!d.bnd.p || d.bnd_stride == 0) return fail(OSWALD_HIP_ESTATE, "device %d has no spill scratch (an earlier allocation failed)", d.id);
    if (c.down_pending) { HIP_TRY(hipStreamWaitEvent(d.stream, c.ev_down, 0)); c.down_pending = false; } // the table of the slot's last search is still on its way out

    OswSearchArgs a;
    memset(&a, 0, sizeof a);
    a.tiled = (const uint16_t *)c.tiled.p;
    a.blocks = (const OswBlock *)c.blocks.p;
    a.sub_cols = c.sub_cols_dev();
    a.items = c.items_ptr();
    a.nitems = c.nitems;
    a.nitems_wg = c.nitems_wg;
    a.two_ended_waves = ctx->tun.two_ended;
    a.one_ended_wg = ctx->tun.one_ended_wg;
    a.force_all = ctx->cell_bits == 32 ? 1u : 0u;
    a.prof_i32 = (const uint2 *)d.prof_alt.p; // S + ge: the int32 cell (re-run and cell_bits = 32)
    a.floor_i32 = (const uint2 *)d.floor_i32.p;
    a.debug_nospill = ctx->tun.debug_nospill ? 1u : 0u; // -DOSW_DIAG builds only (timing experiment: results are wrong); always 0 otherwise
    a.prof = (const uint2 *)d.prof.p;
    a.prof_off = (const uint32_t *)d.prof_off.p;
    a.qlen = (const uint16_t *)d.qlen.p;
    a.bnd = (uint2 *)d.bnd.p;
    a.top_pages = (const uint2 *)d.top_pages.p;
    a.bnd_stride = d.bnd_stride + OSW_SCRATCH_DATA;
    a.scores = (int32_t *)c.scores.p;
    a.score_stride = c.score_stride;
    a.scores_host = table_dev;
    a.host_stride = (uint32_t)table_stride;
    a.host_cols = c.ngroups * c.W;
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

#ifdef OSW_DIAG
    const bool dbg_times = ctx->tun.debug_times;
    if (dbg_times) {
        HIP_TRY(d.wg_times.reserve(((size_t)d.grid * 5 + 2) * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d.wg_times.p, 0, ((size_t)d.grid * 5 + 2) * sizeof(unsigned long long), d.stream));
        a.wg_times = (unsigned long long *)d.wg_times.p;
    }
#endif
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
    // (a launch whose queue holds SHORT items runs the kernel that carries the single-query cell for their tails beside the pair cell)
    const auto launch_pair = c.nitems_short > 0 ? (frame ? osw_launch_s16qt : osw_launch_pk16qt) : (frame ? osw_launch_s16q : osw_launch_pk16q);
    OswSearchArgs as = a; // single queries on the int16 cells: {S, 1} entries (`a` itself stays on the plain integer profile for the int32 kernel)
    as.prof = (const uint2 *)d.prof_seq.p;
    if (frame) { as.prof = (const uint2 *)d.prof_seq_alt.p; as.prof_fb = (const uint2 *)d.prof_seq.p; }
    if (first_pass_is_q8(ctx)) {
        // 8-bit first pass over the query pairs, then -- all on this stream, each kernel reading what the one before
        // queued -- a leftover unpaired query on the plain int16 kernel, the int16 re-run of what left the 7-bit range
        // (queue length on the device), and below the int32 re-run of what reached the int16 ceiling
        if (c.nitems_q > 0) {
            OswSearchArgs aq = a;
            aq.items = c.items_q_ptr();
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
        aq.items = c.items_q_ptr();
        aq.nitems = c.nitems_q;
        aq.nitems_wg = c.nitems_q_wg;
        aq.prof = (const uint2 *)d.prof_pair.p;
        aq.prof_fb = (const uint2 *)d.prof_pair_i16.p;
        aq.prof_off = (const uint32_t *)d.pair_off.p;
        aq.qlen = (const uint16_t *)d.pair_len.p;
        aq.pair_q = (const uint32_t *)d.pair_q.p;
        aq.counters = (uint32_t *)d.counters.p + OSW_CTR_COUNT;
        if (c.nitems_short > 0) {
            // SHORT pair items run their pair's tail on the single-query cell (OswSearchArgs::hand): the single-query profiles of the same
            // set, and a hand region per resident wave -- the half of the spill scratch a launch of single queries beside this one
            // would use (the planner marks items SHORT only where there is no such launch)
            if (c.nitems + c.nitems_wg > 0) return fail(OSWALD_HIP_ERUNTIME, "planner: SHORT pair items beside a launch of single queries");
            aq.hand = a.bnd + (size_t)d.grid * (OSW_WG_THREADS / 64) * a.bnd_stride;
            aq.pair_rows = (const uint16_t *)d.pair_rows.p;
            aq.tail_len = (const uint16_t *)d.tail_len.p;
            aq.tail_off = (const uint32_t *)d.tail_off.p;
            aq.tail_prof = as.prof;
            aq.tail_prof_fb = as.prof_fb;
        }
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
    // the set of work queues these launches pull from is busy until here (build_items)
    {
        const int k = c.items_cur;
        if (!c.ev_set_read[k]) HIP_TRY(hipEventCreateWithFlags(&c.ev_set_read[k], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c.ev_set_read[k], d.stream));
        c.set_read_pending[k] = true;
    }
    ht.lap("search: launches");
    if (!c.is_group) { if (int r = topr_after_search(ctx, d, c)) return r; } // (a group's members fold their columns: search_resident_impl)
    ht.lap("search: top-r launches");
    if (!c.ev_use) HIP_TRY(hipEventCreateWithFlags(&c.ev_use, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c.ev_use, d.stream)); // an upload into this slot waits for it
    c.use_pending = true;
#ifdef OSW_DIAG
    if (dbg_times) {
        // diagnostics only (liboswald_hip_diag.so): when did the workgroups of the DP launch start / leave phase 1 / finish
        HIP_TRY(hipStreamSynchronize(d.stream));
        // (the launch the stamps are of: the single-query launch, or -- a set without single queries -- the query-pair launch)
        const uint32_t g_rep = c.nitems + c.nitems_wg > 0 ? grid : std::min<uint32_t>(grid_cap, (c.nitems_q + 3) / 4 + c.nitems_q_wg);
        std::vector<unsigned long long> t((size_t)g_rep * 5 + 2);
        HIP_TRY(hipMemcpy(t.data(), d.wg_times.p, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        uint32_t ctr[8] = {0};
        HIP_TRY(hipMemcpy(ctr, (uint32_t *)d.counters.p + OSW_CTR_BLOCKS * OSW_CTR_COUNT, sizeof ctr, hipMemcpyDeviceToHost));
        osw_diag_report_times(t.data(), g_rep, ctr[4]);
        if (t[(size_t)g_rep * 5 + 1]) fprintf(stderr, "[oswald_hip]   core clock inside the kernel (cycle counter / 100-MHz counter over its workgroups): %.0f MHz\n", 100.0 * (double)t[(size_t)g_rep * 5] / (double)t[(size_t)g_rep * 5 + 1]);
    }
#endif
    return 0;
}

// The chunk's (this slot's) table to the caller: columns [col0, col0 + ngroups * W) of rows of out_stride scores.
static int search_download(oswald_hip_ctx *ctx, Device &d, Chunk &c, int32_t *scores_out, size_t out_stride, size_t col0, PhaseTimer &pt, HoldTimer &ht)
{
    // The caller's table is [nq][out_stride]; ours is pitched to whole wave blocks.  It leaves on the download stream,
    // behind the search (ev_use) and beside whatever the search stream runs next: row by row when the rows are few (plain
    // DMA, nothing that needs a wave slot), packed on the device first when they are many (a pitched copy into
    // pageable memory runs at a fraction of a GB/s).
    const uint32_t row = c.ngroups * c.W;
    if (row == 0 || ctx->nq == 0) return 0;
    int32_t *dst = scores_out + col0;
    HIP_TRY(hipStreamWaitEvent(d.stream_down, c.ev_use, 0));
    if (out_stride == row && (row == c.score_stride || ctx->nq == 1)) {
        HIP_TRY(hipMemcpyAsync(dst, c.scores.p, (size_t)ctx->nq * row * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream_down));
    } else if (ctx->nq <= 64) {
        for (uint32_t q = 0; q < ctx->nq; ++q)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)q * out_stride, (const int32_t *)c.scores.p + (size_t)q * c.score_stride, (size_t)row * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, d.stream_down));
    } else {
        const size_t bytes = (size_t)ctx->nq * row * sizeof(int32_t);
        HIP_TRY(d.scores_packed.reserve(bytes));
        HIP_TRY(hipMemcpy2DAsync(d.scores_packed.p, (size_t)row * sizeof(int32_t), c.scores.p, (size_t)c.score_stride * sizeof(int32_t),
                                 (size_t)row * sizeof(int32_t), ctx->nq, hipMemcpyDeviceToDevice, d.stream_down));
        if (out_stride == row) HIP_TRY(hipMemcpyAsync(dst, d.scores_packed.p, bytes, hipMemcpyDeviceToHost, d.stream_down));
        else HIP_TRY(hipMemcpy2DAsync(dst, out_stride * sizeof(int32_t), d.scores_packed.p, (size_t)row * sizeof(int32_t), (size_t)row * sizeof(int32_t), ctx->nq,
                                      hipMemcpyDeviceToHost, d.stream_down));
    }
    HIP_TRY(hipEventRecord(c.ev_down, d.stream_down));
    c.down_pending = true;
    ht.lap("search: table download queued");
    if (pt.on) { HIP_TRY(hipStreamSynchronize(d.stream_down)); pt.lap("search: D2H of the score table"); }
    return 0;
}

// a caller's handle: the slot of a live chunk that is not the rest of another one
static int check_chunk(oswald_hip_ctx *ctx, int dev, int chunk)
{
    if (int r = check_dev(ctx, dev)) return r;
    const Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || !d.chunks[chunk].live || d.chunks[chunk].is_cont) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    return 0;
}

static int chunk_search_impl(oswald_hip_ctx *ctx, int dev, int chunk, int32_t *scores_out)
{
    if (int r = check_chunk(ctx, dev, chunk)) return r;
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    PhaseTimer pt(ctx->tun.debug_phases);
    if (ctx->topr_r && d.chunks[chunk].has_index && ctx->topr_queries_version != ctx->queries_version)
        return fail(OSWALD_HIP_ESTATE, "the query set changed since oswald_hip_topr_begin: call it again before searching");
    HoldTimer ht(g_debug_slow);
    size_t out_stride = 0;
    for (int k = chunk; k >= 0; k = d.chunks[k].next) out_stride += (size_t)d.chunks[k].ngroups * d.chunks[k].W;
    if (scores_out && out_stride && ctx->have_queries) {
        // pin the caller's table for the copy unless it is page-locked already (oswald_hip_host_alloc / _register): the DMA engine
        // then writes it directly (a copy into a pageable buffer it has not seen before runs at ~1 GB/s: 8.9 ms for the 8 MB of
        // C2; this way 0.5 ms)
        const size_t bytes = (size_t)ctx->nq * out_stride * sizeof(int32_t);
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
    }
    // A table in page-locked memory is WRITTEN BY THE KERNELS as they finish their items (round 5): no copy, no download stream,
    // no event between a search and the next one -- a table that left by DMA behind its search cost a one-query search 0.3 ms
    // per chunk (0.17 ms before the next search could start, 0.1 ms in the search itself: profiles/r05_inclusive_probe_q1.txt).
    // Anything else (a small pageable table, OSWALD_HIP_NO_DIRECT_TABLE) leaves by DMA as before.
    int32_t *table_dev = nullptr;
    if (scores_out && out_stride && out_stride <= 0xffffffffull && !ctx->tun.no_direct_table) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, scores_out, 0) == hipSuccess && dp) table_dev = (int32_t *)dp;
        else (void)hipGetLastError();
    }
    // a chunk the library cut in two at its upload: head, then rest, behind one another on the device
    size_t col0 = 0;
    for (int k = chunk; k >= 0; k = d.chunks[k].next) {
        Chunk &c = d.chunks[k];
        if (int r = search_plan(ctx, d, c, pt, ht)) return r;
        if (int r = search_launch(ctx, d, c, ht, table_dev ? table_dev + col0 : nullptr, out_stride)) return r;
        if (pt.on) { HIP_TRY(hipStreamSynchronize(d.stream)); pt.lap("search: kernels"); }
        if (scores_out && !table_dev) if (int r = search_download(ctx, d, c, scores_out, out_stride, col0, pt, ht)) return r;
        col0 += (size_t)c.ngroups * c.W;
    }
    return 0;
}

// All queries against SEVERAL resident chunks of a device as ONE launch (see Group).  The chunks' uploads must have landed -- the plan is
// made on the live extents of all of them: this is the entry for a database that STAYS on the device; a caller that streams chunks in
// searches them one by one (oswald_hip_chunk_search), each behind its upload.  Top lists: every member that has an index folds its
// columns of the group's score table into the device's running list, in the order given.  scores_out (or null): int32 [nq][sum of the
// members' ngroups * W], the members' columns side by side in the order given (a plain download: the entry is for top lists).
static int search_resident_impl(oswald_hip_ctx *ctx, int dev, const int *handles, uint32_t nhandles, int32_t *scores_out)
{
    if (int r = check_dev(ctx, dev)) return r;
    if (!handles || nhandles == 0) return fail(OSWALD_HIP_EINVAL, "no chunk handles");
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    stop_warm(ctx);
    if (ctx->topr_r && ctx->topr_queries_version != ctx->queries_version)
        return fail(OSWALD_HIP_ESTATE, "the query set changed since oswald_hip_topr_begin: call it again before searching");
    std::vector<int> members;
    for (uint32_t i = 0; i < nhandles; ++i) {
        if (int r = check_chunk(ctx, dev, handles[i])) return r;
        for (int k = handles[i]; k >= 0; k = d.chunks[k].next) {
            if (std::find(members.begin(), members.end(), k) != members.end()) return fail(OSWALD_HIP_EINVAL, "chunk handle %d given twice", handles[i]);
            members.push_back(k);
        }
    }
    for (int k : members) if (int r = finish_upload(d, d.chunks[k])) return r;
    if (int r = sync_queries(ctx, d)) return r;
    if (!d.group) d.group.reset(new Group);
    Group &G = *d.group;
    Chunk &g = G.g;
    bool same = G.valid && G.members == members;
    for (size_t i = 0; same && i < members.size(); ++i) same = G.up_seq[i] == d.chunks[members[i]].up_seq;
    if (!same) {
        G.valid = false;
        // whoever still reads the group's tables -- a search of the group as it was -- is through
        HIP_TRY(hipStreamSynchronize(d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream2));
        HIP_TRY(hipStreamSynchronize(d.stream_down));
        g.use_pending = g.down_pending = g.set_read_pending[0] = g.set_read_pending[1] = false;
        const uint32_t W = d.chunks[members[0]].W;
        uintptr_t base = ~(uintptr_t)0;
        uint64_t nblocks = 0, stride = 0, ngroups = 0;
        for (int k : members) {
            const Chunk &c = d.chunks[k];
            if (c.W != W) return fail(OSWALD_HIP_EINVAL, "the chunks of a combined search must have one lane width (%u and %u)", W, c.W);
            if (c.nblocks) base = std::min(base, (uintptr_t)c.tiled.p);
            nblocks += c.nblocks; stride += c.score_stride; ngroups += c.ngroups;
        }
        if (nblocks >= 0x7fffffffull || stride >= 0xffffffffull || ngroups >= 0xffffffffull)
            return fail(OSWALD_HIP_EINVAL, "too many sequences for one combined search (%llu blocks)", (unsigned long long)nblocks);
        g.is_group = true;
        g.live = true;
        g.W = W;
        g.ngroups = (uint32_t)ngroups;
        g.nblocks = (uint32_t)nblocks;
        g.score_stride = (uint32_t)stride;
        g.max_ncols4 = 0;
        g.ncols4_alloc.clear();
        G.col0.clear(); G.blk0.clear(); G.up_seq.clear();
        G.sub_cols_host.assign((size_t)nblocks * 128, 0);
        // the members' block tables as the DEVICE holds them (their re-tile wrote the live extents), offsets moved into the group's frame
        std::vector<OswBlock> all(nblocks);
        uint32_t b0 = 0, c0 = 0;
        for (int k : members) {
            const Chunk &c = d.chunks[k];
            G.blk0.push_back(b0); G.col0.push_back(c0); G.up_seq.push_back(c.up_seq);
            if (c.nblocks) {
                HIP_TRY(hipMemcpy(all.data() + b0, c.blocks.p, (size_t)c.nblocks * sizeof(OswBlock), hipMemcpyDeviceToHost));
                const uintptr_t delta = (uintptr_t)c.tiled.p - base;
                if (delta % 512u) return fail(OSWALD_HIP_ERUNTIME, "chunk buffers are not 512-byte aligned to each other");
                const uint64_t d4 = delta / 512u;
                for (uint32_t B = 0; B < c.nblocks; ++B) {
                    OswBlock &blk = all[b0 + B];
                    if ((uint64_t)blk.col4_off + d4 > 0xfffffff0ull) return fail(OSWALD_HIP_EINVAL, "the chunks of a combined search lie too far apart in device memory");
                    blk.col4_off = (uint32_t)(blk.col4_off + d4);
                    blk.seq0 += c0;
                }
                memcpy(G.sub_cols_host.data() + (size_t)b0 * 128, c.sub_cols, (size_t)c.nblocks * 128 * sizeof(uint16_t));
                g.ncols4_alloc.insert(g.ncols4_alloc.end(), c.ncols4_alloc.begin(), c.ncols4_alloc.end());
                g.max_ncols4 = std::max(g.max_ncols4, c.max_ncols4);
            }
            b0 += c.nblocks;
            c0 += c.score_stride;
        }
        g.sub_cols = G.sub_cols_host.data(); // (plain host memory: only the planner reads it; never given to hipHostFree, see free_group)
        g.sub_cols_cap = G.sub_cols_host.size();
        g.tiled.assign((void *)base, ~(size_t)0 >> 1); // a view: the members own their residues
        HIP_TRY(g.blocks.reserve((size_t)nblocks * sizeof(OswBlock) + 16));
        HIP_TRY(g.sub_cols_buf.reserve((size_t)nblocks * 128 * sizeof(uint16_t) + 16));
        if (nblocks > g.blocks_pin_cap) {
            if (g.blocks_pin) HIP_TRY(hipHostFree(g.blocks_pin));
            g.blocks_pin = nullptr;
            g.blocks_pin_cap = 0;
            const size_t want = (size_t)nblocks + nblocks / 8 + 16;
            HIP_TRY(hipHostMalloc((void **)&g.blocks_pin, want * sizeof(OswBlock), hipHostMallocPortable));
            g.blocks_pin_cap = want;
        }
        if (nblocks) memcpy(g.blocks_pin, all.data(), (size_t)nblocks * sizeof(OswBlock));
        HIP_TRY(osw_launch_copy16(g.blocks_pin, g.blocks.p, (size_t)nblocks * sizeof(OswBlock), d.stream));
        for (size_t i = 0; i < members.size(); ++i) {
            const Chunk &c = d.chunks[members[i]];
            if (c.nblocks) HIP_TRY(hipMemcpyAsync((uint16_t *)g.sub_cols_buf.p + (size_t)G.blk0[i] * 128, c.sub_cols_buf.p, (size_t)c.nblocks * 128 * sizeof(uint16_t), hipMemcpyDeviceToDevice, d.stream));
        }
        g.items_version = ~0ull;
        g.items_exact = false;
        G.members = members;
        G.valid = true;
    }
    if (int r = build_items(ctx, d, g, true)) return r;
    HoldTimer ht(g_debug_slow);
    if (int r = search_launch(ctx, d, g, ht, nullptr, 0)) return r;
    const bool searched = g.nitems + g.nitems_wg + g.nitems_q + g.nitems_q_wg != 0;
    for (size_t i = 0; i < members.size(); ++i) {
        Chunk &c = d.chunks[members[i]];
        if (int r = topr_after_search(ctx, d, c, (const int32_t *)g.scores.p + G.col0[i], g.score_stride, searched)) return r;
        // the member's residues are in use until here: an upload into its slot waits for it; its OWN score table was not written
        if (!c.ev_use) HIP_TRY(hipEventCreateWithFlags(&c.ev_use, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c.ev_use, d.stream));
        c.use_pending = true;
        c.searched = false;
    }
    if (scores_out && ctx->nq && searched) {
        size_t out_stride = 0;
        for (int k : members) out_stride += (size_t)d.chunks[k].ngroups * d.chunks[k].W;
        if (!g.ev_down) HIP_TRY(hipEventCreateWithFlags(&g.ev_down, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g.ev_use, d.stream));
        HIP_TRY(hipStreamWaitEvent(d.stream_down, g.ev_use, 0));
        size_t oc = 0;
        for (size_t i = 0; i < members.size(); ++i) {
            const Chunk &c = d.chunks[members[i]];
            const size_t row = (size_t)c.ngroups * c.W;
            for (uint32_t q = 0; q < ctx->nq && row; ++q)
                HIP_TRY(hipMemcpyAsync(scores_out + (size_t)q * out_stride + oc, (const int32_t *)g.scores.p + (size_t)q * g.score_stride + G.col0[i], row * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, d.stream_down));
            oc += row;
        }
        HIP_TRY(hipEventRecord(g.ev_down, d.stream_down));
        g.down_pending = true;
    }
    return 0;
}

// what a Group holds beyond its members' buffers (oswald_hip_finalize)
static void free_group(Device &d)
{
    if (!d.group) return;
    Chunk &g = d.group->g;
    for (hipEvent_t *e : {&g.ev_use, &g.ev_down, &g.ev_set_read[0], &g.ev_set_read[1]}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    for (int k = 0; k < 2; ++k) { if (g.items_pin[k]) (void)hipHostFree(g.items_pin[k]); g.items_pin[k] = nullptr; }
    if (g.blocks_pin) (void)hipHostFree(g.blocks_pin);
    g.blocks_pin = nullptr;
    g.sub_cols = nullptr; // (a std::vector's memory)
    for (DevBuf *b : {&g.tiled, &g.blocks, &g.sub_cols_buf, &g.scores, &g.ovf, &g.ovf8}) b->release();
    d.group.reset();
}

static int chunk_release_impl(oswald_hip_ctx *ctx, int dev, int chunk)
{
    if (int r = check_chunk(ctx, dev, chunk)) return r;
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    // the caller's b / n / disp are free once the upload has landed; a search of the chunk may still be running -- the
    // next upload into the slot waits for it on the device (ev_use), the host does not
    for (int k = chunk; k >= 0; k = d.chunks[k].next) if (int r = finish_upload(d, d.chunks[k])) return r;
    for (int k = chunk; k >= 0;) {
        Chunk &c = d.chunks[k];
        k = c.next;
        c.live = false; // buffers are kept for the next upload into this slot
        c.next = -1;
        c.is_cont = false;
    }
    return 0;
}

// Upload, search and release in one call, and asynchronous throughout: the copies are queued on the copy stream, the search
// behind them on the device, and the slot goes back to the pool when the host next learns that the upload has landed
// (oswald_hip_wait, or a later upload's completion on the same in-order stream) -- a following upload into it waits for this
// search on the device.  With several devices the calls for device d+1 are made while device d's chunk is still on the link:
// the reference's four clEnqueueWriteBuffer per device are non-blocking too (FPGAsearch.c:180-198).
static int search_chunk_async_impl(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n,
                                  const uint32_t *disp, uint32_t ngroups, uint32_t W, int32_t *scores_out)
{
    int h = -1;
    if (int r = oswald_hip_chunk_upload_async(ctx, dev, b, vD, n, disp, ngroups, W, &h)) return r;
    const int r = oswald_hip_chunk_search(ctx, dev, h, scores_out);
    Device &d = ctx->dev[dev];
    for (int k = h; k >= 0;) { // (upload_pending stays set: the slots are not re-used before the upload has landed)
        Chunk &c = d.chunks[k];
        k = c.next;
        c.live = false;
        c.next = -1;
        c.is_cont = false;
    }
    return r;
}

// The reverse of oswald_hip_reserve_chunks' device part: the DEVICE buffers of every chunk slot that holds no chunk go back to the device
// (the reference releases its six buffers at the end of a search, FPGAsearch.c:361-368, and creates them again inside the
// next search's clock, :85-96).  Waits for the device first: a released slot's last search may still be running.
static int release_chunks_impl(oswald_hip_ctx *ctx, int dev)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (dev >= (int)ctx->dev.size()) return fail(OSWALD_HIP_ENODEV, "device index %d out of range", dev);
    for (int i = 0; i < (int)ctx->dev.size(); ++i) {
        if (dev >= 0 && i != dev) continue;
        Device &d = ctx->dev[i];
        HIP_TRY(hipSetDevice(d.id));
        for (hipStream_t st : {d.stream_copy, d.stream_up, d.stream, d.stream2, d.stream_down}) HIP_TRY(hipStreamSynchronize(st));
        release_registered(d);
        for (Chunk &c : d.chunks) {
            c.upload_pending = c.use_pending = c.down_pending = c.map_pending = false;
            c.set_read_pending[0] = c.set_read_pending[1] = false;
            if (c.live) continue;
            // (the DEVICE buffers: the slot's page-locked host staging stays, like the reference's host buffers, which outlive its device
            // buffers -- FPGAsearch.c:361-368 releases the cl_mem objects only)
            for (DevBuf *b : {&c.tiled, &c.blocks, &c.sub_cols_buf, &c.scores, &c.ovf, &c.ovf8, &c.st_b, &c.index_map_dev[0], &c.index_map_dev[1], &c.slab}) b->release();
            c.items_version = ~0ull;
            c.searched = false;
        }
    }
    return 0;
}

static int chunk_wait_impl(oswald_hip_ctx *ctx, int dev, int chunk)
{
    if (int r = check_chunk(ctx, dev, chunk)) return r;
    Device &d = ctx->dev[dev];
    HIP_TRY(hipSetDevice(d.id));
    for (int k = chunk; k >= 0; k = d.chunks[k].next) {
        Chunk &c = d.chunks[k];
        if (int r = finish_upload(d, c)) return r;
        if (c.use_pending) { HIP_TRY(hipEventSynchronize(c.ev_use)); c.use_pending = false; }
        if (c.down_pending) { HIP_TRY(hipEventSynchronize(c.ev_down)); c.down_pending = false; }
    }
    return 0;
}

static int wait_impl(oswald_hip_ctx *ctx, int dev)
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
        for (Chunk &c : ctx->dev[i].chunks) { c.upload_pending = false; c.use_pending = false; c.down_pending = false; c.set_read_pending[0] = c.set_read_pending[1] = false; c.map_read_pending[0] = c.map_read_pending[1] = false; }
        if (ctx->dev[i].group) { Chunk &g = ctx->dev[i].group->g; g.use_pending = g.down_pending = g.set_read_pending[0] = g.set_read_pending[1] = false; }
        release_registered(ctx->dev[i]);
    }
    return 0;
}

static int chunk_topr_impl(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t nvalid, uint32_t r, int32_t *scores, uint32_t *index)
{
    if (int rc = check_dev(ctx, dev)) return rc;
    Device &d = ctx->dev[dev];
    if (chunk < 0 || chunk >= (int)d.chunks.size() || d.chunks[chunk].is_cont) return fail(OSWALD_HIP_EINVAL, "invalid chunk handle %d", chunk);
    uint32_t lanes = 0;
    for (int k = chunk; k >= 0; k = d.chunks[k].next) {
        if (!d.chunks[k].searched) return fail(OSWALD_HIP_ESTATE, "chunk %d has not been searched", chunk);
        lanes += d.chunks[k].ngroups * d.chunks[k].W;
    }
    if (!scores || !index) return fail(OSWALD_HIP_EINVAL, "null output");
    if (nvalid > lanes) return fail(OSWALD_HIP_EINVAL, "nvalid %u exceeds the chunk's %u lanes", nvalid, lanes);
    if (r == 0 || ctx->nq == 0) return 0;
    HIP_TRY(hipSetDevice(d.id));
    const size_t cnt = (size_t)ctx->nq * r;
    if (r > 1024) return fail(OSWALD_HIP_EINVAL, "top-r on the device supports r <= 1024 (asked for %u)", r);
    if (d.chunks[chunk].next < 0) {
        if (int rc = queue_topr(ctx, d, d.chunks[chunk], nvalid, r)) return rc;
        HIP_TRY(hipMemcpyAsync(scores, d.topr_scores.p, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipMemcpyAsync(index, d.topr_index.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        return 0;
    }
    // a chunk in two pieces: the r best of each, indices counted from the chunk's first sequence, merged in the same order
    std::vector<int32_t> cs, ps(cnt);
    std::vector<uint32_t> ci, pi(cnt);
    uint32_t first = 0, pieces = 0;
    for (int k = chunk; k >= 0; k = d.chunks[k].next) {
        Chunk &c = d.chunks[k];
        const uint32_t have = c.ngroups * c.W, nv = nvalid > first ? std::min(nvalid - first, have) : 0u;
        if (nv > 0) {
            if (int rc = queue_topr(ctx, d, c, nv, r)) return rc;
            HIP_TRY(hipMemcpyAsync(ps.data(), d.topr_scores.p, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipMemcpyAsync(pi.data(), d.topr_index.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipStreamSynchronize(d.stream));
            for (size_t i = 0; i < cnt; ++i) if (ps[i] >= 0) pi[i] += first;
        } else {
            std::fill(ps.begin(), ps.end(), -1);
            std::fill(pi.begin(), pi.end(), 0u);
        }
        cs.insert(cs.end(), ps.begin(), ps.end());
        ci.insert(ci.end(), pi.begin(), pi.end());
        first += have;
        ++pieces;
    }
    // [piece][nq][r] -> [nq][pieces * r]
    std::vector<int32_t> ms(cs.size());
    std::vector<uint32_t> mi(ci.size());
    for (uint32_t pc = 0; pc < pieces; ++pc)
        for (uint32_t q = 0; q < ctx->nq; ++q)
            for (uint32_t j = 0; j < r; ++j) {
                ms[((size_t)q * pieces + pc) * r + j] = cs[((size_t)pc * ctx->nq + q) * r + j];
                mi[((size_t)q * pieces + pc) * r + j] = ci[((size_t)pc * ctx->nq + q) * r + j];
            }
    merge_candidates(ctx->nq, (size_t)pieces * r, ms.data(), mi.data(), r, scores, index);
    return 0;
}

static int set_index_slot(Device &d, Chunk &c, uint32_t first_index, uint32_t nvalid, const uint32_t *index_map)
{
    c.first_index = first_index;
    c.nvalid = nvalid;
    c.index_map = false;
    if (index_map && nvalid > 0) {
        // the map goes to the device (the chunk's top list is selected there, on database keys) by way of a page-locked copy
        // made here: the caller's array is free when the call returns, and the copy engine reads nothing pageable
        const int k = c.map_cur ^ 1;
        if (c.ev_map[k]) HIP_TRY(hipEventSynchronize(c.ev_map[k])); // (the copy that last read this staging buffer: two maps ago)
        else HIP_TRY(hipEventCreateWithFlags(&c.ev_map[k], hipEventDisableTiming));
        if (c.map_read_pending[k]) { HIP_TRY(hipEventSynchronize(c.ev_map_read[k])); c.map_read_pending[k] = false; } // (the fold that read device buffer k last)
        HIP_TRY(c.map_pin[k].reserve((size_t)nvalid * sizeof(uint32_t)));
        memcpy(c.map_pin[k].p, index_map, (size_t)nvalid * sizeof(uint32_t));
        // (the device buffer: the search that read it last -- of the chunk this slot held two maps ago -- is long through; a map
        // given twice to one upload flips to the other buffer, which a running search of the chunk does not read)
        HIP_TRY(c.index_map_dev[k].reserve((size_t)nvalid * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(c.index_map_dev[k].p, c.map_pin[k].p, (size_t)nvalid * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream_copy));
        HIP_TRY(hipEventRecord(c.ev_map[k], d.stream_copy));
        c.map_cur = k;
        c.map_pending = true;
        c.index_map = true;
    }
    c.has_index = true;
    return 0;
}

static int chunk_set_index_impl(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t first_index, uint32_t nvalid, const uint32_t *index_map)
{
    if (int r = check_chunk(ctx, dev, chunk)) return r;
    Device &d = ctx->dev[dev];
    uint32_t lanes = 0;
    for (int k = chunk; k >= 0; k = d.chunks[k].next) lanes += d.chunks[k].ngroups * d.chunks[k].W;
    if (nvalid > lanes) return fail(OSWALD_HIP_EINVAL, "nvalid %u exceeds the chunk's %u lanes", nvalid, lanes);
    if (!index_map && (uint64_t)first_index + nvalid > 0xffffffffull) return fail(OSWALD_HIP_EINVAL, "database indices must fit 32 bits");
    HIP_TRY(hipSetDevice(d.id));
    uint32_t first = 0; // sequences of the chunk in front of the piece
    for (int k = chunk; k >= 0; k = d.chunks[k].next) {
        Chunk &c = d.chunks[k];
        const uint32_t have = c.ngroups * c.W, nv = nvalid > first ? std::min(nvalid - first, have) : 0u;
        if (int r = set_index_slot(d, c, index_map ? 0u : first_index + first, nv, index_map ? index_map + first : nullptr)) return r;
        first += have;
    }
    return 0;
}

static int topr_begin_impl(oswald_hip_ctx *ctx, uint32_t r)
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
static int topr_impl(oswald_hip_ctx *ctx, uint32_t r, int32_t *scores, uint32_t *db_index)
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
        for (Chunk &c : d.chunks) c.down_pending = false;
        release_registered(d);
        if (g_debug_slow) drain_events(d);
    }
    memcpy(scores, ctx->top_host, out_cnt * sizeof(int32_t));
    memcpy(db_index, (const char *)ctx->top_host + out_cnt * sizeof(int32_t), out_cnt * sizeof(uint32_t));
    return 0;
}

static int comm_unique_id_impl(void *id, size_t id_bytes)
{
    if (!id || id_bytes < sizeof(ncclUniqueId)) return fail(OSWALD_HIP_EINVAL, "the id buffer must hold %zu bytes", sizeof(ncclUniqueId));
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memset(id, 0, id_bytes);
    memcpy(id, &u, sizeof u);
    return 0;
}

static int comm_init_rank_impl(oswald_hip_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank)
{
    if (!ctx || !id) return fail(OSWALD_HIP_EINVAL, "null argument");
    if (id_bytes < sizeof(ncclUniqueId)) return fail(OSWALD_HIP_EINVAL, "the id must be the %zu bytes oswald_hip_comm_unique_id wrote", sizeof(ncclUniqueId));
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(OSWALD_HIP_EINVAL, "rank %d of %d", rank, nranks);
    if (ctx->pcomm) return fail(OSWALD_HIP_ESTATE, "the context already has a process-level communicator");
    stop_warm(ctx); // (the communicator's bring-up works on the device: no warm-up kernel beside it)
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

static int comm_destroy_impl(oswald_hip_ctx *ctx)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    if (!ctx->pcomm) return 0;
    HIP_TRY(hipSetDevice(ctx->dev[0].id));
    HIP_TRY(hipStreamSynchronize(ctx->dev[0].stream)); // (an all-gather of an earlier oswald_hip_topr has long finished: it waits for its devices)
    const ncclResult_t nr = ncclCommAbort(ctx->pcomm); // no rank waits for another one: a rank that never joined cannot keep this one here
    ctx->pcomm = nullptr;
    ctx->pcomm_nranks = 0;
    ctx->pcomm_rank = -1;
    if (nr != ncclSuccess) return fail(OSWALD_HIP_ECOMM, "ncclCommAbort: %s", ncclGetErrorString(nr));
    return 0;
}

static int comm_info_impl(oswald_hip_ctx *ctx, int *out4)
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

static int merge_candidates_impl(uint32_t nq, uint64_t ncand, const int32_t *cand_scores, const uint32_t *cand_index, uint32_t r,
                                int32_t *scores, uint32_t *db_index)
{
    if (nq == 0 || r == 0) return 0;
    if ((ncand > 0 && (!cand_scores || !cand_index)) || !scores || !db_index) return fail(OSWALD_HIP_EINVAL, "null argument");
    merge_candidates(nq, (size_t)ncand, cand_scores, cand_index, r, scores, db_index);
    return 0;
}

static int set_profiling_impl(oswald_hip_ctx *ctx, int enable)
{
    if (!ctx) return fail(OSWALD_HIP_EINVAL, "null context");
    ctx->profiling = enable != 0;
    return 0;
}

static int kernel_stats_impl(oswald_hip_ctx *ctx, int dev, double *dp_kernel_ms, uint64_t *dp_launches, uint64_t *rerun_items, int reset)
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

static int rerun_stats_impl(oswald_hip_ctx *ctx, int dev, double *ms2)
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

static int rerun_counts_impl(oswald_hip_ctx *ctx, int dev, uint64_t *out2)
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

static int chunk_geometry_impl(oswald_hip_ctx *ctx, int dev, int chunk, uint64_t *out8)
{
    if (int r = check_chunk(ctx, dev, chunk)) return r;
    Device &d = ctx->dev[dev];
    if (!out8) return fail(OSWALD_HIP_EINVAL, "null output");
    HIP_TRY(hipSetDevice(d.id));
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    for (int k = chunk; k >= 0; k = d.chunks[k].next) { // (a chunk the library cut in two: the sums over its pieces)
        Chunk &c = d.chunks[k];
        if (int r = finish_upload(d, c)) return r;
        HIP_TRY(hipStreamSynchronize(d.stream));
        std::vector<OswBlock> blocks(c.nblocks);
        if (c.nblocks) HIP_TRY(hipMemcpy(blocks.data(), c.blocks.p, c.nblocks * sizeof(OswBlock), hipMemcpyDeviceToHost));
        uint64_t alloc = 0, live = 0;
        for (const OswBlock &b : blocks) { alloc += b.ncols4_alloc; live += b.ncols4; }
        out8[0] += c.nblocks;
        out8[1] += alloc;
        out8[2] += live;
        out8[3] += live * 64 * sizeof(uint2);
        out8[4] += c.nitems + 4ull * c.nitems_wg + c.nitems_q + 4ull * c.nitems_q_wg; // wave-level work items (a phase-1 entry is four)
        out8[5] = std::max<uint64_t>(out8[5], c.max_lg);
        out8[6] += c.planned_spill_bytes;
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// The exported entry points.  No exception crosses the C ABI: the implementations above use the standard containers (work
// queues, block tables, host copies of the query set), whose allocations throw; an exception that left an extern "C"
// function would be std::terminate -- a dead caller where include/oswald_hip.h promises an error code.
// Every entry is its implementation inside guarded(): std::bad_alloc / std::length_error -> OSWALD_HIP_ENOMEM, anything
// else -> OSWALD_HIP_ERUNTIME, with the entry's name and what() in oswald_hip_last_error().
// ---------------------------------------------------------------------------------------------------------------
int oswald_hip_host_alloc(size_t bytes, void **ptr)
{
    return guarded("oswald_hip_host_alloc", [&] { return host_alloc_impl(bytes, ptr); });
}

int oswald_hip_host_free(void *ptr)
{
    return guarded("oswald_hip_host_free", [&] { return host_free_impl(ptr); });
}

int oswald_hip_host_register(void *ptr, size_t bytes)
{
    return guarded("oswald_hip_host_register", [&] { return host_register_impl(ptr, bytes); });
}

int oswald_hip_host_unregister(void *ptr)
{
    return guarded("oswald_hip_host_unregister", [&] { return host_unregister_impl(ptr); });
}

int oswald_hip_device_count(int *count)
{
    return guarded("oswald_hip_device_count", [&] { return device_count_impl(count); });
}

int oswald_hip_init(int ndev, const int *device_ids, oswald_hip_ctx **out)
{
    return guarded("oswald_hip_init", [&] { return init_impl(ndev, device_ids, out); });
}

int oswald_hip_finalize(oswald_hip_ctx *ctx)
{
    return guarded("oswald_hip_finalize", [&] { return finalize_impl(ctx); });
}

int oswald_hip_info(oswald_hip_ctx *ctx, int dev, char *buf, size_t buflen)
{
    return guarded("oswald_hip_info", [&] { return info_impl(ctx, dev, buf, buflen); });
}

int oswald_hip_set_scoring(oswald_hip_ctx *ctx, const int8_t *submat, int open_gap, int extend_gap, int cell_bits)
{
    return guarded("oswald_hip_set_scoring", [&] { return set_scoring_impl(ctx, submat, open_gap, extend_gap, cell_bits); });
}

int oswald_hip_set_queries(oswald_hip_ctx *ctx, const uint8_t *a, uint64_t Q, const uint16_t *m, const uint32_t *a_disp, uint32_t nq)
{
    return guarded("oswald_hip_set_queries", [&] { return set_queries_impl(ctx, a, Q, m, a_disp, nq); });
}

int oswald_hip_chunk_upload(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp, uint32_t ngroups, uint32_t W, int *chunk)
{
    return guarded("oswald_hip_chunk_upload", [&] { return chunk_upload_impl(ctx, dev, b, vD, n, disp, ngroups, W, chunk); });
}

int oswald_hip_chunk_upload_async(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp, uint32_t ngroups, uint32_t W, int *chunk)
{
    return guarded("oswald_hip_chunk_upload_async", [&] { return chunk_upload_async_impl(ctx, dev, b, vD, n, disp, ngroups, W, chunk); });
}

int oswald_hip_reserve(oswald_hip_ctx *ctx, int dev, uint32_t max_sequence_length)
{
    return guarded("oswald_hip_reserve", [&] { return reserve_impl(ctx, dev, max_sequence_length); });
}

int oswald_hip_reserve_chunks(oswald_hip_ctx *ctx, int dev, uint64_t chunk_bytes, uint32_t ngroups, uint32_t W, uint32_t nq, uint32_t slots)
{
    return guarded("oswald_hip_reserve_chunks", [&] { return reserve_chunks_impl(ctx, dev, chunk_bytes, ngroups, W, nq, slots); });
}

int oswald_hip_max_chunk_size(oswald_hip_ctx *ctx, int dev, uint32_t nq, uint32_t max_sequence_length, uint64_t *bytes)
{
    return guarded("oswald_hip_max_chunk_size", [&] { return max_chunk_size_impl(ctx, dev, nq, max_sequence_length, bytes); });
}

int oswald_hip_chunk_search(oswald_hip_ctx *ctx, int dev, int chunk, int32_t *scores_out)
{
    return guarded("oswald_hip_chunk_search", [&] { return chunk_search_impl(ctx, dev, chunk, scores_out); });
}

int oswald_hip_chunk_release(oswald_hip_ctx *ctx, int dev, int chunk)
{
    return guarded("oswald_hip_chunk_release", [&] { return chunk_release_impl(ctx, dev, chunk); });
}

int oswald_hip_search_chunk_async(oswald_hip_ctx *ctx, int dev, const uint8_t *b, uint64_t vD, const uint16_t *n, const uint32_t *disp, uint32_t ngroups, uint32_t W, int32_t *scores_out)
{
    return guarded("oswald_hip_search_chunk_async", [&] { return search_chunk_async_impl(ctx, dev, b, vD, n, disp, ngroups, W, scores_out); });
}

int oswald_hip_reserve_host(oswald_hip_ctx *ctx, int dev, uint32_t ngroups, uint32_t W, uint32_t nq, uint32_t slots)
{
    return guarded("oswald_hip_reserve_host", [&] { return reserve_host_impl(ctx, dev, ngroups, W, nq, slots); });
}

int oswald_hip_search_resident(oswald_hip_ctx *ctx, int dev, const int *chunks, uint32_t nchunks, int32_t *scores_out)
{
    return guarded("oswald_hip_search_resident", [&] { return search_resident_impl(ctx, dev, chunks, nchunks, scores_out); });
}

int oswald_hip_release_chunks(oswald_hip_ctx *ctx, int dev)
{
    return guarded("oswald_hip_release_chunks", [&] { return release_chunks_impl(ctx, dev); });
}

int oswald_hip_chunk_wait(oswald_hip_ctx *ctx, int dev, int chunk)
{
    return guarded("oswald_hip_chunk_wait", [&] { return chunk_wait_impl(ctx, dev, chunk); });
}

int oswald_hip_wait(oswald_hip_ctx *ctx, int dev)
{
    return guarded("oswald_hip_wait", [&] { return wait_impl(ctx, dev); });
}

int oswald_hip_chunk_topr(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t nvalid, uint32_t r, int32_t *scores, uint32_t *index)
{
    return guarded("oswald_hip_chunk_topr", [&] { return chunk_topr_impl(ctx, dev, chunk, nvalid, r, scores, index); });
}

int oswald_hip_chunk_set_index(oswald_hip_ctx *ctx, int dev, int chunk, uint32_t first_index, uint32_t nvalid, const uint32_t *index_map)
{
    return guarded("oswald_hip_chunk_set_index", [&] { return chunk_set_index_impl(ctx, dev, chunk, first_index, nvalid, index_map); });
}

int oswald_hip_topr_begin(oswald_hip_ctx *ctx, uint32_t r)
{
    return guarded("oswald_hip_topr_begin", [&] { return topr_begin_impl(ctx, r); });
}

int oswald_hip_topr(oswald_hip_ctx *ctx, uint32_t r, int32_t *scores, uint32_t *db_index)
{
    return guarded("oswald_hip_topr", [&] { return topr_impl(ctx, r, scores, db_index); });
}

int oswald_hip_comm_unique_id(void *id, size_t id_bytes)
{
    return guarded("oswald_hip_comm_unique_id", [&] { return comm_unique_id_impl(id, id_bytes); });
}

int oswald_hip_comm_init_rank(oswald_hip_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank)
{
    return guarded("oswald_hip_comm_init_rank", [&] { return comm_init_rank_impl(ctx, id, id_bytes, nranks, rank); });
}

int oswald_hip_comm_destroy(oswald_hip_ctx *ctx)
{
    return guarded("oswald_hip_comm_destroy", [&] { return comm_destroy_impl(ctx); });
}

int oswald_hip_comm_info(oswald_hip_ctx *ctx, int *out4)
{
    return guarded("oswald_hip_comm_info", [&] { return comm_info_impl(ctx, out4); });
}

int oswald_hip_merge_candidates(uint32_t nq, uint64_t ncand, const int32_t *cand_scores, const uint32_t *cand_index, uint32_t r, int32_t *scores, uint32_t *db_index)
{
    return guarded("oswald_hip_merge_candidates", [&] { return merge_candidates_impl(nq, ncand, cand_scores, cand_index, r, scores, db_index); });
}

int oswald_hip_set_profiling(oswald_hip_ctx *ctx, int enable)
{
    return guarded("oswald_hip_set_profiling", [&] { return set_profiling_impl(ctx, enable); });
}

int oswald_hip_kernel_stats(oswald_hip_ctx *ctx, int dev, double *dp_kernel_ms, uint64_t *dp_launches, uint64_t *rerun_items, int reset)
{
    return guarded("oswald_hip_kernel_stats", [&] { return kernel_stats_impl(ctx, dev, dp_kernel_ms, dp_launches, rerun_items, reset); });
}

int oswald_hip_rerun_stats(oswald_hip_ctx *ctx, int dev, double *ms2)
{
    return guarded("oswald_hip_rerun_stats", [&] { return rerun_stats_impl(ctx, dev, ms2); });
}

int oswald_hip_rerun_counts(oswald_hip_ctx *ctx, int dev, uint64_t *out2)
{
    return guarded("oswald_hip_rerun_counts", [&] { return rerun_counts_impl(ctx, dev, out2); });
}

int oswald_hip_chunk_geometry(oswald_hip_ctx *ctx, int dev, int chunk, uint64_t *out8)
{
    return guarded("oswald_hip_chunk_geometry", [&] { return chunk_geometry_impl(ctx, dev, chunk, out8); });
}

} // extern "C"
