// oswald_amd/csrc/sw_kernels.h -- device-side data layout shared by the HIP
// kernels (sw_kernels.hip) and the C-ABI layer (oswald_hip.cpp).
#ifndef OSWALD_SW_KERNELS_H
#define OSWALD_SW_KERNELS_H

#include <stdint.h>
#include <hip/hip_runtime.h>

#define OSW_WG_THREADS 256   // 4 waves per workgroup, each wave independent
#define OSW_RMAX16 48        // query rows per strip, packed int16 kernels (2 state registers per row; 168 VGPRs = three waves per SIMD)
#define OSW_LDS_ROWS16 192   // profile rows a wave keeps in LDS per round (12 KB; three workgroups of four waves per CU), packed int16 kernels
#define OSW_SEQ_CODES 24     // residue codes per row-block of the single-query int16 kernels' profile (16 B each: 96 B per query row)
#define OSW_LDS_ROWS16_SEQ 128 // ... of which a wave's 12 KB hold 128 rows (the query-pair profile: 32 codes x 16 B, OSW_LDS_ROWS16 / 2 rows)
#define OSW_LDS_ROWS32 256   // profile rows of a wave of the int32 re-run pipeline (osw_sw_i32r: 64 lane groups x 4 rows, 16 KB)
#define OSW_RMAX32F 48       // query rows per strip of the hand-scheduled int32 cell (whole searches with cell_bits = 32: the int16 kernels' register budget)
#define OSW_LDS_ROWS32F 192   // ... and its profile rows per wave (64 B per row: 12 KB, three workgroups per CU)
#define OSW_RMAX8 12         // query rows per strip, SWAR 8-bit kernel (compiler-scheduled; 12 rows keep it within the 80 VGPRs of six waves per SIMD)
#define OSW_LDS_ROWS8 96     // profile rows a wave keeps in LDS per round, SWAR 8-bit kernel (6 KB: six workgroups per CU)
#define OSW_LDS_SKEW8 128    // extra 8-byte units per wave region: group g's slice sits g entries (<= 16 B) further on, G <= 64
#define OSW_LDS_SKEW_Q8 64  // the same for the 8-bit kernel (8-byte entries)
#define OSW_BLOCK_SEQS 128   // database sequences per wave block (2 per lane)
#define OSW_SCRATCH_PAD_COLS 72  // spill scratch columns past the longest block (prefetch + drain of G <= 64), kept zero
// (a wave's spill region holds (columns + OSW_SCRATCH_PAD_COLS) x 32 {H,F} entries, sized from the longest sequence by
// oswald_hip.cpp::ensure_scratch: the boundary row of the longest block at two lane groups; longer blocks, or G = 1
// beyond half of it, run at a geometry with fewer lanes per group)
// A wave's spill region: 64 reserved entries, 64 entries that absorb the stores of steps / rounds that
// have nothing to spill, then the columns.  (The "row above" of a first round comes from top_pages.)
#define OSW_SCRATCH_ZERO 0
#define OSW_SCRATCH_TRASH 64
#define OSW_SCRATCH_DATA 128
#define OSW_TILED_PAD_GROUPS 18  // all-dummy 4-column groups stored after every block (prefetch + drain of G <= 64: 66 columns)
#define OSW_TILED_TAIL_GROUPS 2  // readable groups past the last block
#define OSW_I16S_TABLE 8448u     // entries of the column-frame cell's floor table (frame offsets stay <= 8192, + drain)
#define OSW_I32F_TABLE 66048u    // entries of the hand-scheduled int32 cell's floor table (cell_bits = 32 only): 65 535 columns + prefetch + drain of G <= 64
#define OSW_DUMMY_CODE8 0xB8u    // residue code 23 (dummy), pre-multiplied by 8 as stored in `tiled`

// Residue codes inside the device (round 5): the re-tile kernels store, and every profile is indexed by, a RELABELLED code.
// A wave's profile lookup is a ds_read_b128 per lane: 16-byte entries, a 256-byte bank row = 16 slots, lanes served in four
// groups of 16; two lanes of a group on one slot with DIFFERENT entries cost an extra LDS cycle.  The alphabet has 24 codes:
// codes c and c + 16 share a slot.  In the reference's order (A B C D E F G H I K L M N P Q R S T V W X Y Z) those pairs are
// (A,S) (B,T) (C,V) (D,W) (E,X) (F,Y) (G,Z) (H,dummy): A + S, 15 % of all residues, on one slot.  Relabelled by frequency
// (Robinson-Robinson background), the eight commonest residues L A G S V E T K get slots of their own (8..15), and every
// shared slot holds one of the rarest with one of the next rarest: (F,Y) (Q,M) (N,H) (R,C) (I,W) (P,X) (D,Z) (B,dummy) --
// the dummy, which every lane past the end of its sequence reads, shares with B.  Simulated with the hardware's lane groups
// (tools/lds_sim.py): 6.9 -> 5.6 LDS cycles per ds_read_b128 for a lane group of 16 on one table.  The dummy stays 23.
// Scores do not depend on it: the profile builder looks the matrix up by the ORIGINAL code of a slot.
// OSW_RELABEL_*: relabelled code of codes 0..23, four per word; OSW_ORIGINAL_*: original code of slots 0..31 (24..31: the dummy's)
#define OSW_RELABEL_W0 0x06130709u
#define OSW_RELABEL_W1 0x120a000du
#define OSW_RELABEL_W2 0x11080f04u
#define OSW_RELABEL_W3 0x03010502u
#define OSW_RELABEL_W4 0x140c0e0bu
#define OSW_RELABEL_W5 0x17161015u
#define OSW_ORIGINAL_W0 0x0f0c0e05u
#define OSW_ORIGINAL_W1 0x01030d08u
#define OSW_ORIGINAL_W2 0x1006000au
#define OSW_ORIGINAL_W3 0x09110412u
#define OSW_ORIGINAL_W4 0x02070b15u
#define OSW_ORIGINAL_W5 0x17161413u

// Work item: x = query | sub-block << 16 | log2(G) << 24 | halves << 28 | priority << 30, y = block.
// G = lane groups of the wave geometry; sub-block = which 64/G lanes (sequence
// pairs) of the block; halves (int32 kernel) = which sequence of each pair.
#define OSW_ITEM_PACK(q, sigma, lg, halves) ((uint32_t)(q) | ((uint32_t)(sigma) << 16) | ((uint32_t)(lg) << 24) | ((uint32_t)(halves) << 28))
#define OSW_ITEM_Q(x) ((x) & 0xffffu)
#define OSW_ITEM_SIGMA(x) (((x) >> 16) & 0xffu)
#define OSW_ITEM_LG(x) (((x) >> 24) & 0xfu)
#define OSW_ITEM_HALVES(x) (((x) >> 28) & 3u)
#define OSW_ITEM_PRIO(x) (((x) >> 30) & 3u)
#define OSW_ITEM_WG_FLAG 0x80000000u  // in y of a phase-1 slot: workgroup item (shared profile slice, waves in step)
#define OSW_ITEM_NONE 0x7fffffffu     // y of an empty slot
#define OSW_ITEM_SHORT 1u             // `halves` field of a pair item: the item runs the shorter query's rows as a pair and the rest of the longer query as its TAIL (below)
#define OSW_TAIL_GEOMS 7u             // a pair's tail has a length and a place in the profile for each log2 geometry 0 .. 6 of the item that runs it

// Strip plan of a query of m rows at geometry G: every lane group owns
// T = ceil4(ceil4(m) / G) rows, cut into `rounds` strips whose heights are
// multiples of 4, differ by at most 4 and never exceed min(rmax, lds_rows / G)
// (G x height rows of profile must fit the LDS slice).  Rows past the query are
// zero-score rows: they cannot raise a maximum; at most 4G-1 of them exist.
// Shared by the host (work-queue costs) and the kernels.
struct OswPlan {
    uint32_t rounds;  // strips per lane group
    uint32_t base;    // every round has 4*base rows ...
    uint32_t extra;   // ... and the first `extra` rounds 4 more
    uint32_t m4;      // query rows rounded up to a multiple of 4
};

static __host__ __device__ inline OswPlan osw_plan(uint32_t m, uint32_t G, uint32_t lds_rows, uint32_t rmax)
{
    OswPlan p;
    p.m4 = (m + 3u) & ~3u;
    if (p.m4 == 0) p.m4 = 4;
    uint32_t rcap = (lds_rows / G) & ~3u;
    if (rcap > rmax) rcap = rmax;
    if (rcap < 4) rcap = 4;
    const uint32_t units = ((p.m4 + G - 1) / G + 3u) / 4u; // rows per group / 4
    p.rounds = (units * 4 + rcap - 1) / rcap;
    p.base = units / p.rounds;
    p.extra = units % p.rounds;
    return p;
}
// height of round rho and the number of rows each group has done before it
static __host__ __device__ inline uint32_t osw_round_rows(const OswPlan &p, uint32_t rho) { return 4u * (p.base + (rho < p.extra ? 1u : 0u)); }
static __host__ __device__ inline uint32_t osw_round_row0(const OswPlan &p, uint32_t rho) { return 4u * (rho * p.base + (rho < p.extra ? rho : p.extra)); }
static __host__ __device__ inline uint32_t osw_plan_maxrows(const OswPlan &p) { return 4u * (p.base + (p.extra ? 1u : 0u)); }

// device counters (uint32), zeroed before every search launch pair
#define OSW_CTR_WORK 0       // next item of the pk16 queue
#define OSW_CTR_OVF 1        // number of items queued for the int32 re-run
#define OSW_CTR_WORK32 2     // next item of the int32 queue
#define OSW_CTR_WORK_WG 3     // next workgroup-cooperative (heavy) item of the pk16 queue
#define OSW_CTR_FRONT 4       // wave items taken from the heavy end / light end of the queue
#define OSW_CTR_BACK 5
#define OSW_CTR_FRONT_WG 6    // same for workgroup items
#define OSW_CTR_BACK_WG 7
#define OSW_CTR_CU0 8         // per-CU arrival counters (which workgroup came first on a CU)
#define OSW_CTR_CUS 4096
#define OSW_CTR_COUNT (OSW_CTR_CU0 + OSW_CTR_CUS)
#define OSW_CTR_BLOCKS 3      // counter blocks per device: single-query launch, query-pair launch, int16 re-run of the 8-bit pass
// behind the blocks, shared by all launches of a search: [0] = items queued for the int32 kernel,
// [1] = items queued by the 8-bit pass for the int16 re-run

// One wave block of the re-tiled chunk: 128 consecutive sequences of the
// (length-sorted) chunk, stored column-major: tiled[col*64 + lane] = uint16 {8*residue of
// sequence 2*lane, 8*residue of sequence 2*lane+1} (8*code = the byte offset of the code's
// profile entry); allocation is counted in groups of 4 columns (512 B).
struct OswBlock {
    uint32_t col4_off;      // first 4-column group of the block in `tiled`
    uint32_t ncols4_alloc;  // groups stored (from the caller's padded lengths)
    uint32_t ncols4;        // groups that hold at least one real residue
    uint32_t seq0;          // first sequence (column of the score row)
};

struct OswSearchArgs {
    const uint16_t *tiled;     // [col][64 lanes] {8*residue of seq 2l, 8*residue of seq 2l+1}
    const OswBlock *blocks;
    const uint16_t *sub_cols;  // [block][128]: live columns of sub-block sigma at geometry G, at (G-1) + sigma
    const uint2 *items;        // work queues, heaviest first: nitems_wg phase-1 entries of four slots (one per wave of the
                               // workgroup: the four sub-blocks of a workgroup item, or a quad of independent heavy
                               // wave items), then nitems wave items
    uint32_t nitems;
    uint32_t nitems_wg;
    uint32_t two_ended_waves;  // wave items: eat the queue from both ends (see osw_sw_pk16) or heaviest-first only
    uint32_t one_ended_wg;     // phase-1 entries: every workgroup takes the heaviest entry left (else: the first workgroup of a CU the heaviest, the others the lightest)
    uint32_t force_all;        // int32 kernel: run `items` instead of the overflow queue
    const uint2 *prof;         // [(prof_off[q] + i/4)*32 + code] = 4 x int16 (column-frame kernels: S + ge); the single-query int16 kernels
                               // (osw_sw_pk16 / osw_sw_s16): uint4 entries, a 32-bit word {S, 1} per row; query-pair kernels: uint4, 4 x (S_A, S_B)
    const uint2 *prof_fb;      // column-frame kernels: the plain profile of the same queries / pairs (blocks run on the plain cell)
    const uint2 *prof_i32;     // the hand-scheduled int32 cell's profile: S + ge, 8 B per code = 4 rows x int16 (osw_sw_i32, osw_sw_i32r)
    const uint2 *floor_i32;    // ... and its floor table: entry k = {k * ge, k * ge}, OSW_I32F_TABLE entries (the row above a first round)
    const uint32_t *prof_off;
    const uint16_t *qlen;
    const uint2 *top_pages;    // constant {H,F} entries, the row above a first round: 64 of zeros, 64 of the biased-int16 floor,
                               // then the column-frame cell's floor table (entry k = 1024 + k * ge)
    uint2 *bnd;                // strip-boundary spill {H,F} per column and lane, one region per resident wave
    uint64_t bnd_stride;       // uint2 per region (OSW_SCRATCH_DATA + columns x lanes per group)
    int32_t *scores;           // [nq][score_stride]
    uint32_t score_stride;
    // the caller's table, when it lies in page-locked memory the device can write (oswald_hip_host_alloc / _register, or pinned by
    // the library for the call): every score is ALSO stored there by the kernel that computes it -- column seq of row q at
    // scores_host[q * host_stride + seq] for seq < host_cols (the chunk's ngroups x W lanes; a wave block's padding lanes beyond them
    // have no place in the caller's row) -- and no copy, no download stream and no event stand between a search and the next one
    int32_t *scores_host;
    uint32_t host_stride, host_cols;
    uint32_t *counters;        // this launch's queue counters (OSW_CTR_*)
    uint32_t *counters_ovf;    // shared by all launches of a search: [0] = items queued for the int32 kernel, [1] = for the int16 re-run
    const uint32_t *nitems_dev;// packed-int16 kernels: if set, the number of workgroup entries is read from the device (re-run queue of the 8-bit pass)
    uint2 *ovf8_items;         // 8-bit kernel: the queue it fills for the int16 re-run
    uint32_t bias8, go8, ge8, off8; // 8-bit kernel: profile bias, gap open, gap extend and the cell's offset c (q8_cell.h), plain values
    const uint32_t *pair_q;    // query-pair kernel: the two queries of pair i (rows of the score table)
    // Tails.  A query pair pads its shorter query to the longer one (two passes of the pair cell over rows that hold one query).  A pair
    // item marked OSW_ITEM_SHORT instead runs the SHORTER query's rows as a pair (pair_rows, rounded up to 4; at geometry G its strips
    // end on row X(G) = 4 G ceil(ceil(rows / G) / 4) of the longer query, whose profile rows are all there) and then -- the same wave,
    // the same sub-block, the same geometry -- the rest of the longer query on the SINGLE-query cell, both sequences of a lane at once
    // (6.5 instead of 13 instructions per row and sequence pair).  The bottom row {H, F} of the pair's last round is the tail's row
    // above: pass 0 (a lane's first sequence) leaves it in the wave's HAND region, pass 1 in the wave's own spill region (in place,
    // like every round boundary); the wave merges the longer query's halves of the two into its spill region and goes on from there.
    // (Round 5 ran the tails as a launch of their own and handed over through planes as large as 8 x the chunk per pair: 11.7 GB for
    // the BASELINE set.  Here nothing leaves the wave: the hand regions are the half of the spill scratch that a launch of single
    // queries beside the pairs' would use -- tails are planned for query sets without such a launch.)
    uint2 *hand;               // pair launch: a second region per resident wave, laid out like `bnd` (same stride); null: no SHORT items in the queue
    const uint16_t *pair_rows; // [pair] rows a SHORT item runs as a pair
    const uint16_t *tail_len;  // [pair * OSW_TAIL_GEOMS + lg] rows of the tail behind a SHORT item of geometry 2^lg (0: the strips took the whole query)
    const uint32_t *tail_off;  // ... and its first row-block in the single-query profiles (tail_prof / tail_prof_fb)
    const uint2 *tail_prof;    // the single-query kernels' profile of the same query set ({S, 1} entries; column-frame kernels: S + ge)
    const uint2 *tail_prof_fb; // column-frame kernels: ... and the plain one (blocks run on the plain cell)
    uint2 *ovf_items;
    uint32_t goe_pk, ge_pk;    // (open+extend, extend) replicated in both halves; column-frame kernels: (open, extend)
    uint32_t goe_fb, ge_fb;    // column-frame kernels: (open+extend, extend) for the plain cell
    int32_t goe, ge;
    uint32_t debug_nospill;    // -DOSW_DIAG builds only: every round reads the constant top row and stores to the trash page (WRONG scores; timing only)
    unsigned long long *wg_times; // -DOSW_DIAG builds only (or null): per workgroup {start, end of phase 1, end, end} in 100 MHz ticks
};

// host-side launchers, defined in sw_kernels.hip
hipError_t osw_launch_pk16(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_i32(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_i32r(const OswSearchArgs &a, uint32_t regions, hipStream_t s); // the re-run queue: workgroups of eight waves, one item each
hipError_t osw_launch_pk16q(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_s16(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_s16q(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_s16qt(const OswSearchArgs &a, uint32_t grid, hipStream_t s);   // the query-pair kernels whose queue may hold SHORT items (tails, OswSearchArgs::hand)
hipError_t osw_launch_pk16qt(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_q8(const OswSearchArgs &a, uint32_t grid, hipStream_t s);
hipError_t osw_launch_build_pair_profile8(const uint2 *prof, const uint32_t *prof_off, const uint16_t *qlen, const uint32_t *pair_q,
                                          const uint32_t *pair_off, const uint16_t *pair_len, uint32_t npairs, uint32_t max_rowblocks,
                                          int bias, uint2 *prof_pair8, hipStream_t s);
hipError_t osw_launch_build_pair_profile(const uint2 *prof, const uint32_t *prof_off, const uint16_t *qlen, const uint32_t *pair_q,
                                         const uint32_t *pair_off, const uint16_t *pair_len, uint32_t npairs, uint32_t max_rowblocks,
                                         bool intsum, uint4 *prof_pair, hipStream_t s);
hipError_t osw_launch_fill(void *p, uint8_t byte, size_t bytes, hipStream_t s); // a search-shaped fill (see sw_kernels.hip)
// the int32 cell's floor table, built on the device: t[k] = {k * ge, k * ge}, k < n
hipError_t osw_launch_floor_i32(uint2 *t, uint32_t n, uint32_t ge, hipStream_t s);
hipError_t osw_launch_copy16(const void *src_host_pinned, void *dst, size_t bytes, hipStream_t s); // page-locked host -> device by a kernel that reads the host buffer in place
hipError_t osw_launch_spin(uint32_t *sink, uint32_t grid, double ms, const uint32_t *stop_host_pinned, hipStream_t s); // every CU busy for `ms`, or until *stop_host_pinned != 0 (the warm-up of oswald_hip_init: clocks up before a first search)
hipError_t osw_warm_aux_kernels(hipStream_t s); // first launches of the profile / top-list kernels (bring-up)
hipError_t osw_launch_retile(const uint8_t *b, const uint16_t *n, const uint32_t *disp, uint32_t ngroups, uint32_t W,
                             OswBlock *blocks, uint32_t nblocks, uint16_t *tiled, uint16_t *sub_cols, hipStream_t s);
hipError_t osw_launch_build_profile(const uint8_t *a, const uint32_t *a_disp, const uint16_t *qlen, const uint32_t *prof_off,
                                    const int8_t *submat, uint32_t nq, uint32_t max_rowblocks, int add, uint2 *prof, uint4 *prof_seq, hipStream_t s);
// (prof_seq, or null: the same entries as the sequence-pair cell reads them: [(prof_off[q] + i/4) * OSW_SEQ_CODES + code], 16 B per code
// and 4 rows, a 32-bit word {S, 1} per row; codes 0..23 only -- the re-tile kernels store every residue code >= 24 as 23, the dummy)
uint32_t osw_topr_parts(uint32_t nvalid); // partitions per score row; `cand` holds nq * parts * r tagged keys
hipError_t osw_launch_topr(const int32_t *scores, uint32_t score_stride, uint32_t nvalid, uint32_t r, uint32_t nq,
                           unsigned long long *cand, int32_t *out_scores, uint32_t *out_index, hipStream_t s);
// context-level top-r (oswald_hip_topr): lists of r tagged keys ((score << 32 | database index) << 1 | 1; 0 = none) per query
hipError_t osw_launch_topr_fold_chunk(const int32_t *scores, uint32_t score_stride, uint32_t nvalid, uint32_t r, uint32_t nq,
                                      const uint32_t *index_map, uint32_t first_index, unsigned long long *cand,
                                      const unsigned long long *run_in, unsigned long long *run_out, hipStream_t s);
hipError_t osw_launch_topr_fold_lists(const unsigned long long *lists, uint32_t L, uint64_t list_stride, uint32_t r, uint32_t nq,
                                      unsigned long long *out, hipStream_t s);
hipError_t osw_launch_topr_fold_lists2(const unsigned long long *a, const unsigned long long *b, uint32_t r, uint32_t nq, unsigned long long *out, hipStream_t s);
hipError_t osw_launch_topr_untag(const unsigned long long *keys, uint32_t nq, uint32_t r, uint32_t r_out, int32_t *out_scores, uint32_t *out_index,
                                 hipStream_t s);
int osw_occupancy_pk16(int *blocks_per_cu);
#ifdef OSW_DIAG
// osw_diag.cpp (the -DOSW_DIAG build only): per-workgroup time stamps of a DP launch as text on stderr
void osw_diag_report_times(const unsigned long long *t, uint32_t grid, uint32_t reload_k);
#endif
int osw_occupancy_q8(int *blocks_per_cu);

#endif
