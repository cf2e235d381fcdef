// oswald_amd/csrc/sw_kernels.hip
//
// Hand-written HIP kernels for gfx950 (MI355X / CDNA4) that replace OSWALD's
// FPGA kernel `sw` (reference device/sw.cl:16-94) and the host-side score
// profile build feeding it (reference host/src/FPGAsearch.c:143-177).
//
// Formulation (MI355X-first, not a translation of the 28-wide FPGA pipeline):
//   * inter-sequence parallel: a wave works on a "block" of 128 database
//     sequences, 2 per lane, packed as the two 16-bit halves of a VGPR.  No
//     MFMA: this is a max-plus DP.  Packed 16-bit VALU issues at ~4 cycles per
//     wave instruction per SIMD on gfx950 (measured, tools/ubench.hip), so the
//     cell is kept as short as the ISA allows and hand-scheduled:
//       - packed int16 values carrying a bias of 1024, on which gfx950's
//         v_pk_maximum3_f16 is an integer max3, stored relative to a per-column
//         frame that makes the horizontal gap decay free: 6.5
//         instructions per row -- also when a lane holds two sequences of ONE
//         query: a v_pk_mad_i16 with half selects pairs their two scores up and
//         adds them to the diagonal (OSW_TADD_MAD) --,
//         exact below 22256 (plain biased cell for blocks whose frame would not
//         fit: 7.5 per row, exact below 30576);
//       - sequences that reach the ceiling are queued for the int32 kernel
//     (the reference's int8->int16->int32 escalation, host/src/HybridSearch.c:
//     1670-1680,:1774-1784, yields exact scores; so does this).
//   * the query is cut into strips of R <= 48 rows held in registers (E and
//     the diagonal H of every row, 2 VGPRs per row; three waves per SIMD: 168
//     VGPRs and a 12 KB profile slice in LDS per wave -- round 3; it was 32 rows,
//     four waves and 8 KB, which cost a third more round boundaries and gained
//     nothing: a gfx950 SIMD issues these instruction streams at the same rate
//     from three waves as from four); database columns stream through.  The strip's slice of the query profile lives in a wave-private
//     LDS region and is read with ds_read_b128 (the 16-byte entry of a residue
//     code: 4 rows; address = 2 * (8*residue) + imm; `tiled` stores 8*residue).
//   * "wave geometry" G (1,2,4,...,64): the 64 lanes form G groups of 64/G
//     lanes.  Group g runs strip (round*G + g) of the SAME 128/G sequences, one
//     column behind group g-1, and receives that group's bottom row (H, F)
//     through ds_bpermute -- a systolic array inside the wave, no inter-wave
//     synchronisation.  G = 1 is the plain case; G groups keep G strips'
//     boundaries in registers (only every G-th strip boundary touches HBM) and
//     shorten the critical path of heavy (long query x long sequence) items by
//     G so that the work queue balances; G = 64 is the exact int32 re-run of
//     single sequences.
//   * query pairs: the two halves of a register can also hold two QUERIES of
//     similar length against one sequence per lane (no v_perm_b32).
//   * workgroup items: the four waves of a workgroup run four sub-blocks of one
//     heavy item and share ONE 4x larger profile slice (taller rounds at the
//     same G); two workgroup barriers per round are the only synchronisation.
//   * between rounds the bottom row of the last group spills to a wave-private
//     HBM scratch {H, F} per column and lane and is loaded back two columns
//     ahead, straight into fixed registers (see "Input registers of a column
//     step"): the column loop spends 5 VALU instructions besides the cells.
//   * work items (query or pair, block, sub-block, G) are pulled from atomic
//     queues sorted by cost (planned on the host, oswald_hip.cpp::build_items),
//     so one launch covers all queries of a chunk; every wave exits when the
//     queues are drained: nothing ever waits for another workgroup.
//
// Recurrence (reference sw.cl:60-78): H = max(0, Hdiag + S, E, F);
// E,F <- max(E|F - ge, H - (go+ge)).  E and F are kept clamped at >= 0, which
// is equivalent because they only ever enter a max with H >= 0; the floor is
// the third operand of the maximum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sw_kernels.h"

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 *lds_u2p;
typedef __attribute__((address_space(3))) const char *lds_cp;
typedef __attribute__((address_space(3))) volatile uint32_t *lds_flagp; // a progress word of the re-run pipeline (LDS: ds_read / ds_write, not a flat access)

static __device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
static __device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }

// ---------------------------------------------------------------------------
// Input registers of a column step.  H and F of the row above and the residues
// of the column live in two alternating register sets (even / odd column):
//   H, F : written for ALL lanes by the ds_bpermute hand-off at the end of the
//          previous step (lane groups 1.. take the bottom row of the group
//          below; the lanes of group 0 receive junk); one register each: a
//          step reads them before its own hand-off overwrites them,
//   LH,LF: for the lanes of group 0, the boundary row of the previous round,
//          LOADED from the spill scratch two columns ahead; merged into H / F
//          with one EXEC-masked v_mov each,
//   C    : the residues, loaded two columns ahead by every lane itself (group g
//          is g columns behind, so its lanes read g columns further back).
// A register with a load in flight must not be touched by anything else, and
// the compiler cannot know about loads issued from inline asm: it would be
// free to copy such a register (reading it too early) or to park a temporary
// in it.  The sets are therefore FIXED physical registers that never appear as
// asm operands; every asm statement of the packed-int16 kernels names them as
// clobbers, so the compiler keeps no value in them, and they are read and
// written by name inside the asm text only.  tests/test_isa_inflight.py checks
// the generated ISA: no compiler-scheduled instruction touches them.
// ---------------------------------------------------------------------------
#define OSW_VH "v160"   // H of the row above: handed over at the end of a step, read at the end of the next
#define OSW_VF "v161"   // F of the row above, then the F chain of the step (in place), then handed on
#define OSW_VC0 "v162"  // residues, even / odd column (loaded two columns ahead)
#define OSW_VC1 "v163"
#define OSW_VLH0 "v164" // boundary row of the previous round, group 0, even / odd column (loaded two ahead)
#define OSW_VLF0 "v165"
#define OSW_VLH1 "v166"
#define OSW_VLF1 "v167"
// ... and, since round 4, the registers the cells of a column work through besides their per-row state:
#define OSW_VT "v159"   // a row's temporary (u = H - gap penalty)
#define OSW_PA0 "v150"  // profile buffer A: the substitution scores of one 4-row block (ds_read_b128, or two ds_read_b64)
#define OSW_PA1 "v151"
#define OSW_PA2 "v152"
#define OSW_PA3 "v153"
#define OSW_PA_ALL "v[150:153]"
#define OSW_PB0 "v154"  // profile buffer B: the next block's (the two alternate)
#define OSW_PB1 "v155"
#define OSW_PB2 "v156"
#define OSW_PB3 "v157"
#define OSW_PB_ALL "v[154:157]"
// sequence-pair cell (round 4, third session): a profile buffer is EIGHT registers -- the 4-row entries of the lane's first and of its
// second sequence (two ds_read_b128) -- so its even blocks use v150..v157 as {first: v150..v153, second: v154..v157} and its odd
// blocks a second set, v140..v147; v148 / v149 (the v_perm_b32 pair registers of the old cell) are spare
#define OSW_SX0 "v150, v154" // even blocks: the (first, second sequence) entries of rows 0..3 as the two sources of the row's v_pk_mad_i16
#define OSW_SX1 "v151, v155"
#define OSW_SX2 "v152, v156"
#define OSW_SX3 "v153, v157"
#define OSW_SX_A "v[150:153]"
#define OSW_SX_B "v[154:157]"
#define OSW_SY0 "v140, v144" // odd blocks
#define OSW_SY1 "v141, v145"
#define OSW_SY2 "v142, v146"
#define OSW_SY3 "v143, v147"
#define OSW_SY_A "v[140:143]"
#define OSW_SY_B "v[144:147]"
#define OSW_INFLIGHT "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
// ... and the compiler is given v0..v139 only (amdgpu_num_vgpr caps what its register allocator may use; the kernel's
// register count still comes out as 168 -- three waves per SIMD -- because the asm statements clobber the fixed ones): the
// fixed registers are out of its reach by construction, not by the luck of an allocation order (round 2 had them at
// v120..v127 inside the compiler's range; with a larger budget the allocator did park temporaries there between asm
// statements).  tools/isa_check.py verifies it on the ISA.  (The compiler uses 125 - 139 of the 140.)
#define OSW_COMPILER_VGPRS __attribute__((amdgpu_num_vgpr(140)))

// ---------------------------------------------------------------------------
// The cell, hand-scheduled.  State per row r: E[r] and D[r] = H(i0+r-1, j-1),
// the diagonal input of row r (D[0] is only a name: the top input comes from
// the strip above).  A row also issues the diagonal add of row r+1 *before* it
// overwrites D[r+1] with its own H, so every state register is updated in place
// (no copies), and every result is consumed at a distance of >= 2 issue slots,
// which is what gfx950 needs between a packed-math write and a dependent VALU
// read (no s_nop inside the cell).  F runs down the column in the F register of
// the step's input set (FREG), where the next strip picks it up.
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// Packed-int16 cell with the three-operand maximum ("biased int16").  For
// non-negative 16-bit patterns below 0x7C00 the fp16 ordering IS the integer
// ordering, so v_pk_maximum3_f16 computes an integer max3 -- as long as no
// operand is a NaN pattern (negative small integers are) or a flushed
// denormal (integers below 1024 are denormal patterns).  All values therefore
// carry a bias B = 1024 (the smallest normal fp16): H, E, F >= B always; the
// diagonal sum D + S >= B - 128 may be a denormal, but then it loses against E
// and F either way; the gap terms use the unsigned saturating subtract, so they
// are never negative.  Adds and subtracts are plain packed int16: nothing is
// rounded.  Patterns from 0x7C00 (31744) on would be inf / NaN: a sequence whose
// biased score reaches 31600 (true score 30576) is re-run in int32, and below
// that threshold no sum can reach 31744 (|S| <= 128).  7.5 VOP3P instructions
// per row (9 with two-operand maxima).
//   floor (the value of "zero"): B in both halves; profile: the plain int16 one
// ---------------------------------------------------------------------------
#define OSW_I16B_BIAS 0x04000400u

// ---------------------------------------------------------------------------
// "Column frame" variant of the biased int16 cell: 6.5 instructions per row.
// Every value of database column j is stored as  true + 1024 + (j + G) * ge.
// The horizontal gap E then needs no decay at all -- in the frame of column
// j+1 the same pattern stands for E - ge -- and the vertical one is taken out
// of the maximum:  F' = max(F - ge, H - goe, 0)  =  max3(F, H - go, fl1) - ge,
// with the SAME  u = H - go  (go = gap open) and the same floor fl1 ("zero" in
// the frame of column j+1, a per-lane register bumped once per column) as
// E' = max3(E, u, fl1).  The diagonal H(i-1, j-1) sits one frame back, which the
// profile absorbs (it stores S + ge).  Per row: diagonal add, H = max3, u,
// E = max3, F = max3, F - ge, and 1/2 for the column maximum; per column the
// maximum is brought back to a true score (cm - "zero of column j") and folded
// into the running one.  The frame offset must stay small: an item takes this
// cell only if (columns + 2G + 2) * ge <= 8192 and ge <= 64 (else the plain biased
// cell), and a sequence scoring 22256 or more is re-run in int32: the largest
// pattern of a sequence that stays below is 22255 + 1024 + 8192, and a diagonal
// sum adds at most 127 + ge to it: below 0x7C00.
// The subtracts are plain 32-bit VOP2 on the packed pair: gfx950 issues v_add_u32 / v_sub_u32 in ~2 cycles against
// ~4 for any VOP3P instruction (profiles/r02_oprate_valu_issue.txt; in the cell's mix a row goes from 27.7 to
// 25.3 cycles, profiles/r02_oprate2_valu_mix.txt).  They are exact whenever no borrow crosses the halves:
// every H, E, F pattern is >= 1024 >= go, ge (the kernel checks go <= 1024; ge <= 64 anyway).  The diagonal add is a
// v_add_u32 too when the profile stores its (S_lo, S_hi) pairs as the 32-bit INTEGER S_lo + 65536 * S_hi (query-pair
// profile, osw_build_pair_profile), so that the sum is right in both halves although S_lo may be negative; the
// sequence-pair cell's diagonal add is the v_pk_mad_i16 that also pairs the lane's two scores up (OSW_TADD_MAD).
// ---------------------------------------------------------------------------
#define OSW_I16S_FRAME_MAX 8192u

// ---------------------------------------------------------------------------
// The rows as text.  A block of four rows is ONE asm statement (round 4; it was one per row).  gfx950 has a
// forwarding hazard behind instructions that write part of a register (SDWA dst_sel, op_sel), and LLVM's hazard
// recogniser, which cannot see into an asm statement, assumes every statement ends in one: it puts an `s_nop 0` in front
// of any instruction or statement that touches a register the statement before it defines -- outputs and clobbers alike,
// other asm statements in between counting for nothing.  With a statement per row, per load and per wait that was 73-75
// s_nop in a 48-row column, ~0.9 cycles each at three waves per SIMD (tools/oprate4.hip, probe fs_nop; the statements
// all end in full 32-bit writes: none has the hazard).  With the loads and the waits inside the block
// statement -- the profile buffers and the temporaries are fixed registers for that -- 15 are left.
//   register / operand names are strings: XN = diagonal sum of the next row (out), DN = D[r+1] (in: H(r, j-1), out:
//   H(r, j)), SN = score pair of the next row, X = this row's diagonal sum, E = E[r]; HOOK = text issued behind the
//   add (loads, waits), SCMAX = the half instruction of the running / column maximum or ""
// ---------------------------------------------------------------------------
#define OSW_TADD_PK(XN, DN, SN) "v_pk_add_i16 " XN ", " DN ", " SN " clamp\n\t"
#define OSW_TADD_32(XN, DN, SN) "v_add_u32 " XN ", " DN ", " SN "\n\t"
// sequence-pair cell: SN = "Ea, Eb", the 32-bit profile entries {low half: S, high half: 1} of the row for the lane's first and second
// sequence; x.lo = Ea.lo * Eb.hi + D.lo = S(a) + D.lo, x.hi = Ea.hi * Eb.lo + D.hi = S(b) + D.hi: the half selects of v_pk_mad_i16 pair
// the two scores up AND add them to the diagonal in one instruction (tools/oprate8.hip checks the instruction and prices the row:
// 29.1 cycles against 33.1 with v_perm_b32 + packed add, profiles/r04_oprate8_pk_mad.txt).  An entry beyond the query is all zero
// (fill_profile_slice): 0 * 0 + D = D, a zero score.
#define OSW_TADD_MAD(XN, DN, SN) "v_pk_mad_i16 " XN ", " SN ", " DN " op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
#define OSW_SCMAX(SC, SCI, DP, DN) "v_pk_maximum3_f16 " SC ", " SCI ", " DP ", " DN "\n\t"
// plain biased cell (go_ slot: open + extend; fl_: the bias, a constant)
#define OSW_B_ROW(ADD, XN, DN, SN, X, E, HOOK, SCMAX)                              \
    ADD(XN, DN, SN) HOOK                                                           \
    "v_pk_maximum3_f16 " DN ", " X ", " E ", " OSW_VF "\n\t"                       \
    "v_pk_sub_u16 " E ", " E ", %[ge_] clamp\n\t"                                  \
    "v_pk_sub_u16 " OSW_VT ", " DN ", %[go_] clamp\n\t"                            \
    "v_pk_sub_u16 " OSW_VF ", " OSW_VF ", %[ge_] clamp\n\t"                        \
    SCMAX                                                                          \
    "v_pk_maximum3_f16 " E ", " E ", " OSW_VT ", %[fl_]\n\t"                       \
    "v_pk_maximum3_f16 " OSW_VF ", " OSW_VF ", " OSW_VT ", %[fl_]\n\t"
#define OSW_B_ROW_LAST(HL, X, E, SCMAX)                                            \
    "v_pk_maximum3_f16 " HL ", " X ", " E ", " OSW_VF "\n\t"                       \
    "v_pk_sub_u16 " E ", " E ", %[ge_] clamp\n\t"                                  \
    "v_pk_sub_u16 " OSW_VT ", " HL ", %[go_] clamp\n\t"                            \
    "v_pk_sub_u16 " OSW_VF ", " OSW_VF ", %[ge_] clamp\n\t"                        \
    SCMAX                                                                          \
    "v_pk_maximum3_f16 " E ", " E ", " OSW_VT ", %[fl_]\n\t"                       \
    "v_pk_maximum3_f16 " OSW_VF ", " OSW_VF ", " OSW_VT ", %[fl_]"
// column-frame cell (go_ slot: gap open; fl_: the floor of the next column's frame, a register)
#define OSW_S_ROW_T(ADD, PAUSE1, PAUSE2, XN, DN, SN, X, E, HOOK, SCMAX)             \
    ADD(XN, DN, SN) HOOK                                                           \
    "v_pk_maximum3_f16 " DN ", " X ", " E ", " OSW_VF "\n\t"                       \
    "v_subrev_u32 " OSW_VT ", %[go_], " DN "\n\t" PAUSE1                           \
    SCMAX                                                                          \
    "v_pk_maximum3_f16 " E ", " E ", " OSW_VT ", %[fl_]\n\t"                       \
    "v_pk_maximum3_f16 " OSW_VF ", " OSW_VF ", " OSW_VT ", %[fl_]\n\t"             \
    "v_subrev_u32 " OSW_VF ", %[ge_], " OSW_VF "\n\t" PAUSE2
// The rate at which three waves get these rows through a SIMD depends on where a wave steps aside (tools/oprate5.hip: 25.5 - 27.5
// cycles per row over the placements of an s_nop 0, 26.2 with none; behind a 32-bit VOP2 it helps, behind a VOP3P it hurts).
// Query-pair kernel (profiles/r04_nop_sweep.txt): a one-cycle pause behind each of the two subtracts -- none costs it 3.5 %, this
// placement is the best of the 14 tried.  Sequence-pair kernel (its diagonal add is the VOP3P v_pk_mad_i16; tools/ab_seq_pause.sh,
// two runs each on Q1 / C5 at 1 M sequences): behind the first subtract only +0.8 / +0.9 %, behind the second only +0.7 / +0.7 %,
// behind both -1.2 / -0.8 %.
#define OSW_S_ROW(ADD, XN, DN, SN, X, E, HOOK, SCMAX) OSW_S_ROW_T(ADD, "s_nop 0\n\t", "", XN, DN, SN, X, E, HOOK, SCMAX)
#define OSW_S_ROWP(ADD, XN, DN, SN, X, E, HOOK, SCMAX) OSW_S_ROW_T(ADD, "s_nop 0\n\t", "s_nop 0\n\t", XN, DN, SN, X, E, HOOK, SCMAX)
#define OSW_S_ROW_LAST(HL, X, E, SCMAX)                                            \
    "v_pk_maximum3_f16 " HL ", " X ", " E ", " OSW_VF "\n\t"                       \
    "v_subrev_u32 " OSW_VT ", %[go_], " HL "\n\t"                                  \
    SCMAX                                                                          \
    "v_pk_maximum3_f16 " E ", " E ", " OSW_VT ", %[fl_]\n\t"                       \
    "v_pk_maximum3_f16 " OSW_VF ", " OSW_VF ", " OSW_VT ", %[fl_]\n\t"             \
    "v_subrev_u32 " OSW_VF ", %[ge_], " OSW_VF

// Four rows of a QUERY-PAIR cell: C1..C3 = the score pairs of the block's rows 1..3 (its profile buffer's .y .z .w),
// N0 = row 0 of the next block (the other buffer's .x); MID = the load of the block after next into this block's
// buffer (all of it has been read by then) and the wait for the next block's, in front of the last row.
#define OSW_SC_RUN(DP, DN) OSW_SCMAX("%[sc_]", "%[sc_]", DP, DN)
#define OSW_SC_START(DP, DN) OSW_SCMAX("%[sc_]", DP, DN, DN) /* first odd row of a column-frame column: starts the column maximum */
#define OSW_QP_ROWS3(ROW, ADD, C1, C2, C3, SC1)                                    \
    ROW(ADD, "%[xb_]", "%[D1_]", C1, "%[x_]", "%[E0_]", "", "")                    \
    ROW(ADD, "%[x_]", "%[D2_]", C2, "%[xb_]", "%[E1_]", "", SC1("%[D1_]", "%[D2_]")) \
    ROW(ADD, "%[xb_]", "%[D3_]", C3, "%[x_]", "%[E2_]", "", "")
#define OSW_QP_MID(ROW, ADD, C1, C2, C3, N0, SC1, MID)                             \
    OSW_QP_ROWS3(ROW, ADD, C1, C2, C3, SC1) MID                                    \
    ROW(ADD, "%[x_]", "%[D4_]", N0, "%[xb_]", "%[E3_]", "", OSW_SC_RUN("%[D3_]", "%[D4_]"))
#define OSW_QP_LAST(ROW, ROWL, ADD, C1, C2, C3, SC1)                               \
    OSW_QP_ROWS3(ROW, ADD, C1, C2, C3, SC1)                                        \
    ROWL("%[hl_]", "%[xb_]", "%[E3_]", OSW_SC_RUN("%[D3_]", "%[hl_]"))
#define OSW_QP_LD(BUF) "ds_read_b128 " BUF ", %[a0_] offset:%[off_]\n\ts_waitcnt lgkmcnt(1)\n\t"
#define OSW_WAIT0 "s_waitcnt lgkmcnt(0)\n\t"

// Four rows of a SEQUENCE-PAIR cell: C1..C3 = the entry pairs ("Ea, Eb") of the block's rows 1..3, N0 = row 0 of the next block (the
// other buffer set); MID = the loads of the block after next into this block's set (all of it has been read by then) and the wait for
// the next block's, in front of the last row -- the query-pair cell's schedule with two ds_read_b128 per block.
#define OSW_SP_ROWS3(ROW, C1, C2, C3, SC1)                                         \
    ROW(OSW_TADD_MAD, "%[xb_]", "%[D1_]", C1, "%[x_]", "%[E0_]", "", "")           \
    ROW(OSW_TADD_MAD, "%[x_]", "%[D2_]", C2, "%[xb_]", "%[E1_]", "", SC1("%[D1_]", "%[D2_]")) \
    ROW(OSW_TADD_MAD, "%[xb_]", "%[D3_]", C3, "%[x_]", "%[E2_]", "", "")
#define OSW_SP_MID(ROW, C1, C2, C3, N0, SC1, MID)                                  \
    OSW_SP_ROWS3(ROW, C1, C2, C3, SC1) MID                                         \
    ROW(OSW_TADD_MAD, "%[x_]", "%[D4_]", N0, "%[xb_]", "%[E3_]", "", OSW_SC_RUN("%[D3_]", "%[D4_]"))
#define OSW_SP_LAST(ROW, ROWL, C1, C2, C3, SC1)                                    \
    OSW_SP_ROWS3(ROW, C1, C2, C3, SC1)                                             \
    ROWL("%[hl_]", "%[xb_]", "%[E3_]", OSW_SC_RUN("%[D3_]", "%[hl_]"))
#define OSW_SP_LD(BA, BB) "ds_read_b128 " BA ", %[a0_] offset:%[off_]\n\tds_read_b128 " BB ", %[a1_] offset:%[off_]\n\ts_waitcnt lgkmcnt(2)\n\t"

// the statement around a block's text: rows RB*4 .. RB*4+3 of the strip
#define OSW_BLOCK_STMT_MID(TXT, SCC, FLC)                                                                                        \
    asm volatile(TXT                                                                                                             \
                 : [x_] "+v"(x), [xb_] "=&v"(xb), [E0_] "+v"(E[RB * 4]), [E1_] "+v"(E[RB * 4 + 1]), [E2_] "+v"(E[RB * 4 + 2]),   \
                   [E3_] "+v"(E[RB * 4 + 3]), [D1_] "+v"(D[RB * 4 + 1]), [D2_] "+v"(D[RB * 4 + 2]), [D3_] "+v"(D[RB * 4 + 3]),   \
                   [D4_] "+v"(D[RB * 4 + 4]), [sc_] SCC(sc)                                                                      \
                 : [a0_] "v"(a0), [a1_] "v"(a1), [off_] "i"((RB + 2) * BLOCK_BYTES), [ge_] "s"(ge), [go_] "s"(go), [fl_] FLC(fl)   \
                 : "memory", OSW_INFLIGHT)
#define OSW_BLOCK_STMT_LAST(TXT, SCC, FLC)                                                                                       \
    asm volatile(TXT                                                                                                             \
                 : [x_] "+v"(x), [xb_] "=&v"(xb), [E0_] "+v"(E[RB * 4]), [E1_] "+v"(E[RB * 4 + 1]), [E2_] "+v"(E[RB * 4 + 2]),   \
                   [E3_] "+v"(E[RB * 4 + 3]), [D1_] "+v"(D[RB * 4 + 1]), [D2_] "+v"(D[RB * 4 + 2]), [D3_] "+v"(D[RB * 4 + 3]),   \
                   [hl_] "=&v"(hl), [sc_] SCC(sc)                                                                                \
                 : [ge_] "s"(ge), [go_] "s"(go), [fl_] FLC(fl)                                                                   \
                 : "memory", OSW_INFLIGHT)

// One block of either cell kind with buffers CUR* (this block's) and NXT* (the next one's); SC1 = how the block's
// first odd row folds into the maximum (OSW_SC_RUN, or OSW_SC_START in block 0 of the column-frame cell)
#define OSW_BLOCK_BODY(ROW, ROWL, ADD, SC1, SCC1, FLC, C1, C2, C3, CALL, N0, S1, S2, S3, SA, SB, SN0)                              \
    do {                                                                                                                         \
        if constexpr (!SEQ) {                                                                                                    \
            if constexpr (LAST) OSW_BLOCK_STMT_LAST(OSW_QP_LAST(ROW, ROWL, ADD, C1, C2, C3, SC1), SCC1, FLC);                    \
            else if constexpr (LD) OSW_BLOCK_STMT_MID(OSW_QP_MID(ROW, ADD, C1, C2, C3, N0, SC1, OSW_QP_LD(CALL)), SCC1, FLC);    \
            else OSW_BLOCK_STMT_MID(OSW_QP_MID(ROW, ADD, C1, C2, C3, N0, SC1, OSW_WAIT0), SCC1, FLC);                            \
        } else {                                                                                                                 \
            if constexpr (LAST) OSW_BLOCK_STMT_LAST(OSW_SP_LAST(ROW, ROWL, S1, S2, S3, SC1), SCC1, FLC);                         \
            else if constexpr (LD) OSW_BLOCK_STMT_MID(OSW_SP_MID(ROW, S1, S2, S3, SN0, SC1, OSW_SP_LD(SA, SB)), SCC1, FLC);      \
            else OSW_BLOCK_STMT_MID(OSW_SP_MID(ROW, S1, S2, S3, SN0, SC1, OSW_WAIT0), SCC1, FLC);                                \
        }                                                                                                                        \
    } while (0)
#define OSW_BLOCK_EVEN(ROW, ROWL, ADD, SC1, SCC1, FLC) \
    OSW_BLOCK_BODY(ROW, ROWL, ADD, SC1, SCC1, FLC, OSW_PA1, OSW_PA2, OSW_PA3, OSW_PA_ALL, OSW_PB0, OSW_SX1, OSW_SX2, OSW_SX3, OSW_SX_A, OSW_SX_B, OSW_SY0)
#define OSW_BLOCK_ODD(ROW, ROWL, ADD, SC1, SCC1, FLC) \
    OSW_BLOCK_BODY(ROW, ROWL, ADD, SC1, SCC1, FLC, OSW_PB1, OSW_PB2, OSW_PB3, OSW_PB_ALL, OSW_PA0, OSW_SY1, OSW_SY2, OSW_SY3, OSW_SY_A, OSW_SY_B, OSW_SX0)

// The head of a column: LDS addresses of the lane's residue(s), the loads of blocks 0 and 1, the first diagonal sum
// (and, for the sequence-pair cell, the score pairs of rows 0 and 1).  VC = the residue register of the step.
#define OSW_QP_HEAD(ADD, VC, LD1)                                                  \
    "v_bfe_u32 %[a0_], " VC ", %[sh_], 8\n\t"                                      \
    "v_lshl_add_u32 %[a0_], %[a0_], 1, %[base_]\n\t"                               \
    "ds_read_b128 " OSW_PA_ALL ", %[a0_]\n\t"                                      \
    LD1                                                                            \
    ADD("%[x_]", "%[tp_]", OSW_PA0)
#define OSW_QP_HEAD_LD1 "ds_read_b128 " OSW_PB_ALL ", %[a0_] offset:%[off_]\n\ts_waitcnt lgkmcnt(1)\n\t"
#define OSW_SP_HEAD(VC, LD1)                                                       \
    "v_and_b32 %[a0_], 0xff, " VC "\n\t"                                           \
    "v_lshrrev_b32 %[a1_], 8, " VC "\n\t"                                          \
    "v_lshl_add_u32 %[a0_], %[a0_], 1, %[base_]\n\t"                               \
    "v_lshl_add_u32 %[a1_], %[a1_], 1, %[base_]\n\t"                               \
    "ds_read_b128 " OSW_SX_A ", %[a0_]\n\t"                                        \
    "ds_read_b128 " OSW_SX_B ", %[a1_]\n\t"                                        \
    LD1                                                                            \
    OSW_TADD_MAD("%[x_]", "%[tp_]", OSW_SX0)
#define OSW_SP_HEAD_LD1 "ds_read_b128 " OSW_SY_A ", %[a0_] offset:%[off_]\n\tds_read_b128 " OSW_SY_B ", %[a1_] offset:%[off_]\n\ts_waitcnt lgkmcnt(2)\n\t"
#define OSW_HEAD_STMT(TXT)                                                                                                       \
    asm volatile(TXT                                                                                                             \
                 : [a0_] "=&v"(a0), [a1_] "=&v"(a1), [x_] "=&v"(x)                                                               \
                 : [base_] "v"(base), [tp_] "v"(top_prev), [sh_] "s"(sh), [off_] "i"(BLOCK_BYTES)                                \
                 : "memory", OSW_INFLIGHT)
#define OSW_HEAD_BODY(ADD)                                                                                                       \
    do {                                                                                                                         \
        if constexpr (!SEQ) {                                                                                                    \
            if constexpr (P == 0 && NB > 1) OSW_HEAD_STMT(OSW_QP_HEAD(ADD, OSW_VC0, OSW_QP_HEAD_LD1));                           \
            else if constexpr (P == 0) OSW_HEAD_STMT(OSW_QP_HEAD(ADD, OSW_VC0, OSW_WAIT0));                                      \
            else if constexpr (NB > 1) OSW_HEAD_STMT(OSW_QP_HEAD(ADD, OSW_VC1, OSW_QP_HEAD_LD1));                                \
            else OSW_HEAD_STMT(OSW_QP_HEAD(ADD, OSW_VC1, OSW_WAIT0));                                                            \
        } else {                                                                                                                 \
            if constexpr (P == 0 && NB > 1) OSW_HEAD_STMT(OSW_SP_HEAD(OSW_VC0, OSW_SP_HEAD_LD1));                           \
            else if constexpr (P == 0) OSW_HEAD_STMT(OSW_SP_HEAD(OSW_VC0, OSW_WAIT0));                                      \
            else if constexpr (NB > 1) OSW_HEAD_STMT(OSW_SP_HEAD(OSW_VC1, OSW_SP_HEAD_LD1));                                \
            else OSW_HEAD_STMT(OSW_SP_HEAD(OSW_VC1, OSW_WAIT0));                                                            \
        }                                                                                                                        \
    } while (0)

// ---------------------------------------------------------------------------
// The int32 cell, hand-scheduled (round 5, second session): the column-frame row of the int16 cell with v_max3_i32 where that has
// v_pk_maximum3_f16 -- ONE sequence per lane (the `half` of its pair), exact for any score.  The diagonal add takes the row's
// int16 score S + ge straight out of a half of the profile entry's register: v_add_u32_sdwa with a sign-extended WORD_0 / WORD_1
// source (no unpacking: the compiler's version spent 1 of its 9.3 instructions per row on v_bfe_i32 / v_ashrrev_i32 and ~2 on
// copies around the D[r + 1] shuffle).  6.5 instructions per row of 64 cells: add, H = max3, u = H - go, E = max3, F = max3,
// F - ge, 1/2 column maximum.  Profile entry: 8 B per code = 4 rows x int16 (`prof_alt`: S + ge), ds_read_b64, two blocks ahead.
// ---------------------------------------------------------------------------
#define OSW_WA0 "v150"
#define OSW_WA1 "v151"
#define OSW_WA_ALL "v[150:151]"
#define OSW_WB0 "v154"
#define OSW_WB1 "v155"
#define OSW_WB_ALL "v[154:155]"
#define OSW_W_SEL(REG, W) "sext(" REG ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" W
#define OSW_TADD_SDWA(XN, DN, SN) "v_add_u32_sdwa " XN ", " DN ", " SN "\n\t"
#define OSW_WMAX(SC, SCI, DP, DN) "v_max3_i32 " SC ", " SCI ", " DP ", " DN "\n\t"
#define OSW_WC_RUN(DP, DN) OSW_WMAX("%[sc_]", "%[sc_]", DP, DN)
#define OSW_WC_START(DP, DN) OSW_WMAX("%[sc_]", DP, DN, DN)
#define OSW_W_ROW(XN, DN, SN, X, E, CMAX)                                          \
    OSW_TADD_SDWA(XN, DN, SN)                                                      \
    "v_max3_i32 " DN ", " X ", " E ", " OSW_VF "\n\t"                              \
    "v_subrev_u32 " OSW_VT ", %[go_], " DN "\n\t"                                  \
    CMAX                                                                           \
    "v_max3_i32 " E ", " E ", " OSW_VT ", %[fl_]\n\t"                              \
    "v_max3_i32 " OSW_VF ", " OSW_VF ", " OSW_VT ", %[fl_]\n\t"                    \
    "v_subrev_u32 " OSW_VF ", %[ge_], " OSW_VF "\n\t"
#define OSW_W_ROW_LAST(HL, X, E, CMAX)                                             \
    "v_max3_i32 " HL ", " X ", " E ", " OSW_VF "\n\t"                              \
    "v_subrev_u32 " OSW_VT ", %[go_], " HL "\n\t"                                  \
    CMAX                                                                           \
    "v_max3_i32 " E ", " E ", " OSW_VT ", %[fl_]\n\t"                              \
    "v_max3_i32 " OSW_VF ", " OSW_VF ", " OSW_VT ", %[fl_]\n\t"                    \
    "v_subrev_u32 " OSW_VF ", %[ge_], " OSW_VF
// four rows: C1..C3 = the score operands of the block's rows 1..3, N0 = row 0 of the next block (the other buffer), MID = the load of
// the block after next into this block's buffer and the wait for the next block's (the query-pair cell's schedule)
#define OSW_W_ROWS3(C1, C2, C3, SC1)                                               \
    OSW_W_ROW("%[xb_]", "%[D1_]", C1, "%[x_]", "%[E0_]", "")                       \
    OSW_W_ROW("%[x_]", "%[D2_]", C2, "%[xb_]", "%[E1_]", SC1("%[D1_]", "%[D2_]"))  \
    OSW_W_ROW("%[xb_]", "%[D3_]", C3, "%[x_]", "%[E2_]", "")
#define OSW_W_MID(C1, C2, C3, N0, SC1, MID)                                        \
    OSW_W_ROWS3(C1, C2, C3, SC1) MID                                               \
    OSW_W_ROW("%[x_]", "%[D4_]", N0, "%[xb_]", "%[E3_]", OSW_WC_RUN("%[D3_]", "%[D4_]"))
#define OSW_W_LAST(C1, C2, C3, SC1)                                                \
    OSW_W_ROWS3(C1, C2, C3, SC1)                                                   \
    OSW_W_ROW_LAST("%[hl_]", "%[xb_]", "%[E3_]", OSW_WC_RUN("%[D3_]", "%[hl_]"))
#define OSW_W_LD(BUF) "ds_read_b64 " BUF ", %[a0_] offset:%[off_]\n\ts_waitcnt lgkmcnt(1)\n\t"
#define OSW_W_BLOCK(SC1, SCC1, P0, P1, PALL, N0R)                                                                                      \
    do {                                                                                                                             \
        if constexpr (LAST) OSW_BLOCK_STMT_LAST(OSW_W_LAST(OSW_W_SEL(P0, "1"), OSW_W_SEL(P1, "0"), OSW_W_SEL(P1, "1"), SC1), SCC1, "v"); \
        else if constexpr (LD) OSW_BLOCK_STMT_MID(OSW_W_MID(OSW_W_SEL(P0, "1"), OSW_W_SEL(P1, "0"), OSW_W_SEL(P1, "1"), OSW_W_SEL(N0R, "0"), SC1, OSW_W_LD(PALL)), SCC1, "v"); \
        else OSW_BLOCK_STMT_MID(OSW_W_MID(OSW_W_SEL(P0, "1"), OSW_W_SEL(P1, "0"), OSW_W_SEL(P1, "1"), OSW_W_SEL(N0R, "0"), SC1, OSW_WAIT0), SCC1, "v"); \
    } while (0)
// head of a column: LDS address of the lane's residue (`tiled` holds 8 * code = the byte offset of the code's 8-byte entry), the
// loads of blocks 0 and 1, the first diagonal sum
#define OSW_W_HEAD(VC, LD1)                                                        \
    "v_bfe_u32 %[a0_], " VC ", %[sh_], 8\n\t"                                      \
    "v_add_u32 %[a0_], %[a0_], %[base_]\n\t"                                       \
    "ds_read_b64 " OSW_WA_ALL ", %[a0_]\n\t"                                       \
    LD1                                                                            \
    OSW_TADD_SDWA("%[x_]", "%[tp_]", OSW_W_SEL(OSW_WA0, "0"))
#define OSW_W_HEAD_LD1 "ds_read_b64 " OSW_WB_ALL ", %[a0_] offset:%[off_]\n\ts_waitcnt lgkmcnt(1)\n\t"
#define OSW_W_HEAD_STMT(TXT)                                                                                                     \
    asm volatile(TXT                                                                                                             \
                 : [a0_] "=&v"(a0), [x_] "=&v"(x)                                                                                \
                 : [base_] "v"(base), [tp_] "v"(top_prev), [sh_] "s"(sh), [off_] "i"(BLOCK_BYTES)                                \
                 : "memory", OSW_INFLIGHT)

struct ArithI32F {
    static constexpr int BLOCK_BYTES = 256; // 32 codes x 8 B (4 rows x int16)
    template <int P, int NB>
    static __device__ __forceinline__ void head(uint32_t base, uint32_t sh, int top_prev, uint32_t &a0, int &x)
    {
        if constexpr (P == 0 && NB > 1) OSW_W_HEAD_STMT(OSW_W_HEAD(OSW_VC0, OSW_W_HEAD_LD1));
        else if constexpr (P == 0) OSW_W_HEAD_STMT(OSW_W_HEAD(OSW_VC0, OSW_WAIT0));
        else if constexpr (NB > 1) OSW_W_HEAD_STMT(OSW_W_HEAD(OSW_VC1, OSW_W_HEAD_LD1));
        else OSW_W_HEAD_STMT(OSW_W_HEAD(OSW_VC1, OSW_WAIT0));
    }
    template <int RB, int NB>
    static __device__ __forceinline__ void block(uint32_t a0, int (&D)[NB * 4], int (&E)[NB * 4], int &x, int &hl, int &sc, uint32_t ge, uint32_t go, int fl)
    {
        constexpr bool LAST = RB == NB - 1, LD = RB + 2 < NB;
        const uint32_t a1 = a0; // (the block statements name a second address operand: the sequence-pair cell's)
        (void)a1;
        int xb;
        if constexpr (RB == 0) OSW_W_BLOCK(OSW_WC_START, "=&v", OSW_WA0, OSW_WA1, OSW_WA_ALL, OSW_WB0);
        else if constexpr ((RB & 1) == 0) OSW_W_BLOCK(OSW_WC_RUN, "+v", OSW_WA0, OSW_WA1, OSW_WA_ALL, OSW_WB0);
        else OSW_W_BLOCK(OSW_WC_RUN, "+v", OSW_WB0, OSW_WB1, OSW_WB_ALL, OSW_WA0);
    }
};

// Cell arithmetic policies: the blocks of a column and what a finished score means.
//   head<P, NB, SEQ>: a0 / a1 out (LDS addresses), x out (diagonal sum of row 0); sh = bit offset of the lane's residue
//   block<RB, NB, SEQ>: rows RB*4..RB*4+3 of NB*4; x in/out, hl out (last block), sc = running / column maximum
struct ArithI16B {
    static constexpr int kCeiling = 31600 - 1024; // true score from which a sequence is re-run in int32
    static constexpr bool kShifted = false;
    static constexpr uint32_t kFloor = OSW_I16B_BIAS;
    template <int P, int NB, bool SEQ>
    static __device__ __forceinline__ void head(uint32_t base, uint32_t sh, v2s top_prev, uint32_t &a0, uint32_t &a1, v2s &x)
    {
        constexpr int BLOCK_BYTES = SEQ ? 16 * OSW_SEQ_CODES : 512; // 16 B per code: 4 rows x a {S, 1} entry (24 codes), or x the int16 scores of two queries (32)
        OSW_HEAD_BODY(OSW_TADD_PK);
    }
    template <int RB, int NB, bool SEQ>
    static __device__ __forceinline__ void block(uint32_t a0, uint32_t a1, v2s (&D)[NB * 4], v2s (&E)[NB * 4], v2s &x, v2s &hl, v2s &sc, uint32_t ge,
                                                 uint32_t go, v2s /*aux*/)
    {
        constexpr int BLOCK_BYTES = SEQ ? 16 * OSW_SEQ_CODES : 512; // 16 B per code: 4 rows x a {S, 1} entry (24 codes), or x the int16 scores of two queries (32)
        constexpr bool LAST = RB == NB - 1, LD = RB + 2 < NB;
        const uint32_t fl = OSW_I16B_BIAS;
        v2s xb;
        if constexpr ((RB & 1) == 0) OSW_BLOCK_EVEN(OSW_B_ROW, OSW_B_ROW_LAST, OSW_TADD_PK, OSW_SC_RUN, "+v", "s");
        else OSW_BLOCK_ODD(OSW_B_ROW, OSW_B_ROW_LAST, OSW_TADD_PK, OSW_SC_RUN, "+v", "s");
    }
    static __device__ __forceinline__ int to_int(short bits) { return (int)bits - 1024; }
    // at or past the threshold -- or any pattern the maximum may have turned into a NaN of either sign
    static __device__ __forceinline__ bool over(short bits) { return (uint16_t)bits >= 31600u; }
};

// `go` carries the gap OPEN penalty for this cell, `fl` the floor of the next column's frame, and the rows
// accumulate the COLUMN maximum (in the column's frame) into `sc` (written, not read, by block 0); sw_round_fast turns
// it into a true score.  INTSUM: the profile stores its score pairs as 32-bit integer sums: the diagonal add is a
// v_add_u32 too; otherwise only the two subtracts are.
template <bool INTSUM>
struct ArithI16S {
    static constexpr int kCeiling = 22256; // true score from which a sequence is re-run in int32 (31600 - 1024 - 8192 - 128)
    static constexpr uint32_t kFloor = OSW_I16B_BIAS;
    static constexpr bool kShifted = true;
    template <int P, int NB, bool SEQ>
    static __device__ __forceinline__ void head(uint32_t base, uint32_t sh, v2s top_prev, uint32_t &a0, uint32_t &a1, v2s &x)
    {
        constexpr int BLOCK_BYTES = SEQ ? 16 * OSW_SEQ_CODES : 512; // 16 B per code: 4 rows x a {S, 1} entry (24 codes), or x the int16 scores of two queries (32)
        if constexpr (INTSUM) OSW_HEAD_BODY(OSW_TADD_32);
        else OSW_HEAD_BODY(OSW_TADD_PK);
    }
    template <int RB, int NB, bool SEQ>
    static __device__ __forceinline__ void block(uint32_t a0, uint32_t a1, v2s (&D)[NB * 4], v2s (&E)[NB * 4], v2s &x, v2s &hl, v2s &sc, uint32_t ge,
                                                 uint32_t go, v2s fl)
    {
        constexpr int BLOCK_BYTES = SEQ ? 16 * OSW_SEQ_CODES : 512; // 16 B per code: 4 rows x a {S, 1} entry (24 codes), or x the int16 scores of two queries (32)
        constexpr bool LAST = RB == NB - 1, LD = RB + 2 < NB;
        v2s xb;
        if constexpr (INTSUM) {
            if constexpr (RB == 0) OSW_BLOCK_EVEN(OSW_S_ROWP, OSW_S_ROW_LAST, OSW_TADD_32, OSW_SC_START, "=&v", "v");
            else if constexpr ((RB & 1) == 0) OSW_BLOCK_EVEN(OSW_S_ROWP, OSW_S_ROW_LAST, OSW_TADD_32, OSW_SC_RUN, "+v", "v");
            else OSW_BLOCK_ODD(OSW_S_ROWP, OSW_S_ROW_LAST, OSW_TADD_32, OSW_SC_RUN, "+v", "v");
        } else {
            if constexpr (RB == 0) OSW_BLOCK_EVEN(OSW_S_ROW, OSW_S_ROW_LAST, OSW_TADD_PK, OSW_SC_START, "=&v", "v");
            else if constexpr ((RB & 1) == 0) OSW_BLOCK_EVEN(OSW_S_ROW, OSW_S_ROW_LAST, OSW_TADD_PK, OSW_SC_RUN, "+v", "v");
            else OSW_BLOCK_ODD(OSW_S_ROW, OSW_S_ROW_LAST, OSW_TADD_PK, OSW_SC_RUN, "+v", "v");
        }
    }
    static __device__ __forceinline__ int to_int(short bits) { return (int)(uint16_t)bits; } // the running score is a true one
    static __device__ __forceinline__ bool over(short bits) { return (uint16_t)bits >= 22256u; }
};

template <class A>
struct CellSeqPair {
    typedef A Arith;
    typedef v2s T;
    typedef uint32_t GapT; // (value, value) packed, wave-uniform
    static constexpr bool kFast = true;
    static constexpr uint32_t kFloorBits = A::kFloor;
    static constexpr bool kShifted = A::kShifted;
    static constexpr int kRows = OSW_RMAX16;
    static constexpr int kLdsRows = OSW_LDS_ROWS16_SEQ;
    static constexpr int kCodes = OSW_SEQ_CODES; // entries per row-block: the 24 residue codes the re-tiled database can hold
    static constexpr int kRowBytes = 4 * OSW_SEQ_CODES; // profile bytes per query row: 24 codes x a 32-bit entry {low half: S, high half: 1}
    typedef uint4 Entry;                  // one code, 4 rows (osw_build_profile's prof_seq)
    static __device__ __forceinline__ T zero() { return as_v2s(A::kFloor); } // the value that stands for 0
    static __device__ __forceinline__ T from_bits(uint32_t x) { return as_v2s(x); }
    static __device__ __forceinline__ uint32_t to_bits(T x) { return as_u32(x); }
    static __device__ __forceinline__ T vmax(T a, T b) { return __builtin_elementwise_max(a, b); }
    // column frames: the floor of the next column's frame; the column's maximum (in its frame) folded into the true running score
    static constexpr uint32_t kTopTable = 128; // first entry of the cell's floor table in top_pages
    static __device__ __forceinline__ const uint2 *top_pages(const OswSearchArgs &p) { return p.top_pages; }
    static __device__ __forceinline__ T frame_next(T fl, GapT ge) { return fl + as_v2s(ge); }
    static __device__ __forceinline__ void fold(T &score, T cm, T fl)
    {
        const v2u tru = __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, cm), __builtin_bit_cast(v2u, fl));
        score = __builtin_bit_cast(v2s, __builtin_elementwise_max(__builtin_bit_cast(v2u, score), tru));
    }

    // One database column against the R rows of the strip, inputs in register set P:
    //   residues {8*code of the lane's first sequence, 8*code of the second} in bytes 0 / 1 of
    //   the set's C register (8*code = byte offset of the code's profile entry), F(i0, j) in its
    //   F register (out: F(i0+R, j)).  base = LDS byte address of the lane's profile slice;
    //   top_prev = H(i0-1, j-1); hl = H(i0+R-1, j).
    // Four rows of profile entries for both sequences of the lane are two ds_read_b128 (one per sequence; 16 B per
    // residue code: address = base + 2 * (8*code)); the row's v_pk_mad_i16 pairs the two scores up and adds them to the
    // diagonal (OSW_TADD_MAD: 6.5 instructions per row; until the third session of round 4 a v_perm_b32 per row paired
    // 16-bit scores from 8-byte entries and a packed add followed: 7.5); the loads run two blocks ahead, inside the
    // blocks' statements (ArithI16*::block).
    template <int R, int RB>
    struct Blocks {
        static __device__ __forceinline__ void run(uint32_t a0, uint32_t a1, T (&D)[R], T (&E)[R], T &x, T &hl, GapT goe, GapT ge, T &score, T aux)
        {
            A::template block<RB, R / 4, true>(a0, a1, D, E, x, hl, score, ge, goe, aux);
            if constexpr (RB + 1 < R / 4) Blocks<R, RB + 1>::run(a0, a1, D, E, x, hl, goe, ge, score, aux);
        }
    };
    template <int R, int P>
    static __device__ __forceinline__ void column(uint32_t base, int /*half*/, T (&D)[R], T (&E)[R], T top_prev, T &hl, GapT goe, GapT ge,
                                                  T &score, T aux)
    {
        uint32_t a0, a1;
        T x;
        A::template head<P, R / 4, true>(base, 0u, top_prev, a0, a1, x);
        Blocks<R, 0>::run(a0, a1, D, E, x, hl, goe, ge, score, aux);
    }
};

// ---------------------------------------------------------------------------
// Query-pair cell (query batching): the two halves of a register
// hold two QUERIES of similar length against ONE database sequence per lane
// (the `half` of the lane's sequence pair; the item runs both halves one after
// the other).  The pair profile stores (S_A, S_B) already packed, 4 rows x
// 2 queries = 16 B per residue code, so one ds_read_b128 feeds four rows and
// the v_perm_b32 disappears: 9 VALU instructions per wave per 128 cells.
// ---------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <class A>
struct CellQueryPair {
    typedef A Arith;
    typedef v2s T;
    typedef uint32_t GapT;
    static constexpr bool kFast = true;
    static constexpr uint32_t kFloorBits = A::kFloor;
    static constexpr bool kShifted = A::kShifted;
    static constexpr int kRows = OSW_RMAX16;
    static constexpr int kLdsRows = OSW_LDS_ROWS16 / 2;
    static constexpr int kCodes = 32;
    static constexpr int kRowBytes = 128; // 32 codes x 2 queries x int16
    typedef uint4 Entry;                  // one code, 4 rows x 2 queries
    static __device__ __forceinline__ T zero() { return as_v2s(A::kFloor); } // the value that stands for 0
    static __device__ __forceinline__ T from_bits(uint32_t x) { return as_v2s(x); }
    static __device__ __forceinline__ uint32_t to_bits(T x) { return as_u32(x); }
    static __device__ __forceinline__ T vmax(T a, T b) { return __builtin_elementwise_max(a, b); }
    // column frames: the floor of the next column's frame; the column's maximum (in its frame) folded into the true running score
    static constexpr uint32_t kTopTable = 128; // first entry of the cell's floor table in top_pages
    static __device__ __forceinline__ const uint2 *top_pages(const OswSearchArgs &p) { return p.top_pages; }
    static __device__ __forceinline__ T frame_next(T fl, GapT ge) { return fl + as_v2s(ge); }
    static __device__ __forceinline__ void fold(T &score, T cm, T fl)
    {
        const v2u tru = __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, cm), __builtin_bit_cast(v2u, fl));
        score = __builtin_bit_cast(v2s, __builtin_elementwise_max(__builtin_bit_cast(v2u, score), tru));
    }

    // profile entry of the lane's residue: 16 B per code = 2 x (8*code); the loads run two blocks ahead, inside
    // the blocks' statements (ArithI16*::block)
    template <int R, int RB>
    struct Blocks {
        static __device__ __forceinline__ void run(uint32_t a, T (&D)[R], T (&E)[R], T &x, T &hl, GapT goe, GapT ge, T &score, T aux)
        {
            A::template block<RB, R / 4, false>(a, a, D, E, x, hl, score, ge, goe, aux);
            if constexpr (RB + 1 < R / 4) Blocks<R, RB + 1>::run(a, D, E, x, hl, goe, ge, score, aux);
        }
    };
    template <int R, int P>
    static __device__ __forceinline__ void column(uint32_t base, int half, T (&D)[R], T (&E)[R], T top_prev, T &hl, GapT goe, GapT ge,
                                                  T &score, T aux)
    {
        uint32_t a, unused;
        T x;
        A::template head<P, R / 4, false>(base, (uint32_t)half * 8u, top_prev, a, unused, x);
        Blocks<R, 0>::run(a, D, E, x, hl, goe, ge, score, aux);
    }
};

typedef CellSeqPair<ArithI16B> CellPK16B;
typedef CellQueryPair<ArithI16B> CellPK16BQ;
typedef CellSeqPair<ArithI16S<false>> CellPK16S;
typedef CellQueryPair<ArithI16S<true>> CellPK16SQ;

// The hand-scheduled int32 cell (ArithI32F) for whole searches with cell_bits = 32: 48-row strips and the packed-int16 kernels'
// column loop (sw_round_fast: fixed registers, loads two columns ahead, three waves per SIMD).  go = gap OPEN, ge = gap extend,
// plain 32-bit values (wave-uniform); floors / frames as in ArithI16S without the fp16 bias; a first round reads the row above
// it from the int32 floor table (OswSearchArgs::floor_i32, entry k = k * ge; OSW_I32F_TABLE entries: every block the library
// accepts -- columns are 16-bit -- fits, so this cell needs no fallback).
struct CellI32F {
    typedef int T;
    typedef uint32_t GapT;
    static constexpr bool kFast = true;
    static constexpr uint32_t kFloorBits = 0;
    static constexpr bool kShifted = true;
    static constexpr int kRows = OSW_RMAX32F;
    static constexpr int kLdsRows = OSW_LDS_ROWS32F;
    static constexpr int kCodes = 32;
    static constexpr int kRowBytes = 64;
    typedef uint2 Entry;
    static constexpr uint32_t kTopTable = 0; // a table of its own (built on the device by osw_floor_i32)
    static __device__ __forceinline__ const uint2 *top_pages(const OswSearchArgs &p) { return p.floor_i32; }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ T from_bits(uint32_t x) { return (int)x; }
    static __device__ __forceinline__ uint32_t to_bits(T x) { return (uint32_t)x; }
    static __device__ __forceinline__ T vmax(T a, T b) { return a > b ? a : b; }
    static __device__ __forceinline__ T frame_next(T fl, GapT ge) { return fl + (int)ge; }
    static __device__ __forceinline__ void fold(T &score, T cm, T fl) { score = cm - fl > score ? cm - fl : score; }
    template <int R, int RB>
    struct Blocks {
        static __device__ __forceinline__ void run(uint32_t a, T (&D)[R], T (&E)[R], T &x, T &hl, GapT go, GapT ge, T &cm, T fl1)
        {
            ArithI32F::template block<RB, R / 4>(a, D, E, x, hl, cm, ge, go, fl1);
            if constexpr (RB + 1 < R / 4) Blocks<R, RB + 1>::run(a, D, E, x, hl, go, ge, cm, fl1);
        }
    };
    template <int R, int P>
    static __device__ __forceinline__ void column(uint32_t base, int half, T (&D)[R], T (&E)[R], T top_prev, T &hl, GapT go, GapT ge, T &cm, T fl1)
    {
        uint32_t a;
        T x;
        ArithI32F::template head<P, R / 4>(base, (uint32_t)half * 8u, top_prev, a, x);
        Blocks<R, 0>::run(a, D, E, x, hl, go, ge, cm, fl1);
    }
};

// The 8-bit cell of cell_bits = 8 (SWAR: four 7-bit cells per register, offset domain): q8_cell.h.
#include <type_traits>
#include "q8_cell.h"

// ---------------------------------------------------------------------------
// Logical lanes (round 5).  The LDS serves a ds_read_b128 in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and lanes conflict only within a group.  The packed-int16 kernels
// therefore number their lanes so that every such group is 16 CONSECUTIVE logical lanes: a wave geometry of G lane groups of
// 64 / G logical lanes then puts whole lane groups (strips of the query: one profile table each) into a hardware group --
// one table at G = 4, two at G = 8 -- instead of pieces of four (G = 8: lanes 0-3 | 12-15 | 20-23 | 24-27 belong to four
// strips).  Fewer tables in a group = fewer distinct entries on its 16 slots: simulated 10.7 -> 8.4 LDS cycles per
// ds_read_b128 at G = 8, 8.9 -> 5.6 at G = 4 (tools/lds_sim.py, with the relabelled codes).  Everything that names a lane --
// the lane groups' masks, the ds_bpermute hand-off, the residues' and the spill's addresses, the score's place -- goes by
// the logical number; physical numbers appear only as ds_bpermute sources.
//   physical p: block b = bits 4..2 of p; hardware group of the half = parity of b, place in it = b >> 1
// ---------------------------------------------------------------------------
static __device__ __forceinline__ int osw_logical_lane(int p)
{
    const int b = (p >> 2) & 7, par = (b ^ (b >> 1) ^ (b >> 2)) & 1;
    return (p & 32) | (par << 4) | ((b >> 1) << 2) | (p & 3);
}
static __device__ __forceinline__ int osw_physical_lane(int l)
{
    const int r = (l >> 2) & 3, h = (l >> 4) & 1, b = (r << 1) | (h ^ ((r ^ (r >> 1)) & 1));
    return (l & 32) | (b << 2) | (l & 3);
}

// the same cell in the re-run pipeline (osw_sw_i32r): geometry 64, 4-row strips -- one 4-row block per column, so a lane group's table is
// a single row-block and may be COMPACT: the 24 residue codes the re-tiled database can hold (OSW_SEQ_CODES), 8 B each = 200 B per lane
// group with its skew entry, 12.5 KB per wave: twelve waves of a workgroup fit the CU's LDS (with 32 codes: nine)
struct CellI32FP : CellI32F {
    static constexpr int kRows = 4;
    static constexpr int kLdsRows = OSW_LDS_ROWS32;
    static constexpr int kCodes = OSW_SEQ_CODES;
    static constexpr int kRowBytes = 2 * OSW_SEQ_CODES;
};

static __device__ __forceinline__ uint64_t osw_uniform64(uint64_t x)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------------------
// One round: G lane groups, group g runs R rows of strip (round*G + g), one
// column behind group g-1.
//   tcol : the block's residues at this sub-block, column 0 (wave-uniform)
//   u    : lane's index inside its group (its sequence pair of the sub-block)
//   base : LDS byte address of this lane's profile slice (wave region + g*R rows)
//   bnd  : wave's scratch region (wave-uniform), layout [column][gl lanes of a
//          group] behind the zero and trash pages (OSW_SCRATCH_*)
//
// Packed-int16 version.  All vector-memory instructions of the loop are inline
// asm and their waits are counted by hand.  A step issues, in this order:
// 2 stores (bottom row {H, F} of the last group; into the trash page while
// there is nothing to spill, so that the count never changes) and 3 loads for
// the next-but-one column (LH, LF: lanes of group 0; C: all lanes).  A step
// therefore starts with vmcnt <= 5: everything older than the 5 operations of
// the previous step has landed.  No memory instruction is ever issued with an
// empty EXEC mask.  A first round reads its (all-zero) row above from the zero
// page with a zero stride; the columns read past the block's end (prefetch,
// drain steps of the lane groups) are zero in the scratch, and `tiled` holds
// dummy residues before and after every block (warm-up / drain of the groups).
// Per column on top of the cells: 2 VALU (LDS addresses) + 1 (first diagonal
// add) + 1 (top_prev, merged with group 0's loaded H by a v_cndmask) + 1 (F
// merge); the rest is SALU, LDS and VMEM.
// ---------------------------------------------------------------------------
#define OSW_STEP_BEGIN_ASM(LFP)                                                                              \
    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\t"                                                         \
                 "v_cndmask_b32 " OSW_VF ", " OSW_VF ", " LFP ", %[mg0]"                                     \
                 :                                                                                           \
                 : [mg0] "s"(m_g0)                                                                           \
                 : "memory", OSW_INFLIGHT)

// end of a step: top_prev for the next step (H of the row above of THIS step), spill of the last group's bottom
// row, hand-off to the group above (H and F are overwritten by it: everything that reads them comes first), then
// the loads for the next-but-one column
#define OSW_STEP_END_ASM(CP, LHP, LFP)                                                                       \
    asm volatile("v_cndmask_b32 %[tp], " OSW_VH ", " LHP ", %[mg0]\n\t"                                       \
                 "s_mov_b64 %[sv], exec\n\t"                                                                 \
                 "s_mov_b64 exec, %[mst]\n\t"                                                                \
                 "global_store_dword %[voff], %[ho], %[sptr]\n\t"                                            \
                 "global_store_dword %[voff], " OSW_VF ", %[sptr] offset:4\n\t"                              \
                 "s_not_b64 exec, %[mg0]\n\t"                                                                \
                 "s_cbranch_execz 1f\n\t"                                                                    \
                 "s_mov_b64 exec, %[sv]\n\t"                                                                 \
                 "ds_bpermute_b32 " OSW_VH ", %[src], %[ho]\n\t"                                             \
                 "ds_bpermute_b32 " OSW_VF ", %[src], " OSW_VF "\n"                                           \
                 "1:\n\t"                                                                                    \
                 "s_mov_b64 exec, %[mg0]\n\t"                                                                \
                 "global_load_dword " LHP ", %[voffl], %[lptr]\n\t"                                          \
                 "global_load_dword " LFP ", %[voffl], %[lptr] offset:4\n\t"                                 \
                 "s_mov_b64 exec, %[sv]\n\t"                                                                 \
                 "global_load_ushort " CP ", %[voffc], %[tptr]"                                              \
                 : [tp] "=&v"(tp), [sv] "=&s"(sv)                                                            \
                 : [src] "v"(src), [ho] "v"(ho), [voff] "v"(voff), [voffl] "v"(voffl), [voffc] "v"(voffc), [mst] "s"(m_st), \
                   [mg0] "s"(m_g0), [sptr] "s"(sptr), [lptr] "s"(lptr), [tptr] "s"(tptr)                      \
                 : "memory", "scc", OSW_INFLIGHT)

#define OSW_PIPE_BATCH 32u
// PIPE (the re-run pipeline, run_item_i32f_pipe): the round's row above comes from ANOTHER wave's region (src_region), column by
// column while that wave is still producing it: before every batch of OSW_PIPE_BATCH steps the wave waits until the columns the
// batch loads (two ahead of the steps) are published in *prog_src under the tag rho, and behind every batch it publishes how many
// columns of its own bottom row are stored (s_waitcnt vmcnt(0) first: the stores have been performed; the waves of a workgroup share
// the CU's L1, and a workgroup-scope fence orders the flag behind them).
template <class C, int R, bool PIPE = false>
static __device__ __forceinline__ void sw_round_fast(const uint16_t *tcol, uint32_t u, uint32_t ncols, uint32_t base, uint2 *bnd,
                                                     const uint2 *top_pages, bool first, bool last, uint32_t G, uint32_t gl, int lane, int half,
                                                     typename C::GapT goe, typename C::GapT ge, typename C::T &score,
                                                     const uint2 *src_region = nullptr, lds_flagp prog_src = nullptr,
                                                     lds_flagp prog_mine = nullptr, uint32_t rho = 0, uint2 *hand_store = nullptr)
{
    typedef typename C::T T;
    // (`lane` is the LOGICAL lane, see osw_logical_lane: the masks are over physical lanes)
    const uint64_t m_g0 = __builtin_amdgcn_ballot_w64((uint32_t)lane < gl);        // lanes of group 0
    const uint64_t m_st = __builtin_amdgcn_ballot_w64((uint32_t)lane >= 64u - gl); // lanes of the last group
    const uint32_t src = (uint32_t)osw_physical_lane((lane - (int)gl) & 63) << 2;  // ds_bpermute source: the lane one group below
    const uint32_t g = (uint32_t)lane / gl;
    // "zero" for this lane's first step.  Column-frame cell: the lane starts at column -g, whose frame offset is
    // (G - g) * ge (see ArithI16S); the state that belongs to the column before it sits one frame back.
    T fl = C::from_bits(C::kFloorBits), fl_prev = fl;
    if constexpr (C::kShifted) {
        fl = C::from_bits(C::kFloorBits + (G - g) * ge);        // (packed cells: ge holds the penalty in both halves: no carry between them below 2^15)
        fl_prev = C::from_bits(C::kFloorBits + (G - g - 1u) * ge);
    }
    // (int32 cell: the floors are the same in every round of an item, and the compiler would set the 2 R state registers of every strip
    // height up ONCE, outside the rounds, and keep them: ~40 registers spilled to scratch.  Opaque values stay where they are used.)
    if constexpr (std::is_same<T, int>::value) asm volatile("" : "+v"(fl), "+v"(fl_prev));
    T D[R], E[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { D[r] = fl_prev; E[r] = fl; }
    T top_prev = fl_prev; // H(i0-1, j-1)
    // group g reads its residues g columns behind group 0: the pointer runs G-1 columns behind,
    // the lanes' offsets make up for it (columns before the block are the dummy pad in front of it)
    // (PIPE: `tcol` is a compact copy of the ONE sequence the workgroup works on, 2 bytes per column -- a lane's residue load of a step
    // is then one cache line for the whole wave, not a line per lane: at geometry 64 every lane reads a different column)
    constexpr uint32_t CS = PIPE ? 2u : 128u; // bytes from a column's residues to the next column's
    const uint32_t voff = u * 8u, voffc = (PIPE ? 0u : u * 2u) + (G - 1u - g) * CS;
    // a last round stores into the trash page; so do the G-1 warm-up steps in which the last group is
    // still before column 0 (the store pointer starts G-1 columns before the data: inside the trash page)
    const uint64_t data = (uint64_t)(bnd + OSW_SCRATCH_DATA);
    // (hand_store: a LAST round whose bottom row is wanted all the same -- a pair's longer query goes on as a tail item -- stores it
    // there, column 0 at hand_store, instead of into the trash page)
    const uint64_t hs = osw_uniform64((uint64_t)hand_store);
    const bool keep = last && hs != 0;
    const uint32_t sstep = last && !keep ? 0u : gl * 8u;
    // A first round reads the row above it -- "zero" in the cell's representation -- from constant memory: one
    // entry for every column (zero stride), or, for the column-frame cell, the table of per-column floors
    // (entry k = 1024 + k * ge; column 0 is entry G), 8 B per column, the same entry for every lane.
    const uint32_t lstep = !first ? gl * 8u : C::kShifted ? 8u : 0u;
    const uint32_t voffl = first && C::kShifted ? 0u : voff;
    // (wave-uniform by construction; said so explicitly: the asm statements take them in scalar registers)
    uint64_t lptr = osw_uniform64(!first ? (PIPE ? (uint64_t)(src_region + OSW_SCRATCH_DATA) : data)
                                         : C::kShifted ? (uint64_t)(top_pages + C::kTopTable + G) : (uint64_t)(top_pages + (C::kFloorBits ? 64 : 0)));
    [[maybe_unused]] auto wait_for = [&](uint32_t cols) { // columns 0 .. cols-1 of round rho - 1 (published under the tag rho) are stored and visible
        const uint32_t need = (rho << 20) | (cols < ncols ? cols : ncols);
        while (*prog_src < need) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    if constexpr (PIPE) { if (!first) wait_for(2); }
    uint64_t sptr = osw_uniform64(keep ? hs - (uint64_t)(G - 1u) * gl * 8u : last ? (uint64_t)(bnd + OSW_SCRATCH_TRASH) : data - (uint64_t)(G - 1u) * gl * 8u);
    uint64_t tptr = (uint64_t)tcol - (uint64_t)(G - 1u) * CS;
    uint64_t sv;
    // nothing has been handed over yet: zeros; columns 0 and 1 of the stream
    asm volatile("v_mov_b32 " OSW_VH ", %[fl]\n\t"
                 "v_mov_b32 " OSW_VF ", %[fl]\n\t"
                 "global_load_ushort " OSW_VC0 ", %[voffc], %[tptr]\n\t"
                 "global_load_ushort " OSW_VC1 ", %[voffc], %[tptr] offset:%[cs]\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[mg0]\n\t"
                 "global_load_dword " OSW_VLH0 ", %[voffl], %[lptr]\n\t"
                 "global_load_dword " OSW_VLF0 ", %[voffl], %[lptr] offset:4\n\t"
                 "global_load_dword " OSW_VLH1 ", %[voffl], %[lptr2]\n\t"
                 "global_load_dword " OSW_VLF1 ", %[voffl], %[lptr2] offset:4\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : [sv] "=&s"(sv)
                 : [voffl] "v"(voffl), [voffc] "v"(voffc), [mg0] "s"(m_g0), [lptr] "s"(lptr), [lptr2] "s"(lptr + lstep), [tptr] "s"(tptr),
                   [fl] "v"(fl), [cs] "i"(CS)
                 : "memory", OSW_INFLIGHT);
    lptr += 2 * lstep;
    tptr += 2 * CS;
    const uint32_t nsteps = ncols + G - 1;
    // one column on input set P: the plain cells keep the running maximum themselves; the column-frame cell
    // returns the column's maximum in its frame, which is turned into a true score here
#define OSW_COLUMN(P)                                                                                              \
    if constexpr (C::kShifted) {                                                                                   \
        const T fl1 = C::frame_next(fl, ge);                                                                       \
        T cm; /* started by the first odd row */                                                                   \
        C::template column<R, P>(base, half, D, E, top_prev, hl, goe, ge, cm, fl1);                                \
        C::fold(score, cm, fl);                                                                                    \
        fl = fl1;                                                                                                  \
    } else {                                                                                                       \
        C::template column<R, P>(base, half, D, E, top_prev, hl, goe, ge, score, fl);                              \
    }
#define OSW_STEP_PAIR(T_END)                                    \
        {                                                       \
            OSW_STEP_BEGIN_ASM(OSW_VLF0);                       \
            T hl, tp;                                           \
            OSW_COLUMN(0)                                       \
            const uint32_t ho = C::to_bits(hl);                 \
            OSW_STEP_END_ASM(OSW_VC0, OSW_VLH0, OSW_VLF0);      \
            top_prev = tp;                                      \
            sptr += sstep;                                      \
            lptr += lstep;                                      \
            tptr += CS;                                         \
        }                                                       \
        if (t + 1 < (T_END)) {                                  \
            OSW_STEP_BEGIN_ASM(OSW_VLF1);                       \
            T hl, tp;                                           \
            OSW_COLUMN(1)                                       \
            const uint32_t ho = C::to_bits(hl);                 \
            OSW_STEP_END_ASM(OSW_VC1, OSW_VLH1, OSW_VLF1);      \
            top_prev = tp;                                      \
            sptr += sstep;                                      \
            lptr += lstep;                                      \
            tptr += CS;                                         \
        }
    if constexpr (!PIPE) {
#pragma unroll 1
        for (uint32_t t = 0; t < nsteps; t += 2) {
            OSW_STEP_PAIR(nsteps)
        }
    } else {
        static_assert(OSW_PIPE_BATCH % 2 == 0, "a batch is whole pairs of steps");
#pragma unroll 1
        for (uint32_t t0 = 0; t0 < nsteps; t0 += OSW_PIPE_BATCH) {
            const uint32_t t1 = t0 + OSW_PIPE_BATCH < nsteps ? t0 + OSW_PIPE_BATCH : nsteps;
            if (!first) wait_for(t1 + 2); // the batch's steps load the row above two columns ahead: up to column t1 + 1
#pragma unroll 1
            for (uint32_t t = t0; t < t1; t += 2) {
                OSW_STEP_PAIR(t1)
            }
            if (!last) {
                // step t stores column t + 1 - G of this round's bottom row: columns 0 .. t1 - G are on their way; performed, then published
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory", OSW_INFLIGHT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 63) *prog_mine = ((rho + 1u) << 20) | (t1 >= G ? t1 + 1u - G : 0u);
            }
        }
    }
#undef OSW_STEP_PAIR
#undef OSW_COLUMN
    // the prefetches of the two columns past the end are still in flight
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory", OSW_INFLIGHT);
}

// ---------------------------------------------------------------------------
// The same round for the hand-scheduled 8-bit cell (CellQ8F, q8_cell.h): identical step structure -- 2 stores + 3
// loads per step through fixed registers, hand-counted waits -- on the register set v72..v79 that six waves per SIMD
// leave room for.  H and F travel between steps / lane groups / rounds as 7-bit offset values ("zero" = the offset c
// in every byte); a first round reads its row above from the constant page of that value (top_pages, behind the
// column-frame cell's floor table).
// ---------------------------------------------------------------------------
#define OSW8_STEP_BEGIN_ASM(LFP)                                                                             \
    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\t"                                                         \
                 "v_cndmask_b32 " OSW8_VF ", " OSW8_VF ", " LFP ", %[mg0]"                                   \
                 :                                                                                           \
                 : [mg0] "s"(m_g0)                                                                           \
                 : "memory", OSW8_INFLIGHT)

#define OSW8_STEP_END_ASM(CP, LHP, LFP)                                                                      \
    asm volatile("v_cndmask_b32 %[tp], " OSW8_VH ", " LHP ", %[mg0]\n\t"                                      \
                 "s_mov_b64 %[sv], exec\n\t"                                                                 \
                 "s_mov_b64 exec, %[mst]\n\t"                                                                \
                 "global_store_dword %[voff], %[ho], %[sptr]\n\t"                                            \
                 "global_store_dword %[voff], " OSW8_VF ", %[sptr] offset:4\n\t"                             \
                 "s_not_b64 exec, %[mg0]\n\t"                                                                \
                 "s_cbranch_execz 1f\n\t"                                                                    \
                 "s_mov_b64 exec, %[sv]\n\t"                                                                 \
                 "ds_bpermute_b32 " OSW8_VH ", %[src], %[ho]\n\t"                                            \
                 "ds_bpermute_b32 " OSW8_VF ", %[src], " OSW8_VF "\n"                                         \
                 "1:\n\t"                                                                                    \
                 "s_mov_b64 exec, %[mg0]\n\t"                                                                \
                 "global_load_dword " LHP ", %[voffl], %[lptr]\n\t"                                          \
                 "global_load_dword " LFP ", %[voffl], %[lptr] offset:4\n\t"                                 \
                 "s_mov_b64 exec, %[sv]\n\t"                                                                 \
                 "global_load_ushort " CP ", %[voffc], %[tptr]"                                              \
                 : [tp] "=&v"(tp), [sv] "=&s"(sv)                                                            \
                 : [src] "v"(src), [ho] "v"(ho), [voff] "v"(voff), [voffl] "v"(voffl), [voffc] "v"(voffc), [mst] "s"(m_st), \
                   [mg0] "s"(m_g0), [sptr] "s"(sptr), [lptr] "s"(lptr), [tptr] "s"(tptr)                      \
                 : "memory", "scc", OSW8_INFLIGHT)

template <int R>
static __device__ __forceinline__ void sw_round_q8f(const uint16_t *tcol, uint32_t u, uint32_t ncols, uint32_t base, uint2 *bnd,
                                                    const uint2 *top_pages, bool first, bool last, uint32_t G, uint32_t gl, int lane,
                                                    const CellQ8::GapT &gp, uint32_t &score)
{
    typedef CellQ8F C;
    const uint64_t m_g0 = gl >= 64 ? ~0ull : (1ull << gl) - 1ull; // lanes of group 0
    const uint64_t m_st = ~0ull << (64u - gl);                     // lanes of the last group
    const uint32_t src = (uint32_t)((lane - (int)gl) & 63) << 2;   // ds_bpermute source: the lane one group below
    const uint32_t g = (uint32_t)lane / gl;
    const uint32_t zb = C::zero_bits(gp);
    const uint32_t voff = u * 8u, voffc = u * 2u + (G - 1u - g) * 128u;
    const uint64_t data = (uint64_t)(bnd + OSW_SCRATCH_DATA);
    const uint32_t sstep = last ? 0u : gl * 8u;
    const uint32_t lstep = first ? 0u : gl * 8u;
    const uint32_t voffl = first ? 0u : voff;
    uint64_t lptr = first ? (uint64_t)(top_pages + 128 + OSW_I16S_TABLE) : data;
    uint64_t sptr = last ? (uint64_t)(bnd + OSW_SCRATCH_TRASH) : data - (uint64_t)(G - 1u) * gl * 8u;
    uint64_t tptr = (uint64_t)tcol - (uint64_t)(G - 1u) * 128u;
    uint64_t sv;
    asm volatile("v_mov_b32 " OSW8_VH ", %[fl]\n\t"
                 "v_mov_b32 " OSW8_VF ", %[fl]\n\t"
                 "global_load_ushort " OSW8_VC0 ", %[voffc], %[tptr]\n\t"
                 "global_load_ushort " OSW8_VC1 ", %[voffc], %[tptr] offset:128\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[mg0]\n\t"
                 "global_load_dword " OSW8_VLH0 ", %[voffl], %[lptr]\n\t"
                 "global_load_dword " OSW8_VLF0 ", %[voffl], %[lptr] offset:4\n\t"
                 "global_load_dword " OSW8_VLH1 ", %[voffl], %[lptr2]\n\t"
                 "global_load_dword " OSW8_VLF1 ", %[voffl], %[lptr2] offset:4\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : [sv] "=&s"(sv)
                 : [voffl] "v"(voffl), [voffc] "v"(voffc), [mg0] "s"(m_g0), [lptr] "s"(lptr), [lptr2] "s"(lptr + lstep), [tptr] "s"(tptr),
                   [fl] "v"(zb)
                 : "memory", OSW8_INFLIGHT);
    lptr += 2 * lstep;
    tptr += 256;
    const uint32_t nsteps = ncols + G - 1;
    // the strip's state is set up HERE, from values that come out of an asm statement ordered behind the prologue's:
    // set up earlier, its 2R registers would be live across the prologue and spilled around it
    uint32_t cg = gp.cG, d0 = zb - gp.bias;
    asm volatile("" : "+v"(cg), "+v"(d0) : : OSW8_INFLIGHT);
    uint32_t D[R], E[R], top_prev = zb;
#pragma unroll
    for (int r = 0; r < R; ++r) { D[r] = d0; E[r] = cg; }
    uint32_t sc = (score & gp.L) | gp.G, fl = score; // running score with its guard; the sticky flags of earlier rounds ride in fl
#pragma unroll 1
    for (uint32_t t = 0; t < nsteps; t += 2) {
        {
            OSW8_STEP_BEGIN_ASM(OSW8_VLF0);
            uint32_t hl, tp;
            C::template column<R, 0>(base, D, E, top_prev, hl, gp, sc, fl);
            const uint32_t ho = hl;
            OSW8_STEP_END_ASM(OSW8_VC0, OSW8_VLH0, OSW8_VLF0);
            top_prev = tp;
            sptr += sstep;
            lptr += lstep;
            tptr += 128;
        }
        if (t + 1 < nsteps) {
            OSW8_STEP_BEGIN_ASM(OSW8_VLF1);
            uint32_t hl, tp;
            C::template column<R, 1>(base, D, E, top_prev, hl, gp, sc, fl);
            const uint32_t ho = hl;
            OSW8_STEP_END_ASM(OSW8_VC1, OSW8_VLH1, OSW8_VLF1);
            top_prev = tp;
            sptr += sstep;
            lptr += lstep;
            tptr += 128;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory", OSW8_INFLIGHT);
    score = (sc & gp.L) | (fl & gp.G);
}

template <class C>
static __device__ __forceinline__ void sw_round_dispatch(uint32_t R, const uint16_t *tcol, uint32_t u, uint32_t ncols, uint32_t base, uint2 *bnd,
                                                         const uint2 *top_pages, bool first, bool last, uint32_t G, uint32_t gl, int lane, int half,
                                                         typename C::GapT goe, typename C::GapT ge, typename C::T &score, uint2 *hand_store = nullptr)
{
#define OSW_ROUND_CASE(RR)                                                                                                      \
    case RR:                                                                                                                    \
        if constexpr (RR > C::kRows) break; /* taller than the cell's strips: never planned, not compiled */                    \
        else if constexpr (std::is_same<C, CellQ8F>::value) sw_round_q8f<RR>(tcol, u, ncols, base, bnd, top_pages, first, last, G, gl, lane, goe, score); \
        else sw_round_fast<C, RR>(tcol, u, ncols, base, bnd, top_pages, first, last, G, gl, lane, half, goe, ge, score, nullptr, nullptr, nullptr, 0u, hand_store); \
        break
    switch (R) {
        OSW_ROUND_CASE(4);
        OSW_ROUND_CASE(8);
        OSW_ROUND_CASE(12);
    default:
        if constexpr (C::kRows >= 16) {
            if (R == 16) {
                sw_round_fast<C, 16>(tcol, u, ncols, base, bnd, top_pages, first, last, G, gl, lane, half, goe, ge, score, nullptr, nullptr, nullptr, 0u, hand_store);
                break;
            }
        }
        if constexpr (C::kRows > 16) {
            switch (R) {
                OSW_ROUND_CASE(20);
                OSW_ROUND_CASE(24);
                OSW_ROUND_CASE(28);
                OSW_ROUND_CASE(32);
                OSW_ROUND_CASE(36);
                OSW_ROUND_CASE(40);
                OSW_ROUND_CASE(44);
                OSW_ROUND_CASE(48);
            default: break;
            }
        }
        break;
    }
#undef OSW_ROUND_CASE
}

// Copy the round's profile slice into LDS: G lane groups x rbg row-blocks (4 rows x 32 codes) each, taken
// from row-block rb0 of the query's profile on (row-blocks at or past rb_end are beyond the query and read as
// zero scores).  Group g's part starts g ENTRIES later than a dense layout would put it (one entry = the
// scores of one residue code for 4 rows: 8 B, 16 B for query pairs): lanes of different groups that look up
// the same code -- all lanes past the end of their sequence read the dummy code -- would otherwise hit the same
// LDS banks at different addresses (a G-way conflict); shifted by one entry per group they hit neighbouring
// banks.  E = uint2 / uint4 (the entry); tid / nthr: the threads that copy (a wave, or the whole workgroup).
template <class E, uint32_t NC = 32u> // NC = entries (residue codes) per row-block
static __device__ __forceinline__ void fill_profile_slice(const E *prof_q, uint32_t rb0, uint32_t rbg, uint32_t G, uint32_t rb_end, E *dst,
                                                          uint32_t tid, uint32_t nthr)
{
    const uint32_t per_group = rbg * NC; // entries
    for (uint32_t g = 0; g < G; ++g) {
        const uint32_t rbs = rb0 + g * rbg; // first row-block of the group
        const uint32_t valid = rb_end > rbs ? ((rb_end - rbs) < rbg ? (rb_end - rbs) : rbg) * NC : 0u;
        const E *src = prof_q + (size_t)rbs * NC;
        E *d = dst + (size_t)g * (per_group + 1u);
        for (uint32_t e = tid; e < per_group; e += nthr) d[e] = e < valid ? src[e] : E{};
    }
}

// The same for ONE row-block per lane group, keeping the first NCD of the source's NCS codes per row-block (the re-run pipeline's compact tables)
template <class E, uint32_t NCS, uint32_t NCD>
static __device__ __forceinline__ void fill_profile_slice_compact(const E *prof_q, uint32_t rb0, uint32_t G, uint32_t rb_end, E *dst, uint32_t tid, uint32_t nthr)
{
    for (uint32_t i = tid; i < G * NCD; i += nthr) {
        const uint32_t g = i / NCD, code = i - g * NCD, rb = rb0 + g; // group g: row-block rb0 + g of the query
        dst[(size_t)g * (NCD + 1u) + code] = rb < rb_end ? prof_q[(size_t)rb * NCS + code] : E{};
    }
}

// Diagnostics that exist only in the -DOSW_DIAG build of the library (liboswald_hip_diag.so, tools/): per-workgroup
// time stamps, and a timing experiment that makes every round read the constant top row and store to the trash page
// (WRONG scores, same instruction stream, no spill traffic).  The shipped library contains neither.
#ifdef OSW_DIAG
#define OSW_DIAG_NOSPILL(p) ((p).debug_nospill != 0)
#define OSW_DIAG_STAMP(cond, slot) do { if (p.wg_times && (cond)) p.wg_times[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OSW_DIAG_NOSPILL(p) false
#define OSW_DIAG_STAMP(cond, slot) do { } while (0)
#endif

// the constant "row above a first round" of a cell: top_pages, or the cell's own table
template <class C, class = void> struct osw_has_top_pages : std::false_type {};
template <class C> struct osw_has_top_pages<C, std::void_t<decltype(&C::top_pages)>> : std::true_type {};
template <class C>
static __device__ __forceinline__ const uint2 *osw_top_pages(const OswSearchArgs &p)
{
    if constexpr (osw_has_top_pages<C>::value) return C::top_pages(p);
    else return p.top_pages;
}

// One work item: all rounds of (query q, block B, sub-block sigma) at geometry G.
// Returns the lane's best score (valid in the lanes of group 0 after the
// cross-group reduction).
//   wg = false: the wave works alone; its profile slice lives in its private
//               LDS region (kLdsRows rows).
//   wg = true : the four waves of the workgroup run four sub-blocks of the same
//               (query, block, G) item in step; they share ONE profile slice in
//               the workgroup's whole LDS (4 x kLdsRows rows), which allows
//               4x taller rounds for heavy items.  The only synchronisation
//               is a pair of workgroup barriers around the slice reload.
// (wg is a run-time, workgroup-uniform flag: both kinds run the same round code.)
// HWL: `lane` is a logical lane number (osw_logical_lane; the packed-int16 kernels), else the physical one
// Tails (OswSearchArgs::hand): what a SHORT pair item and the tail behind it need beyond an ordinary item's arguments.
struct OswHand {
    uint2 *region;     // HMODE 1: the wave's hand region; the LAST round of pass 0 leaves its bottom row there, of pass 1 in the wave's own spill region (in place)
    uint32_t mlen;     // HMODE 1: the rows the item runs as a pair (the shorter query's, rounded up to 4); HMODE 2: the rows of the tail
    uint32_t rb_end;   // HMODE 1: row-blocks of the pair's profile that are real (the LONGER query's: the strips run on into its own rows)
    uint32_t prof_rb0; // HMODE 2: the tail's first row-block in the single-query profile
};

// The tail's row above: the longer query's halves (the high halves of a pair's packed {H, F}) of what pass 0 left in the hand region
// (the lane's first sequence -> low half) and pass 1 in the spill region (second sequence -> high half), merged in place in the spill
// region.  Same wave, same geometry, same columns: entry (column, lane of the group) stays where it is, the frames are the same.
static __device__ __forceinline__ void osw_merge_hand(uint2 *bnd, const uint2 *hand_region, uint32_t ncols, uint32_t gl, int lane)
{
    uint2 *row = bnd + OSW_SCRATCH_DATA;
    const uint2 *h0 = hand_region + OSW_SCRATCH_DATA;
    const uint32_t n = ncols * gl;
    // eight entries a turn per lane: sixteen independent loads in flight
    for (uint32_t i0 = (uint32_t)lane; i0 < n; i0 += 8u * 64u) {
        uint2 e0[8], e1[8];
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
            const uint32_t i = i0 + t * 64u;
            const bool live = i < n;
            e0[t] = live ? h0[i] : make_uint2(0u, 0u);
            e1[t] = live ? row[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) {
            const uint32_t i = i0 + t * 64u;
            if (i < n) row[i] = make_uint2((e0[t].x >> 16) | (e1[t].x & 0xffff0000u), (e0[t].y >> 16) | (e1[t].y & 0xffff0000u));
        }
    }
}

// HMODE: 0 = an ordinary item; 1 = a SHORT pair item (hand.mlen rows, the last round's bottom row is kept); 2 = the tail behind it
// (the single-query cell: rows hand.mlen from row-block hand.prof_rb0 of `prof` on; its row above is in the spill region already)
template <class C, bool HWL = false, int HMODE = 0>
static __device__ __forceinline__ typename C::T run_item(const OswSearchArgs &p, const uint2 *prof, uint32_t q, uint32_t B, const OswBlock &blk, uint32_t sigma,
                                                         uint32_t lg, int lane, int half, bool wg, uint2 *lds_region, uint2 *bnd_wave,
                                                         typename C::GapT goe, typename C::GapT ge, OswHand hand = OswHand{nullptr, 0u, 0u, 0u})
{
    typedef typename C::T T;
    const uint32_t kLds = wg ? C::kLdsRows * (OSW_WG_THREADS / 64) : C::kLdsRows;
    const uint32_t G = 1u << lg, gl = 64u >> lg;
    const uint32_t u = (uint32_t)lane & (gl - 1), g = (uint32_t)lane >> (6 - lg);
    // the item's extent: the longest sequence of its sub-block
    const uint32_t ncols = __builtin_amdgcn_readfirstlane((uint32_t)p.sub_cols[(size_t)B * 128 + (G - 1u) + sigma]);
    const uint16_t *tcol = (const uint16_t *)osw_uniform64((uint64_t)(p.tiled + (size_t)blk.col4_off * 256 + sigma * gl));
    uint2 *bnd = (uint2 *)osw_uniform64((uint64_t)bnd_wave);
    uint32_t mlen, prof_rb0;
    if constexpr (HMODE == 0) { mlen = p.qlen[q]; prof_rb0 = p.prof_off[q]; }
    else if constexpr (HMODE == 1) { mlen = hand.mlen; prof_rb0 = p.prof_off[q]; }
    else { mlen = hand.mlen; prof_rb0 = hand.prof_rb0; }
    const OswPlan plan = osw_plan(mlen, G, kLds, C::kRows);
    typedef typename C::Entry Entry; // the profile scores of one residue code for 4 rows
    const Entry *prof_q = (const Entry *)prof + (size_t)prof_rb0 * (uint32_t)C::kCodes;
    constexpr bool tail = HMODE == 2;
    if (plan.rounds > 1 || tail) {
        // the scratch columns the prefetch and the drain steps read past the block's last one are the
        // row above of dummy columns: "zero" in the cell's representation (other items have written here)
        uint2 *pad = bnd + OSW_SCRATCH_DATA + (size_t)ncols * gl;
        for (uint32_t k = lane; k < (G + 2u) * gl; k += 64) {
            uint32_t z = C::kFloorBits;
            if constexpr (std::is_same<C, CellQ8F>::value) z = C::zero_bits(goe); // the 8-bit cell's "zero" is its offset c in every byte
            if constexpr (C::kShifted) z += (ncols + k / gl + G) * (uint32_t)ge; // zero in the frame of that column
            pad[k] = make_uint2(z, z);
        }
    }
    T score;
    if constexpr (std::is_same<C, CellQ8F>::value) score = C::score_init(goe);   // "zero" may depend on the scoring system (8-bit cell)
    else if constexpr (C::kShifted) score = C::from_bits(0u); // the column-frame cell keeps a true (unbiased) running score
    else score = C::zero();
    for (uint32_t rho = 0; rho < plan.rounds; ++rho) {
        // round rho: group g runs rows [G*row0 + g*R, +R) of the query
        uint32_t rb_end = plan.m4 / 4;
        if constexpr (HMODE == 1) rb_end = hand.rb_end; // (a SHORT pair item: its strips run on into the longer query's own rows)
        const uint32_t R = osw_round_rows(plan, rho), rb0 = G * osw_round_row0(plan, rho) / 4;
        if (wg) {
#ifdef OSW_DIAG // (OSWALD_HIP_DEBUG_TIMES: core-clock cycles this wave spends in the slice reload incl. both barriers)
            const unsigned long long tb = p.wg_times ? __builtin_readcyclecounter() : 0ull;
#endif
            __syncthreads(); // every wave is done with the previous slice
            uint32_t tid = threadIdx.x;
            asm volatile("" : "+v"(tid)); // keep the per-thread source address out of the registers that live across the rounds
            fill_profile_slice<Entry, (uint32_t)C::kCodes>(prof_q, rb0, R / 4, G, rb_end, (Entry *)lds_region, tid, OSW_WG_THREADS);
            __syncthreads();
#ifdef OSW_DIAG
            if (p.wg_times && lane == 0) atomicAdd(&p.counters_ovf[4], (uint32_t)((__builtin_readcyclecounter() - tb) >> 10));
#endif
        } else {
            // only this wave touches its region; LDS operations of one wave execute in order, the wave barriers
            // only pin the compiler's order
            __builtin_amdgcn_wave_barrier();
            fill_profile_slice<Entry, (uint32_t)C::kCodes>(prof_q, rb0, R / 4, G, rb_end, (Entry *)lds_region, (uint32_t)lane, 64u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t base = (uint32_t)(uintptr_t)((lds_cp)lds_region + g * (R * C::kRowBytes + (uint32_t)sizeof(Entry)));
        sw_round_dispatch<C>(R, tcol, u, ncols, base, bnd, osw_top_pages<C>(p), (rho == 0 && !tail) || OSW_DIAG_NOSPILL(p), rho + 1 == plan.rounds || OSW_DIAG_NOSPILL(p), G, gl, lane, half, goe, ge, score,
                             HMODE == 1 && rho + 1 == plan.rounds ? (half ? bnd : hand.region) + OSW_SCRATCH_DATA : nullptr);
    }
    // best over the strips = best over the lane groups (the lane index is laundered so that the permute
    // addresses are computed here instead of being kept in registers across all the rounds)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    for (uint32_t off = gl; off < 64; off <<= 1)
        score = C::vmax(score, C::from_bits((uint32_t)__builtin_amdgcn_ds_bpermute((HWL ? osw_physical_lane(ln ^ (int)off) : (ln ^ (int)off)) << 2, (int)C::to_bits(score))));
    return score;
}

// Issue priority of the wave for the duration of an item (0..3).  The queue
// planner raises it for the few items that are long compared with a wave's fair
// share of the launch: up to four waves share a SIMD's VALU issue, and a long
// item that had to take turns with three short-item waves would define the
// launch time.
static __device__ __forceinline__ void set_wave_prio(uint32_t prio)
{
    switch (prio) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// A score goes to the device's table (the top lists are selected there, the re-run tiers overwrite what they redo) and, if the
// caller's table is page-locked, straight into it as well (OswSearchArgs::scores_host): 4 bytes per (query, sequence) over the link.
static __device__ __forceinline__ void osw_store_score(const OswSearchArgs &p, uint32_t q, size_t seq, int v)
{
    p.scores[(size_t)q * p.score_stride + seq] = v;
    if (p.scores_host && seq < p.host_cols) p.scores_host[(size_t)q * p.host_stride + seq] = v;
}
static __device__ __forceinline__ void osw_store_score2(const OswSearchArgs &p, uint32_t q, size_t seq /* even */, int2 v)
{
    *(int2 *)(p.scores + (size_t)q * p.score_stride + seq) = v;
    if (p.scores_host) {
        int32_t *h = p.scores_host + (size_t)q * p.host_stride + seq;
        if (seq + 1 < p.host_cols && (((uintptr_t)h) & 7u) == 0) *(int2 *)h = v;
        else { if (seq < p.host_cols) h[0] = v.x; if (seq + 1 < p.host_cols) h[1] = v.y; }
    }
}

// Scores of one packed 16-bit item: written for the lanes of group 0; lanes at
// the ceiling of the cell arithmetic are queued for the exact int32 kernel.
// (tail: the rows of a pair's longer query beyond the pair's own: the pair item has stored the best score of the rows before)
template <class A>
static __device__ __forceinline__ void pk16_finish(const OswSearchArgs &p, uint32_t q, uint32_t B, const OswBlock &blk, uint32_t sigma,
                                                   uint32_t lg, int lane, v2s score, bool tail = false)
{
    const uint32_t gl = 64u >> lg;
    if ((uint32_t)lane < gl) {
        const uint32_t lam = sigma * gl + lane; // lane of the block = sequence pair
        int2 out;
        out.x = A::to_int(score.x);
        out.y = A::to_int(score.y);
        uint32_t queued = 0u; // (tail) sequences of the lane that the pair part of the item has sent to the int32 re-run already: not twice
        if (tail) {
            const int2 before = *(const int2 *)(p.scores + (size_t)q * p.score_stride + (size_t)blk.seq0 + 2 * lam);
            out.x = out.x > before.x ? out.x : before.x;
            out.y = out.y > before.y ? out.y : before.y;
            queued = (before.x < 0 || before.x >= A::kCeiling ? 1u : 0u) | (before.y < 0 || before.y >= A::kCeiling ? 2u : 0u);
        }
        osw_store_score2(p, q, (size_t)blk.seq0 + 2 * lam, out);
        const uint32_t hm = ((A::over(score.x) ? 1u : 0u) | (A::over(score.y) ? 2u : 0u)) & ~queued;
        // (an entry per sequence: the re-run gives an entry to one workgroup, and the two sequences of a lane -- neighbours in a length-sorted
        // database: two near-copies of one query -- in one entry would run one after the other: `hi`'s int32 tier 8.0 instead of 4.2 ms)
        if (hm & 1u) {
            const uint32_t k = atomicAdd(&p.counters_ovf[0], 1u);
            p.ovf_items[k] = make_uint2(OSW_ITEM_PACK(q, lam, 6u, 1u), B);
        }
        if (hm & 2u) {
            const uint32_t k = atomicAdd(&p.counters_ovf[0], 1u);
            p.ovf_items[k] = make_uint2(OSW_ITEM_PACK(q, lam, 6u, 2u), B);
        }
    }
}

// Scores of one query-pair item and one sequence half: lane's low half = query A,
// high half = query B, both against sequence 2*lam + half of the block.
template <class A>
static __device__ __forceinline__ void pk16q_finish(const OswSearchArgs &p, uint32_t pair, uint32_t B, const OswBlock &blk, uint32_t sigma,
                                                    uint32_t lg, int lane, int half, v2s score)
{
    const uint32_t gl = 64u >> lg;
    if ((uint32_t)lane < gl) {
        const uint32_t lam = sigma * gl + lane;
        const uint32_t qa = p.pair_q[2 * pair], qb = p.pair_q[2 * pair + 1];
        const size_t seq = (size_t)blk.seq0 + 2 * lam + half;
        const int sa = A::to_int(score.x), sb = A::to_int(score.y);
        osw_store_score(p, qa, seq, sa);
        osw_store_score(p, qb, seq, sb);
        if (A::over(score.x)) {
            const uint32_t k = atomicAdd(&p.counters_ovf[0], 1u);
            p.ovf_items[k] = make_uint2(OSW_ITEM_PACK(qa, lam, 6u, 1u << half), B);
        }
        if (A::over(score.y)) {
            const uint32_t k = atomicAdd(&p.counters_ovf[0], 1u);
            p.ovf_items[k] = make_uint2(OSW_ITEM_PACK(qb, lam, 6u, 1u << half), B);
        }
    }
}

// Which blocks the column-frame cell may take (ArithI16S; p.goe_pk holds the gap OPEN penalty there): the frame offset (columns + 2G + 2) * ge must
// stay within OSW_I16S_FRAME_MAX, and -- whatever ge is, 0 included -- a first round indexes the floor table
// BY COLUMN (entry G + column, up to two columns of prefetch and G - 1 drain steps past the block), so the
// block must also fit the table's OSW_I16S_TABLE entries.  Everything else runs on the plain biased cell.
// go <= 1024: the cell subtracts the gap penalties with 32-bit instructions on the packed pair (OSW_SUBU_32).
static __device__ __forceinline__ bool osw_frame_cell_takes(uint32_t cols, uint32_t lg, uint32_t ge, uint32_t go)
{
    const uint32_t span = cols + 2u * (1u << lg) + 2u;
    return ge <= 64u && go <= 1024u && span + 2u <= OSW_I16S_TABLE && span * ge <= OSW_I16S_FRAME_MAX;
}

// ---------------------------------------------------------------------------
// Main kernel: packed int16.
// ---------------------------------------------------------------------------
// C = the cell; CF = the cell for the blocks C cannot take (CF = C: none)
// TAILS (query-pair kernels): the queue may hold SHORT pair items (OswSearchArgs::hand), whose tails run on the single-query cells TC / TCF.
// A kernel of its own beside the plain pair kernel: with the single-query cell's code beside the pair cell's the compiler spills more
// item-level state around the rounds (SGPR spills 95 -> 118, scratch 12 -> 52 B), which items of a hundred rounds feel (`hi`: 2 %);
// a launch without SHORT items runs the plain kernel.
template <class C, class CF, bool PAIR, bool TAILS = false, class TC = C, class TCF = CF>
static __device__ __forceinline__ void pk16_body(const OswSearchArgs &p)
{
    __shared__ uint2 lds_prof[OSW_WG_THREADS / 64][OSW_LDS_ROWS16 * 8 + OSW_LDS_SKEW8]; // + one entry per lane group (fill_profile_slice)
    __shared__ uint32_t wg_item;
    const int lane = osw_logical_lane(threadIdx.x & 63); // (logical: every 16-lane group of the LDS is 16 consecutive lanes; lane 0 is lane 0)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slot = blockIdx.x * (OSW_WG_THREADS / 64) + wv;
    uint2 *bnd_wave = p.bnd + (size_t)slot * p.bnd_stride;
    [[maybe_unused]] uint2 *hand_wave = PAIR && TAILS && p.hand ? p.hand + (size_t)slot * p.bnd_stride : nullptr; // (null: the queue holds no SHORT item)

    // (-DOSW_DIAG: when each workgroup started, left phase 1 and finished, 100 MHz ticks)
    OSW_DIAG_STAMP(threadIdx.x == 0, blockIdx.x * 4 + 0);
#ifdef OSW_DIAG // (... and the core clock its waves ran at: cycle counter over the 100-MHz counter, summed over the workgroups behind the per-workgroup entries)
    const unsigned long long diag_c0 = __builtin_readcyclecounter(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Which end of the (cost-sorted) queues this workgroup eats from: the first workgroup to
    // arrive on a CU takes the heavy end, later arrivals the light end, so that a long item
    // shares its SIMD with short ones (and, with its raised priority, runs at nearly the full
    // issue rate) instead of with three other long ones.  Placement is read from the hardware
    // id registers; it only steers speed, never correctness.
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));   // HW_REG_HW_ID[15:8]: cu, sh, se
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID[3:0]
        const uint32_t cu = ((xcc & 15u) << 8) | (hw & 255u);
        wg_item = atomicAdd(&p.counters[OSW_CTR_CU0 + (cu & (OSW_CTR_CUS - 1))], 1u);
#ifdef OSW_DIAG
        if (p.wg_times) p.wg_times[(size_t)gridDim.x * 4 + blockIdx.x] = cu; // (behind the time stamps: which CU the workgroup ran on)
#endif
    }
    __syncthreads();
    const bool heavy_end = wg_item == 0 || p.one_ended_wg != 0;
    __syncthreads();

    // Phase 1: workgroup entries, heaviest first -- the workgroup's four waves on four sub-blocks of one item
    // (shared profile slice, rounds in step), or on a quad of four independent heavy wave items.  Phase 2: every
    // wave on its own, light wave items.  One loop, so that the (large, fully unrolled) round code exists once.
    // (the re-run queue of the 8-bit pass -- workgroup entries only -- was filled by the previous kernel on this stream)
    const uint32_t nitems_wg = p.nitems_dev ? __builtin_amdgcn_readfirstlane(*p.nitems_dev) : p.nitems_wg;
    const uint2 *wave_items = p.items + (size_t)nitems_wg * 4;
    const uint32_t nitems = p.nitems;
    bool phase1 = true;
    for (;;) {
        uint2 item;
        bool shared = false;
        if (phase1) {
            if (threadIdx.x == 0) {
                uint32_t t = atomicAdd(&p.counters[OSW_CTR_WORK_WG], 1u);
                if (t < nitems_wg) t = heavy_end ? atomicAdd(&p.counters[OSW_CTR_FRONT_WG], 1u) : nitems_wg - 1 - atomicAdd(&p.counters[OSW_CTR_BACK_WG], 1u);
                wg_item = t;
            }
            __syncthreads();
            const uint32_t it = wg_item;
            __syncthreads();
            if (it >= nitems_wg) { // all four waves see this together
                phase1 = false;
                OSW_DIAG_STAMP(threadIdx.x == 0, blockIdx.x * 4 + 1);
                continue;
            }
            item = p.items[(size_t)it * 4 + wv];
            shared = (item.y & OSW_ITEM_WG_FLAG) != 0; // the same for the four slots of an entry
            if (!shared && item.y == OSW_ITEM_NONE) continue;
        } else {
            uint32_t it = 0;
            if (lane == 0) {
                it = atomicAdd(&p.counters[OSW_CTR_WORK], 1u);
                if (it < nitems && p.two_ended_waves) it = heavy_end ? atomicAdd(&p.counters[OSW_CTR_FRONT], 1u) : nitems - 1 - atomicAdd(&p.counters[OSW_CTR_BACK], 1u);
            }
            it = __builtin_amdgcn_readfirstlane(it);
            if (it >= nitems) break;
            item = wave_items[it];
        }
        const uint32_t q = OSW_ITEM_Q(item.x), lg = OSW_ITEM_LG(item.x), sigma = OSW_ITEM_SIGMA(item.x);
        const uint32_t B = item.y & ~OSW_ITEM_WG_FLAG;
        set_wave_prio(OSW_ITEM_PRIO(item.x));
        uint2 *lds_region = shared ? &lds_prof[0][0] : lds_prof[wv];
        const OswBlock blk = p.blocks[B];
        // the column-frame cell only takes blocks whose frame offset stays small (ArithI16S); the rest run on CF
        const bool cf_only = C::kShifted && !osw_frame_cell_takes(blk.ncols4 * 4u, lg, (uint32_t)p.ge, p.goe_pk & 0xffffu);
        // Tails (OswSearchArgs::hand): a SHORT pair item runs the shorter query's rows as a pair -- its last rounds keep their bottom rows --
        // and then the rest of the longer query on the single-query cell (TC / TCF), same wave, same sub-block, same geometry, same
        // representation (what decides the cell, `cf_only`, is the same for both).
        // (separate instantiations of the item code: the ordinary items run the one without any hand-over state -- kept through the rounds, that
        // state costs the query-pair kernel 3 % on many-round items: registers it does not have)
        bool shorty = false;
        if constexpr (PAIR && TAILS) shorty = OSW_ITEM_HALVES(item.x) == OSW_ITEM_SHORT; // (the planner's choice, item by item)
        for (int half = 0; half < (PAIR ? 2 : 1); ++half) {
            v2s score;
            if constexpr (PAIR) {
                if (cf_only) {
                    if constexpr (TAILS) {
                        const OswHand hand = {hand_wave, shorty ? (uint32_t)p.pair_rows[q] : 0u, (p.qlen[q] + 3u) / 4u, 0u};
                        if (shorty) score = run_item<CF, true, 1>(p, p.prof_fb, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_fb, p.ge_fb, hand);
                        else score = run_item<CF, true, 0>(p, p.prof_fb, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_fb, p.ge_fb);
                    } else score = run_item<CF, true, 0>(p, p.prof_fb, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_fb, p.ge_fb);
                    pk16q_finish<typename CF::Arith>(p, q, B, blk, sigma, lg, lane, half, score);
                } else {
                    if constexpr (TAILS) {
                        const OswHand hand = {hand_wave, shorty ? (uint32_t)p.pair_rows[q] : 0u, (p.qlen[q] + 3u) / 4u, 0u};
                        if (shorty) score = run_item<C, true, 1>(p, p.prof, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_pk, p.ge_pk, hand);
                        else score = run_item<C, true, 0>(p, p.prof, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_pk, p.ge_pk);
                    } else score = run_item<C, true, 0>(p, p.prof, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_pk, p.ge_pk);
                    pk16q_finish<typename C::Arith>(p, q, B, blk, sigma, lg, lane, half, score);
                }
            } else {
                if (cf_only) {
                    score = run_item<CF, true, 0>(p, p.prof_fb, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_fb, p.ge_fb);
                    pk16_finish<typename CF::Arith>(p, q, B, blk, sigma, lg, lane, score);
                } else {
                    score = run_item<C, true, 0>(p, p.prof, q, B, blk, sigma, lg, lane, half, shared, lds_region, bnd_wave, p.goe_pk, p.ge_pk);
                    pk16_finish<typename C::Arith>(p, q, B, blk, sigma, lg, lane, score);
                }
            }
        }
        if constexpr (PAIR && TAILS) {
            // (wave-uniform, and for a workgroup item the same in its four waves: one entity, one geometry)
            const uint32_t mt = shorty ? __builtin_amdgcn_readfirstlane((uint32_t)p.tail_len[q * OSW_TAIL_GEOMS + lg]) : 0u;
            if (mt) {
                const uint32_t ncols = __builtin_amdgcn_readfirstlane((uint32_t)p.sub_cols[(size_t)B * 128 + ((1u << lg) - 1u) + sigma]);
                osw_merge_hand(bnd_wave, hand_wave, ncols, 64u >> lg, lane);
                const OswHand th = {nullptr, mt, 0u, p.tail_off[q * OSW_TAIL_GEOMS + lg]};
                const uint32_t qb = p.pair_q[2 * q + 1]; // the longer query of the pair: the row of the table the tail's scores belong to
                v2s score;
                if (cf_only) {
                    score = run_item<TCF, true, 2>(p, p.tail_prof_fb, q, B, blk, sigma, lg, lane, 0, shared, lds_region, bnd_wave, p.goe_fb, p.ge_fb, th);
                    pk16_finish<typename TCF::Arith>(p, qb, B, blk, sigma, lg, lane, score, true);
                } else {
                    score = run_item<TC, true, 2>(p, p.tail_prof, q, B, blk, sigma, lg, lane, 0, shared, lds_region, bnd_wave, p.goe_pk, p.ge_pk, th);
                    pk16_finish<typename TC::Arith>(p, qb, B, blk, sigma, lg, lane, score, true);
                }
            }
        }
        set_wave_prio(0);
    }
    OSW_DIAG_STAMP(lane == 0, blockIdx.x * 4 + 2 + (wv & 1)); // waves 0/1 (or 2/3) race: any is fine
#ifdef OSW_DIAG
    if (p.wg_times && threadIdx.x == 0) {
        atomicAdd(&p.wg_times[(size_t)gridDim.x * 5 + 0], __builtin_readcyclecounter() - diag_c0);
        atomicAdd(&p.wg_times[(size_t)gridDim.x * 5 + 1], __builtin_amdgcn_s_memrealtime() - diag_r0);
    }
#endif
}

extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_pk16(OswSearchArgs p) { pk16_body<CellPK16B, CellPK16B, false>(p); }

// Query pairs: `items` / `qlen` / `prof` / `prof_off` describe pairs (length = the longer query,
// profile = packed (A, B) scores); pair_q maps a pair to its two query rows of the score table.
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_pk16q(OswSearchArgs p) { pk16_body<CellPK16BQ, CellPK16BQ, true>(p); }
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_pk16qt(OswSearchArgs p) { pk16_body<CellPK16BQ, CellPK16BQ, true, true, CellPK16B, CellPK16B>(p); } // ... whose queue may hold SHORT items

// Column-frame cell (6.5 instructions per row) with the plain biased cell for the blocks it cannot take:
// `prof` holds S + ge, goe_pk the gap OPEN penalty; prof_fb / goe_fb / ge_fb serve the plain cell.
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_s16(OswSearchArgs p) { pk16_body<CellPK16S, CellPK16B, false>(p); }
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_s16q(OswSearchArgs p) { pk16_body<CellPK16SQ, CellPK16BQ, true>(p); }
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_s16qt(OswSearchArgs p) { pk16_body<CellPK16SQ, CellPK16BQ, true, true, CellPK16S, CellPK16B>(p); } // ... whose queue may hold SHORT items

// ---------------------------------------------------------------------------
// The int32 re-run of ONE (query, sequence) on the TWELVE waves of a workgroup.  A sequence that reaches the int16 cells'
// ceiling is a near-copy of a long query: thousands of rows against thousands of columns, 20+ rounds of 256 rows at
// geometry 64 (every lane a 4-row strip), which one wave ran one after the other until round 4 -- 30-60 ms per item; with
// a hundred such items on a device of 4096 wave slots the re-run took a quarter of the whole search although it is 0.2 %
// of its cells (bench.py --workload hi).  Here wave w of the workgroup runs rounds w, w+12, w+24, ... of the item, each
// behind the round before it by a few dozen columns: the boundary row {H, F} of round rho goes through the spill scratch
// of the wave that ran it (a region per wave, as always) to the wave that runs round rho+1, column by column
// (sw_round_fast<.., PIPE>).  The producer publishes, in LDS, how many columns of its round are stored (its stores have
// been performed; the waves of a workgroup share the CU's L1); the consumer waits for the columns of its next batch of 32
// steps.  No wave ever waits for a later round, the first round waits for nothing, and the waves of a workgroup are
// resident together: the waits cannot deadlock.  A region is overwritten by its owner's NEXT round (rho+12), which depends
// -- through rounds rho+11 ... rho+1 -- on the reader of this round's row having been there already.
//   Round 5, second session: the hand-scheduled cell (CellI32FP) in the packed-int16 kernels' column loop.  The compiler-
// scheduled round before it handed the residues from lane to lane together with the row, so the profile read of a step
// waited for the hand-off of the step before (~1 050 cycles per step); here every lane loads its own residues two columns
// ahead, from a compact copy of the sequence (560 cycles per step: hi's 133 items 8.75 -> 4.67 ms), and the lane groups' tables hold the
// 24 residue codes the database can contain instead of 32, which makes room in LDS for twelve waves where there were eight (a 5 000-row
// query: two rounds per wave instead of three; 4.07 ms).  `lane` is the logical lane.
// ---------------------------------------------------------------------------
template <int NW>
static __device__ __forceinline__ int run_item_i32f_pipe(const OswSearchArgs &p, uint32_t q, uint32_t B, const OswBlock &blk, uint32_t sigma, uint32_t lg,
                                                         int lane, int wv, int half, uint2 *lds_wave, uint2 *bnd_wg, volatile uint32_t *prog, int *red)
{
    typedef CellI32FP C;
    const uint32_t G = 1u << lg, gl = 64u >> lg;
    const uint32_t u = (uint32_t)lane & (gl - 1), g = (uint32_t)lane >> (6 - lg);
    const uint32_t ncols = __builtin_amdgcn_readfirstlane((uint32_t)p.sub_cols[(size_t)B * 128 + (G - 1u) + sigma]);
    const uint16_t *tcol = (const uint16_t *)osw_uniform64((uint64_t)(p.tiled + (size_t)blk.col4_off * 256 + sigma * gl));
    const OswPlan plan = osw_plan(p.qlen[q], G, C::kLdsRows, C::kRows);
    const uint2 *prof_q = p.prof_i32 + (size_t)p.prof_off[q] * 32u;
    uint2 *mine = (uint2 *)osw_uniform64((uint64_t)(bnd_wg + (size_t)wv * p.bnd_stride));
    const uint2 *prev = (const uint2 *)osw_uniform64((uint64_t)(bnd_wg + (size_t)((wv + NW - 1) % NW) * p.bnd_stride));
    const uint32_t go = (uint32_t)(p.goe - p.ge), ge = (uint32_t)p.ge;
    {
        // the columns the prefetch and the drain steps of the NEXT round read past the block's last one: zero in their frames
        uint2 *pad = mine + OSW_SCRATCH_DATA + (size_t)ncols * gl;
        for (uint32_t k = (uint32_t)lane; k < (G + 2u) * gl; k += 64) {
            const uint32_t z = (ncols + k / gl + G) * ge;
            pad[k] = make_uint2(z, z);
        }
    }
    // A compact copy of the item's residues: one uint16 per column (`tiled` keeps a 128-byte row per column for the block's 64 lanes; at
    // geometry 64 every lane of a wave reads a different column of the SAME lane of the block: 64 cache lines per load).  It lives
    // behind the columns of wave 0's spill region -- at one lane per group a region is used to a 32nd --, with the dummy columns the
    // lane groups' warm-up and drain steps read in front of it and behind it (`tiled` has them around every block).
    uint16_t *codes = (uint16_t *)osw_uniform64((uint64_t)((uint16_t *)(bnd_wg + OSW_SCRATCH_DATA + ncols + OSW_SCRATCH_PAD_COLS + 8u) + 64));
    for (int c = (int)threadIdx.x - 64; c < (int)ncols + (int)OSW_SCRATCH_PAD_COLS; c += NW * 64) codes[c] = tcol[(ptrdiff_t)c * 64 + u];
    if (lane == 0) prog[wv] = 0;
    __syncthreads();
    int score = 0;
    for (uint32_t rho = (uint32_t)wv; rho < plan.rounds; rho += NW) {
        const uint32_t R = osw_round_rows(plan, rho), rb0 = G * osw_round_row0(plan, rho) / 4, rb_end = plan.m4 / 4;
        __builtin_amdgcn_wave_barrier();
        fill_profile_slice_compact<uint2, 32u, (uint32_t)C::kCodes>(prof_q, rb0, G, rb_end, lds_wave, (uint32_t)lane, 64u); // (R == 4: one row-block per group)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t base = (uint32_t)(uintptr_t)((lds_cp)lds_wave + g * (R * C::kRowBytes + (uint32_t)sizeof(uint2)));
        const bool first = rho == 0, last = rho + 1 == plan.rounds;
        const lds_flagp ps = (lds_flagp)(prog + ((wv + NW - 1) % NW)), pm = (lds_flagp)(prog + wv);
        sw_round_fast<C, 4, true>(codes, u, ncols, base, mine, p.floor_i32, first, last, G, gl, lane, half, go, ge, score, prev, ps, pm, rho); // (R == 4: C::kRows)
    }
    // best over the strips = best over the lane groups, then over the waves
    int ln = lane;
    asm volatile("" : "+v"(ln));
    for (uint32_t off = gl; off < 64; off <<= 1) {
        const int o = __builtin_amdgcn_ds_bpermute(osw_physical_lane(ln ^ (int)off) << 2, score);
        score = o > score ? o : score;
    }
    red[wv * 64 + lane] = score;
    __syncthreads();
    int best = red[lane];
    for (int w = 1; w < NW; ++w) best = red[w * 64 + lane] > best ? red[w * 64 + lane] : best;
    __syncthreads(); // (red and prog are free for the next item)
    return best;
}

// ---------------------------------------------------------------------------
// Exact int32 kernel.  Default: re-run of the lanes queued by osw_sw_pk16 at
// geometry 64 (each lane one strip of the same sequence: the whole wave works
// on one sequence at a time).  force_all: run `items` (cell_bits = 32 mode).
// ---------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 3) OSW_COMPILER_VGPRS void osw_sw_i32(OswSearchArgs p)
{
    __shared__ uint2 lds_prof[OSW_WG_THREADS / 64][OSW_LDS_ROWS32F * 8 + OSW_LDS_SKEW8];
    const int lane = osw_logical_lane(threadIdx.x & 63); // (the hand-scheduled cell numbers its lanes like the packed-int16 kernels; lane 0 is lane 0)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slot = blockIdx.x * (OSW_WG_THREADS / 64) + wv;
    uint2 *bnd_wave = p.bnd + (size_t)slot * p.bnd_stride;
    uint2 *lds_wave = lds_prof[wv];
    const uint32_t nitems = p.nitems;                  // cell_bits = 32: the item list of the plan (wave items)
    const uint2 *items = p.items + (size_t)p.nitems_wg * 4;
    const uint32_t go = (uint32_t)(p.goe - p.ge), ge = (uint32_t)p.ge;
    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(&p.counters[OSW_CTR_WORK32], 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= nitems) break;
        const uint2 item = items[it];
        const uint32_t q = OSW_ITEM_Q(item.x), sigma = OSW_ITEM_SIGMA(item.x), lg = OSW_ITEM_LG(item.x), B = item.y;
        const uint32_t hm = OSW_ITEM_HALVES(item.x);
        const OswBlock blk = p.blocks[B];
        const uint32_t gl = 64u >> lg;
        for (int half = 0; half < 2; ++half) {
            if (!((hm >> half) & 1u)) continue;
            const int score = run_item<CellI32F, true>(p, p.prof_i32, q, B, blk, sigma, lg, lane, half, false, lds_wave, bnd_wave, go, ge);
            if ((uint32_t)lane < gl) osw_store_score(p, q, (size_t)blk.seq0 + 2 * (sigma * gl + lane) + half, score);
        }
    }
}

// The re-run queue of a search (what reached the int16 cells' ceiling): few, long items -- a workgroup of TWELVE waves per
// item, a pipeline over the item's rounds (run_item_i32f_pipe); three waves per SIMD hide each other's latencies.  The queue
// length was produced by the kernels before it on this stream.  Wave w of workgroup b uses spill region 12 b + w: the grid
// is at most a twelfth of the regions (osw_launch_i32r).
#define OSW_I32R_WAVES 12
#define OSW_I32R_SLICE (64u * (OSW_SEQ_CODES + 1u)) // uint2 entries of a wave's profile slice: 64 lane groups x (24 codes + the skew entry)
extern "C" __global__ __launch_bounds__(OSW_I32R_WAVES * 64) OSW_COMPILER_VGPRS void osw_sw_i32r(OswSearchArgs p)
{
    extern __shared__ uint2 lds_dyn[]; // OSW_I32R_WAVES profile slices of OSW_I32R_SLICE entries (154 KB: above the static limit)
    __shared__ uint32_t prog[OSW_I32R_WAVES];
    __shared__ uint32_t wg_it;
    __shared__ int red[OSW_I32R_WAVES * 64];
    const int lane = osw_logical_lane(threadIdx.x & 63); // (the hand-scheduled cell's lane numbering; lane 0 is lane 0)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint2 *lds_wave = lds_dyn + (size_t)wv * OSW_I32R_SLICE;
    uint2 *bnd_wg = p.bnd + (size_t)blockIdx.x * OSW_I32R_WAVES * p.bnd_stride;
    const uint32_t nitems = p.counters_ovf[0];
    const uint2 *items = p.ovf_items;
    for (;;) {
        if (threadIdx.x == 0) wg_it = atomicAdd(&p.counters[OSW_CTR_WORK32], 1u);
        __syncthreads();
        const uint32_t it = wg_it;
        __syncthreads();
        if (it >= nitems) break;
        const uint2 item = items[it];
        const uint32_t q = OSW_ITEM_Q(item.x), sigma = OSW_ITEM_SIGMA(item.x), lg = OSW_ITEM_LG(item.x), B = item.y;
        const uint32_t hm = OSW_ITEM_HALVES(item.x);
        const OswBlock blk = p.blocks[B];
        const uint32_t gl = 64u >> lg;
        for (int half = 0; half < 2; ++half) {
            if (!((hm >> half) & 1u)) continue;
            const int score = run_item_i32f_pipe<OSW_I32R_WAVES>(p, q, B, blk, sigma, lg, lane, wv, half, lds_wave, bnd_wg, prog, red);
            if (wv == 0 && (uint32_t)lane < gl)
                osw_store_score(p, q, (size_t)blk.seq0 + 2 * (sigma * gl + lane) + half, score);
        }
    }
}

// ---------------------------------------------------------------------------
// 8-bit first pass (cell_bits = 8): query-pair items of the wave queue, one 2 x 2 tile per lane (CellQ8, q8_cell.h).
// Every (query, sequence) that left the cell's range is queued for the packed-int16 kernel, which re-runs the aligned
// quad of lanes (eight sequences) around it as ONE workgroup entry at geometry 64 and queues what reaches ITS ceiling
// for the int32 kernel.  The four lanes combine their flags (homologous sequences sit next to each other in a sorted
// database and usually flag together) and the first one queues the entry once.
// ---------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(OSW_WG_THREADS, 6) OSW8_COMPILER_VGPRS void osw_sw_q8(OswSearchArgs p)
{
    __shared__ uint2 lds_prof[OSW_WG_THREADS / 64][OSW_LDS_ROWS8 * 8 + OSW_LDS_SKEW_Q8];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slot = blockIdx.x * (OSW_WG_THREADS / 64) + wv;
    uint2 *bnd_wave = p.bnd + (size_t)slot * p.bnd_stride;
    const uint2 *items = p.items + (size_t)p.nitems_wg * 4;
    const CellQ8::GapT gp = CellQ8::make_gap(p.go8, p.ge8, p.bias8, p.off8);
    const int off = (int)p.off8;
    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(&p.counters[OSW_CTR_WORK], 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= p.nitems) break;
        const uint2 item = items[it];
        const uint32_t pair = OSW_ITEM_Q(item.x), sigma = OSW_ITEM_SIGMA(item.x), lg = OSW_ITEM_LG(item.x), B = item.y;
        const OswBlock blk = p.blocks[B];
        const uint32_t gl = 64u >> lg;
        const uint32_t score = run_item<CellQ8F>(p, p.prof, pair, B, blk, sigma, lg, lane, 0, false, lds_prof[wv], bnd_wave, gp, gp);
        const bool mine = (uint32_t)lane < gl;            // the lanes of group 0 hold the sub-block's scores
        uint32_t flags = mine ? score & 0x80808080u : 0u; // which of the lane's four (query, sequence) left the range
        // a re-run entry covers an aligned quad of lanes (eight sequences): combine the quad's flags
        flags |= (uint32_t)__builtin_amdgcn_ds_swizzle((int)flags, 0x041F); // | lane ^ 1
        flags |= (uint32_t)__builtin_amdgcn_ds_swizzle((int)flags, 0x081F); // | lane ^ 2
        if (mine) {
            const uint32_t lam = sigma * gl + lane; // lane of the block = sequence pair
            const uint32_t qa = p.pair_q[2 * pair], qb = p.pair_q[2 * pair + 1];
            const size_t seq = (size_t)blk.seq0 + 2 * lam;
            int2 ra, rb; // offset scores back to true ones (a flagged one is garbage and will be overwritten by the re-run)
            ra.x = (int)(score & 0x7fu) - off;         // A . s0
            rb.x = (int)((score >> 8) & 0x7fu) - off;  // B . s0
            ra.y = (int)((score >> 16) & 0x7fu) - off; // A . s1
            rb.y = (int)((score >> 24) & 0x7fu) - off; // B . s1
            osw_store_score2(p, qa, seq, ra);
            osw_store_score2(p, qb, seq, rb);
            if (!(lane & 3)) { // (gl < 4: the other lanes of the quad belong to other items, which queue it again if they flag: harmless)
                // one workgroup entry of the packed-int16 kernel: its four waves take the four lanes of the quad at
                // geometry 64 (every lane group 1/64 of the query: the shortest critical path the kernel offers)
                const uint32_t first = lam & ~3u;
                if (flags & 0x00800080u) { // query A, any sequence of the quad
                    const uint32_t k = atomicAdd(&p.counters_ovf[1], 1u);
                    for (uint32_t w = 0; w < 4; ++w) p.ovf8_items[4 * k + w] = make_uint2(OSW_ITEM_PACK(qa, first + w, 6u, 3u) | (3u << 30), B | OSW_ITEM_WG_FLAG);
                }
                if (flags & 0x80008000u) {
                    const uint32_t k = atomicAdd(&p.counters_ovf[1], 1u);
                    for (uint32_t w = 0; w < 4; ++w) p.ovf8_items[4 * k + w] = make_uint2(OSW_ITEM_PACK(qb, first + w, 6u, 3u) | (3u << 30), B | OSW_ITEM_WG_FLAG);
                }
            }
        }
    }
}

// 8-bit pair profile: prof8[(pair_off[p] + i/4)*32 + code] = 8 bytes {A r0, B r0, A r1, B r1, ...}, every byte
// S + bias (rows past a query's end: bias, i.e. score 0), built from the plain int16 single-query profiles.
extern "C" __global__ __launch_bounds__(256) void osw_build_pair_profile8(const uint2 *__restrict__ prof, const uint32_t *__restrict__ prof_off,
                                                                           const uint16_t *__restrict__ qlen, const uint32_t *__restrict__ pair_q,
                                                                           const uint32_t *__restrict__ pair_off, const uint16_t *__restrict__ pair_len,
                                                                           uint32_t npairs, int bias, uint2 *__restrict__ prof8)
{
    const uint32_t pr = blockIdx.y;
    if (pr >= npairs) return;
    const uint32_t qa = pair_q[2 * pr], qb = pair_q[2 * pr + 1];
    const uint32_t nrb = (pair_len[pr] + 3u) / 4u > 0 ? (pair_len[pr] + 3u) / 4u : 1u;
    const uint32_t na = (qlen[qa] + 3u) / 4u, nb = (qlen[qb] + 3u) / 4u;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < nrb * 32; e += gridDim.x * blockDim.x) {
        const uint32_t rb = e >> 5, code = e & 31;
        const uint2 A = rb < na ? prof[(size_t)(prof_off[qa] + rb) * 32 + code] : make_uint2(0, 0);
        const uint2 Bv = rb < nb ? prof[(size_t)(prof_off[qb] + rb) * 32 + code] : make_uint2(0, 0);
        auto pack = [&](uint32_t a16, uint32_t b16) { // -> {A + bias, B + bias} as two bytes
            return (uint32_t)(((int)(int16_t)a16 + bias) & 0xff) | ((uint32_t)(((int)(int16_t)b16 + bias) & 0xff) << 8);
        };
        uint2 o;
        o.x = pack(A.x & 0xffffu, Bv.x & 0xffffu) | (pack(A.x >> 16, Bv.x >> 16) << 16);
        o.y = pack(A.y & 0xffffu, Bv.y & 0xffffu) | (pack(A.y >> 16, Bv.y >> 16) << 16);
        prof8[(size_t)(pair_off[pr] + rb) * 32 + code] = o;
    }
}

// ---------------------------------------------------------------------------
// Upload-side kernels.
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// Upload-side kernels are "search-shaped" (second session of round 4): 256 threads that allocate what a workgroup of the
// packed-int16 search kernels allocates -- 168 VGPRs per wave, 53 KB of LDS (the launchers add dynamic LDS up to the search
// kernel's own figure).  A CU holds exactly three search workgroups; their waves and their LDS are placed contiguously and
// never move (a persistent grid).  Thousands of short workgroups with a small footprint that start TOGETHER with a search
// land between the search's workgroups while those are being placed, and the holes they leave behind (16 VGPRs here, 256 B
// of LDS there) are too small for a third search workgroup: such a CU runs two for the whole launch, the search 12 %
// slower (tools/plan_probe3.py, orders C / D: 123 instead of 110 ms).  Workgroups of the search's own shape leave holes a
// search workgroup fits exactly.  The re-tile runs a little slower for it (three workgroups per CU).
// ---------------------------------------------------------------------------

#define OSW_SEARCH_SHAPED_VGPRS() asm volatile("v_mov_b32 v167, 0" ::: "v167")

// the hand-scheduled int32 cell's floor table (CellI32F): entry k = "zero" in the frame of column k - G
extern "C" __global__ __launch_bounds__(256) void osw_floor_i32(uint2 *__restrict__ t, uint32_t n, uint32_t ge)
{
    OSW_SEARCH_SHAPED_VGPRS();
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) t[k] = make_uint2(k * ge, k * ge);
}

// keeps every CU issuing vector instructions for `ticks` of the 100-MHz clock, or until the host says stop (*stop != 0: a word in
// page-locked host memory, looked at by one lane per workgroup every ~15 us and passed on through LDS; null: never).  Every wave
// leaves by itself -- on the clock at the latest --, nothing waits for anything (osw_launch_spin: the warm-up of oswald_hip_init).
extern "C" __global__ __launch_bounds__(256) void osw_spin(uint32_t *sink, uint64_t ticks, const uint32_t *stop)
{
    OSW_SEARCH_SHAPED_VGPRS();
    __shared__ uint32_t quit;
    if (threadIdx.x == 0) quit = 0u;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = threadIdx.x, y = blockIdx.x, it = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int k = 0; k < 64; ++k) { x = x * 1664525u + y; y = (y ^ x) + 1013904223u; }
        if ((++it & 63u) == 0u && threadIdx.x == 0 && stop && *(const volatile uint32_t *)stop) *(volatile uint32_t *)&quit = 1u;
        if (*(volatile uint32_t *)&quit) break;
    }
    if (x == 0x12345678u && y == 0x9abcdef0u) *sink = x; // (never: keeps the loop)
}

// the pad columns of `tiled` (and everything else a chunk's re-tile does not write): dummy residues; instead of the runtime's fill
// kernel, which is not search-shaped
extern "C" __global__ __launch_bounds__(256) void osw_fill16(uint4 *__restrict__ p, uint32_t word, size_t n16)
{
    OSW_SEARCH_SHAPED_VGPRS();
    const uint4 v = make_uint4(word, word, word, word);
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) p[k] = v;
}


// page-locked host memory -> device (or back), by a kernel that reads / writes the host buffer in place (the small inputs of a query set:
// oswald_hip.cpp::sync_queries); n16 = 16-byte words
extern "C" __global__ __launch_bounds__(256) void osw_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    OSW_SEARCH_SHAPED_VGPRS(); // (it runs on the upload stream beside searches: see "search-shaped" above)
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}

// Re-tile the reference's W-lane interleaved groups (a4: b[disp_g + j*W + l],
// reference host/src/sequences.c:479-498) into 128-sequence wave blocks:
// tiled[(col4_off*4 + j)*64 + lane] = uint16 {8*residue j of seq 2*lane, 8*residue j of
// seq 2*lane+1}.  Missing groups / columns past a group's length read as the
// dummy residue 23, exactly the reference's padding value.  The buffer is
// pre-filled with the dummy residue: OSW_TILED_PAD_GROUPS all-dummy groups lie
// in front of the first block and behind every block (the columns the lane
// groups of the search kernels warm up, prefetch and drain through).
// four residue codes in the bytes of a word, each masked to five bits and 24..31 replaced by 23
static __device__ __forceinline__ uint32_t osw_clamp_codes4(uint32_t v)
{
    v &= 0x1f1f1f1fu;
    const uint32_t m = ((v >> 3) & (v >> 4) & 0x01010101u) * 0x1fu; // 0x1f in the bytes that hold 24..31
    return (v & ~m) | (m & 0x17171717u);
}
// four codes 0..23 in the bytes of a word -> their relabelled codes (sw_kernels.h: OSW_RELABEL): a 24-byte table looked up
// eight entries at a time by v_perm_b32 (selector = code & 7), the right third picked by the code's upper bits
static __device__ __forceinline__ uint32_t osw_relabel_codes4(uint32_t v)
{
    const uint32_t sel = v & 0x07070707u;
    const uint32_t t0 = __builtin_amdgcn_perm(OSW_RELABEL_W1, OSW_RELABEL_W0, sel), t1 = __builtin_amdgcn_perm(OSW_RELABEL_W3, OSW_RELABEL_W2, sel),
                   t2 = __builtin_amdgcn_perm(OSW_RELABEL_W5, OSW_RELABEL_W4, sel);
    const uint32_t m1 = ((v >> 3) & 0x01010101u) * 0xffu, m2 = ((v >> 4) & 0x01010101u) * 0xffu; // bytes that hold 8..15 / 16..23
    return (t0 & ~(m1 | m2)) | (t1 & m1) | (t2 & m2);
}
// original code of a profile slot 0..31
static __device__ __forceinline__ uint32_t osw_original_code(uint32_t slot)
{
    const uint32_t w = slot < 4 ? OSW_ORIGINAL_W0 : slot < 8 ? OSW_ORIGINAL_W1 : slot < 12 ? OSW_ORIGINAL_W2 : slot < 16 ? OSW_ORIGINAL_W3
                     : slot < 20 ? OSW_ORIGINAL_W4 : slot < 24 ? OSW_ORIGINAL_W5 : 0x17171717u;
    return (w >> (8u * (slot & 3u))) & 0xffu;
}

// The all-dummy columns around a block -- OSW_TILED_PAD_GROUPS groups behind it (and, block 0, in front of it; the last block,
// the readable tail too): what the lane groups of the search kernels warm up, prefetch and drain through.  Written by the block's
// own re-tile workgroup since round 5: until then a fill kernel wrote the WHOLE buffer with dummies first (a second pass over
// 128 MiB per chunk, and one more kernel that wants wave slots while the search before is draining).
static __device__ __forceinline__ void osw_write_pads(uint16_t *__restrict__ tiled, const OswBlock &blk, uint32_t B, uint32_t nblocks, uint32_t t)
{
    const uint32_t D = OSW_DUMMY_CODE8 * 0x01010101u;
    const uint4 v = make_uint4(D, D, D, D);
    uint4 *behind = (uint4 *)(tiled + ((size_t)blk.col4_off + blk.ncols4_alloc) * 4 * 64); // 8 x 16 B per column
    const uint32_t n_behind = (OSW_TILED_PAD_GROUPS + (B + 1 == nblocks ? OSW_TILED_TAIL_GROUPS : 0)) * 4 * 8;
    for (uint32_t k = t; k < n_behind; k += 256) behind[k] = v;
    if (B == 0) {
        uint4 *front = (uint4 *)tiled;
        for (uint32_t k = t; k < blk.col4_off * 4 * 8; k += 256) front[k] = v;
    }
}

extern "C" __global__ __launch_bounds__(256) void osw_retile(const uint8_t *__restrict__ b, const uint16_t *__restrict__ n,
                                                              const uint32_t *__restrict__ disp, uint32_t ngroups, uint32_t W,
                                                              const OswBlock *__restrict__ blocks, uint16_t *__restrict__ tiled)
{
    OSW_SEARCH_SHAPED_VGPRS();
    const uint32_t B = blockIdx.x;
    const OswBlock blk = blocks[B];
    osw_write_pads(tiled, blk, B, gridDim.x, threadIdx.x);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t gpb = 128 / W;                  // groups per block
    const uint32_t g = B * gpb + (2 * lane) / W;   // this lane's group
    const uint32_t l = (2 * lane) % W;             // lane pair inside the group
    const bool have = g < ngroups;
    const uint32_t ng = have ? n[g] : 0;
    const uint8_t *src = b + (have ? disp[g] : 0) + l;
    const uint32_t ncols = blk.ncols4_alloc * 4;
    for (uint32_t j = wv; j < ncols; j += 4) {
        uint32_t r0 = 23, r1 = 23;
        if (j < ng) {
            r0 = src[(size_t)j * W];
            r1 = src[(size_t)j * W + 1];
        }
        // codes >= 24 cannot come from the reference's preprocessing (0..23) and score like the dummy, 23, in every matrix
        // oswald_hip_set_scoring accepts: they are stored as 23 (the single-query kernels' profile holds 24 entries per row-block)
        r0 &= 31u; r1 &= 31u;
        const uint32_t two = osw_relabel_codes4((r0 < 24u ? r0 : 23u) | ((r1 < 24u ? r1 : 23u) << 8)); // (relabelled: sw_kernels.h)
        tiled[((size_t)blk.col4_off * 4 + j) * 64 + lane] = (uint16_t)((two & 0xffffu) << 3);
    }
}

// The same for the reference's own layout, W = 16 (round 4): vectorised, and with the live extents computed on the way
// (below: osw_block_extent walks every lane back from the end of the block with dependent loads -- 0.86 ms for a
// 100 000-sequence chunk, twice the time of the re-tile itself and half of a single-query search).  A thread takes
// one COLUMN of one 16-sequence group: 16 residues = one 16-byte load, 8 lanes x {8 * code, 8 * code} = one 16-byte
// store (the byte order of a group's column IS the lane order of the block's row); threads of a wave cover 32
// consecutive columns of two neighbouring groups, so loads run over 512 contiguous bytes and stores fill whole
// 32-byte sectors.  Every thread keeps, for the 8 lanes of its group, the last column with a real residue; the
// workgroup combines them in LDS and writes sub_cols / blocks[B].ncols4 exactly as osw_block_extent does.
extern "C" __global__ __launch_bounds__(256) void osw_retile16(const uint8_t *__restrict__ b, const uint16_t *__restrict__ n,
                                                                const uint32_t *__restrict__ disp, uint32_t ngroups,
                                                                OswBlock *__restrict__ blocks, uint16_t *__restrict__ tiled, uint16_t *__restrict__ sub_cols)
{
    __shared__ uint32_t lane_n[64];
    OSW_SEARCH_SHAPED_VGPRS();
    const uint32_t B = blockIdx.x, t = threadIdx.x;
    const OswBlock blk = blocks[B];
    if (t < 64) lane_n[t] = 0;
    osw_write_pads(tiled, blk, B, gridDim.x, t);
    __syncthreads();
    const uint32_t gi = t >> 5, jl = t & 31;      // group of the block (0..7), column inside a run of 32
    const uint32_t g = B * 8 + gi;
    const bool have = g < ngroups;
    const uint32_t ng = have ? n[g] : 0;
    const uint8_t *src = b + (have ? disp[g] : 0);
    const uint32_t ncols = blk.ncols4_alloc * 4;
    const bool aligned = (((uintptr_t)src) & 15u) == 0; // (the caller's buffer and displacements are multiples of 16 in the reference's layout; anything else takes byte loads)
    uint4 *dst = (uint4 *)(tiled + (size_t)blk.col4_off * 4 * 64) + gi;    // + j * 8: row j is 128 B = 8 x 16 B
    uint32_t last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t D4 = 0x17171717u;              // four dummy residues (code 23)
    for (uint32_t j = jl; j < ncols; j += 32) {
        uint4 v = make_uint4(D4, D4, D4, D4);
        if (j < ng) {
            if (aligned) v = *(const uint4 *)(src + (size_t)j * 16);
            else {
                const uint8_t *q = src + (size_t)j * 16;
                uint32_t w[4];
                for (int k = 0; k < 4; ++k) w[k] = (uint32_t)q[4 * k] | ((uint32_t)q[4 * k + 1] << 8) | ((uint32_t)q[4 * k + 2] << 16) | ((uint32_t)q[4 * k + 3] << 24);
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // word k = lanes 2k, 2k+1 of the group (two residues each)
            if ((w[k] & 0xffffu) != 0x1717u) last[2 * k] = j + 1;
            if ((w[k] >> 16) != 0x1717u) last[2 * k + 1] = j + 1;
        }
        // codes >= 24 (bits 4 and 3 both set) cannot come from the reference's preprocessing (0..23); they are stored as 23, see osw_retile
        // ... and relabelled by frequency (sw_kernels.h: OSW_RELABEL)
        dst[(size_t)j * 8] = make_uint4(osw_relabel_codes4(osw_clamp_codes4(v.x)) << 3, osw_relabel_codes4(osw_clamp_codes4(v.y)) << 3,
                                        osw_relabel_codes4(osw_clamp_codes4(v.z)) << 3, osw_relabel_codes4(osw_clamp_codes4(v.w)) << 3);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (last[k]) atomicMax(&lane_n[gi * 8 + k], last[k]);
    __syncthreads();
    if (t < 127) {
        const uint32_t lg = 31u - (uint32_t)__builtin_clz(t + 1u), sigma = t + 1u - (1u << lg), gl = 64u >> lg;
        uint32_t mx = 0;
        for (uint32_t k = 0; k < gl; ++k) mx = lane_n[sigma * gl + k] > mx ? lane_n[sigma * gl + k] : mx;
        sub_cols[(size_t)B * 128 + t] = (uint16_t)mx;
        if (t == 0) blocks[B].ncols4 = (mx + 3) / 4;
    }
}

// Live extents.  Columns past a sequence's end hold the dummy residue, which scores 0 against
// everything (reference submat.c: column 23 is zero) and therefore cannot raise any maximum: an item
// stops at the longest sequence of ITS sub-block (the sequences are sorted by length, so this trims
// almost all padding).  sub_cols[B*128 + (G-1) + sigma] = columns up to the last real residue of
// sub-block sigma at geometry G (a binary heap over the 64 lanes: G = 1 is the whole block, G = 64 a
// single lane); blocks[B].ncols4 = 4-column groups of the whole block.
extern "C" __global__ __launch_bounds__(128) void osw_block_extent(OswBlock *blocks, const uint16_t *__restrict__ tiled, uint16_t *__restrict__ sub_cols)
{
    __shared__ uint32_t lane_n[64];
    OSW_SEARCH_SHAPED_VGPRS();
    const uint32_t B = blockIdx.x;
    const uint32_t t = threadIdx.x;
    const OswBlock blk = blocks[B];
    const uint16_t dummy = (uint16_t)(OSW_DUMMY_CODE8 | (OSW_DUMMY_CODE8 << 8));
    if (t < 64) {
        uint32_t n = blk.ncols4_alloc * 4;
        while (n > 0 && tiled[((size_t)blk.col4_off * 4 + n - 1) * 64 + t] == dummy) --n;
        lane_n[t] = n;
    }
    __syncthreads();
    if (t < 127) {
        const uint32_t lg = 31u - (uint32_t)__builtin_clz(t + 1u), sigma = t + 1u - (1u << lg), gl = 64u >> lg;
        uint32_t mx = 0;
        for (uint32_t k = 0; k < gl; ++k) mx = lane_n[sigma * gl + k] > mx ? lane_n[sigma * gl + k] : mx;
        sub_cols[(size_t)B * 128 + t] = (uint16_t)mx;
        if (t == 0) blocks[B].ncols4 = (mx + 3) / 4;
    }
}

// Query profile in the layout the search kernels read:
// prof[(prof_off[q] + i/4)*32 + code] = 4 x int16 = S(a[i..i+3], code);
// rows past the query end and query codes >= 24 score 0 (the reference's
// 24th matrix row is all zero, submat.c).  add: a constant added to every entry
// (the column-frame int16 cell wants S + ge, also in the zero rows).
extern "C" __global__ __launch_bounds__(256) void osw_build_profile(const uint8_t *__restrict__ a, const uint32_t *__restrict__ a_disp,
                                                                     const uint16_t *__restrict__ qlen, const uint32_t *__restrict__ prof_off,
                                                                     const int8_t *__restrict__ submat, uint32_t nq, int add,
                                                                     uint2 *__restrict__ prof, uint4 *__restrict__ prof_seq)
{
    const uint32_t q = blockIdx.y;
    if (q >= nq) return;
    const uint32_t m = qlen[q];
    const uint32_t nrb = (m + 3) / 4 > 0 ? (m + 3) / 4 : 1;
    const uint8_t *aq = a + a_disp[q];
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < nrb * 32; e += gridDim.x * blockDim.x) {
        const uint32_t rb = e >> 5, code = e & 31;
        short s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = rb * 4 + k;
            int v = 0;
            if (i < m) {
                const uint32_t ai = aq[i];
                if (ai < 24) v = submat[ai * 32 + osw_original_code(code)]; // `code` is the SLOT: the relabelled code the re-tiled database holds
            }
            s[k] = (short)(v + add); // add: the column-frame cell wants S + ge
        }
        uint2 o;
        o.x = (uint32_t)(uint16_t)s[0] | ((uint32_t)(uint16_t)s[1] << 16);
        o.y = (uint32_t)(uint16_t)s[2] | ((uint32_t)(uint16_t)s[3] << 16);
        prof[(size_t)(prof_off[q] + rb) * 32 + code] = o;
        // the sequence-pair cell's form of the same entry: a 32-bit word per row, {low half: S, high half: 1} (OSW_TADD_MAD)
        if (prof_seq && code < OSW_SEQ_CODES) {
            uint4 w;
            w.x = (uint32_t)(uint16_t)s[0] | 0x10000u;
            w.y = (uint32_t)(uint16_t)s[1] | 0x10000u;
            w.z = (uint32_t)(uint16_t)s[2] | 0x10000u;
            w.w = (uint32_t)(uint16_t)s[3] | 0x10000u;
            prof_seq[(size_t)(prof_off[q] + rb) * OSW_SEQ_CODES + code] = w;
        }
    }
}

// Pair profile: prof_pair[(pair_off[p] + i/4)*32 + code] = 16 B = 4 rows x (S_A, S_B) int16 pairs.
// Built from the single-query profiles (rows past a query's end are already zero there, and
// a row-block past its last one reads as zero).
extern "C" __global__ __launch_bounds__(256) void osw_build_pair_profile(const uint2 *__restrict__ prof, const uint32_t *__restrict__ prof_off,
                                                                          const uint16_t *__restrict__ qlen, const uint32_t *__restrict__ pair_q,
                                                                          const uint32_t *__restrict__ pair_off, const uint16_t *__restrict__ pair_len,
                                                                          uint32_t npairs, uint32_t intsum, uint4 *__restrict__ prof_pair)
{
    const uint32_t pr = blockIdx.y;
    if (pr >= npairs) return;
    const uint32_t qa = pair_q[2 * pr], qb = pair_q[2 * pr + 1];
    const uint32_t nrb = (pair_len[pr] + 3u) / 4u > 0 ? (pair_len[pr] + 3u) / 4u : 1u;
    const uint32_t na = (qlen[qa] + 3u) / 4u, nb = (qlen[qb] + 3u) / 4u;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < nrb * 32; e += gridDim.x * blockDim.x) {
        const uint32_t rb = e >> 5, code = e & 31;
        const uint2 A = rb < na ? prof[(size_t)(prof_off[qa] + rb) * 32 + code] : make_uint2(0, 0);
        const uint2 Bv = rb < nb ? prof[(size_t)(prof_off[qb] + rb) * 32 + code] : make_uint2(0, 0);
        // (S_A, S_B) as two int16 halves -- or, intsum, as the 32-bit integer S_A + 65536 * S_B, which a plain
        // 32-bit add turns into the right sums in both halves of a non-negative packed pair (OSW_ADD_32)
        auto pack = [&](uint32_t a16, uint32_t b16) {
            return intsum ? (uint32_t)((int32_t)(int16_t)a16 + (int32_t)(int16_t)b16 * 65536) : (a16 & 0xffffu) | (b16 << 16);
        };
        uint4 o;
        o.x = pack(A.x & 0xffffu, Bv.x & 0xffffu);
        o.y = pack(A.x >> 16, Bv.x >> 16);
        o.z = pack(A.y & 0xffffu, Bv.y & 0xffffu);
        o.w = pack(A.y >> 16, Bv.y >> 16);
        prof_pair[(size_t)(pair_off[pr] + rb) * 32 + code] = o;
    }
}

// ---------------------------------------------------------------------------
// Top-r selection with the reference's tie rule (reference host/src/utils.c:
// 3-86 sorts descending and, on equal scores, puts the LATER database index
// first).  key = score << 32 | index, so "descending key" is exactly that
// order.  Two levels: every (query, partition of the score row) workgroup picks
// its r largest keys in r rounds of a coalesced max-scan (round k takes the
// largest key below the key of round k-1), then one workgroup per query picks
// the r largest of the partitions' candidates the same way.  Keys are stored
// as (key << 1) | 1, 0 = none (scores are >= 0 and < 2^31: the shift is lossless).
// ---------------------------------------------------------------------------
template <class KeyAt>
static __device__ __forceinline__ void osw_select_top(KeyAt key_at, uint32_t n, uint32_t r, unsigned long long *out /* r tagged keys */)
{
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long bound_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long bound = ~0ull; // tagged keys below this one are still available
    for (uint32_t k = 0; k < r; ++k) {
        unsigned long long v = 0;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned long long key = key_at(i); // tagged, 0 = none
            if (key < bound && key > v) v = key;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(v, off);
            v = o > v ? o : v;
        }
        if (lane == 0) red[wv] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = 0;
            for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) m = red[w] > m ? red[w] : m;
            bound_s = m;
            out[k] = m;
        }
        __syncthreads();
        bound = bound_s ? bound_s : 0ull; // nothing left: every later round finds nothing either
        __syncthreads();
    }
}

// key of score-row element i: index-in-chunk i, or -- for the context-level lists -- its DATABASE index (first + i, or
// map[i] for a chunk that is not one contiguous run of the database): the selection then breaks ties by database index
// whatever the order of the map.
extern "C" __global__ __launch_bounds__(256) void osw_topr_part(const int32_t *__restrict__ scores, uint32_t score_stride, uint32_t nvalid,
                                                                 uint32_t r, uint32_t part, const uint32_t *__restrict__ index_map, uint32_t first_index,
                                                                 unsigned long long *__restrict__ cand)
{
    const uint32_t q = blockIdx.x, p = blockIdx.y, P = gridDim.y;
    const uint32_t i0 = p * part, n = i0 < nvalid ? (nvalid - i0 < part ? nvalid - i0 : part) : 0;
    const int32_t *row = scores + (size_t)q * score_stride + i0;
    if (index_map) {
        const uint32_t *map = index_map + i0;
        osw_select_top([&](uint32_t i) { return ((((unsigned long long)(uint32_t)row[i] << 32) | map[i]) << 1) | 1ull; }, n, r, cand + ((size_t)q * P + p) * r);
    } else {
        const uint32_t base = first_index + i0;
        osw_select_top([&](uint32_t i) { return ((((unsigned long long)(uint32_t)row[i] << 32) | (base + i)) << 1) | 1ull; }, n, r, cand + ((size_t)q * P + p) * r);
    }
}

extern "C" __global__ __launch_bounds__(256) void osw_topr_merge(const unsigned long long *__restrict__ cand, uint32_t ncand, uint32_t r,
                                                                  int32_t *__restrict__ out_scores, uint32_t *__restrict__ out_index)
{
    __shared__ unsigned long long best[1024];
    const uint32_t q = blockIdx.x;
    const unsigned long long *c = cand + (size_t)q * ncand;
    osw_select_top([&](uint32_t i) { return c[i]; }, ncand, r, best);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < r; k += blockDim.x) {
        const unsigned long long m = best[k];
        out_scores[(size_t)q * r + k] = (m & 1ull) ? (int32_t)((m >> 1) >> 32) : -1;
        out_index[(size_t)q * r + k] = (m & 1ull) ? (uint32_t)((m >> 1) & 0xffffffffull) : 0xffffffffu;
    }
}

// Fold lists of tagged keys into one: query q's candidates are `ncand` keys at cand + q * ncand (may be 0) and L lists
// of r keys each at lists + l * list_stride + q * r -- a device's running list, the lists of the GPUs an all-gather
// brought together, ...  The r largest DISTINCT keys go to out + q * r (a key that appears twice -- a chunk searched
// twice -- counts once: every round takes the largest key below the one before).  `out` must not alias an input.
extern "C" __global__ __launch_bounds__(256) void osw_topr_fold(const unsigned long long *__restrict__ cand, uint32_t ncand,
                                                                 const unsigned long long *__restrict__ lists, uint32_t L, uint64_t list_stride,
                                                                 uint32_t r, unsigned long long *__restrict__ out)
{
    const uint32_t q = blockIdx.x;
    const unsigned long long *c = cand + (size_t)q * ncand, *l0 = lists + (size_t)q * r;
    osw_select_top([&](uint32_t i) { return i < ncand ? c[i] : l0[(size_t)((i - ncand) / r) * list_stride + (i - ncand) % r]; }, ncand + L * r, r,
                   out + (size_t)q * r);
}

// tagged keys -> (score, index); the first r_out of every r keys; empty slots: score -1, index 0xffffffff
extern "C" __global__ __launch_bounds__(256) void osw_topr_untag(const unsigned long long *__restrict__ keys, uint32_t nq, uint32_t r, uint32_t r_out,
                                                                  int32_t *__restrict__ out_scores, uint32_t *__restrict__ out_index)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nq * r_out) return;
    const unsigned long long m = keys[(size_t)(k / r_out) * r + k % r_out];
    out_scores[k] = (m & 1ull) ? (int32_t)((m >> 1) >> 32) : -1;
    out_index[k] = (m & 1ull) ? (uint32_t)((m >> 1) & 0xffffffffull) : 0xffffffffu;
}

// ---------------------------------------------------------------------------
// Host-side launchers (the kernels are only launched from this translation
// unit; oswald_hip.cpp calls these).
// ---------------------------------------------------------------------------
#define OSW_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t osw_launch_pk16(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_pk16, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_pk16q(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_pk16q, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_pk16qt(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_pk16qt, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_s16qt(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_s16qt, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_s16(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_s16, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_s16q(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_s16q, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_build_pair_profile(const uint2 *prof, const uint32_t *prof_off, const uint16_t *qlen, const uint32_t *pair_q,
                                         const uint32_t *pair_off, const uint16_t *pair_len, uint32_t npairs, uint32_t max_rowblocks,
                                         bool intsum, uint4 *prof_pair, hipStream_t s)
{
    if (npairs == 0) return hipSuccess;
    uint32_t gx = (max_rowblocks * 32 + 255) / 256;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL(osw_build_pair_profile, dim3(gx, npairs), dim3(256), 0, s, prof, prof_off, qlen, pair_q, pair_off, pair_len, npairs, intsum ? 1u : 0u, prof_pair);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_q8(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_q8, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_build_pair_profile8(const uint2 *prof, const uint32_t *prof_off, const uint16_t *qlen, const uint32_t *pair_q,
                                          const uint32_t *pair_off, const uint16_t *pair_len, uint32_t npairs, uint32_t max_rowblocks,
                                          int bias, uint2 *prof_pair8, hipStream_t s)
{
    if (npairs == 0) return hipSuccess;
    uint32_t gx = (max_rowblocks * 32 + 255) / 256;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL(osw_build_pair_profile8, dim3(gx, npairs), dim3(256), 0, s, prof, prof_off, qlen, pair_q, pair_off, pair_len, npairs, bias, prof_pair8);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_i32(const OswSearchArgs &a, uint32_t grid, hipStream_t s)
{
    hipLaunchKernelGGL(osw_sw_i32, dim3(grid), dim3(OSW_WG_THREADS), 0, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

// the re-run of the int32 queue: `regions` spill regions exist on the device (one per wave of a DP launch, two launches)
hipError_t osw_launch_i32r(const OswSearchArgs &a, uint32_t regions, hipStream_t s)
{
    const size_t lds = (size_t)OSW_I32R_WAVES * OSW_I32R_SLICE * sizeof(uint2);
    // (per launch: the attribute belongs to the function ON THE CURRENT DEVICE; a context may drive several)
    const hipError_t attr = hipFuncSetAttribute((const void *)osw_sw_i32r, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    uint32_t grid = regions / OSW_I32R_WAVES;
    if (grid > 512u) grid = 512u;
    if (grid == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(osw_sw_i32r, dim3(grid), dim3(OSW_I32R_WAVES * 64), lds, s, a);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

// dynamic LDS that brings a workgroup of `kern` up to the LDS allocation of a search workgroup (see "search-shaped" above)
static size_t osw_shape_lds(const void *kern)
{
    hipFuncAttributes fs, fk;
    if (hipFuncGetAttributes(&fs, (const void *)osw_sw_s16q) != hipSuccess || hipFuncGetAttributes(&fk, kern) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const size_t want = fs.sharedSizeBytes, have = fk.sharedSizeBytes;
    const size_t dyn = want > have ? want - have : 0;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); // (per function and device: cheap, and the device may have changed)
    return dyn;
}

hipError_t osw_launch_fill(void *p, uint8_t byte, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if ((bytes & 15u) || ((uintptr_t)p & 15u)) return hipMemsetAsync(p, byte, bytes, s); // (never: the library's buffers and sizes are multiples of 16)
    const size_t n16 = bytes / 16;
    const uint32_t grid = (uint32_t)std::min<size_t>((n16 + 255) / 256, 768u * 4u);
    hipLaunchKernelGGL(osw_fill16, dim3(grid), dim3(256), osw_shape_lds((const void *)osw_fill16), s, (uint4 *)p, (uint32_t)byte * 0x01010101u, n16);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_spin(uint32_t *sink, uint32_t grid, double ms, const uint32_t *stop_host_pinned, hipStream_t s)
{
    hipLaunchKernelGGL(osw_spin, dim3(grid), dim3(256), osw_shape_lds((const void *)osw_spin), s, sink, (uint64_t)(ms * 1.0e5), stop_host_pinned);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_floor_i32(uint2 *t, uint32_t n, uint32_t ge, hipStream_t s)
{
    if (n == 0 && t) return hipSuccess; // (null, 0: the first launch of the kernel, at bring-up)
    hipLaunchKernelGGL(osw_floor_i32, dim3(64), dim3(256), osw_shape_lds((const void *)osw_floor_i32), s, t, n, ge);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

static size_t osw_shape_lds(const void *kern);
hipError_t osw_launch_copy16(const void *src_host_pinned, void *dst, size_t bytes, hipStream_t s)
{
    const size_t n16 = (bytes + 15) / 16; // (both buffers are sized with slack beyond a multiple of 16)
    if (n16 == 0 && src_host_pinned) return hipSuccess; // (null, 0: the first launch of the kernel, at bring-up)
    hipLaunchKernelGGL(osw_copy16, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n16 + 255) / 256, 64))), dim3(256), osw_shape_lds((const void *)osw_copy16), s, (const uint4 *)src_host_pinned, (uint4 *)dst, n16);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

// first launches of the kernels a query set's bring-up and a search's top list use, on empty inputs (oswald_hip_init: before any clock)
hipError_t osw_warm_aux_kernels(hipStream_t s)
{
    hipLaunchKernelGGL(osw_copy16, dim3(1), dim3(256), 0, s, (const uint4 *)nullptr, (uint4 *)nullptr, (size_t)0);
    hipLaunchKernelGGL(osw_build_profile, dim3(1, 1), dim3(256), 0, s, (const uint8_t *)nullptr, (const uint32_t *)nullptr, (const uint16_t *)nullptr, (const uint32_t *)nullptr,
                       (const int8_t *)nullptr, 0u, 0, (uint2 *)nullptr, (uint4 *)nullptr);
    hipLaunchKernelGGL(osw_build_pair_profile, dim3(1, 1), dim3(256), 0, s, (const uint2 *)nullptr, (const uint32_t *)nullptr, (const uint16_t *)nullptr, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, (const uint16_t *)nullptr, 0u, 0u, (uint4 *)nullptr);
    hipLaunchKernelGGL(osw_build_pair_profile8, dim3(1, 1), dim3(256), 0, s, (const uint2 *)nullptr, (const uint32_t *)nullptr, (const uint16_t *)nullptr, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, (const uint16_t *)nullptr, 0u, 0, (uint2 *)nullptr);
    hipLaunchKernelGGL(osw_floor_i32, dim3(1), dim3(256), 0, s, (uint2 *)nullptr, 0u, 0u);
    hipLaunchKernelGGL(osw_topr_part, dim3(1, 1), dim3(256), 0, s, (const int32_t *)nullptr, 0u, 0u, 0u, 1u, (const uint32_t *)nullptr, 0u, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(osw_topr_fold, dim3(1), dim3(256), 0, s, (const unsigned long long *)nullptr, 0u, (const unsigned long long *)nullptr, 0u, (uint64_t)0, 0u, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(osw_topr_untag, dim3(1), dim3(256), 0, s, (const unsigned long long *)nullptr, 0u, 0u, 0u, (int32_t *)nullptr, (uint32_t *)nullptr);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_retile(const uint8_t *b, const uint16_t *n, const uint32_t *disp, uint32_t ngroups, uint32_t W,
                             OswBlock *blocks, uint32_t nblocks, uint16_t *tiled, uint16_t *sub_cols, hipStream_t s)
{
    if (nblocks == 0) return hipSuccess;
    if (W == 16) { // the reference's layout: one kernel re-tiles and finds the live extents
        hipLaunchKernelGGL(osw_retile16, dim3(nblocks), dim3(256), osw_shape_lds((const void *)osw_retile16), s, b, n, disp, ngroups, blocks, tiled, sub_cols);
        OSW_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(osw_retile, dim3(nblocks), dim3(256), osw_shape_lds((const void *)osw_retile), s, b, n, disp, ngroups, W, (const OswBlock *)blocks, tiled);
    OSW_LAUNCH_CHECK();
    hipLaunchKernelGGL(osw_block_extent, dim3(nblocks), dim3(128), osw_shape_lds((const void *)osw_block_extent), s, blocks, (const uint16_t *)tiled, sub_cols);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_build_profile(const uint8_t *a, const uint32_t *a_disp, const uint16_t *qlen, const uint32_t *prof_off,
                                    const int8_t *submat, uint32_t nq, uint32_t max_rowblocks, int add, uint2 *prof, uint4 *prof_seq, hipStream_t s)
{
    if (nq == 0) return hipSuccess;
    uint32_t gx = (max_rowblocks * 32 + 255) / 256;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL(osw_build_profile, dim3(gx, nq), dim3(256), 0, s, a, a_disp, qlen, prof_off, submat, nq, add, prof, prof_seq);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

// candidates buffer: osw_topr_cand_count(nvalid, r, nq) tagged keys
uint32_t osw_topr_parts(uint32_t nvalid)
{
    uint32_t P = nvalid / 4096;
    return P < 1 ? 1 : P > 64 ? 64 : P;
}

hipError_t osw_launch_topr(const int32_t *scores, uint32_t score_stride, uint32_t nvalid, uint32_t r, uint32_t nq,
                           unsigned long long *cand, int32_t *out_scores, uint32_t *out_index, hipStream_t s)
{
    if (nq == 0 || r == 0) return hipSuccess;
    const uint32_t P = osw_topr_parts(nvalid), part = (nvalid + P - 1) / P;
    hipLaunchKernelGGL(osw_topr_part, dim3(nq, P), dim3(256), 0, s, scores, score_stride, nvalid, r, part, (const uint32_t *)nullptr, 0u, cand);
    OSW_LAUNCH_CHECK();
    hipLaunchKernelGGL(osw_topr_merge, dim3(nq), dim3(256), 0, s, (const unsigned long long *)cand, P * r, r, out_scores, out_index);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

// the chunk's r best per query as tagged DATABASE keys, folded into the device's running list: run_out = top r of
// (run_in, chunk)
hipError_t osw_launch_topr_fold_chunk(const int32_t *scores, uint32_t score_stride, uint32_t nvalid, uint32_t r, uint32_t nq,
                                      const uint32_t *index_map, uint32_t first_index, unsigned long long *cand,
                                      const unsigned long long *run_in, unsigned long long *run_out, hipStream_t s)
{
    if (nq == 0 || r == 0) return hipSuccess;
    const uint32_t P = osw_topr_parts(nvalid), part = (nvalid + P - 1) / P;
    hipLaunchKernelGGL(osw_topr_part, dim3(nq, P), dim3(256), 0, s, scores, score_stride, nvalid, r, part, index_map, first_index, cand);
    OSW_LAUNCH_CHECK();
    hipLaunchKernelGGL(osw_topr_fold, dim3(nq), dim3(256), 0, s, (const unsigned long long *)cand, P * r, run_in, 1u, (uint64_t)0, r, run_out);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_topr_fold_lists(const unsigned long long *lists, uint32_t L, uint64_t list_stride, uint32_t r, uint32_t nq,
                                      unsigned long long *out, hipStream_t s)
{
    if (nq == 0 || r == 0) return hipSuccess;
    hipLaunchKernelGGL(osw_topr_fold, dim3(nq), dim3(256), 0, s, (const unsigned long long *)nullptr, 0u, lists, L, list_stride, r, out);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

// two lists of r keys per query, both laid out [nq][r]
hipError_t osw_launch_topr_fold_lists2(const unsigned long long *a, const unsigned long long *b, uint32_t r, uint32_t nq, unsigned long long *out, hipStream_t s)
{
    if (nq == 0 || r == 0) return hipSuccess;
    hipLaunchKernelGGL(osw_topr_fold, dim3(nq), dim3(256), 0, s, a, r, b, 1u, (uint64_t)0, r, out);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t osw_launch_topr_untag(const unsigned long long *keys, uint32_t nq, uint32_t r, uint32_t r_out, int32_t *out_scores, uint32_t *out_index,
                                 hipStream_t s)
{
    if (nq == 0 || r_out == 0) return hipSuccess;
    hipLaunchKernelGGL(osw_topr_untag, dim3((nq * r_out + 255) / 256), dim3(256), 0, s, keys, nq, r, r_out, out_scores, out_index);
    OSW_LAUNCH_CHECK();
    return hipSuccess;
}

int osw_occupancy_q8(int *blocks_per_cu)
{
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, osw_sw_q8, OSW_WG_THREADS, 0);
}

int osw_occupancy_pk16(int *blocks_per_cu)
{
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, osw_sw_pk16, OSW_WG_THREADS, 0);
}
