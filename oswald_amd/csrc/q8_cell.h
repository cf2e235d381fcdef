// oswald_amd/csrc/q8_cell.h -- the 8-bit cell of `cell_bits = 8` (BASELINE configs[2]: "int8 packed cells with
// int16 overflow re-run"; the reference's first pass is 16 x int8 saturating at 127, device/sw.cl:60-78,
// host/src/HybridSearch.c:1618-1633, and everything that reaches 127 is redone in int16, :1670-1680).
// Included by sw_kernels.hip (the osw_sw_q8 kernel) and by tools/oprate_q8.hip (its issue-rate microbenchmark).
//
// gfx950 has no packed 8-bit maximum and no packed 8-bit saturating add / subtract, so four 7-bit values ride in the
// four bytes of a register, bit 7 of every byte is a guard, and a maximum is a SWAR sequence of plain 32-bit
// instructions.  A lane works on a 2 x 2 tile: the two queries of a pair against its two sequences, bytes
// {A.s0, B.s0, A.s1, B.s1}; one pass covers what the packed-int16 query-pair kernel does in two.
//
// Round 3 formulation ("offset domain", 43 instead of 57 instructions per row, and none of them a slow one):
//   * every H, E, F is stored as  true value + c  with  c = max(gap open + gap extend, bias)  (bias = -min S: the
//     profile stores S + bias >= 0).  A value that stands for zero is c, so "max(., 0)" is a maximum with the
//     constant c -- needed once per cell, on the diagonal sum -- and every SUBTRACTION of a penalty is a plain 32-bit
//     subtract that cannot borrow across bytes:  H >= c >= go + ge,  max(E, H - go) >= c - go >= ge.  (A horizontal /
//     vertical gap state whose true value has dropped below zero keeps decaying inside [0, c): it can never win a
//     maximum against H >= c, exactly like the clamped state it stands for.)  The previous formulation clamped with
//     a 6-instruction saturating subtract in four places per cell.
//   * the maximum:  t = (a | G) - b  keeps its guard bit G = 0x80 exactly in the bytes where a >= b;
//     k = m - (m >> 7) with m = t & G is 0x7f there and 0 elsewhere;  max = b + (t & k).  Six instructions, five when
//     a already carries its guard -- so E, F and the running score are KEPT with the guard set: the "- ge" of the gap
//     recurrences and the "| G" merge into one add of the constant (G - go - ge) to H.
//   * only v_add_u32 / v_sub_u32 / v_and_b32 / v_or_b32 / v_lshrrev_b32 with VGPR (or inline-constant) operands:
//     these issue in ~2.1-2.5 cycles per wave instruction on a gfx950 SIMD, against 4.25 for anything VOP3 (v_bfi_b32,
//     v_perm_b32, v_pk_*) or with an SGPR / literal operand (profiles/r02_oprate_valu_issue.txt, r03_oprate_q8.txt).
//     The constants therefore live in VGPRs (laundered through empty asm so that the compiler keeps them there); the
//     one VOP3 left per row is the v_perm_b32 that pairs the two sequences' profile entries.
//   * a diagonal sum that reaches 128 sets the guard bit of its byte, which is OR-ed into a sticky flag: that
//     (query, sequence) has left the cell's range [0, 127 - c] at some point and is queued for the packed-int16 kernel;
//     one that was never flagged is exact.
// Between strips / lane groups / rounds (spill scratch, ds_bpermute hand-off) H and F travel as plain 7-bit offset
// values (guard clear); "zero" there is c in every byte (zero_bits).
#ifndef OSWALD_Q8_CELL_H
#define OSWALD_Q8_CELL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sw_kernels.h"

struct CellQ8 {
    typedef uint32_t T;
    // all replicated in the four bytes and held in VGPRs
    struct GapT {
        uint32_t go;     // gap open
        uint32_t ujoin;  // 0x80 - (gap open + gap extend): H + ujoin = (H - go - ge) with the guard set
        uint32_t bias;   // profile bias
        uint32_t cG;     // the offset c with the guard set
        uint32_t G, L;   // 0x80808080, 0x7f7f7f7f
    };
    static constexpr bool kFast = false;
    static constexpr uint32_t kFloorBits = 0;
    static constexpr bool kShifted = false;
    static constexpr int kRows = OSW_RMAX8;
    static constexpr int kLdsRows = OSW_LDS_ROWS8;
    static constexpr int kCodes = 32;
    static constexpr int kRowBytes = 64; // 32 codes x 2 queries x 1 byte
    typedef uint2 Entry; // one residue code: 4 rows x (S_A, S_B) bytes, each S + bias

    // host and device: the offset for a scoring system, or -1 if the 8-bit cells cannot run it
    static __host__ __device__ inline int offset_for(int bias, int go, int ge)
    {
        if (bias < 0 || go < 0 || ge < 0 || go > 127 || ge > 127) return -1;
        const int c = go + ge > bias ? go + ge : bias;
        return c <= 64 ? c : -1; // beyond that hardly a score would fit the remaining range
    }
    static __device__ __forceinline__ GapT make_gap(uint32_t go, uint32_t ge, uint32_t bias, uint32_t c)
    {
        GapT g;
        g.go = go * 0x01010101u;
        g.ujoin = (0x80u - go - ge) * 0x01010101u;
        g.bias = bias * 0x01010101u;
        g.cG = (c | 0x80u) * 0x01010101u;
        g.G = 0x80808080u;
        g.L = 0x7f7f7f7fu;
        // keep them in VGPRs: a VOP2 instruction with an SGPR or literal operand issues at half the rate
        asm volatile("" : "+v"(g.go), "+v"(g.ujoin), "+v"(g.bias), "+v"(g.cG), "+v"(g.G), "+v"(g.L));
        return g;
    }
    static __device__ __forceinline__ uint32_t zero_bits(const GapT &g) { return g.cG & g.L; } // "zero" between strips: c
    static __device__ __forceinline__ T score_init(const GapT &g) { return g.cG & g.L; }
    static __device__ __forceinline__ T from_bits(uint32_t x) { return x; }
    static __device__ __forceinline__ uint32_t to_bits(T x) { return x; }
    template <int R>
    static __device__ __forceinline__ void init_state(T (&D)[R], T (&E)[R], T &top_prev, const GapT &g)
    {
        const uint32_t c = g.cG & g.L;
#pragma unroll
        for (int r = 0; r < R; ++r) { D[r] = c - g.bias; E[r] = g.cG; } // D: H - bias (what the diagonal add wants); E: guard set
        top_prev = c;
    }
    // a >= b per byte (a with its guard set, b 7-bit): b + ((a - b) where a >= b, else 0) = max(a, b), guard clear
    static __device__ __forceinline__ uint32_t maxg(uint32_t aG, uint32_t b, uint32_t G)
    {
        const uint32_t t = aG - b, m = t & G, k = m - (m >> 7);
        return b + (t & k);
    }
    // the same, added to `base` instead of b: base + max(a - b, 0)
    static __device__ __forceinline__ uint32_t excess(uint32_t aG, uint32_t b, uint32_t G, uint32_t base)
    {
        const uint32_t t = aG - b, m = t & G, k = m - (m >> 7);
        return base + (t & k);
    }
    // running score between items / lane groups: low 7 bits of every byte the best offset score, bit 7 the sticky flag
    static __device__ __forceinline__ T vmax(T a, T b)
    {
        const uint32_t G = 0x80808080u, L = 0x7f7f7f7fu;
        return maxg((a & L) | G, b & L, G) | ((a | b) & G);
    }

    // One database column against R rows.  top_prev = H(i0-1, j-1), f = F(i0, j) in, F(i0+R, j) out, hl = H(i0+R-1, j),
    // all 7-bit offset values; D[r] = H(i0+r-1, j-1) - bias, E[r] with the guard set; score: see vmax.
    template <int R>
    static __device__ __forceinline__ void column(uint32_t base, uint32_t codes, int /*half*/, T (&D)[R], T (&E)[R], T top_prev, T &f, T &hl,
                                                  const GapT &g, const GapT & /*same*/, T &score)
    {
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const u32x2_ *ldsp;
        typedef __attribute__((address_space(3))) const char *ldsc;
        const uint32_t G = g.G, L = g.L, go = g.go, uj = g.ujoin, bias = g.bias, cG = g.cG;
        const ldsc l0 = (ldsc)(uintptr_t)(base + (codes & 0xffu)), l1 = (ldsc)(uintptr_t)(base + ((codes >> 8) & 0xffu));
        uint32_t diag = top_prev - bias;
        uint32_t fG = f | G, scG = score | G, fl = score;
#pragma unroll
        for (int rb = 0; rb < R / 4; ++rb) {
            const u32x2_ p0 = *(ldsp)(l0 + rb * 256), p1 = *(ldsp)(l1 + rb * 256);
            const uint32_t s[4] = {__builtin_amdgcn_perm(p1.x, p0.x, 0x05040100u), __builtin_amdgcn_perm(p1.x, p0.x, 0x07060302u),
                                   __builtin_amdgcn_perm(p1.y, p0.y, 0x05040100u), __builtin_amdgcn_perm(p1.y, p0.y, 0x07060302u)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rb * 4 + k;
                uint32_t y = diag + s[k];            // H(i-1,j-1) + S + c: at most 127 + 127, no carry between the bytes
                fl |= y;                             // guard bit set: out of range from here on (sticky)
                y &= L;
                uint32_t h = maxg(cG, y, G);         // max(., "zero")
                h = maxg(E[r], h, G);
                h = maxg(fG, h, G);
                const uint32_t u = h - go;           // H - gap open: >= c - go >= ge
                const uint32_t ug = h + uj;          // (H - go - ge) with the guard set
                E[r] = excess(E[r], u, G, ug);       // max(E, u) - ge, guard set
                fG = excess(fG, u, G, ug);
                scG = excess(scG, h, G, h + G);      // max(score, H), guard set
                if (r + 1 < R) { diag = D[r + 1]; D[r + 1] = h - bias; } else { hl = h; }
            }
        }
        f = fG & L;
        score = (scG & L) | (fl & G);
    }
};

// ---------------------------------------------------------------------------
// Round 4: the same cell, HAND-SCHEDULED (CellQ8F), for the column loop with fixed in-flight registers (sw_round_q8f
// in sw_kernels.hip).  The compiler's version of CellQ8::column came out at ~50 VALU instructions per row (register
// copies around the D[r+1] shuffle, three-operand adds that issue at half rate) and, at the 80 registers six waves per
// SIMD leave, with 30 spilled registers and scratch traffic inside the column loop.  Here a row is 39 instructions +
// the v_perm_b32 that pairs the two sequences' profile entries, every state register is updated in place, and nothing
// spills:
//   * a maximum whose result is wanted as a plain 7-bit value is a SELECT: with k = 0x7f in the bytes where a >= b,
//     max(a, b) = (a & k) | (b & ~k)  -- one v_bitop3_b32 (gfx950: an arbitrary three-input boolean function, and the
//     one VOP3 that issues at the rate of the plain VOP2 instructions, profiles/r03_oprate_q8.txt) instead of and + add:
//     five instructions per maximum, t = aG - b, m = t & G, s = m >> 7, k = m - s, select;
//   * the running score keeps its guard bit: score' = select(score, H + G, k) -- six with the add that sets H's guard;
//   * E' = (u - ge + G) + ((E - u) & k) and F' likewise stay and + add (six): their winner is not one of the operands;
//   * the diagonal sum of row r+1 is issued BEFORE row r overwrites D[r+1] (in place, no copies).
// F runs down the column in the fixed F register of the step (guard set inside the column, cleared for the hand-off).
// ---------------------------------------------------------------------------
#define OSW_Q8_SEL "bitop3:0xe4" /* (src0 & src2) | (src1 & ~src2) */
// one maximum step of the H chain: H = max(A, H) with A carrying its guard; t1..t3 scratch
#define OSW_Q8_HMAX(A, H)                                   \
    "v_sub_u32 %[t1], " A ", " H "\n\t"                     \
    "v_and_b32 %[t2], %[t1], %[G_]\n\t"                     \
    "v_lshrrev_b32 %[t3], 7, %[t2]\n\t"                     \
    "v_sub_u32 %[t2], %[t2], %[t3]\n\t"                     \
    "v_bitop3_b32 " H ", " A ", " H ", %[t2] " OSW_Q8_SEL "\n\t"
// X = (u - ge + G) + max(X - u, 0): the gap recurrences, X with its guard in and out
#define OSW_Q8_GAP(X)                                       \
    "v_sub_u32 %[t1], " X ", %[u_]\n\t"                     \
    "v_and_b32 %[t2], %[t1], %[G_]\n\t"                     \
    "v_lshrrev_b32 %[t3], 7, %[t2]\n\t"                     \
    "v_sub_u32 %[t2], %[t2], %[t3]\n\t"                     \
    "v_and_b32 %[t1], %[t1], %[t2]\n\t"                     \
    "v_add_u32 " X ", %[ug_], %[t1]\n\t"
// the body of a row after its diagonal sum x: flag, clamp at "zero", H chain, gaps, running score; H ends up in %[h_]
#define OSW_Q8_ROW_BODY(FREG)                               \
    "v_or_b32 %[fl_], %[fl_], %[x_]\n\t"                    \
    "v_and_b32 %[h_], %[x_], %[L_]\n\t"                     \
    OSW_Q8_HMAX("%[cG_]", "%[h_]")                          \
    OSW_Q8_HMAX("%[E_]", "%[h_]")                           \
    OSW_Q8_HMAX(FREG, "%[h_]")                              \
    "v_sub_u32 %[u_], %[h_], %[go_]\n\t"                    \
    "v_add_u32 %[ug_], %[h_], %[uj_]\n\t"                   \
    OSW_Q8_GAP("%[E_]")                                     \
    OSW_Q8_GAP(FREG)                                        \
    "v_add_u32 %[ug_], %[h_], %[G_]\n\t"                    \
    "v_sub_u32 %[t1], %[sc_], %[h_]\n\t"                    \
    "v_and_b32 %[t2], %[t1], %[G_]\n\t"                     \
    "v_lshrrev_b32 %[t3], 7, %[t2]\n\t"                     \
    "v_sub_u32 %[t2], %[t2], %[t3]\n\t"                     \
    "v_bitop3_b32 %[sc_], %[sc_], %[ug_], %[t2] " OSW_Q8_SEL "\n\t"

// Input registers of a column step of the hand-scheduled 8-bit loop: the same scheme as the packed-int16 kernels'
// (sw_kernels.hip, OSW_VH ...), at the top of the 80 registers six waves per SIMD leave; the compiler is given v0..v71
// (amdgpu_num_vgpr(72)), these eight are read and written by name inside asm text only.
#define OSW8_VH "v72"   // H of the row above (7-bit offset values): handed over at the end of a step
#define OSW8_VF "v73"   // F of the row above, then the F chain of the step (guard set inside the column), then handed on
#define OSW8_VC0 "v74"  // residues, even / odd column
#define OSW8_VC1 "v75"
#define OSW8_VLH0 "v76" // boundary row of the previous round, group 0, even / odd column
#define OSW8_VLF0 "v77"
#define OSW8_VLH1 "v78"
#define OSW8_VLF1 "v79"
#define OSW8_INFLIGHT "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define OSW8_COMPILER_VGPRS __attribute__((amdgpu_num_vgpr(72)))

struct CellQ8F : CellQ8 {
    static constexpr bool kFast = true;
    static constexpr bool kQ8Fast = true;

    // row r < R-1: x = diagonal sum of this row (consumed), on return that of the next row (computed from the OLD
    // D[r+1] = Dn, which then receives this row's H - bias)
    static __device__ __forceinline__ void row(uint32_t &x, uint32_t &Er, uint32_t &Dn, uint32_t s_next, uint32_t &sc, uint32_t &fl, const GapT &g)
    {
        uint32_t xn, t1, t2, t3, u, ug;
        asm volatile("v_add_u32 %[xn_], %[Dn_], %[sn_]\n\t"
                     "v_or_b32 %[fl_], %[fl_], %[x_]\n\t"
                     "v_and_b32 %[Dn_], %[x_], %[L_]\n\t"
                     OSW_Q8_HMAX("%[cG_]", "%[Dn_]")
                     OSW_Q8_HMAX("%[E_]", "%[Dn_]")
                     OSW_Q8_HMAX(OSW8_VF, "%[Dn_]")
                     "v_sub_u32 %[u_], %[Dn_], %[go_]\n\t"
                     "v_add_u32 %[ug_], %[Dn_], %[uj_]\n\t"
                     OSW_Q8_GAP("%[E_]")
                     OSW_Q8_GAP(OSW8_VF)
                     "v_add_u32 %[ug_], %[Dn_], %[G_]\n\t"
                     "v_sub_u32 %[t1], %[sc_], %[Dn_]\n\t"
                     "v_and_b32 %[t2], %[t1], %[G_]\n\t"
                     "v_lshrrev_b32 %[t3], 7, %[t2]\n\t"
                     "v_sub_u32 %[t2], %[t2], %[t3]\n\t"
                     "v_bitop3_b32 %[sc_], %[sc_], %[ug_], %[t2] " OSW_Q8_SEL "\n\t"
                     "v_sub_u32 %[Dn_], %[Dn_], %[b_]"
                     : [xn_] "=&v"(xn), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u_] "=&v"(u), [ug_] "=&v"(ug),
                       [E_] "+v"(Er), [Dn_] "+v"(Dn), [sc_] "+v"(sc), [fl_] "+v"(fl)
                     : [x_] "v"(x), [sn_] "v"(s_next), [G_] "v"(g.G), [L_] "v"(g.L), [cG_] "v"(g.cG), [go_] "v"(g.go), [uj_] "v"(g.ujoin), [b_] "v"(g.bias)
                     : OSW8_INFLIGHT);
        x = xn;
    }
    // last row of the strip: its H goes to hl (7-bit), for the hand-off / the spill
    static __device__ __forceinline__ void row_last(uint32_t x, uint32_t &Er, uint32_t &hl, uint32_t &sc, uint32_t &fl, const GapT &g)
    {
        uint32_t t1, t2, t3, u, ug;
        asm volatile("v_or_b32 %[fl_], %[fl_], %[x_]\n\t"
                     "v_and_b32 %[h_], %[x_], %[L_]\n\t"
                     OSW_Q8_HMAX("%[cG_]", "%[h_]")
                     OSW_Q8_HMAX("%[E_]", "%[h_]")
                     OSW_Q8_HMAX(OSW8_VF, "%[h_]")
                     "v_sub_u32 %[u_], %[h_], %[go_]\n\t"
                     "v_add_u32 %[ug_], %[h_], %[uj_]\n\t"
                     OSW_Q8_GAP("%[E_]")
                     OSW_Q8_GAP(OSW8_VF)
                     "v_add_u32 %[ug_], %[h_], %[G_]\n\t"
                     "v_sub_u32 %[t1], %[sc_], %[h_]\n\t"
                     "v_and_b32 %[t2], %[t1], %[G_]\n\t"
                     "v_lshrrev_b32 %[t3], 7, %[t2]\n\t"
                     "v_sub_u32 %[t2], %[t2], %[t3]\n\t"
                     "v_bitop3_b32 %[sc_], %[sc_], %[ug_], %[t2] " OSW_Q8_SEL
                     : [h_] "=&v"(hl), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u_] "=&v"(u), [ug_] "=&v"(ug),
                       [E_] "+v"(Er), [sc_] "+v"(sc), [fl_] "+v"(fl)
                     : [x_] "v"(x), [G_] "v"(g.G), [L_] "v"(g.L), [cG_] "v"(g.cG), [go_] "v"(g.go), [uj_] "v"(g.ujoin)
                     : OSW8_INFLIGHT);
    }

    // Four rows of (S + bias) bytes {A.s0, B.s0, A.s1, B.s1} from the two sequences' profile entries: two ds_read_b64
    // (kept single: ds_read2_b64's 32-bank addressing makes codes c and c+16 collide) + four v_perm_b32.
    struct Raw { unsigned int lo __attribute__((ext_vector_type(2))), hi __attribute__((ext_vector_type(2))); };
    template <int RB>
    static __device__ __forceinline__ void ld(uint32_t a_lo, uint32_t a_hi, Raw &r)
    {
        asm volatile("ds_read_b64 %0, %2 offset:%4\n\t"
                     "ds_read_b64 %1, %3 offset:%4"
                     : "=&v"(r.lo), "=&v"(r.hi)
                     : "v"(a_lo), "v"(a_hi), "i"(RB * 256)
                     : "memory", OSW8_INFLIGHT);
    }
    template <int Newest>
    static __device__ __forceinline__ void landed(Raw &r)
    {
        if constexpr (Newest == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.lo), "+v"(r.hi)::OSW8_INFLIGHT);
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r.lo), "+v"(r.hi)::OSW8_INFLIGHT);
    }
    static __device__ __forceinline__ void pair_up(const Raw &r, uint32_t (&s)[4])
    {
        s[0] = __builtin_amdgcn_perm(r.hi.x, r.lo.x, 0x05040100u);
        s[1] = __builtin_amdgcn_perm(r.hi.x, r.lo.x, 0x07060302u);
        s[2] = __builtin_amdgcn_perm(r.hi.y, r.lo.y, 0x05040100u);
        s[3] = __builtin_amdgcn_perm(r.hi.y, r.lo.y, 0x07060302u);
    }
    template <int R, int RB>
    struct Batch {
        static __device__ __forceinline__ void run(uint32_t a_lo, uint32_t a_hi, uint32_t (&D)[R], uint32_t (&E)[R], uint32_t &x, uint32_t &hl,
                                                   const GapT &g, uint32_t &sc, uint32_t &fl, uint32_t (&s)[4], Raw &r1)
        {
            uint32_t sn[4];
            if constexpr (RB + 1 < R / 4) {
                landed<0>(r1);
                pair_up(r1, sn);
                if constexpr (RB + 2 < R / 4) ld<RB + 2>(a_lo, a_hi, r1);
            }
            row(x, E[RB * 4 + 0], D[RB * 4 + 1], s[1], sc, fl, g);
            row(x, E[RB * 4 + 1], D[RB * 4 + 2], s[2], sc, fl, g);
            row(x, E[RB * 4 + 2], D[RB * 4 + 3], s[3], sc, fl, g);
            if constexpr (RB + 1 < R / 4) {
                row(x, E[RB * 4 + 3], D[RB * 4 + 4], sn[0], sc, fl, g);
                Batch<R, RB + 1>::run(a_lo, a_hi, D, E, x, hl, g, sc, fl, sn, r1);
            } else {
                row_last(x, E[RB * 4 + 3], hl, sc, fl, g);
            }
        }
    };
    // One database column against the R rows of the strip, inputs in register set P: residues {8*code of the lane's
    // first sequence, 8*code of the second} in bytes 0 / 1 of the set's C register, F(i0, j) as a 7-bit value in the
    // fixed F register (out: F(i0+R, j), 7-bit).  D[r] = H(i0+r-1, j-1) - bias (D[0] is only a name), E[r] with the
    // guard; top_prev = H(i0-1, j-1); hl = H(i0+R-1, j); sc = running score with the guard; fl = sticky flags.
    template <int R, int P>
    static __device__ __forceinline__ void column(uint32_t base, uint32_t (&D)[R], uint32_t (&E)[R], uint32_t top_prev, uint32_t &hl, const GapT &g,
                                                  uint32_t &sc, uint32_t &fl)
    {
        uint32_t a_lo, a_hi;
        if constexpr (P == 0)
            asm volatile("v_add_u32_sdwa %0, %2, " OSW8_VC0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_add_u32_sdwa %1, %2, " OSW8_VC0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                         "v_or_b32 " OSW8_VF ", " OSW8_VF ", %3"
                         : "=&v"(a_lo), "=&v"(a_hi) : "v"(base), "v"(g.G) : OSW8_INFLIGHT);
        else
            asm volatile("v_add_u32_sdwa %0, %2, " OSW8_VC1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_add_u32_sdwa %1, %2, " OSW8_VC1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                         "v_or_b32 " OSW8_VF ", " OSW8_VF ", %3"
                         : "=&v"(a_lo), "=&v"(a_hi) : "v"(base), "v"(g.G) : OSW8_INFLIGHT);
        Raw r0, r1;
        uint32_t s[4];
        ld<0>(a_lo, a_hi, r0);
        if constexpr (R / 4 > 1) { ld<1>(a_lo, a_hi, r1); landed<2>(r0); } else { landed<0>(r0); }
        pair_up(r0, s);
        uint32_t x = top_prev - g.bias + s[0];
        Batch<R, 0>::run(a_lo, a_hi, D, E, x, hl, g, sc, fl, s, r1);
        asm volatile("v_and_b32 " OSW8_VF ", " OSW8_VF ", %0" : : "v"(g.L) : OSW8_INFLIGHT); // F goes on as a 7-bit value
    }
};

#endif
