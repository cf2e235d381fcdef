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

#endif
