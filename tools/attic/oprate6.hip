// tools/oprate6.hip -- round 4: can the LDS pair up the two sequences of a lane instead of the VALU?
// The sequence-pair cell (one query against the two sequences of a lane: osw_sw_s16 / osw_sw_pk16) spends one
// v_perm_b32 per row on pairing the substitution scores of the lane's two residues (two ds_read_b64 per 4 rows + four
// v_perm_b32): 7.5 VALU instructions per row where the query-pair cell has 6.5.  gfx950 has 16-bit LDS loads that
// write ONE half of a register and keep the other (ds_read_u16_d16 / ds_read_u16_d16_hi): two of them per row put the
// pair together with no VALU instruction at all -- at four times the LDS instructions (8 per 4 rows instead of 2).
// This probe runs the column loop of the cell (48-row strip = 12 blocks of 4 rows, state updated in place, loads two
// blocks ahead, three waves per SIMD like the kernel: 168 VGPRs) on random residues in four forms:
//   perm    the kernel's: 2 ds_read_b64 + 4 v_perm_b32 per block (profile entries of 8 B: 4 rows of one residue code)
//   d16     8 ds_read_u16_d16[_hi] per block from the SAME profile layout (codes c and c+16 share a bank for 4-byte-class reads)
//   d16r    the same from a [row][32 codes] int16 layout (a row's 24 codes lie in 12 dwords: conflict-free)
//   none    no loads, no perms (the VALU floor of the 6.5-instruction row)
//   mix     the kernel's 2 ds_read_b64 + ONE 16-bit load per row that completes a pair in place, with 1 / 4 / 8 lane groups
// TIMING ONLY: on this part (SRAM-ECC) a d16 load zeroes the other half of its register instead of keeping it (LABNOTES,
// "What did not work" (1)), so the d16 forms do not compute the pairs they are priced for; the probe answers whether the
// LDS could carry the pairing if they did.  Result (profiles/r04_oprate6_lds_pairing.txt): only with one lane group.
// and reports core-clock cycles per row per SIMD (median / slowest / fastest SIMD of the chip), like tools/oprate4.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate6 tools/oprate6.hip ; run: tools/oprate6 [columns]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

struct Stamp { unsigned long long cyc, real, r0, r1; uint32_t hwid, pad; };

// fixed registers (the compiler gets v0..v147): profile buffers A / B, the perm variant's pair registers, temporary, F
#define PA0 "v150"
#define PA1 "v151"
#define PA2 "v152"
#define PA3 "v153"
#define PB0 "v154"
#define PB1 "v155"
#define PB2 "v156"
#define PB3 "v157"
#define PS0 "v148"
#define PS1 "v149"
#define VT "v159"
#define VF "v161"
#define FIXED "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v159", "v161"

#define ROWT(XN, DN, SN, X, E, PRE, HOOK, SCMAX)                        \
    PRE "v_pk_add_i16 " XN ", " DN ", " SN " clamp\n\t" HOOK            \
    "v_pk_maximum3_f16 " DN ", " X ", " E ", " VF "\n\t"                \
    "v_subrev_u32 " VT ", %[go_], " DN "\n\t" SCMAX                     \
    "v_pk_maximum3_f16 " E ", " E ", " VT ", %[fl_]\n\t"                \
    "v_pk_maximum3_f16 " VF ", " VF ", " VT ", %[fl_]\n\t"              \
    "v_subrev_u32 " VF ", %[ge_], " VF "\n\t"
#define SCRUN(A, B) "v_pk_maximum3_f16 %[sc_], %[sc_], " A ", " B "\n\t"
#define PERM_A(PS, HI, LO) "v_perm_b32 " PS ", " HI ", " LO ", %[sela_]\n\t"
#define PERM_B(PS, HI, LO) "v_perm_b32 " PS ", " HI ", " LO ", %[selb_]\n\t"

// --- perm form: buffer = {lo.x, lo.y, hi.x, hi.y}; loads of the block after next in front of row 1's body
#define LD_B64(LO, HI) "ds_read_b64 " LO ", %[a0_] offset:%[off_]\n\tds_read_b64 " HI ", %[a1_] offset:%[off_]\n\ts_waitcnt lgkmcnt(2)\n\t"
#define BLK_PERM(C1, C3, CLO, CHI, N0, N2)                                                                       \
    ROWT("%[xb_]", "%[D1_]", PS0, "%[x_]", "%[E0_]", "", PERM_A(PS1, C3, C1), "")                                \
    ROWT("%[x_]", "%[D2_]", PS1, "%[xb_]", "%[E1_]", "", PERM_B(PS0, C3, C1) LD_B64(CLO, CHI), SCRUN("%[D1_]", "%[D2_]")) \
    ROWT("%[xb_]", "%[D3_]", PS0, "%[x_]", "%[E2_]", "", PERM_A(PS1, N2, N0), "")                                \
    ROWT("%[x_]", "%[D4_]", PS1, "%[xb_]", "%[E3_]", "", PERM_B(PS0, N2, N0), SCRUN("%[D3_]", "%[D4_]"))

// --- d16 forms: buffer = the (first, second sequence) score pairs of rows 0..3; the register of row k is free behind
// the add of row k-1 and is loaded for the block after next right there; one counted wait per block
#define LD_D16(R, OFF) "ds_read_u16_d16 " R ", %[a0_] offset:" OFF "\n\tds_read_u16_d16_hi " R ", %[a1_] offset:" OFF "\n\t"
#define BLK_D16(C0, C1, C2, C3, N0)                                                                              \
    ROWT("%[xb_]", "%[D1_]", C1, "%[x_]", "%[E0_]", "", LD_D16(C1, "%[o1_]"), "")                                \
    ROWT("%[x_]", "%[D2_]", C2, "%[xb_]", "%[E1_]", "", LD_D16(C2, "%[o2_]"), SCRUN("%[D1_]", "%[D2_]"))         \
    ROWT("%[xb_]", "%[D3_]", C3, "%[x_]", "%[E2_]", "", LD_D16(C3, "%[o3_]"), "")                                \
    ROWT("%[x_]", "%[D4_]", N0, "%[xb_]", "%[E3_]", "s_waitcnt lgkmcnt(6)\n\t", LD_D16(C0, "%[o0_]"), SCRUN("%[D3_]", "%[D4_]"))

// --- mix form (the candidate): the two ds_read_b64 of the perm form put rows (0,1) / (2,3) of the lane's FIRST sequence into
// {c0, c1} and of its SECOND into {c2, c3}; one 16-bit load per row then completes a pair in place: row 0 = c0 with its high
// half overwritten by the second sequence's row 0, row 1 = c2 with its low half overwritten by the first sequence's row 1,
// rows 2 / 3 likewise in c1 / c3.  Per block: 2 ds_read_b64 + 4 ds_read_u16_d16[_hi], no v_perm_b32.  Order of use: c0 c2 c1 c3.
// The loads of the block after next go out behind the add that consumes the block's last register (row 2's), the wait for
// the next block's in front of row 3's add.
#define LD_MIX(C0, C1, C2, C3, CLO, CHI)                                                                          \
    "ds_read_b64 " CLO ", %[a0_] offset:%[o0_]\n\tds_read_b64 " CHI ", %[a1_] offset:%[o0_]\n\t"              \
    "ds_read_u16_d16_hi " C0 ", %[a1_] offset:%[o0_]\n\tds_read_u16_d16 " C2 ", %[a0_] offset:%[m1_]\n\t"     \
    "ds_read_u16_d16_hi " C1 ", %[a1_] offset:%[m2_]\n\tds_read_u16_d16 " C3 ", %[a0_] offset:%[m3_]\n\t"
#define BLK_MIX(C0, C1, C2, C3, CLO, CHI, N0)                                                                     \
    ROWT("%[xb_]", "%[D1_]", C2, "%[x_]", "%[E0_]", "", "", "")                                                  \
    ROWT("%[x_]", "%[D2_]", C1, "%[xb_]", "%[E1_]", "", "", SCRUN("%[D1_]", "%[D2_]"))                           \
    ROWT("%[xb_]", "%[D3_]", C3, "%[x_]", "%[E2_]", "", LD_MIX(C0, C1, C2, C3, CLO, CHI), "")                    \
    ROWT("%[x_]", "%[D4_]", N0, "%[xb_]", "%[E3_]", "s_waitcnt lgkmcnt(6)\n\t", "", SCRUN("%[D3_]", "%[D4_]"))

#define BLK_NONE(C1, C2, C3, N0)                                                                                 \
    ROWT("%[xb_]", "%[D1_]", "%[s1_]", "%[x_]", "%[E0_]", "", "", "")                                            \
    ROWT("%[x_]", "%[D2_]", "%[s2_]", "%[xb_]", "%[E1_]", "", "", SCRUN("%[D1_]", "%[D2_]"))                     \
    ROWT("%[xb_]", "%[D3_]", "%[s1_]", "%[x_]", "%[E2_]", "", "", "")                                            \
    ROWT("%[x_]", "%[D4_]", "%[s2_]", "%[xb_]", "%[E3_]", "", "", SCRUN("%[D3_]", "%[D4_]"))

#define STMT(TXT, RB, NEXTOFF)                                                                                   \
    asm volatile(TXT                                                                                             \
                 : [x_] "+v"(x), [xb_] "=&v"(xb), [E0_] "+v"(E[RB * 4]), [E1_] "+v"(E[RB * 4 + 1]), [E2_] "+v"(E[RB * 4 + 2]),  \
                   [E3_] "+v"(E[RB * 4 + 3]), [D1_] "+v"(D[RB * 4 + 1]), [D2_] "+v"(D[RB * 4 + 2]), [D3_] "+v"(D[RB * 4 + 3]),  \
                   [D4_] "+v"(D[RB * 4 + 4]), [sc_] "+v"(sc)                                                    \
                 : [a0_] "v"(a0), [a1_] "v"(a1), [off_] "i"(NEXTOFF * 256), [o0_] "i"(NEXTOFF * BS), [o1_] "i"(NEXTOFF * BS + RS), \
                   [o2_] "i"(NEXTOFF * BS + 2 * RS), [o3_] "i"(NEXTOFF * BS + 3 * RS), [m1_] "i"(NEXTOFF * 256 + 2), [m2_] "i"(NEXTOFF * 256 + 4), [m3_] "i"(NEXTOFF * 256 + 6), [ge_] "s"(ge), [go_] "s"(go), [fl_] "v"(fl), \
                   [sela_] "s"(0x05040100u), [selb_] "s"(0x07060302u), [s1_] "v"(s1), [s2_] "v"(s2)                \
                 : "memory", FIXED)

enum { V_PERM = 0, V_D16 = 1, V_D16R = 2, V_NONE = 3, V_MIX = 4 };

template <int V, int RB>
static __device__ __forceinline__ void block(uint32_t a0, uint32_t a1, uint32_t (&D)[49], uint32_t (&E)[48], uint32_t &x, uint32_t &sc, uint32_t ge,
                                             uint32_t go, uint32_t fl, uint32_t s1, uint32_t s2)
{
    // BS = bytes from a block's entry to the next block's, RS = bytes from a row to the next inside a block
    constexpr int BS = V == V_D16R ? 256 : 256, RS = V == V_D16R ? 64 : 2;
    constexpr int NEXT = (RB + 2) % 12; // (the last two blocks of a column load blocks 0 / 1 again: the stream never stops)
    uint32_t xb;
    if constexpr (V == V_PERM) {
        if constexpr ((RB & 1) == 0) STMT(BLK_PERM(PA1, PA3, "v[150:151]", "v[152:153]", PB0, PB2), RB, NEXT);
        else STMT(BLK_PERM(PB1, PB3, "v[154:155]", "v[156:157]", PA0, PA2), RB, NEXT);
    } else if constexpr (V == V_D16 || V == V_D16R) {
        if constexpr ((RB & 1) == 0) STMT(BLK_D16(PA0, PA1, PA2, PA3, PB0), RB, NEXT);
        else STMT(BLK_D16(PB0, PB1, PB2, PB3, PA0), RB, NEXT);
    } else if constexpr (V == V_MIX) {
        if constexpr ((RB & 1) == 0) STMT(BLK_MIX(PA0, PA1, PA2, PA3, "v[150:151]", "v[152:153]", PB0), RB, NEXT);
        else STMT(BLK_MIX(PB0, PB1, PB2, PB3, "v[154:155]", "v[156:157]", PA0), RB, NEXT);
    } else {
        STMT(BLK_NONE(0, 0, 0, 0), RB, NEXT);
    }
    if constexpr (RB + 1 < 12) block<V, RB + 1>(a0, a1, D, E, x, sc, ge, go, fl, s1, s2);
}

template <int V, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(148))) void probe(Stamp *out, uint32_t seed, int ncols)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // G = 1: every wave has a slice of its own (a wave item); G > 1: the lanes form G groups whose 48-row slices lie one
    // behind the other, each 8 bytes further on (fill_profile_slice), in ONE region that the four waves share (a workgroup item)
    for (uint32_t i = threadIdx.x; i < 7168; i += 256) lds[i] = 0x00010001u * ((i * 7 + 3) & 3);
    __syncthreads();
    const uint32_t base = G == 1 ? wave * 4096 : (lane / (64 / G)) * (3072 + 8), stride = V == V_D16R ? 2 : 8;
    uint32_t D[49], E[48], x = 0x04000400u, sc = 0x04000400u, fl = 0x04000400u, s1 = 0x00010002u, s2 = 0x00020001u;
    const uint32_t ge = 2, go = 10;
#pragma unroll
    for (int i = 0; i < 48; ++i) { D[i] = 0x04000400u + lane; E[i] = 0x04000400u; }
    D[48] = 0x04000400u;
    asm volatile("v_mov_b32 " VF ", %0\n\tv_mov_b32 " PS0 ", 0\n\tv_mov_b32 " PS1 ", 0\n\t"
                 "v_mov_b32 " PA0 ", 0\n\tv_mov_b32 " PA1 ", 0\n\tv_mov_b32 " PA2 ", 0\n\tv_mov_b32 " PA3 ", 0\n\t"
                 "v_mov_b32 " PB0 ", 0\n\tv_mov_b32 " PB1 ", 0\n\tv_mov_b32 " PB2 ", 0\n\tv_mov_b32 " PB3 ", 0" ::"v"(fl) : FIXED);
    uint32_t rng = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int col = 0; col < ncols; ++col) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t a0 = base + ((((rng >> 8) & 0xffu) * 20u) >> 8) * stride, a1 = base + ((((rng >> 20) & 0xffu) * 20u) >> 8) * stride;
        // head: the loads of blocks 0 and 1 (the blocks' own loads run two ahead), the first diagonal sum
        if constexpr (V == V_PERM) {
            asm volatile("ds_read_b64 v[150:151], %[a0_]\n\tds_read_b64 v[152:153], %[a1_]\n\t"
                         "ds_read_b64 v[154:155], %[a0_] offset:256\n\tds_read_b64 v[156:157], %[a1_] offset:256\n\ts_waitcnt lgkmcnt(2)\n\t" PERM_A(PS1, PA2, PA0)
                             PERM_B(PS0, PA2, PA0) "v_pk_add_i16 %[x_], %[tp_], " PS1 " clamp"
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [a1_] "v"(a1), [tp_] "v"(D[0]), [sela_] "s"(0x05040100u), [selb_] "s"(0x07060302u)
                         : "memory", FIXED);
        } else if constexpr (V == V_D16 || V == V_D16R) {
            constexpr int RS = V == V_D16R ? 64 : 2;
            asm volatile(LD_D16(PA0, "%[o0_]") LD_D16(PA1, "%[o1_]") LD_D16(PA2, "%[o2_]") LD_D16(PA3, "%[o3_]")
                         LD_D16(PB0, "%[p0_]") LD_D16(PB1, "%[p1_]") LD_D16(PB2, "%[p2_]") "s_waitcnt lgkmcnt(6)\n\t"
                         LD_D16(PB3, "%[p3_]") "v_pk_add_i16 %[x_], %[tp_], " PA0 " clamp"
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [a1_] "v"(a1), [tp_] "v"(D[0]), [o0_] "i"(0), [o1_] "i"(RS), [o2_] "i"(2 * RS), [o3_] "i"(3 * RS),
                           [p0_] "i"(256), [p1_] "i"(256 + RS), [p2_] "i"(256 + 2 * RS), [p3_] "i"(256 + 3 * RS)
                         : "memory", FIXED);
        } else if constexpr (V == V_MIX) {
            asm volatile("ds_read_b64 v[150:151], %[a0_]\n\tds_read_b64 v[152:153], %[a1_]\n\t"
                         "ds_read_u16_d16_hi " PA0 ", %[a1_]\n\tds_read_u16_d16 " PA2 ", %[a0_] offset:2\n\t"
                         "ds_read_u16_d16_hi " PA1 ", %[a1_] offset:4\n\tds_read_u16_d16 " PA3 ", %[a0_] offset:6\n\t"
                         "ds_read_b64 v[154:155], %[a0_] offset:256\n\tds_read_b64 v[156:157], %[a1_] offset:256\n\t"
                         "ds_read_u16_d16_hi " PB0 ", %[a1_] offset:256\n\tds_read_u16_d16 " PB2 ", %[a0_] offset:258\n\t"
                         "ds_read_u16_d16_hi " PB1 ", %[a1_] offset:260\n\tds_read_u16_d16 " PB3 ", %[a0_] offset:262\n\t"
                         "s_waitcnt lgkmcnt(6)\n\tv_pk_add_i16 %[x_], %[tp_], " PA0 " clamp"
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [a1_] "v"(a1), [tp_] "v"(D[0])
                         : "memory", FIXED);
        } else {
            asm volatile("v_pk_add_i16 %[x_], %[tp_], %[s1_] clamp" : [x_] "=&v"(x) : [tp_] "v"(D[0]), [s1_] "v"(s1));
        }
        block<V, 0>(a0, a1, D, E, x, sc, ge, go, fl, s1, s2);
        D[0] = D[48];
        if constexpr (V != V_NONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory", FIXED); // (the wrapped loads of the last two blocks)
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = sc ^ x;
#pragma unroll
    for (int i = 0; i < 48; ++i) acc ^= D[i] ^ E[i];
    if (lane == 0) {
        Stamp s;
        s.cyc = t1 - t0 + (acc == 0x12345678u);
        s.real = r1 - r0;
        s.r0 = r0;
        s.r1 = r1;
        s.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));
        s.pad = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        out[blockIdx.x * 4 + wave] = s;
    }
}

struct Probe { const char *name; void (*kern)(Stamp *, uint32_t, int); const char *what; };

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const Probe probes[] = {
        {"perm", probe<V_PERM, 1>, "2 ds_read_b64 + 4 v_perm_b32 per 4 rows (the kernel's sequence-pair cell, 7.5 VALU per row)"},
        {"perm4", probe<V_PERM, 4>, "... 4 lane groups on one shared slice"},
        {"perm8", probe<V_PERM, 8>, "... 8 lane groups"},
        {"mix", probe<V_MIX, 1>, "2 ds_read_b64 + 4 ds_read_u16_d16[_hi] per 4 rows, no v_perm_b32 (6.5 VALU per row), 8-byte profile entries"},
        {"mix4", probe<V_MIX, 4>, "... 4 lane groups on one shared slice"},
        {"mix8", probe<V_MIX, 8>, "... 8 lane groups"},
        {"d16", probe<V_D16, 1>, "8 ds_read_u16_d16[_hi] per 4 rows, 8-byte profile entries (6.5 VALU per row)"},
        {"d16r", probe<V_D16R, 1>, "8 ds_read_u16_d16[_hi] per 4 rows, [row][code] int16 profile (6.5 VALU per row)"},
        {"none", probe<V_NONE, 1>, "no loads, no perms (6.5 VALU per row)"},
    };
    const int ncols = argc > 1 ? atoi(argv[1]) : 4000;
    Stamp *o;
    (void)hipMalloc(&o, (size_t)cus * 8 * 4 * sizeof(Stamp));
    printf("device %s, %d CUs; core-clock cycles per ROW (48-row columns, %d columns) per SIMD, over the SIMDs of the chip\n", p.gcnArchName, cus, ncols);
    const int wpss[] = {1, 2, 3};
    for (const Probe &pr : probes) {
        printf("%-5s %s\n", pr.name, pr.what);
        for (int wps : wpss) {
            const int nb = cus * wps;
            const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
            (void)hipFuncSetAttribute((const void *)pr.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(pr.kern, dim3(nb), dim3(256), lds, 0, o, 12345u, ncols);
            if (hipDeviceSynchronize() != hipSuccess) { printf("   launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            std::vector<Stamp> h((size_t)nb * 4);
            (void)hipMemcpy(h.data(), o, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
            struct Slot { unsigned long long r0 = ~0ull, r1 = 0; int n = 0; double clk = 0; };
            std::map<uint32_t, Slot> slots;
            double clk = 0;
            for (const Stamp &s : h) {
                Slot &sl = slots[((s.hwid >> 4) & 3u) | (((s.hwid >> 8) & 0xffu) << 2) | (s.pad << 10)];
                sl.r0 = std::min(sl.r0, s.r0);
                sl.r1 = std::max(sl.r1, s.r1);
                sl.n++;
                sl.clk += (double)s.cyc / ((double)s.real / 100.0);
                clk += (double)s.cyc / ((double)s.real / 100.0);
            }
            std::vector<double> cpr;
            int most = 0;
            for (auto &kv : slots) {
                const Slot &sl = kv.second;
                most = std::max(most, sl.n);
                const double cycles = (double)(sl.r1 - sl.r0) / 100.0 * (sl.clk / sl.n);
                cpr.push_back(cycles / ((double)ncols * 48.0 * sl.n));
            }
            std::sort(cpr.begin(), cpr.end());
            printf("   w%d: median %6.2f  slowest %6.2f  fastest %6.2f  @%4.0f MHz  (%zu SIMDs, at most %d waves on one)\n", wps, cpr[cpr.size() / 2],
                   cpr.back(), cpr.front(), clk / h.size(), slots.size(), most);
        }
        fflush(stdout);
    }
    return 0;
}
