// tools/oprate4.hip -- round 4: WHY does a plain 32-bit VOP2 instruction (v_add_u32: 2.13 cycles per wave instruction per
// SIMD on its own) cost ~3.3 cycles inside the DP cell's mix with VOP3P instructions (4.25 on their own)?  VERDICT r03
// item 5.  One binary, every probe a kernel of its own name (so that a rocprofv3 --pmc pass attributes SQ counters per
// probe), every probe run at 1, 2, 3, 4, 6 and 8 waves per SIMD, reporting for every configuration
//   slowest / mean workgroup time in core-clock cycles (s_memtime) per wave instruction per SIMD,
//   the core clock the probe ran at (s_memtime vs s_memrealtime), and
//   the number of distinct (CU, SIMD) slots seen and the waves per slot (HW_ID), i.e. whether the residency is what the
//   LDS size asked for.
// Probes (F = v_add_u32 e32, S = v_pk_maximum3_f16; 16 independent registers unless said otherwise):
//   f16, s16           pure streams
//   fs, ffss, f8s8     8 F + 8 S in three arrangements
//   f1s3, f3s1         other ratios (4 F + 12 S, 12 F + 4 S)
//   fs_dep             F S F S where every F reads the S before it and every S the F before it (one chain per 2 registers)
//   fs_2op             S = v_pk_max_i16 (two operands) instead of the three-operand maximum
//   fs_vop3            S = v_max3_i32 (VOP3, not packed)
//   fmov_s             F = v_mov_b32 (VOP1)
//   fs_nop             F S with an s_nop 0 behind every S
//   row                the column-frame cell's row, dependencies as in the kernel: add, max3, sub, max3, max3, sub (+1/2 max3)
//   row_alls           the same row with the three VOP2 instructions as VOP3P (round 1's row)
//   row_prio           the row with s_setprio 1 on the odd waves of a SIMD
//   row4, row4n, row4n1, row4n3   the rows four to an asm statement, with no / s_nop 0 / s_nop 1 / s_nop 3 behind every row
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate4 tools/oprate4.hip ; run: tools/oprate4 [probe-substring]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define F(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
#define S(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
#define S2(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
#define S3(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
#define FM(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(c1));
#define SN(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n\ts_nop 0" : "+v"(x[i]) : "v"(c1), "v"(c2));
// dependent pair: F writes x[i] from x[j], S writes x[j] from x[i]
#define FD(i, j) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[j]), "v"(c1));
#define SD(j, i) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[i]), "v"(c1), "v"(c2));

struct Stamp { unsigned long long cyc, real, r0, r1; uint32_t hwid, pad; };

#define PROBE_PROLOGUE                                                                     \
    extern __shared__ uint32_t pad_lds[];                                                  \
    if (iters < 0) pad_lds[threadIdx.x] = c1;                                              \
    uint32_t x[16];                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = 0x3c003c00u + threadIdx.x + i;  \
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#define PROBE_EPILOGUE                                                                     \
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime(); \
    uint32_t acc = 0;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc ^= x[i];                           \
    if ((threadIdx.x & 63) == 0) {                                                         \
        Stamp s;                                                                           \
        s.cyc = t1 - t0 + (acc == 0x12345678u);                                            \
        s.real = r1 - r0;                                                                  \
        s.r0 = r0;                                                                         \
        s.r1 = r1;                                                                         \
        s.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)); /* HW_REG_HW_ID[15:0]: wave, simd, pipe, cu, sh, se */ \
        s.pad = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  /* HW_REG_XCC_ID[3:0] */ \
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                      \
    }

#define PROBE(name, BODY)                                                                                         \
    __global__ __launch_bounds__(256) void name(Stamp *out, uint32_t c1, uint32_t c2, int iters)                  \
    {                                                                                                             \
        PROBE_PROLOGUE                                                                                            \
        for (int it = 0; it < iters; ++it) { BODY }                                                               \
        PROBE_EPILOGUE                                                                                            \
    }

PROBE(p_f16, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15))
PROBE(p_s16, S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15))
PROBE(p_fs, F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7) F(8) S(9) F(10) S(11) F(12) S(13) F(14) S(15))
PROBE(p_ffss, F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7) F(8) F(9) S(10) S(11) F(12) F(13) S(14) S(15))
PROBE(p_f8s8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15))
PROBE(p_f1s3, F(0) S(1) S(2) S(3) F(4) S(5) S(6) S(7) F(8) S(9) S(10) S(11) F(12) S(13) S(14) S(15))
PROBE(p_f3s1, F(0) F(1) F(2) S(3) F(4) F(5) F(6) S(7) F(8) F(9) F(10) S(11) F(12) F(13) F(14) S(15))
PROBE(p_fs_dep, FD(0, 1) SD(1, 0) FD(2, 3) SD(3, 2) FD(4, 5) SD(5, 4) FD(6, 7) SD(7, 6) FD(8, 9) SD(9, 8) FD(10, 11) SD(11, 10) FD(12, 13) SD(13, 12) FD(14, 15) SD(15, 14))
PROBE(p_fs_2op, F(0) S2(1) F(2) S2(3) F(4) S2(5) F(6) S2(7) F(8) S2(9) F(10) S2(11) F(12) S2(13) F(14) S2(15))
PROBE(p_fs_vop3, F(0) S3(1) F(2) S3(3) F(4) S3(5) F(6) S3(7) F(8) S3(9) F(10) S3(11) F(12) S3(13) F(14) S3(15))
PROBE(p_fmov_s, FM(0) S(1) FM(2) S(3) FM(4) S(5) FM(6) S(7) FM(8) S(9) FM(10) S(11) FM(12) S(13) FM(14) S(15))
PROBE(p_fs_nop, F(0) SN(1) F(2) SN(3) F(4) SN(5) F(6) SN(7) F(8) SN(9) F(10) SN(11) F(12) SN(13) F(14) SN(15))

// The column-frame cell's row with its dependencies (sw_kernels.hip OSW_I16S_ROW_EVEN / _ODD): D, E per row, F chain, x.
// 8 rows of state, 8 rows per iteration = 52 instructions (6 per row + 1/2 for the running maximum).
#define ROW_V2(r, rn)                                                                                                              \
    asm volatile("v_add_u32 %[xn], %[Dn], %[s]\n\t"                                                                                 \
                 "v_pk_maximum3_f16 %[Dn], %[x], %[E], %[Fc]\n\t"                                                                   \
                 "v_subrev_u32 %[t], %[go], %[Dn]\n\t"                                                                              \
                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\t"                                                                    \
                 "v_pk_maximum3_f16 %[Fc], %[Fc], %[t], %[fl]\n\t"                                                                  \
                 "v_subrev_u32 %[Fc], %[ge], %[Fc]"                                                                                 \
                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[rn]), [Fc] "+v"(Fc)                                    \
                 : [x] "v"(xx), [s] "v"(c1), [go] "v"(go), [ge] "v"(ge), [fl] "v"(fl));                                             \
    xx = xn;
#define ROW_PK(r, rn)                                                                                                              \
    asm volatile("v_pk_add_i16 %[xn], %[Dn], %[s] clamp\n\t"                                                                        \
                 "v_pk_maximum3_f16 %[Dn], %[x], %[E], %[Fc]\n\t"                                                                   \
                 "v_pk_sub_u16 %[t], %[Dn], %[go] clamp\n\t"                                                                        \
                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\t"                                                                    \
                 "v_pk_maximum3_f16 %[Fc], %[Fc], %[t], %[fl]\n\t"                                                                  \
                 "v_pk_sub_u16 %[Fc], %[Fc], %[ge] clamp"                                                                           \
                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[rn]), [Fc] "+v"(Fc)                                    \
                 : [x] "v"(xx), [s] "v"(c1), [go] "v"(go), [ge] "v"(ge), [fl] "v"(fl));                                             \
    xx = xn;
#define ROW_MAX(ra, rb) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(sc) : "v"(D[ra]), "v"(D[rb]));

#define ROW_PROBE(name, ROW, PRIO)                                                                                \
    __global__ __launch_bounds__(256) void name(Stamp *out, uint32_t c1, uint32_t c2, int iters)                  \
    {                                                                                                             \
        extern __shared__ uint32_t pad_lds[];                                                                     \
        if (iters < 0) pad_lds[threadIdx.x] = c1;                                                                 \
        uint32_t x[16], D[8], E[8], Fc = c2, xx = c2, sc = c2, xn, t;                                             \
        uint32_t go = 0x000a000au, ge = 0x00020002u, fl = c2;                                                     \
        asm volatile("" : "+v"(go), "+v"(ge), "+v"(fl));                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = 0;                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { D[i] = c2 + threadIdx.x; E[i] = c2; }                     \
        if (PRIO) { if ((__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1u)) __builtin_amdgcn_s_setprio(1); } \
        const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();        \
        for (int it = 0; it < iters; ++it) {                                                                      \
            ROW(0, 1) ROW(1, 2) ROW_MAX(1, 2) ROW(2, 3) ROW(3, 4) ROW_MAX(3, 4) ROW(4, 5) ROW(5, 6) ROW_MAX(5, 6) ROW(6, 7) ROW(7, 0) ROW_MAX(7, 0) \
        }                                                                                                         \
        x[0] = sc ^ xx ^ Fc;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) x[1] ^= D[i] ^ E[i];                                        \
        PROBE_EPILOGUE                                                                                            \
    }
ROW_PROBE(p_row, ROW_V2, 0)
ROW_PROBE(p_row_alls, ROW_PK, 0)
ROW_PROBE(p_row_prio, ROW_V2, 1)


// The same 8 rows as two statements of four (no compiler-made s_nop between the rows of a statement), with nothing / an
// s_nop 0 / an s_nop 1 behind every row: what the hazard recogniser's s_nop behind every asm statement costs (or buys).
#define RTXT(xn, x, E, Dn, NOP)                                     \
    "v_add_u32 " xn ", " Dn ", %[s]\n\t"                            \
    "v_pk_maximum3_f16 " Dn ", " x ", " E ", %[Fc]\n\t"             \
    "v_subrev_u32 %[t], %[go], " Dn "\n\t"                          \
    "v_pk_maximum3_f16 " E ", " E ", %[t], %[fl]\n\t"               \
    "v_pk_maximum3_f16 %[Fc], %[Fc], %[t], %[fl]\n\t"               \
    "v_subrev_u32 %[Fc], %[ge], %[Fc]\n\t" NOP
#define RMAXT(a, b) "v_pk_maximum3_f16 %[sc], %[sc], " a ", " b "\n\t"
#define ROW4(e0, d1, d2, d3, d4, NOP)                                                                                              \
    asm volatile(RTXT("%[xb]", "%[x]", "%[E0]", "%[D1]", NOP) RTXT("%[x]", "%[xb]", "%[E1]", "%[D2]", NOP) RMAXT("%[D1]", "%[D2]")  \
                 RTXT("%[xb]", "%[x]", "%[E2]", "%[D3]", NOP) RTXT("%[x]", "%[xb]", "%[E3]", "%[D4]", NOP) RMAXT("%[D3]", "%[D4]")  \
                 : [x] "+v"(xx), [xb] "=&v"(xn), [t] "=&v"(t), [E0] "+v"(E[e0]), [E1] "+v"(E[e0 + 1]), [E2] "+v"(E[e0 + 2]),       \
                   [E3] "+v"(E[e0 + 3]), [D1] "+v"(D[d1]), [D2] "+v"(D[d2]), [D3] "+v"(D[d3]), [D4] "+v"(D[d4]), [Fc] "+v"(Fc),    \
                   [sc] "+v"(sc)                                                                                                   \
                 : [s] "v"(c1), [go] "v"(go), [ge] "v"(ge), [fl] "v"(fl));
#define ROW4_PROBE(name, NOP)                                                                                     \
    __global__ __launch_bounds__(256) void name(Stamp *out, uint32_t c1, uint32_t c2, int iters)                  \
    {                                                                                                             \
        extern __shared__ uint32_t pad_lds[];                                                                     \
        if (iters < 0) pad_lds[threadIdx.x] = c1;                                                                 \
        uint32_t x[16], D[8], E[8], Fc = c2, xx = c2, sc = c2, xn, t;                                             \
        uint32_t go = 0x000a000au, ge = 0x00020002u, fl = c2;                                                     \
        asm volatile("" : "+v"(go), "+v"(ge), "+v"(fl));                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = 0;                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { D[i] = c2 + threadIdx.x; E[i] = c2; }                     \
        const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();        \
        for (int it = 0; it < iters; ++it) {                                                                      \
            ROW4(0, 1, 2, 3, 4, NOP) ROW4(4, 5, 6, 7, 0, NOP)                                                     \
        }                                                                                                         \
        x[0] = sc ^ xx ^ Fc;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) x[1] ^= D[i] ^ E[i];                                        \
        PROBE_EPILOGUE                                                                                            \
    }
ROW4_PROBE(p_row4, "")
ROW4_PROBE(p_row4n, "s_nop 0\n\t")
ROW4_PROBE(p_row4n1, "s_nop 1\n\t")
ROW4_PROBE(p_row4n3, "s_nop 3\n\t")

struct Probe { const char *name; void (*kern)(Stamp *, uint32_t, uint32_t, int); double per_iter; const char *what; };

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const Probe probes[] = {
        {"f16", p_f16, 16, "16 F"}, {"s16", p_s16, 16, "16 S"}, {"fs", p_fs, 16, "F S F S ..."}, {"ffss", p_ffss, 16, "F F S S ..."},
        {"f8s8", p_f8s8, 16, "8 F then 8 S"}, {"f1s3", p_f1s3, 16, "F S S S ..."}, {"f3s1", p_f3s1, 16, "F F F S ..."},
        {"fs_dep", p_fs_dep, 16, "F S, each reading the one before"}, {"fs_2op", p_fs_2op, 16, "F + v_pk_max_i16"},
        {"fs_vop3", p_fs_vop3, 16, "F + v_max3_i32"}, {"fmov_s", p_fmov_s, 16, "v_mov_b32 + S"}, {"fs_nop", p_fs_nop, 16, "F + S + s_nop 0"},
        {"row", p_row, 52, "cell row, 3 VOP2 + 3.5 VOP3P (per 6.5 instructions)"}, {"row_alls", p_row_alls, 52, "cell row, all VOP3P"},
        {"row_prio", p_row_prio, 52, "cell row, odd waves at s_setprio 1"},
        {"row4", p_row4, 52, "cell rows, four to a statement (no s_nop)"}, {"row4n", p_row4n, 52, "... with s_nop 0 behind every row"},
        {"row4n1", p_row4n1, 52, "... with s_nop 1 behind every row"}, {"row4n3", p_row4n3, 52, "... with s_nop 3 behind every row"},
    };
    Stamp *o;
    (void)hipMalloc(&o, (size_t)cus * 8 * 4 * sizeof(Stamp));
    printf("device %s, %d CUs; core-clock cycles per wave instruction per SIMD (rows: per row of 6.5 instructions), over the SIMDs of the chip\n", p.gcnArchName, cus);
    const int wpss[] = {1, 2, 3, 4, 6, 8};
    for (const Probe &pr : probes) {
        if (argc > 1 && !strstr(pr.name, argv[1])) continue;
        printf("%-9s %-52s\n", pr.name, pr.what);
        for (int wps : wpss) {
            const int nb = cus * wps, iters = argc > 2 ? atoi(argv[2]) : 400000 / wps;
            const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
            (void)hipFuncSetAttribute((const void *)pr.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(pr.kern, dim3(nb), dim3(256), lds, 0, o, 0x00030003u, 0x04000400u, iters);
            (void)hipDeviceSynchronize();
            std::vector<Stamp> h((size_t)nb * 4);
            (void)hipMemcpy(h.data(), o, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
            // per SIMD of the chip: the waves it held, from the first start to the last end (s_memrealtime: one 100 MHz clock
            // for the whole chip), in cycles of the core clock those waves measured -- launch skew and early finishers do
            // not enter; reported: the median SIMD, and the slowest
            struct Slot { unsigned long long r0 = ~0ull, r1 = 0; int n = 0; double clk = 0; };
            std::map<uint32_t, Slot> slots;
            double clk = 0;
            for (const Stamp &s : h) {
                Slot &sl = slots[((s.hwid >> 4) & 3u) | (((s.hwid >> 8) & 0xffu) << 2) | (s.pad << 10)]; // (SIMD, CU/SH/SE, XCC)
                sl.r0 = std::min(sl.r0, s.r0);
                sl.r1 = std::max(sl.r1, s.r1);
                sl.n++;
                sl.clk += (double)s.cyc / ((double)s.real / 100.0); // MHz
                clk += (double)s.cyc / ((double)s.real / 100.0);
            }
            std::vector<double> cpi;
            int most = 0;
            for (auto &kv : slots) {
                const Slot &sl = kv.second;
                most = std::max(most, sl.n);
                const double cycles = (double)(sl.r1 - sl.r0) / 100.0 * (sl.clk / sl.n); // us x MHz
                cpi.push_back(cycles / ((double)iters * pr.per_iter * sl.n));
            }
            std::sort(cpi.begin(), cpi.end());
            const double scale = pr.per_iter == 52 ? 6.5 : 1.0; // rows: report cycles per ROW (6.5 instructions)
            printf("   w%d: median %6.2f  slowest %6.2f  fastest %6.2f  @%4.0f MHz  (%zu SIMDs, at most %d waves on one)\n", wps, cpi[cpi.size() / 2] * scale,
                   cpi.back() * scale, cpi.front() * scale, clk / h.size(), slots.size(), most);
        }
        fflush(stdout);
    }
    return 0;
}
