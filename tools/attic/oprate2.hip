// tools/oprate2.hip -- how fast (2-cycle: v_add_u32 / v_sub_u32, VGPR operands, e32) and slow (4-cycle: VOP3P
// packed 16-bit, VOP3) VALU instructions mix on one gfx950 SIMD.  tools/oprate.hip showed the per-opcode costs;
// a row of the DP cell interleaves both kinds, and the costs do not simply add.  Patterns of F (fast) and S (slow)
// over 8 independent registers, and candidate orderings of the real cell row, at 2 / 4 / 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate2 tools/oprate2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define F(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
#define S(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));

template <int P>
__global__ __launch_bounds__(256) void pat_probe(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0x3c003c00u + threadIdx.x + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if constexpr (P == 0) { S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) }
            if constexpr (P == 1) { F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) }
            if constexpr (P == 2) { F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7) F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7) }
            if constexpr (P == 3) { F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7) F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7) }
            if constexpr (P == 4) { F(0) F(1) F(2) F(3) S(4) S(5) S(6) S(7) F(0) F(1) F(2) F(3) S(4) S(5) S(6) S(7) }
            if constexpr (P == 5) { F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) }
            if constexpr (P == 6) { F(0) S(1) S(2) F(3) S(4) S(5) F(6) S(7) S(0) F(1) S(2) S(3) F(4) S(5) S(6) S(7) } // 5 F + 11 S
            if constexpr (P == 7) { F(0) F(1) F(2) S(3) S(4) S(5) S(6) F(7) F(0) F(1) S(2) S(3) S(4) S(5) S(6) S(7) } // 6 F + 10 S, runs of 3
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

// Candidate orderings of the column-frame cell row (4 rows of state, 8 rows per iteration).
//   0: today: add(pk) max3 sub(pk) max3 max3 sub(pk)  [+ 1/2 max3]
//   1: VOP2 in the same places
//   2: VOP2, u and the next row's diagonal add adjacent:  max3 sub add | max3 max3 sub
//   3: VOP2, only the two subtracts (the sequence-pair cell keeps v_pk_add_i16 for the diagonal)
//   4: VOP2, only the diagonal add
//   5: as 2 plus the running maximum kept per row pair with v_pk_max (no change in count; order only)
template <int V>
__global__ __launch_bounds__(256) void row_probe(unsigned long long *out, uint32_t go, uint32_t ge, uint32_t fl, uint32_t s, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = go;
    uint32_t D[4], E[4], F = fl, x = fl, sc = fl, t;
    const uint32_t sgo = __builtin_amdgcn_readfirstlane(go), sge = __builtin_amdgcn_readfirstlane(ge);
#pragma unroll
    for (int i = 0; i < 4; ++i) { D[i] = fl + threadIdx.x; E[i] = fl; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t xn;
                if constexpr (V == 0)
                    asm volatile("v_pk_add_i16 %[xn], %[Dn], %[s] clamp\n\tv_pk_maximum3_f16 %[Dn], %[x], %[E], %[F]\n\tv_pk_sub_u16 %[t], %[Dn], %[sgo] clamp\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_pk_sub_u16 %[F], %[F], %[sge] clamp"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [go] "v"(go), [ge] "v"(ge), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                if constexpr (V == 1)
                    asm volatile("v_add_u32 %[xn], %[Dn], %[s]\n\tv_pk_maximum3_f16 %[Dn], %[x], %[E], %[F]\n\tv_sub_u32 %[t], %[Dn], %[go]\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_sub_u32 %[F], %[F], %[ge]"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [go] "v"(go), [ge] "v"(ge), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                if constexpr (V == 2 || V == 5) { // the diagonal add of row r+1 needs D[r+1] BEFORE this row's H overwrites it: use a second register
                    uint32_t h;
                    asm volatile("v_pk_maximum3_f16 %[h], %[x], %[E], %[F]\n\tv_sub_u32 %[t], %[h], %[go]\n\tv_add_u32 %[xn], %[Dn], %[s]\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_sub_u32 %[F], %[F], %[ge]\n\tv_mov_b32 %[Dn], %[h]"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [h] "=&v"(h), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [go] "v"(go), [ge] "v"(ge), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                }
                if constexpr (V == 3)
                    asm volatile("v_pk_add_i16 %[xn], %[Dn], %[s] clamp\n\tv_pk_maximum3_f16 %[Dn], %[x], %[E], %[F]\n\tv_sub_u32 %[t], %[Dn], %[go]\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_sub_u32 %[F], %[F], %[ge]"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [go] "v"(go), [ge] "v"(ge), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                if constexpr (V == 4)
                    asm volatile("v_add_u32 %[xn], %[Dn], %[s]\n\tv_pk_maximum3_f16 %[Dn], %[x], %[E], %[F]\n\tv_pk_sub_u16 %[t], %[Dn], %[sgo] clamp\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_pk_sub_u16 %[F], %[F], %[sge] clamp"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [go] "v"(go), [ge] "v"(ge), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                if (r & 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(sc) : "v"(D[r]), "v"(D[(r + 1) & 3]));
                x = xn;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = sc ^ F ^ x;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= D[i] ^ E[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

template <class L>
static double slowest(L launch, int nb, unsigned long long *o)
{
    launch();
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb);
    (void)hipMemcpy(h.data(), o, nb * 8, hipMemcpyDeviceToHost);
    double cmax = 0;
    for (int i = 0; i < nb; ++i) cmax = std::max(cmax, (double)h[i]);
    return cmax;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    unsigned long long *o;
    (void)hipMalloc(&o, (size_t)cus * 8 * 8);
    const int wpss[3] = {2, 4, 8};
    const char *pn[8] = {"S x16", "F x16", "FSFS...", "FFSS...", "FFFFSSSS...", "F x8 S x8", "5 F + 11 S spread", "6 F + 10 S, runs of 3"};
    const int nf[8] = {0, 16, 8, 8, 8, 8, 5, 6};
    printf("patterns of 16 instructions (F = v_add_u32 e32, S = v_pk_maximum3_f16), 8 independent registers; cycles per 16 [if costs added: 2.13 F + 4.25 S]\n");
    for (int pt = 0; pt < 8; ++pt) {
        printf("%-26s", pn[pt]);
        for (int w = 0; w < 3; ++w) {
            const int wps = wpss[w], nb = cus * wps;
            const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                return slowest([&] { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, o, 0x00030003u, 0x04000400u, iters); }, nb, o);
            };
            double c = 0;
            switch (pt) {
            case 0: c = go(pat_probe<0>); break; case 1: c = go(pat_probe<1>); break; case 2: c = go(pat_probe<2>); break; case 3: c = go(pat_probe<3>); break;
            case 4: c = go(pat_probe<4>); break; case 5: c = go(pat_probe<5>); break; case 6: c = go(pat_probe<6>); break; default: c = go(pat_probe<7>); break;
            }
            printf("  w%d: %6.2f", wps, c / ((double)iters * 2) / wps);
        }
        printf("   [%.1f]\n", nf[pt] * 2.125 + (16 - nf[pt]) * 4.25);
        fflush(stdout);
    }
    const char *rn[6] = {"row today (6.5 VOP3P)", "row 3 VOP2 same places", "row 3 VOP2, u+add adjacent (+mov)", "row 2 VOP2 (subs only)", "row 1 VOP2 (diag add only)", "-"};
    printf("cell rows, cycles per row per SIMD:\n");
    for (int v = 0; v < 5; ++v) {
        printf("%-36s", rn[v]);
        for (int w = 0; w < 3; ++w) {
            const int wps = wpss[w], nb = cus * wps;
            const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                return slowest([&] { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, o, 0x000a000au, 0x00020002u, 0x04000400u, 0x00040004u, iters / 4); }, nb, o);
            };
            double c = v == 0 ? go(row_probe<0>) : v == 1 ? go(row_probe<1>) : v == 2 ? go(row_probe<2>) : v == 3 ? go(row_probe<3>) : go(row_probe<4>);
            printf("  w%d: %6.2f", wps, c / ((double)(iters / 4) * 8) / wps);
        }
        printf("\n");
    }
    return 0;
}
