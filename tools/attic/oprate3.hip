// tools/oprate3.hip -- do the 2-cycle VOP2 instructions of a mixed stream cost 2 cycles when the four waves of a SIMD
// run the SAME stream IN PHASE?  tools/oprate2.hip: with waves from different workgroups (random phase) a fast
// instruction costs ~3.3 cycles inside a mix.  Here one workgroup of 1024 threads per CU (16 waves = 4 per SIMD) starts
// its loop behind a barrier, optionally re-synchronising every `sync_every` iterations.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate3 tools/oprate3.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int V>
__global__ __launch_bounds__(1024) void row_probe(unsigned long long *out, uint32_t go, uint32_t ge, uint32_t fl, uint32_t s, int iters, int sync_every)
{
    uint32_t D[4], E[4], F = fl, x = fl, sc = fl, t;
    const uint32_t sgo = __builtin_amdgcn_readfirstlane(go), sge = __builtin_amdgcn_readfirstlane(ge);
#pragma unroll
    for (int i = 0; i < 4; ++i) { D[i] = fl + threadIdx.x; E[i] = fl; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (sync_every > 0 && (it % sync_every) == 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t xn;
                if constexpr (V == 0)
                    asm volatile("v_pk_add_i16 %[xn], %[Dn], %[s] clamp\n\tv_pk_maximum3_f16 %[Dn], %[x], %[E], %[F]\n\tv_pk_sub_u16 %[t], %[Dn], %[sgo] clamp\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_pk_sub_u16 %[F], %[F], %[sge] clamp"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                else
                    asm volatile("v_add_u32 %[xn], %[Dn], %[s]\n\tv_pk_maximum3_f16 %[Dn], %[x], %[E], %[F]\n\tv_subrev_u32 %[t], %[sgo], %[Dn]\n\t"
                                 "v_pk_maximum3_f16 %[E], %[E], %[t], %[fl]\n\tv_pk_maximum3_f16 %[F], %[F], %[t], %[fl]\n\tv_subrev_u32 %[F], %[sge], %[F]"
                                 : [xn] "=&v"(xn), [t] "=&v"(t), [E] "+v"(E[r]), [Dn] "+v"(D[(r + 1) & 3]), [F] "+v"(F)
                                 : [x] "v"(x), [s] "v"(s), [sgo] "s"(sgo), [sge] "s"(sge), [fl] "v"(fl));
                if (r & 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(sc) : "v"(D[r]), "v"(D[(r + 1) & 3]));
                x = xn;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = sc ^ F ^ x;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= D[i] ^ E[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0 + (acc == 0x12345678u);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    unsigned long long *o;
    (void)hipMalloc(&o, (size_t)cus * 16 * 8);
    printf("one workgroup of 16 waves per CU (4 waves per SIMD, started behind a barrier); cycles per cell row per SIMD (slowest wave)\n");
    const int syncs[4] = {0, 1, 8, 64};
    for (int v = 0; v < 2; ++v)
        for (int si = 0; si < 4; ++si) {
            if (v == 0) hipLaunchKernelGGL(row_probe<0>, dim3(cus), dim3(1024), 0, 0, o, 0x000a000au, 0x00020002u, 0x04000400u, 0x00040004u, iters, syncs[si]);
            else hipLaunchKernelGGL(row_probe<1>, dim3(cus), dim3(1024), 0, 0, o, 0x000a000au, 0x00020002u, 0x04000400u, 0x00040004u, iters, syncs[si]);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h((size_t)cus * 16);
            (void)hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
            double cmax = 0;
            for (auto c : h) cmax = std::max(cmax, (double)c);
            printf("%s, barrier every %2d iterations (of 8 rows): %6.2f cycles per row per SIMD\n", v ? "3 VOP2 + 3.5 VOP3P" : "6.5 VOP3P          ", syncs[si],
                   cmax / ((double)iters * 8) / 4);
        }
    return 0;
}
