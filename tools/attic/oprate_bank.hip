// tools/oprate_bank.hip -- does the fast issue rate of plain 32-bit VOP2 instructions on gfx950 (~2.1-2.5 cycles per
// wave instruction against 4.25 for VOP3 / VOP3P) depend on WHICH registers the operands sit in?  The 8-bit cell
// (q8_cell.h) consists of such instructions only and still costs 3.6 cycles per instruction with compiler-allocated
// registers.  Fixed physical registers, eight instructions per group, varying the (dst, src0, src1) register numbers
// modulo 4 (VGPR banks) and whether dst == src0.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate_bank tools/oprate_bank.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", \
             "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"

// eight instructions "OP vD, vA, vB"; the three register numbers are given per instruction
#define I8(OP, d0, a0, b0, d1, a1, b1, d2, a2, b2, d3, a3, b3, d4, a4, b4, d5, a5, b5, d6, a6, b6, d7, a7, b7)                                   \
    OP " v" #d0 ", v" #a0 ", v" #b0 "\n\t" OP " v" #d1 ", v" #a1 ", v" #b1 "\n\t" OP " v" #d2 ", v" #a2 ", v" #b2 "\n\t" OP " v" #d3 ", v" #a3 ", v" #b3 "\n\t" \
    OP " v" #d4 ", v" #a4 ", v" #b4 "\n\t" OP " v" #d5 ", v" #a5 ", v" #b5 "\n\t" OP " v" #d6 ", v" #a6 ", v" #b6 "\n\t" OP " v" #d7 ", v" #a7 ", v" #b7 "\n\t"

template <int V>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, uint32_t c1, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\t"
                 "v_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\tv_mov_b32 v55, %0\n\t"
                 "v_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\tv_mov_b32 v58, %0\n\tv_mov_b32 v59, %0\n\tv_mov_b32 v60, %0\n\tv_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\t"
                 "v_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\tv_mov_b32 v67, %0\n\tv_mov_b32 v68, %0\n\tv_mov_b32 v69, %0\n\tv_mov_b32 v70, %0\n\tv_mov_b32 v71, %0"
                 : : "v"(c1) : CLOB);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // 0: in place, src1 one shared constant register in another bank:  vN = vN & v64        (N = 40..47)
            if constexpr (V == 0) asm volatile(I8("v_and_b32", 40, 40, 64, 41, 41, 64, 42, 42, 64, 43, 43, 64, 44, 44, 64, 45, 45, 64, 46, 46, 64, 47, 47, 64) ::: CLOB);
            // 1: in place, src1 in the SAME bank as src0 / dst (N and N + 8)
            if constexpr (V == 1) asm volatile(I8("v_and_b32", 40, 40, 48, 41, 41, 49, 42, 42, 50, 43, 43, 51, 44, 44, 52, 45, 45, 53, 46, 46, 54, 47, 47, 55) ::: CLOB);
            // 2: in place, src1 in the next bank (N and N + 9)
            if constexpr (V == 2) asm volatile(I8("v_and_b32", 40, 40, 49, 41, 41, 50, 42, 42, 51, 43, 43, 52, 44, 44, 53, 45, 45, 54, 46, 46, 55, 47, 47, 56) ::: CLOB);
            // 3: three different registers, all in different banks: vN = v(N+9) & v(N+18)
            if constexpr (V == 3) asm volatile(I8("v_and_b32", 40, 49, 58, 41, 50, 59, 42, 51, 60, 43, 52, 61, 44, 53, 62, 45, 54, 63, 46, 55, 64, 47, 56, 65) ::: CLOB);
            // 4: three different registers, src0 and src1 in the same bank, dst in another: vN = v(N+9) & v(N+17)
            if constexpr (V == 4) asm volatile(I8("v_and_b32", 40, 49, 57, 41, 50, 58, 42, 51, 59, 43, 52, 60, 44, 53, 61, 45, 54, 62, 46, 55, 63, 47, 56, 64) ::: CLOB);
            // 5: three different registers, dst in the bank of src0: vN = v(N+8) & v(N+17)
            if constexpr (V == 5) asm volatile(I8("v_and_b32", 40, 48, 57, 41, 49, 58, 42, 50, 59, 43, 51, 60, 44, 52, 61, 45, 53, 62, 46, 54, 63, 47, 55, 64) ::: CLOB);
            // 6: a dependent chain of eight (each reads the previous result), banks rotating: latency, not issue
            if constexpr (V == 6) asm volatile(I8("v_and_b32", 41, 40, 64, 42, 41, 65, 43, 42, 66, 44, 43, 67, 45, 44, 68, 46, 45, 69, 47, 46, 70, 40, 47, 71) ::: CLOB);
            // 7: two interleaved dependent chains
            if constexpr (V == 7) asm volatile(I8("v_and_b32", 41, 40, 64, 49, 48, 65, 42, 41, 66, 50, 49, 67, 43, 42, 68, 51, 50, 69, 40, 43, 70, 48, 51, 71) ::: CLOB);
            // 8: four interleaved dependent chains
            if constexpr (V == 8) asm volatile(I8("v_and_b32", 41, 40, 64, 49, 48, 65, 57, 56, 66, 61, 60, 67, 40, 41, 68, 48, 49, 69, 56, 57, 70, 60, 61, 71) ::: CLOB);
            // 9: v_sub_u32 three-address, different banks
            if constexpr (V == 9) asm volatile(I8("v_sub_u32", 40, 49, 58, 41, 50, 59, 42, 51, 60, 43, 52, 61, 44, 53, 62, 45, 54, 63, 46, 55, 64, 47, 56, 65) ::: CLOB);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc;
    asm volatile("v_xor_b32 %0, v40, v47" : "=v"(acc) : : CLOB);
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
    const char *names[10] = {"in place, src1 = one constant register", "in place, src1 same bank", "in place, src1 next bank", "3 registers, 3 banks",
                             "3 registers, src0/src1 same bank", "3 registers, dst/src0 same bank", "dependent chain of 8", "2 interleaved chains", "4 interleaved chains",
                             "v_sub_u32, 3 registers, 3 banks"};
    printf("v_and_b32 vD, vA, vB with fixed registers; cycles per wave instruction per SIMD at 1 / 2 / 4 waves per SIMD\n");
    for (int v = 0; v < 10; ++v) {
        printf("%-42s", names[v]);
        for (int wps : {1, 2, 4}) {
            const int nb = cus * wps;
            const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                return slowest([&] { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, o, 0x7f7f7f7fu, iters); }, nb, o);
            };
            double c = 0;
            switch (v) {
            case 0: c = go(probe<0>); break; case 1: c = go(probe<1>); break; case 2: c = go(probe<2>); break; case 3: c = go(probe<3>); break;
            case 4: c = go(probe<4>); break; case 5: c = go(probe<5>); break; case 6: c = go(probe<6>); break; case 7: c = go(probe<7>); break;
            case 8: c = go(probe<8>); break; default: c = go(probe<9>); break;
            }
            printf("  w%d: %6.2f", wps, c / ((double)iters * 32) / wps);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
