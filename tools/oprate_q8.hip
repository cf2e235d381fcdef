// tools/oprate_q8.hip -- issue cost of the 8-bit cell (oswald_amd/csrc/q8_cell.h) on one gfx950 SIMD: the bitwise /
// shift instructions it consists of, how a few slow (VOP3) instructions mix into a stream of fast ones, and the cell's
// own column step (CellQ8::column<OSW_RMAX8>: the kernel's strip of query rows x 4 cells per lane), in core-clock cycles per SIMD at 2 / 4 / 6
// waves per SIMD (six is what osw_sw_q8 runs at).  bench.py's ROW_CYCLES[8] comes from the last line.
// Build: hipcc --offload-arch=gfx950 -O3 -I oswald_amd/csrc -I include -o tools/oprate_q8 tools/oprate_q8.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "q8_cell.h"

#define OPV(name, text)                                                                                  \
    struct name {                                                                                        \
        static __device__ __forceinline__ void run(uint32_t &x, uint32_t c1, uint32_t c2, uint32_t sc)   \
        {                                                                                                \
            asm volatile(text : "+v"(x) : "v"(c1), "v"(c2), "s"(sc));                                    \
        }                                                                                                \
    };
OPV(OpAndV, "v_and_b32 %0, %0, %1")
OPV(OpOrV, "v_or_b32 %0, %0, %1")
OPV(OpXorV, "v_xor_b32 %0, %0, %1")
OPV(OpLshrInl, "v_lshrrev_b32 %0, 7, %0")
OPV(OpLshlInl, "v_lshlrev_b32 %0, 1, %0")
OPV(OpAndLit, "v_and_b32 %0, 0x7f7f7f7f, %0")
OPV(OpAndSgpr, "v_and_b32 %0, %3, %0")
OPV(OpSubInl, "v_subrev_u32 %0, 12, %0")
OPV(OpBfi, "v_bfi_b32 %0, %1, %0, %2")
OPV(OpPermV, "v_perm_b32 %0, %0, %1, %2")
OPV(OpPermS, "v_perm_b32 %0, %0, %1, %3")
OPV(OpOr3, "v_or3_b32 %0, %0, %1, %2")
OPV(OpAdd3, "v_add3_u32 %0, %0, %1, %2")
OPV(OpAndOr, "v_and_or_b32 %0, %0, %1, %2")
OPV(OpMulU24, "v_mul_u32_u24 %0, %0, %1")
OPV(OpPkMaxU16, "v_pk_max_u16 %0, %0, %1")
OPV(OpMaxU16, "v_max_u16 %0, %0, %1")
OPV(OpBitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8")

template <class Op>
__global__ __launch_bounds__(256) void op_probe(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[8];
    const uint32_t sc = __builtin_amdgcn_readfirstlane(c2);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0x01020304u * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) Op::run(x[i], c1, c2, sc);
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

// NS slow instructions (v_perm_b32) spread over a stream of 32 - NS fast ones (v_and_b32 / v_sub_u32 alternating)
template <int NS>
__global__ __launch_bounds__(256) void mix_probe(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0x01020304u * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (NS > 0 && k % (32 / (NS > 0 ? NS : 1)) == 0) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(c1), "v"(c2));
            else if (k & 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(c1));
            else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(c2));
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

// the cell's column step on a made-up profile in LDS: OSW_RMAX8 rows per call
__global__ __launch_bounds__(256) void cell_probe(unsigned long long *out, uint32_t go, uint32_t ge, uint32_t bias, uint32_t codes0, int iters)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 4 * 32 * 2 * 4; i += 256) lds[i] = 0x04030201u * ((i * 7) & 7) + 0x01010101u * bias; // 16 rows x 32 codes x 8 B / 4
    __syncthreads();
    const CellQ8::GapT g = CellQ8::make_gap(go, ge, bias, go + ge > bias ? go + ge : bias);
    uint32_t D[OSW_RMAX8], E[OSW_RMAX8], top_prev, f = CellQ8::zero_bits(g), hl = 0, score = CellQ8::score_init(g);
    CellQ8::init_state<OSW_RMAX8>(D, E, top_prev, g);
    uint32_t codes = (codes0 + threadIdx.x * 8) & 0xf8f8u;
    const uint32_t base = 0; // LDS byte address of the slice
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        CellQ8::column<OSW_RMAX8>(base, codes, 0, D, E, top_prev, f, hl, g, g, score);
        top_prev = hl;
        codes = (codes + 0x0808u) & 0xf8f8u;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = score ^ f ^ hl;
#pragma unroll
    for (int i = 0; i < OSW_RMAX8; ++i) acc ^= D[i] ^ E[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

// the hand-scheduled cell (CellQ8F::column<OSW_RMAX8, 0>) on the same made-up profile: residues and F in the fixed registers
__global__ __launch_bounds__(256) OSW8_COMPILER_VGPRS void cell_probe_f(unsigned long long *out, uint32_t go, uint32_t ge, uint32_t bias, uint32_t codes0, int iters)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 4 * 32 * 2 * 4; i += 256) lds[i] = 0x04030201u * ((i * 7) & 7) + 0x01010101u * bias;
    __syncthreads();
    const CellQ8::GapT g = CellQ8::make_gap(go, ge, bias, go + ge > bias ? go + ge : bias);
    uint32_t D[OSW_RMAX8], E[OSW_RMAX8], top_prev, hl = 0;
    CellQ8::init_state<OSW_RMAX8>(D, E, top_prev, g);
    uint32_t sc = CellQ8::score_init(g) | g.G, fl = 0;
    const uint32_t codes = (codes0 + threadIdx.x * 8) & 0xf8f8u, zb = CellQ8::zero_bits(g);
    asm volatile("v_mov_b32 " OSW8_VC0 ", %0\n\tv_mov_b32 " OSW8_VF ", %1" : : "v"(codes), "v"(zb) : OSW8_INFLIGHT);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        CellQ8F::column<OSW_RMAX8, 0>(0u, D, E, top_prev, hl, g, sc, fl);
        top_prev = hl;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = sc ^ fl ^ hl;
#pragma unroll
    for (int i = 0; i < OSW_RMAX8; ++i) acc ^= D[i] ^ E[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

// v_bitop3_b32 bitop3:0xe4 is the byte select of the hand-scheduled cell: (a & k) | (b & ~k)
__global__ void bitop3_selfcheck(uint32_t *out, uint32_t a, uint32_t b, uint32_t k)
{
    uint32_t r;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(r) : "v"(a), "v"(b), "v"(k));
    out[threadIdx.x] = r;
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

template <class K, class... A>
static void sweep(const char *name, K kern, double per, int iters, int cus, unsigned long long *o, A... args)
{
    printf("%-44s", name);
    for (int wps : {2, 4, 6}) {
        const int nb = cus * wps;
        const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const double c = slowest([&] { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, o, args..., iters); }, nb, o);
        printf("  w%d: %7.2f", wps, c / ((double)iters * per) / wps);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    unsigned long long *o;
    (void)hipMalloc(&o, (size_t)cus * 8 * 8);
    printf("device %s, %d CUs; core-clock cycles per wave instruction per SIMD (slowest workgroup) at 2 / 4 / 6 waves per SIMD\n", p.gcnArchName, cus);
#define ONE(T, label) sweep(label, op_probe<T>, 32.0, iters, cus, o, 0x7f7f7f7fu, 0x05040100u)
    ONE(OpAndV, "v_and_b32 vgpr,vgpr (VOP2)");
    ONE(OpOrV, "v_or_b32 vgpr,vgpr (VOP2)");
    ONE(OpXorV, "v_xor_b32 vgpr,vgpr (VOP2)");
    ONE(OpLshrInl, "v_lshrrev_b32 inline 7 (VOP2)");
    ONE(OpLshlInl, "v_lshlrev_b32 inline 1 (VOP2)");
    ONE(OpSubInl, "v_subrev_u32 inline 12 (VOP2)");
    ONE(OpAndLit, "v_and_b32 32-bit literal (VOP2)");
    ONE(OpAndSgpr, "v_and_b32 sgpr operand (VOP2)");
    ONE(OpMaxU16, "v_max_u16 (VOP2)");
    ONE(OpMulU24, "v_mul_u32_u24 (VOP2)");
    ONE(OpBfi, "v_bfi_b32 (VOP3)");
    ONE(OpPermV, "v_perm_b32 vgpr selector (VOP3)");
    ONE(OpPermS, "v_perm_b32 sgpr selector (VOP3)");
    ONE(OpOr3, "v_or3_b32 (VOP3)");
    ONE(OpAdd3, "v_add3_u32 (VOP3)");
    ONE(OpAndOr, "v_and_or_b32 (VOP3)");
    ONE(OpBitop3, "v_bitop3_b32 (VOP3, gfx950)");
    ONE(OpPkMaxU16, "v_pk_max_u16 (VOP3P)");
    printf("32 instructions: NS x v_perm_b32 spread over (32 - NS) x v_and_b32 / v_sub_u32 -- cycles per 32 [if costs added]\n");
    sweep("  NS = 0   [68.0]", mix_probe<0>, 1.0, iters, cus, o, 0x7f7f7f7fu, 0x05040100u);
    sweep("  NS = 1   [70.1]", mix_probe<1>, 1.0, iters, cus, o, 0x7f7f7f7fu, 0x05040100u);
    sweep("  NS = 2   [72.3]", mix_probe<2>, 1.0, iters, cus, o, 0x7f7f7f7fu, 0x05040100u);
    sweep("  NS = 4   [76.5]", mix_probe<4>, 1.0, iters, cus, o, 0x7f7f7f7fu, 0x05040100u);
    sweep("  NS = 8   [85.0]", mix_probe<8>, 1.0, iters, cus, o, 0x7f7f7f7fu, 0x05040100u);
    printf("the 8-bit cell (q8_cell.h), cycles per query row of a 2 x 2 tile (= per 256 cells of a wave) per SIMD:\n");
    sweep("  CellQ8::column<OSW_RMAX8>, PAM250 14/2 (bias 8)", cell_probe, (double)OSW_RMAX8, iters / 8, cus, o, 14u, 2u, 8u, 0x1830u);
    sweep("  CellQ8::column<OSW_RMAX8>, BLOSUM62 10/2 (bias 4)", cell_probe, (double)OSW_RMAX8, iters / 8, cus, o, 10u, 2u, 4u, 0x1830u);
    printf("the hand-scheduled cell (CellQ8F: 39 instructions + 1 v_perm_b32 per row, selects by v_bitop3_b32), same units:\n");
    sweep("  CellQ8F::column<OSW_RMAX8>, PAM250 14/2 (bias 8)", cell_probe_f, (double)OSW_RMAX8, iters / 8, cus, o, 14u, 2u, 8u, 0x1830u);
    sweep("  CellQ8F::column<OSW_RMAX8>, BLOSUM62 10/2 (bias 4)", cell_probe_f, (double)OSW_RMAX8, iters / 8, cus, o, 10u, 2u, 4u, 0x1830u);
    {
        uint32_t *chk, h[1];
        (void)hipMalloc(&chk, 256);
        hipLaunchKernelGGL(bitop3_selfcheck, dim3(1), dim3(1), 0, 0, chk, 0xA1B2C3D4u, 0x11223344u, 0x7f007f00u);
        (void)hipMemcpy(h, chk, 4, hipMemcpyDeviceToHost);
        const uint32_t want = (0xA1B2C3D4u & 0x7f007f00u) | (0x11223344u & ~0x7f007f00u);
        printf("v_bitop3_b32 bitop3:0xe4 (a=0xA1B2C3D4, b=0x11223344, k=0x7f007f00) = 0x%08x, select wants 0x%08x: %s\n", h[0], want, h[0] == want ? "ok" : "MISMATCH");
    }
    return 0;
}
