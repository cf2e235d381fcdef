// tools/oprate7.hip -- round 4: would SDWA byte maxima make the 8-bit cell shorter?  The SWAR maximum of four bytes
// (q8_cell.h) is five fast instructions (sub, and, shift, sub, select); gfx950 still has SDWA, and
//   v_max_u16_sdwa d, a, b dst_sel:BYTE_k dst_unused:UNUSED_PRESERVE src0_sel:BYTE_k src1_sel:BYTE_k
// is the maximum of ONE byte lane written in place: four of them are a packed 8-bit maximum.  What do they cost?
//   sdwa1     pure stream of such instructions, 8 independent registers, one byte each
//   sdwa4     the four bytes of a register in a row (each instruction reads the register the one before it wrote), 8 registers in turn
//   sdwa_mix  one SDWA maximum in four, the rest v_sub_u32 / v_and_b32
//   row_swar  the hand-scheduled row of q8_cell.h (39 instructions), synthetic operands, 12 rows of state
//   row_sdwa  the same recurrences with SDWA maxima (31 instructions)
// cycles per wave instruction (rows: per row) per SIMD at 2 / 4 / 6 waves per SIMD, slowest workgroup.
// Build: make -C tools oprate7
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define MAXB(D, A, B, K) "v_max_u16_sdwa " D ", " A ", " B " dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #K " src1_sel:BYTE_" #K "\n\t"
#define MAX4(D, A, B) MAXB(D, A, B, 0) MAXB(D, A, B, 1) MAXB(D, A, B, 2) MAXB(D, A, B, 3)

__global__ __launch_bounds__(256) void p_sdwa1(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0x01020304u * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile(MAXB("%0", "%0", "%1", 0) : "+v"(x[i]) : "v"(c1));
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

__global__ __launch_bounds__(256) void p_sdwa4(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0x01020304u * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile(MAX4("%0", "%0", "%1") : "+v"(x[i]) : "v"(c1));
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

__global__ __launch_bounds__(256) void p_sdwa_mix(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
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
            if ((k & 3) == 0) asm volatile(MAXB("%0", "%0", "%1", 1) : "+v"(x[k & 7]) : "v"(c1));
            else if (k & 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(c1));
            else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(c2));
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345678u);
}

// the SWAR row of q8_cell.h (OSW_Q8_HMAX / OSW_Q8_GAP), on ordinary operands
#define SEL "bitop3:0xe4"
#define HMAX(A, H) "v_sub_u32 %[t1], " A ", " H "\n\tv_and_b32 %[t2], %[t1], %[G_]\n\tv_lshrrev_b32 %[t3], 7, %[t2]\n\tv_sub_u32 %[t2], %[t2], %[t3]\n\tv_bitop3_b32 " H ", " A ", " H ", %[t2] " SEL "\n\t"
#define GAP(X) "v_sub_u32 %[t1], " X ", %[u_]\n\tv_and_b32 %[t2], %[t1], %[G_]\n\tv_lshrrev_b32 %[t3], 7, %[t2]\n\tv_sub_u32 %[t2], %[t2], %[t3]\n\tv_and_b32 %[t1], %[t1], %[t2]\n\tv_add_u32 " X ", %[ug_], %[t1]\n\t"
#define ROW_SWAR                                                                                          \
    "v_add_u32 %[xn_], %[Dn_], %[sn_]\n\tv_or_b32 %[fl_], %[fl_], %[x_]\n\tv_and_b32 %[Dn_], %[x_], %[L_]\n\t" \
    HMAX("%[cG_]", "%[Dn_]") HMAX("%[E_]", "%[Dn_]") HMAX("%[F_]", "%[Dn_]")                               \
    "v_sub_u32 %[u_], %[Dn_], %[go_]\n\tv_add_u32 %[ug_], %[Dn_], %[uj_]\n\t" GAP("%[E_]") GAP("%[F_]")     \
    "v_add_u32 %[ug_], %[Dn_], %[G_]\n\tv_sub_u32 %[t1], %[sc_], %[Dn_]\n\tv_and_b32 %[t2], %[t1], %[G_]\n\t" \
    "v_lshrrev_b32 %[t3], 7, %[t2]\n\tv_sub_u32 %[t2], %[t2], %[t3]\n\tv_bitop3_b32 %[sc_], %[sc_], %[ug_], %[t2] " SEL "\n\t" \
    "v_sub_u32 %[Dn_], %[Dn_], %[b_]"
// the same recurrences with SDWA maxima: H = max(x & L, c, E, F); u = H - go; E = max(E, u) - ge; F likewise; score
#define ROW_SDWA                                                                                          \
    "v_add_u32 %[xn_], %[Dn_], %[sn_]\n\tv_or_b32 %[fl_], %[fl_], %[x_]\n\tv_and_b32 %[Dn_], %[x_], %[L_]\n\t" \
    MAX4("%[Dn_]", "%[Dn_]", "%[cG_]") MAX4("%[Dn_]", "%[Dn_]", "%[E_]") MAX4("%[Dn_]", "%[Dn_]", "%[F_]")   \
    "v_sub_u32 %[u_], %[Dn_], %[go_]\n\t"                                                                  \
    MAX4("%[E_]", "%[E_]", "%[u_]") "v_sub_u32 %[E_], %[E_], %[uj_]\n\t"                                    \
    MAX4("%[F_]", "%[F_]", "%[u_]") "v_sub_u32 %[F_], %[F_], %[uj_]\n\t"                                    \
    MAX4("%[sc_]", "%[sc_]", "%[Dn_]")                                                                      \
    "v_sub_u32 %[Dn_], %[Dn_], %[b_]"

#define ROW_PROBE(name, TXT)                                                                                        \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(72))) void name(unsigned long long *out, uint32_t c1, uint32_t c2, int iters) \
    {                                                                                                               \
        extern __shared__ uint32_t pad_lds[];                                                                       \
        if (iters < 0) pad_lds[threadIdx.x] = c1;                                                                   \
        uint32_t D[13], E[12], F = 0x10101010u, sc = 0x10101010u, fl = 0, x = 0x11121314u + threadIdx.x, xn, t1, t2, t3, u, ug; \
        uint32_t G = 0x80808080u, L = 0x7f7f7f7fu, cG = 0x90909090u, go = 0x0e0e0e0eu, uj = 0x02020202u, b = 0x08080808u, sn = 0x09070b05u; \
        asm volatile("" : "+v"(G), "+v"(L), "+v"(cG), "+v"(go), "+v"(uj), "+v"(b), "+v"(sn));                       \
        _Pragma("unroll") for (int i = 0; i < 12; ++i) { D[i] = 0x08080808u + threadIdx.x; E[i] = 0x90909090u; }   \
        D[12] = 0x08080808u;                                                                                        \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                 \
        for (int it = 0; it < iters; ++it) {                                                                        \
            _Pragma("unroll") for (int r = 0; r < 12; ++r) {                                                        \
                asm volatile(TXT                                                                                    \
                             : [xn_] "=&v"(xn), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u_] "=&v"(u), [ug_] "=&v"(ug), [E_] "+v"(E[r]), \
                               [Dn_] "+v"(D[r + 1]), [sc_] "+v"(sc), [fl_] "+v"(fl), [F_] "+v"(F)                   \
                             : [x_] "v"(x), [sn_] "v"(sn), [G_] "v"(G), [L_] "v"(L), [cG_] "v"(cG), [go_] "v"(go), [uj_] "v"(uj), [b_] "v"(b)); \
                x = xn;                                                                                             \
            }                                                                                                       \
        }                                                                                                           \
        const unsigned long long t1c = __builtin_readcyclecounter();                                                \
        uint32_t acc = sc ^ fl ^ F ^ x;                                                                             \
        _Pragma("unroll") for (int i = 0; i < 12; ++i) acc ^= D[i] ^ E[i];                                          \
        if (threadIdx.x == 0) out[blockIdx.x] = t1c - t0 + (acc == 0x12345678u);                                    \
    }
ROW_PROBE(p_row_swar, ROW_SWAR)
ROW_PROBE(p_row_sdwa, ROW_SDWA)

__global__ void sdwa_selfcheck(uint32_t *out, uint32_t a, uint32_t b)
{
    uint32_t r = a;
    asm volatile(MAX4("%0", "%0", "%1") : "+v"(r) : "v"(b));
    out[threadIdx.x] = r;
}

template <class K>
static void sweep(const char *name, K kern, double per, int iters, int cus, unsigned long long *o)
{
    printf("%-72s", name);
    for (int wps : {2, 4, 6}) {
        const int nb = cus * wps;
        const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, o, 0x7f7f7f7fu, 0x05040100u, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(nb);
        (void)hipMemcpy(h.data(), o, nb * 8, hipMemcpyDeviceToHost);
        double cmax = 0;
        for (int i = 0; i < nb; ++i) cmax = std::max(cmax, (double)h[i]);
        printf("  w%d: %7.2f", wps, cmax / ((double)iters * per) / wps);
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
    printf("device %s, %d CUs; core-clock cycles per wave instruction (rows: per row) per SIMD, slowest workgroup, at 2 / 4 / 6 waves per SIMD\n", p.gcnArchName, cus);
    sweep("sdwa1: v_max_u16_sdwa one byte in place, independent registers", p_sdwa1, 32.0, iters, cus, o);
    sweep("sdwa4: the four bytes of a register in a row", p_sdwa4, 32.0, iters, cus, o);
    sweep("sdwa_mix: 8 SDWA maxima among 24 v_sub_u32 / v_and_b32 (per instruction)", p_sdwa_mix, 32.0, iters, cus, o);
    sweep("row_swar: the 8-bit cell's row, SWAR maxima (39 instructions), per row", p_row_swar, 12.0, iters / 8, cus, o);
    sweep("row_sdwa: the same recurrences, SDWA maxima (31 instructions), per row", p_row_sdwa, 12.0, iters / 8, cus, o);
    uint32_t *chk, h[1];
    (void)hipMalloc(&chk, 256);
    hipLaunchKernelGGL(sdwa_selfcheck, dim3(1), dim3(1), 0, 0, chk, 0x10F27F03u, 0x11803344u);
    (void)hipMemcpy(h, chk, 4, hipMemcpyDeviceToHost);
    printf("four v_max_u16_sdwa (a=0x10F27F03, b=0x11803344) = 0x%08x, bytewise maximum wants 0x11F27F44: %s\n", h[0], h[0] == 0x11F27F44u ? "ok" : "MISMATCH");
    return 0;
}
