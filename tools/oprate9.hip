// tools/oprate9.hip -- round 5, second session: the hand-scheduled int32 row (CellI32F, sw_kernels.hip).
//     v_add_u32_sdwa x, D, sext(P) src1_sel:WORD_k      x = D + the k-th int16 of the profile entry's register
//     v_max3_i32 H, x, E, F ; v_subrev_u32 u, go, H ; v_max3_i32 E, E, u, fl ; v_max3_i32 F, F, u, fl ; v_subrev_u32 F, ge, F
//     (+ one v_max3_i32 per two rows for the column maximum): 6.5 instructions per row of 64 cells.
// This probe (harness of tools/oprate8.hip) (1) checks on the device that the SDWA add computes D + sext(word) for both words,
// (2) runs the column loop of the cell -- 48-row strip, state in place, 1 / 2 / 3 waves per SIMD (168 VGPRs) -- in two forms:
//   i32    the kernel's: one ds_read_b64 per 4 rows (8-byte entries: 4 rows x int16), loads two blocks ahead; 1 / 8 lane groups
//   none   no loads (the VALU floor of the row)
// and reports core-clock cycles per row per SIMD (median / slowest / fastest SIMD of the chip).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate9 tools/oprate9.hip ; run: tools/oprate9 [columns]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

struct Stamp { unsigned long long cyc, real, r0, r1; uint32_t hwid, pad; };

#define PA0 "v150"
#define PA1 "v151"
#define PB0 "v154"
#define PB1 "v155"
#define VT "v159"
#define VF "v161"
#define FIXED "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v159", "v161"

#define SEL(REG, W) "sext(" REG ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" W
#define ROWBODY(DN, X, E, SCMAX)                                        \
    "v_max3_i32 " DN ", " X ", " E ", " VF "\n\t"                       \
    "v_subrev_u32 " VT ", %[go_], " DN "\n\t" SCMAX                     \
    "v_max3_i32 " E ", " E ", " VT ", %[fl_]\n\t"                       \
    "v_max3_i32 " VF ", " VF ", " VT ", %[fl_]\n\t"                     \
    "v_subrev_u32 " VF ", %[ge_], " VF "\n\t"
#define ROWW(XN, DN, SN, X, E, HOOK, SCMAX) "v_add_u32_sdwa " XN ", " DN ", " SN "\n\t" HOOK ROWBODY(DN, X, E, SCMAX)
#define ROWN(XN, DN, SN, X, E, HOOK, SCMAX) "v_add_u32 " XN ", " DN ", " SN "\n\t" HOOK ROWBODY(DN, X, E, SCMAX)
#define SCRUN(A, B) "v_max3_i32 %[sc_], %[sc_], " A ", " B "\n\t"
#define LD_64(BUF) "ds_read_b64 " BUF ", %[a0_] offset:%[off_]\n\ts_waitcnt lgkmcnt(1)\n\t"
#define BLK_I32(P0, P1, PALL, N0)                                                                                \
    ROWW("%[xb_]", "%[D1_]", SEL(P0, "1"), "%[x_]", "%[E0_]", "", "")                                            \
    ROWW("%[x_]", "%[D2_]", SEL(P1, "0"), "%[xb_]", "%[E1_]", "", SCRUN("%[D1_]", "%[D2_]"))                     \
    ROWW("%[xb_]", "%[D3_]", SEL(P1, "1"), "%[x_]", "%[E2_]", LD_64(PALL), "")                                   \
    ROWW("%[x_]", "%[D4_]", SEL(N0, "0"), "%[xb_]", "%[E3_]", "", SCRUN("%[D3_]", "%[D4_]"))
#define BLK_NONE                                                                                                 \
    ROWN("%[xb_]", "%[D1_]", "%[s1_]", "%[x_]", "%[E0_]", "", "")                                                \
    ROWN("%[x_]", "%[D2_]", "%[s2_]", "%[xb_]", "%[E1_]", "", SCRUN("%[D1_]", "%[D2_]"))                         \
    ROWN("%[xb_]", "%[D3_]", "%[s1_]", "%[x_]", "%[E2_]", "", "")                                                \
    ROWN("%[x_]", "%[D4_]", "%[s2_]", "%[xb_]", "%[E3_]", "", SCRUN("%[D3_]", "%[D4_]"))

#define STMT(TXT, RB, NEXTOFF)                                                                                   \
    asm volatile(TXT                                                                                             \
                 : [x_] "+v"(x), [xb_] "=&v"(xb), [E0_] "+v"(E[RB * 4]), [E1_] "+v"(E[RB * 4 + 1]), [E2_] "+v"(E[RB * 4 + 2]),  \
                   [E3_] "+v"(E[RB * 4 + 3]), [D1_] "+v"(D[RB * 4 + 1]), [D2_] "+v"(D[RB * 4 + 2]), [D3_] "+v"(D[RB * 4 + 3]),  \
                   [D4_] "+v"(D[RB * 4 + 4]), [sc_] "+v"(sc)                                                    \
                 : [a0_] "v"(a0), [off_] "i"(NEXTOFF * 256), [ge_] "s"(ge), [go_] "s"(go), [fl_] "v"(fl), [s1_] "v"(s1), [s2_] "v"(s2) \
                 : "memory", FIXED)

enum { V_I32 = 0, V_NONE = 2 };

template <int V, int RB>
static __device__ __forceinline__ void block(uint32_t a0, uint32_t (&D)[49], uint32_t (&E)[48], uint32_t &x, uint32_t &sc, uint32_t ge,
                                             uint32_t go, uint32_t fl, uint32_t s1, uint32_t s2)
{
    constexpr int NEXT = (RB + 2) % 12; // (the last blocks of a column load the first again: the stream never stops)
    uint32_t xb;
    if constexpr (V == V_I32) {
        if constexpr ((RB & 1) == 0) STMT(BLK_I32(PA0, PA1, "v[150:151]", PB0), RB, NEXT);
        else STMT(BLK_I32(PB0, PB1, "v[154:155]", PA0), RB, NEXT);
    } else {
        STMT(BLK_NONE, RB, NEXT);
    }
    if constexpr (RB + 1 < 12) block<V, RB + 1>(a0, D, E, x, sc, ge, go, fl, s1, s2);
}

#define LDS_WORDS 6656 // 26 KB: eight lane groups x (48 rows x 64 B + 8)

template <int V, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(140))) void probe(Stamp *out, uint32_t seed, int ncols)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < LDS_WORDS; i += 256) lds[i] = 0x00010001u * ((i * 7 + 3) & 3);
    __syncthreads();
    constexpr uint32_t SLICE = 3072u; // bytes of a 48-row slice
    const uint32_t base = G == 1 ? wave * SLICE : (lane / (64 / G)) * (SLICE + 8u);
    uint32_t D[49], E[48], x = 1000u, sc = 0u, fl = 100u, s1 = 3u, s2 = 5u;
    const uint32_t ge = 2, go = 10;
#pragma unroll
    for (int i = 0; i < 48; ++i) { D[i] = 1000u + lane; E[i] = 100u; }
    D[48] = 1000u;
    asm volatile("v_mov_b32 " VF ", %0\n\tv_mov_b32 " PA0 ", 0\n\tv_mov_b32 " PA1 ", 0\n\tv_mov_b32 " PB0 ", 0\n\tv_mov_b32 " PB1 ", 0" ::"v"(fl) : FIXED);
    uint32_t rng = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int col = 0; col < ncols; ++col) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t a0 = base + ((((rng >> 8) & 0xffu) * 20u) >> 8) * 8u;
        if constexpr (V == V_I32) {
            asm volatile("ds_read_b64 v[150:151], %[a0_]\n\tds_read_b64 v[154:155], %[a0_] offset:256\n\ts_waitcnt lgkmcnt(1)\n\t"
                         "v_add_u32_sdwa %[x_], %[tp_], " SEL(PA0, "0")
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [tp_] "v"(D[0])
                         : "memory", FIXED);
        } else {
            asm volatile("v_add_u32 %[x_], %[tp_], %[s1_]" : [x_] "=&v"(x) : [tp_] "v"(D[0]), [s1_] "v"(s1));
        }
        block<V, 0>(a0, D, E, x, sc, ge, go, fl, s1, s2);
        D[0] = D[48];
        if constexpr (V != V_NONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory", FIXED); // (the wrapped loads of the last blocks)
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

// what the instruction computes: out[i] = {D + sext(P.lo), D + sext(P.hi)}
__global__ void sdwa_check(const uint32_t *pp, const uint32_t *d, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x0, x1;
    asm volatile("v_add_u32_sdwa %0, %2, " "sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" "\n\t"
                 "v_add_u32_sdwa %1, %2, " "sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                 : "=&v"(x0), "=&v"(x1) : "v"(d[i]), "v"(pp[i]));
    out[2 * i] = x0;
    out[2 * i + 1] = x1;
}

struct Probe { const char *name; void (*kern)(Stamp *, uint32_t, int); const char *what; };

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    {   // (1) the instruction: every int16 pair pattern of a 16 x 16-bit sample grid against diagonal values of either sign
        const int n = 65536 * 4;
        std::vector<uint32_t> pp(n), d(n), got(2 * n);
        const uint32_t dv[4] = {0u, 123456789u, 0xfffffff0u, 0x7fff0000u};
        for (int i = 0; i < n; ++i) {
            const uint32_t lo = (uint32_t)(i & 0xffff), hi = (uint32_t)((i * 40503u + 77u) & 0xffff);
            pp[i] = lo | (hi << 16);
            d[i] = dv[i >> 16];
        }
        uint32_t *dp, *dd, *dout;
        (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&dd, n * 4); (void)hipMalloc(&dout, n * 8);
        (void)hipMemcpy(dp, pp.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sdwa_check, dim3(n / 256), dim3(256), 0, 0, dp, dd, dout, n);
        (void)hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t w0 = d[i] + (uint32_t)(int32_t)(int16_t)(pp[i] & 0xffffu), w1 = d[i] + (uint32_t)(int32_t)(int16_t)(pp[i] >> 16);
            if ((got[2 * i] != w0 || got[2 * i + 1] != w1) && bad++ < 5) printf("   MISMATCH p %08x d %08x: got %08x %08x want %08x %08x\n", pp[i], d[i], got[2 * i], got[2 * i + 1], w0, w1);
        }
        printf("v_add_u32_sdwa x, D, sext(P) src1_sel:WORD_0 / WORD_1 == D + (int16)P.lo / D + (int16)P.hi on %d cases: %ld mismatches\n", n, bad);
        (void)hipFree(dp); (void)hipFree(dd); (void)hipFree(dout);
    }
    const Probe probes[] = {
        {"i32", probe<V_I32, 1>, "1 ds_read_b64 per 4 rows, v_add_u32_sdwa + 3.5 v_max3_i32 + 2 v_subrev_u32 per row (6.5 VALU per row of 64 cells)"},
        {"i32x8", probe<V_I32, 8>, "... 8 lane groups on one shared slice"},
        {"none", probe<V_NONE, 1>, "no loads, v_add_u32 in place of the SDWA add (6.5 VALU per row)"},
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
