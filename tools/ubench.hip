// tools/ubench.hip -- instruction-rate microbenchmarks that size the DP kernel
// on gfx950: packed-int16 VALU rate vs. waves per SIMD, dependent-issue cost,
// and LDS profile-read rate.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

template <int ILP>
__global__ __launch_bounds__(256) void pk_rate(uint32_t *out, uint32_t c1, uint32_t c2, int iters)
{
    v2s x[ILP];
    const v2s a = __builtin_bit_cast(v2s, c1), b = __builtin_bit_cast(v2s, c2);
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = __builtin_bit_cast(v2s, (uint32_t)(threadIdx.x + i));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            x[i] = __builtin_elementwise_add_sat(x[i], a);
            x[i] = __builtin_elementwise_max(x[i], b);
            x[i] = __builtin_bit_cast(v2s, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, x[i]), __builtin_bit_cast(v2u, a)));
            x[i] = __builtin_elementwise_max(x[i], a);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc ^= __builtin_bit_cast(uint32_t, x[i]);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// plain 32-bit integer VALU (v_add_u32 / v_max_i32 / v_max3_i32 / v_sub_u32): issue rate vs packed int16
template <int ILP>
__global__ __launch_bounds__(256) void i32_rate(uint32_t *out, int c1, int c2, int iters)
{
    int x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            int t;
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(c1));
            asm volatile("v_max_i32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(c2));
            asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(c1));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(t), "v"(c2), "v"(c1));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc ^= (uint32_t)x[i];
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// packed fp16 with the gfx950 three-operand maximum (candidate cell arithmetic for a 2047-ceiling first pass)
template <int ILP>
__global__ __launch_bounds__(256) void f16_rate(uint32_t *out, uint32_t c1, uint32_t c2, int iters)
{
    uint32_t x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = 0x3c003c00u + (threadIdx.x & 1) + i; // ~1.0 in both halves
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            uint32_t t;
            asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(c1));
            asm volatile("v_pk_maximum3_f16 %0, %1, %2, 0" : "=v"(x[i]) : "v"(t), "v"(c2));
            asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x[i]), "v"(c1));
            asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(x[i]) : "v"(t), "v"(c2), "v"(c1));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc ^= x[i];
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// Issue rate of single VALU instruction kinds under a sustained load, in CORE CLOCK cycles: s_memtime
// (core clock counter) and s_memrealtime (100 MHz) are read inside the kernel, so neither launch overhead
// nor an assumed frequency enters.  OP: 0 v_pk_add_f16, 1 v_pk_maximum3_f16, 2 v_pk_add_i16 clamp,
// 3 v_pk_max_i16, 4 v_perm_b32, 5 v_mov_b32, 6 alternating add_f16 / maximum3_f16.  8 independent chains.
template <int OP>
__global__ __launch_bounds__(256) void clk_probe(unsigned long long *out, uint32_t c1, uint32_t c2, int iters)
{
    extern __shared__ uint32_t pad_lds[]; // sized by the host so that exactly `waves per SIMD` workgroups fit a CU
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0x3c003c00u + threadIdx.x + i;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
                if constexpr (OP == 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, 0" : "+v"(x[i]) : "v"(c2));
                if constexpr (OP == 2) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(c1));
                if constexpr (OP == 3) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
                if constexpr (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c2), "s"(0x07060302u));
                if constexpr (OP == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(c2));
                if constexpr (OP == 6) {
                    if (k == 0) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
                    else asm volatile("v_pk_maximum3_f16 %0, %0, %1, 0" : "+v"(x[i]) : "v"(c2));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0 + (acc == 0x12345678u); out[blockIdx.x * 2 + 1] = r1 - r0; }
}

// LDS: ds_read_b64 of 4 profile rows at residue*8 + imm, 16 reads per "column"
__global__ __launch_bounds__(256) void lds_rate(uint32_t *out, const uint32_t *res, int iters)
{
    __shared__ uint2 prof[4][8 * 32];
    for (int i = threadIdx.x; i < 4 * 8 * 32; i += 256) ((uint2 *)prof)[i] = make_uint2(i, i * 3);
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    uint32_t r = res[threadIdx.x];
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t a = (r & 31);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            uint2 p = prof[wv][rb * 32 + a];
            acc += p.x ^ p.y;
        }
        r = (r >> 5) | (r << 27);
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// PMC calibration: a known number of bytes moved with the DP kernel's access width (8 B per lane)
__global__ __launch_bounds__(256) void stream_read8(const uint2 *src, uint32_t *out, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 v = src[i]; acc += v.x ^ v.y; }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void stream_write8(uint2 *dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_uint2((uint32_t)i, 7u);
}

// the same bytes moved the way the DP kernels spill since the fixed-register column loop: {H, F} entries of 8 B,
// read / written as two dword accesses per lane (offset 0 and 4)
__global__ __launch_bounds__(256) void stream_read4x2(const uint32_t *src, uint32_t *out, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t a, b;
        asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:4\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(src + 2 * i) : "memory");
        acc += a ^ b;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void stream_write4x2(uint32_t *dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %0, %2, off offset:4" ::"v"(dst + 2 * i), "v"((uint32_t)i), "v"(7u) : "memory");
    }
}

template <class F>
static double time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "calib")) {
        // 1 GiB read + 1 GiB written, 8 B per lane: compare with FETCH_SIZE / WRITE_SIZE of these two kernels
        const size_t n = (1ull << 30) / 8;
        uint2 *buf; uint32_t *o;
        hipMalloc(&buf, n * 8); hipMalloc(&o, 4096);
        hipMemset(buf, 1, n * 8);
        hipLaunchKernelGGL(stream_write8, dim3(2048), dim3(256), 0, 0, buf, n);
        hipLaunchKernelGGL(stream_read8, dim3(2048), dim3(256), 0, 0, (const uint2 *)buf, o, n);
        hipLaunchKernelGGL(stream_write4x2, dim3(2048), dim3(256), 0, 0, (uint32_t *)buf, n);
        hipLaunchKernelGGL(stream_read4x2, dim3(2048), dim3(256), 0, 0, (const uint32_t *)buf, o, n);
        hipDeviceSynchronize();
        printf("calib: wrote %zu bytes (stream_write8), read %zu bytes (stream_read8)\n", n * 8, n * 8);
        return 0;
    }
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    if (argc > 1 && !strcmp(argv[1], "clk")) {
        const int iters = 100000;
        unsigned long long *o; hipMalloc(&o, cus * 8 * 16);
        const char *names[7] = {"v_pk_add_f16", "v_pk_maximum3_f16", "v_pk_add_i16 clamp", "v_pk_max_i16", "v_perm_b32", "v_mov_b32", "add_f16/maximum3_f16"};
        for (int wps = 1; wps <= 8; wps *= 2) {
            for (int op = 0; op < 7; ++op) {
                const int nb = cus * wps;
                const size_t lds = (size_t)(160 * 1024 / wps) - 1024; // exactly wps workgroups per CU: one wave of each per SIMD
                auto launch = [&](auto kern) {
                    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, o, 0x3c003c00u, 0x40004000u, iters);
                };
                switch (op) {
                case 0: launch(clk_probe<0>); break; case 1: launch(clk_probe<1>); break; case 2: launch(clk_probe<2>); break;
                case 3: launch(clk_probe<3>); break; case 4: launch(clk_probe<4>); break; case 5: launch(clk_probe<5>); break;
                default: launch(clk_probe<6>); break;
                }
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(nb * 2);
                hipMemcpy(h.data(), o, nb * 16, hipMemcpyDeviceToHost);
                double cyc = 0, ref = 0, cmax = 0, cmin = 1e30;
                for (int i = 0; i < nb; ++i) { cyc += (double)h[2 * i]; ref += (double)h[2 * i + 1]; cmax = std::max(cmax, (double)h[2 * i]); cmin = std::min(cmin, (double)h[2 * i]); }
                // workgroups start staggered, so the early and late ones run partly alone: the slowest one shared its
                // SIMDs with wps-1 others all the way and gives the loaded rate
                printf("clk %-22s waves/SIMD=%d: %6.0f MHz, %5.2f core cycles per wave instruction per SIMD (from the slowest workgroup; cycles min %.3g avg %.3g max %.3g)\n",
                       names[op], wps, cyc / ref * 100.0, cmax / ((double)iters * 16) / wps, cmin, cyc / nb, cmax);
            }
        }
        return 0;
    }
    printf("device %s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    uint32_t *out, *res;
    hipMalloc(&out, 4096);
    hipMalloc(&res, 4096);
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 2654435761u * (i + 1);
    hipMemcpy(res, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        auto run = [&](auto kern, int ilp, const char *name) {
            double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, out, 0x00030003u, 0x00010001u, iters); }, 3);
            double ops = (double)cus * wps * 4 /*waves*/ * iters * ilp * 4.0;
            printf("pk_rate %-6s waves/SIMD=%d : %.3f ms  %.2f T wave-instr... %.1f Gwaveinstr/s  = %.2f cycles/instr/SIMD at 2.4GHz\n", name, wps, ms,
                   ops / 1e12, ops / ms / 1e6, (double)cus * 4 * 2.4e9 / (ops / (ms * 1e-3)));
        };
        run(pk_rate<1>, 1, "ilp1");
        run(pk_rate<2>, 2, "ilp2");
        run(pk_rate<8>, 8, "ilp8");
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        auto run = [&](auto kern, int ilp, const char *name) {
            double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, out, 0x3c003c00u, 0x40004000u, iters); }, 3);
            double ops = (double)cus * wps * 4 * iters * ilp * 4.0;
            printf("f16_rate %-6s waves/SIMD=%d : %.3f ms = %.2f cycles/instr/SIMD at 2.4GHz\n", name, wps, ms, (double)cus * 4 * 2.4e9 / (ops / (ms * 1e-3)));
        };
        run(f16_rate<1>, 1, "ilp1");
        run(f16_rate<2>, 2, "ilp2");
        run(f16_rate<8>, 8, "ilp8");
    }
    if (argc > 1 && !strcmp(argv[1], "f16")) return 0;
    for (int wps = 1; wps <= 8; wps *= 2) {
        auto run = [&](auto kern, int ilp, const char *name) {
            double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, out, 3, 1, iters); }, 3);
            double ops = (double)cus * wps * 4 * iters * ilp * 4.0;
            printf("i32_rate %-6s waves/SIMD=%d : %.3f ms = %.2f cycles/instr/SIMD at 2.4GHz\n", name, wps, ms, (double)cus * 4 * 2.4e9 / (ops / (ms * 1e-3)));
        };
        run(i32_rate<1>, 1, "ilp1");
        run(i32_rate<8>, 8, "ilp8");
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        double ms = time_ms([&] { hipLaunchKernelGGL(lds_rate, dim3(cus * wps), dim3(256), 0, 0, out, res, iters); }, 3);
        double reads = (double)cus * wps * 4 * iters * 8;
        printf("lds_rate ds_read_b64 waves/SIMD=%d : %.3f ms, %.2f cycles per wave-read per CU at 2.4GHz\n", wps, ms,
               (double)cus * 2.4e9 / (reads / (ms * 1e-3)));
    }
    return 0;
}
