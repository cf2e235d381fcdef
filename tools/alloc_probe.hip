// tools/alloc_probe.hip -- what does creating device buffers cost on this box?  hipMalloc / hipFree by size, one slab against many
// buffers, hipHostMalloc, hipMallocAsync from a warmed pool, and hipMalloc beside a running kernel.  (round 6: the chunk slots'
// buffers are created inside the caller's clock, like the reference's clCreateBuffer, FPGAsearch.c:85-96)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long cycles, unsigned *sink) { unsigned long long t0 = __builtin_readcyclecounter(); unsigned v = 0; while (__builtin_readcyclecounter() - t0 < cycles) v += 1; if (v == 12345u) *sink = v; }
int main()
{
    CK(hipSetDevice(0));
    void *w; CK(hipMalloc(&w, 1 << 20)); CK(hipFree(w));
    const size_t sizes[] = {1u << 20, 16u << 20, 64u << 20, 128u << 20, 165u << 20, 384u << 20, 1024u << 20, 1536u << 20};
    for (int rep = 0; rep < 2; ++rep)
        for (size_t sz : sizes) {
            void *p; double t0 = now(); CK(hipMalloc(&p, sz)); double t1 = now(); CK(hipMemset(p, 1, 4096)); CK(hipDeviceSynchronize()); double t2 = now(); CK(hipFree(p)); double t3 = now();
            printf("hipMalloc %5zu MB: %7.3f ms   first touch %6.3f ms   hipFree %7.3f ms\n", sz >> 20, t1 - t0, t2 - t1, t3 - t2);
        }
    {   // a slot as the library makes it: nine buffers, against one slab of the same total
        const size_t parts[] = {128u << 20, 165u << 20, 27u << 20, 54u << 20, 1u << 20, 1u << 20, 1u << 20, 1u << 20, 1u << 20};
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<void *> ps; double t0 = now(); size_t tot = 0;
            for (size_t s : parts) { void *p; CK(hipMalloc(&p, s)); ps.push_back(p); tot += s; }
            double t1 = now(); void *slab; CK(hipMalloc(&slab, tot)); double t2 = now();
            printf("nine buffers (%zu MB): %.3f ms; one slab: %.3f ms\n", tot >> 20, t1 - t0, t2 - t1);
            for (void *p : ps) CK(hipFree(p)); CK(hipFree(slab));
        }
    }
    for (size_t sz : {64u << 10, 1u << 20, 8u << 20, 32u << 20}) {
        void *p; double t0 = now(); CK(hipHostMalloc(&p, sz, hipHostMallocPortable)); double t1 = now(); CK(hipHostFree(p)); double t2 = now();
        printf("hipHostMalloc %6zu KB: %7.3f ms   hipHostFree %7.3f ms\n", sz >> 10, t1 - t0, t2 - t1);
    }
    {   // stream-ordered pool, warmed
        hipStream_t s; CK(hipStreamCreate(&s)); hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        void *p; double t0 = now(); CK(hipMallocAsync(&p, 1536u << 20, s)); CK(hipStreamSynchronize(s)); double t1 = now(); CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        printf("hipMallocAsync 1536 MB, cold pool: %.3f ms; hipFreeAsync %.3f ms\n", t1 - t0, t2 - t1);
        for (size_t sz : {128u << 20, 384u << 20, 1024u << 20}) {
            t0 = now(); CK(hipMallocAsync(&p, sz, s)); CK(hipStreamSynchronize(s)); t1 = now(); CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s)); t2 = now();
            printf("hipMallocAsync %4zu MB, warm pool: %.3f ms; hipFreeAsync %.3f ms\n", sz >> 20, t1 - t0, t2 - t1);
        }
    }
    {   // beside a running kernel: does hipMalloc wait, does the kernel notice?
        hipStream_t s; CK(hipStreamCreate(&s)); unsigned *sink; CK(hipMalloc(&sink, 4));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int with = 0; with < 2; ++with) {
            CK(hipEventRecord(a, s)); spin<<<1024, 256, 0, s>>>(50000000ull, sink); CK(hipEventRecord(b, s));
            double t0 = now(); void *p = nullptr; if (with) CK(hipMalloc(&p, 384u << 20)); double t1 = now();
            CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("kernel %.3f ms%s\n", ms, with ? "" : " (alone)"); if (with) { printf("  hipMalloc 384 MB beside it: %.3f ms\n", t1 - t0); CK(hipFree(p)); }
        }
    }
    return 0;
}
