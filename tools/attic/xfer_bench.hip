// tools/xfer_bench.hip -- how fast do host<->device copies of caller-owned (pageable) buffers go on this box?
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/xfer_bench tools/xfer_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 38u << 20, m = 8u << 20;
    char *h = (char *)aligned_alloc(64, n), *ho = (char *)aligned_alloc(64, m);
    memset(h, 1, n); memset(ho, 2, m);
    char *d; hipMalloc(&d, n);
    hipStream_t s; hipStreamCreate(&s);
    void *pin[2]; hipHostMalloc(&pin[0], 4 << 20); hipHostMalloc(&pin[1], 4 << 20);
    hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("H2D pageable 38MB        %7.2f ms\n", now() - t);
        t = now(); hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now(); hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t2 = now(); hipHostUnregister(h);
        printf("H2D register %.2f + copy %.2f + unregister %.2f ms\n", t1 - t, t2 - t1, now() - t2);
        t = now();
        size_t off = 0; const size_t sl = 4 << 20;
        for (int k = 0; off < n; ++k) { int i = k & 1; size_t c = std::min(sl, n - off); if (k >= 2) hipEventSynchronize(ev[i]); memcpy(pin[i], h + off, c); hipMemcpyAsync(d + off, pin[i], c, hipMemcpyHostToDevice, s); hipEventRecord(ev[i], s); off += c; }
        hipStreamSynchronize(s); printf("H2D staged 2x4MB pinned  %7.2f ms\n", now() - t);
        t = now(); memcpy(pin[0], h, 4 << 20); printf("   memcpy 4MB into pinned %7.2f ms\n", now() - t);
        t = now(); hipMemcpyAsync(ho, d, m, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("D2H pageable 8MB         %7.2f ms\n", now() - t);
        t = now(); hipHostRegister(ho, m, hipHostRegisterDefault); t1 = now(); hipMemcpyAsync(ho, d, m, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t2 = now(); hipHostUnregister(ho);
        printf("D2H register %.2f + copy %.2f + unregister %.2f ms\n", t1 - t, t2 - t1, now() - t2);
        t = now(); off = 0;
        for (int k = 0; off < m; ++k) { int i = k & 1; size_t c = std::min(sl, m - off); hipMemcpyAsync(pin[i], d + off, c, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); memcpy(ho + off, pin[i], c); off += c; }
        printf("D2H staged (serial)      %7.2f ms\n", now() - t);
    }
    void *p2; double t = now(); hipHostMalloc(&p2, 4 << 20); printf("hipHostMalloc 4MB %.2f ms\n", now() - t);
    return 0;
}
