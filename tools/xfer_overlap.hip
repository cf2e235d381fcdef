// tools/xfer_overlap.hip -- does an H2D copy on a stream of its own run BESIDE a kernel that fills the GPU?  Round 4: in the
// upload-inclusive pass the copies of chunk k+1 started only when the search of chunk k had ended (rocprofv3 memory-copy
// trace).  Variants: how many non-blocking streams the process has created before (ROCclr maps streams onto a limited number of
// hardware queues: GPU_MAX_HW_QUEUES, default 4), in which order, and whether the copy stream is a high-priority stream.
// Prints, per variant, when the copy completed relative to the kernel's launch and its end.
// Build: make -C tools xfer_overlap ; run: tools/xfer_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(256) void spin(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// nstreams: non-blocking streams created first (the kernel runs on the first); copy_slot: which of them carries the copy
// (-1: an extra high-priority stream)
static void variant(const char *name, int nstreams, int copy_slot, void *pinned, void *dev, size_t bytes, unsigned long long *out)
{
    std::vector<hipStream_t> st(nstreams);
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStream_t cs;
    if (copy_slot < 0) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        (void)hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, hi);
    } else cs = st[copy_slot];
    // touch every stream once (a stream's hardware queue may be chosen at its first use)
    for (auto &s : st) { (void)hipMemsetAsync(dev, 0, 256, s); (void)hipStreamSynchronize(s); }
    (void)hipMemcpyAsync(dev, pinned, 4096, hipMemcpyHostToDevice, cs);
    (void)hipStreamSynchronize(cs);
    hipEvent_t ek, ec;
    (void)hipEventCreate(&ek);
    (void)hipEventCreate(&ec);
    const double t0 = now_ms();
    hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, st[0], out, 100000000ull * 60 / 1000); // ~60 ms at 100 MHz, 8 workgroups per CU
    (void)hipEventRecord(ek, st[0]);
    (void)hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, cs);
    (void)hipEventRecord(ec, cs);
    const double t1 = now_ms();
    (void)hipEventSynchronize(ec);
    const double tc = now_ms();
    (void)hipEventSynchronize(ek);
    const double tk = now_ms();
    printf("%-58s issue %.2f ms, copy done at %7.2f ms, kernel done at %7.2f ms -> %s\n", name, t1 - t0, tc - t0, tk - t0, tc < tk - 5 ? "BESIDE the kernel" : "behind the kernel");
    (void)hipEventDestroy(ek);
    (void)hipEventDestroy(ec);
    for (auto &s : st) (void)hipStreamDestroy(s);
    if (copy_slot < 0) (void)hipStreamDestroy(cs);
}

// Several copies queued back to back on the high-priority stream beside the kernel: two 128 MB page-locked copies, then 1 MB from
// pageable memory (what a plan's queue copy is); when does each complete?
static void back_to_back(void *pinned, void *dev, void *dev2, size_t bytes, unsigned long long *out)
{
    hipStream_t ks, cs;
    int lo = 0, hi = 0;
    (void)hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    (void)hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, hi);
    std::vector<char> pageable(1u << 20, 3);
    (void)hipMemcpyAsync(dev, pinned, 4096, hipMemcpyHostToDevice, cs);
    (void)hipMemcpyAsync(dev2, pageable.data(), 4096, hipMemcpyHostToDevice, cs);
    (void)hipStreamSynchronize(cs);
    hipEvent_t e[3], ek;
    for (auto &x : e) (void)hipEventCreate(&x);
    (void)hipEventCreate(&ek);
    const double t0 = now_ms();
    hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, ks, out, 100000000ull * 60 / 1000);
    (void)hipEventRecord(ek, ks);
    (void)hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, cs);
    (void)hipEventRecord(e[0], cs);
    (void)hipMemcpyAsync(dev2, pinned, bytes, hipMemcpyHostToDevice, cs);
    (void)hipEventRecord(e[1], cs);
    const double t1 = now_ms();
    (void)hipMemcpyAsync(dev, pageable.data(), pageable.size(), hipMemcpyHostToDevice, cs);
    const double t2 = now_ms();
    (void)hipEventRecord(e[2], cs);
    double tc[3];
    for (int k = 0; k < 3; ++k) { (void)hipEventSynchronize(e[k]); tc[k] = now_ms() - t0; }
    (void)hipEventSynchronize(ek);
    printf("back to back on a high-priority stream: issue %.2f ms (+ %.2f ms in the pageable copy's call); 128 MB done at %.2f ms, second 128 MB at %.2f ms, 1 MB from pageable memory at %.2f ms; kernel done at %.2f ms\n",
           t1 - t0, t2 - t1, tc[0], tc[1], tc[2], now_ms() - t0);
    (void)hipStreamDestroy(ks);
    (void)hipStreamDestroy(cs);
}

// The library's slot re-use: a third stream WAITS for the kernel's end (hipStreamWaitEvent) and has work queued behind the wait;
// then a copy goes out on the high-priority stream.  nnormal: ordinary streams created before (kernel stream, [extra], waiting stream)
static void with_waiter(int extra, bool waiter_waits, void *pinned, void *dev, size_t bytes, unsigned long long *out)
{
    hipStream_t ks, us, cs;
    std::vector<hipStream_t> ex(extra);
    int lo = 0, hi = 0;
    (void)hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
    for (auto &x : ex) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&us, hipStreamNonBlocking);
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    (void)hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, hi);
    for (auto &x : ex) { (void)hipMemsetAsync(dev, 0, 256, x); (void)hipStreamSynchronize(x); }
    (void)hipMemsetAsync(dev, 0, 256, us);
    (void)hipStreamSynchronize(us);
    (void)hipMemcpyAsync(dev, pinned, 4096, hipMemcpyHostToDevice, cs);
    (void)hipStreamSynchronize(cs);
    hipEvent_t ek, ec, eu;
    (void)hipEventCreateWithFlags(&ek, hipEventDisableTiming);
    (void)hipEventCreate(&ec);
    (void)hipEventCreateWithFlags(&eu, hipEventDisableTiming);
    const double t0 = now_ms();
    hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, ks, out, 100000000ull * 60 / 1000);
    (void)hipEventRecord(ek, ks);
    if (waiter_waits) (void)hipStreamWaitEvent(us, ek, 0);
    (void)hipMemsetAsync((char *)dev + (64u << 20), 0, 1u << 20, us); // (a fill kernel: finds no wave slot beside the spin kernel either way)
    (void)hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, cs);
    (void)hipEventRecord(ec, cs);
    (void)hipStreamWaitEvent(us, ec, 0);
    (void)hipEventRecord(eu, us);
    const double t1 = now_ms();
    (void)hipEventSynchronize(ec);
    const double tc = now_ms();
    (void)hipStreamSynchronize(ks);
    const double tk = now_ms();
    (void)hipStreamSynchronize(us);
    printf("%d ordinary streams, one of them %s; copy on the high-priority stream: issue %.2f ms, copy done at %.2f ms, kernel at %.2f ms -> %s\n", 2 + extra,
           waiter_waits ? "WAITING for the kernel's end with a fill queued behind the wait" : "with a fill queued (no wait)", t1 - t0, tc - t0, tk - t0, tc < tk - 5 ? "BESIDE" : "behind");
    (void)hipStreamDestroy(ks); (void)hipStreamDestroy(us); (void)hipStreamDestroy(cs);
    for (auto &x : ex) (void)hipStreamDestroy(x);
}

int main()
{
    const size_t bytes = 128u << 20;
    void *pinned, *dev;
    unsigned long long *out;
    (void)hipHostMalloc(&pinned, bytes, hipHostMallocPortable);
    memset(pinned, 1, bytes);
    (void)hipMalloc(&dev, bytes);
    (void)hipMalloc(&out, 256 * 8 * 8);
    (void)hipMemcpy(dev, pinned, bytes, hipMemcpyHostToDevice);
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s; 128 MB page-locked -> device beside a 60 ms kernel that fills every CU\n", q ? q : "(unset)");
    variant("2 streams, copy on the 2nd", 2, 1, pinned, dev, bytes, out);
    variant("4 streams, copy on the 4th", 4, 3, pinned, dev, bytes, out);
    variant("5 streams, copy on the 4th (the library's order)", 5, 3, pinned, dev, bytes, out);
    variant("5 streams, copy on the 5th", 5, 4, pinned, dev, bytes, out);
    variant("5 streams, copy on the 2nd", 5, 1, pinned, dev, bytes, out);
    variant("8 streams, copy on the 8th", 8, 7, pinned, dev, bytes, out);
    variant("5 streams + a high-priority copy stream", 5, -1, pinned, dev, bytes, out);
    void *dev2;
    (void)hipMalloc(&dev2, bytes);
    back_to_back(pinned, dev, dev2, bytes, out);
    for (int extra = 0; extra < 3; ++extra) { with_waiter(extra, false, pinned, dev, bytes, out); with_waiter(extra, true, pinned, dev, bytes, out); }
    return 0;
}
