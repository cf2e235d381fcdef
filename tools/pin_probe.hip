// tools/pin_probe.hip -- how does a database file get into page-locked memory fastest, and what does the DMA make of it?
// Round 5: a one-query search through the CLI is bounded by its uploads (378 MB of group cache leave a MAP_PRIVATE mapping as
// pageable copies at ~18 GB/s; the search itself reads them at ~28 GB/s).  Variants, each on a file of <MiB> MiB written first
// and read back through the page cache:
//   A  mmap PROT_READ, MAP_PRIVATE | MAP_POPULATE                      -> hipMemcpyAsync as is (pageable: the baseline)
//   B  the same mapping                                                 -> hipHostRegister (default / ReadOnly flag), then copies
//   C  mmap PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_POPULATE          -> hipHostRegister, then copies
//   D  mmap PROT_READ, MAP_SHARED | MAP_POPULATE                        -> hipHostRegister, then copies
//   E  hipHostMalloc + pread (1 and 8 threads)                          -> copies
//   F  anonymous memory + pread, hipHostRegister                        -> copies
// Prints what every step takes and the H2D rate of 128 MiB copies from the result.
// Build: make -C tools pin_probe ; run: tools/pin_probe [MiB] [dir]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_rate(const char *what, const void *src, size_t bytes, void *dev, hipStream_t s)
{
    const size_t piece = std::min<size_t>(bytes, 128u << 20);
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now_ms();
        size_t done = 0;
        hipError_t e = hipSuccess;
        for (; done + piece <= bytes && e == hipSuccess; done += piece) e = hipMemcpyAsync(dev, (const char *)src + done, piece, hipMemcpyHostToDevice, s);
        const double t1 = now_ms();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        const double t2 = now_ms();
        if (e != hipSuccess) { printf("    %-40s copy FAILED: %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); return; }
        printf("    %-40s pass %d: %zu MiB in %.2f ms = %.1f GB/s (calls held the host %.2f ms)\n", what, rep, done >> 20, t2 - t0, (double)done / (t2 - t0) / 1e6, t1 - t0);
    }
}

static void try_register(const char *what, void *p, size_t bytes, unsigned flags, void *dev, hipStream_t s)
{
    // chunk by chunk, as the CLI would do it ahead of use
    const size_t piece = 128u << 20;
    const double t0 = now_ms();
    hipError_t e = hipSuccess;
    size_t done = 0;
    for (; done < bytes && e == hipSuccess; done += piece) e = hipHostRegister((char *)p + done, std::min(piece, bytes - done), flags);
    const double t1 = now_ms();
    if (e != hipSuccess) {
        printf("  %-42s hipHostRegister(flags %u) FAILED at %zu MiB: %s\n", what, flags, done >> 20, hipGetErrorString(e));
        (void)hipGetLastError();
        for (size_t u = 0; u + piece < done; u += piece) (void)hipHostUnregister((char *)p + u);
        return;
    }
    printf("  %-42s hipHostRegister(flags %u) of %zu MiB in %.1f ms (%.2f ms per 128 MiB)\n", what, flags, bytes >> 20, t1 - t0, (t1 - t0) * (double)piece / (double)bytes);
    copy_rate("registered", p, bytes, dev, s);
    const double t2 = now_ms();
    for (size_t u = 0; u < bytes; u += piece) (void)hipHostUnregister((char *)p + u);
    printf("    unregister %.1f ms\n", now_ms() - t2);
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 384;
    const std::string dir = argc > 2 ? argv[2] : "/tmp";
    const size_t bytes = mib << 20;
    const std::string path = dir + "/pin_probe.bin";
    {
        std::vector<char> buf(1u << 20);
        for (size_t i = 0; i < buf.size(); ++i) buf[i] = (char)(i * 131 + 7);
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) { perror("fopen"); return 1; }
        for (size_t k = 0; k < mib; ++k) fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
    }
    double t0 = now_ms();
    if (hipSetDevice(0) != hipSuccess) { printf("no GPU\n"); return 1; }
    void *dev = nullptr;
    if (hipMalloc(&dev, 128u << 20) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipStream_t s;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
    printf("runtime bring-up %.1f ms; file %zu MiB at %s\n", now_ms() - t0, mib, path.c_str());
    { // warm the pageable staging of the stream
        std::vector<char> w(16u << 20, 1);
        (void)hipMemcpyAsync(dev, w.data(), w.size(), hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
    }
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { perror("open"); return 1; }

    struct Map { const char *name; int prot, flags; };
    const Map maps[] = {{"A/B mmap PROT_READ MAP_PRIVATE|POPULATE", PROT_READ, MAP_PRIVATE | MAP_POPULATE},
                        {"C   mmap PROT_READ|WRITE MAP_PRIVATE|POPULATE", PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_POPULATE},
                        {"D   mmap PROT_READ MAP_SHARED|POPULATE", PROT_READ, MAP_SHARED | MAP_POPULATE}};
    for (const Map &m : maps) {
        t0 = now_ms();
        void *p = mmap(nullptr, bytes, m.prot, m.flags, fd, 0);
        if (p == MAP_FAILED) { perror("mmap"); continue; }
        printf("%s: mapped in %.1f ms\n", m.name, now_ms() - t0);
        if (m.prot == PROT_READ && (m.flags & MAP_PRIVATE)) copy_rate("pageable (as the CLI of round 4)", p, bytes, dev, s);
        try_register(m.name, p, bytes, hipHostRegisterDefault, dev, s);
        try_register(m.name, p, bytes, hipHostRegisterPortable, dev, s);
#ifdef hipHostRegisterReadOnly
        try_register(m.name, p, bytes, hipHostRegisterReadOnly, dev, s);
#endif
        munmap(p, bytes);
    }
    for (int threads : {1, 8}) {
        t0 = now_ms();
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { printf("hipHostMalloc failed\n"); (void)hipGetLastError(); continue; }
        const double t1 = now_ms();
        std::vector<std::thread> th;
        for (int k = 0; k < threads; ++k)
            th.emplace_back([&, k] {
                const size_t a = bytes * k / threads, b = bytes * (k + 1) / threads;
                for (size_t off = a; off < b;) { const ssize_t r = pread(fd, (char *)p + off, std::min<size_t>(b - off, 8u << 20), (off_t)off); if (r <= 0) break; off += (size_t)r; }
            });
        for (auto &t : th) t.join();
        printf("E   hipHostMalloc %.1f ms + pread with %d threads %.1f ms\n", t1 - t0, threads, now_ms() - t1);
        copy_rate("hipHostMalloc", p, bytes, dev, s);
        t0 = now_ms();
        (void)hipHostFree(p);
        printf("    hipHostFree %.1f ms\n", now_ms() - t0);
    }
    {
        t0 = now_ms();
        void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        std::vector<std::thread> th;
        for (int k = 0; k < 8; ++k)
            th.emplace_back([&, k] {
                const size_t a = bytes * k / 8, b = bytes * (k + 1) / 8;
                for (size_t off = a; off < b;) { const ssize_t r = pread(fd, (char *)p + off, std::min<size_t>(b - off, 8u << 20), (off_t)off); if (r <= 0) break; off += (size_t)r; }
            });
        for (auto &t : th) t.join();
        printf("F   anonymous memory + pread with 8 threads %.1f ms\n", now_ms() - t0);
        try_register("F   anonymous", p, bytes, hipHostRegisterPortable, dev, s);
        munmap(p, bytes);
    }
    close(fd);
    unlink(path.c_str());
    (void)hipFree(dev);
    return 0;
}
