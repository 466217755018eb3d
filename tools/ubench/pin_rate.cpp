// pin_rate.cpp — what page-locking the pipe's rings costs in a fresh process, three ways (hipcc tools/ubench/pin_rate.cpp -o /tmp/pin_rate):
//   A  hipHostMalloc(portable)                                  what aqc_host_alloc does
//   B  mmap + first touch (4 KiB pages) + hipHostRegister
//   C  mmap 2 MiB-aligned + MADV_HUGEPAGE + first touch + hipHostRegister
// each for 16 buffers of 64 MiB, then one H2D copy from every buffer (the DMA rate must not differ).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    const size_t N = 64u << 20;
    const int K = 16;
    CK(hipSetDevice(0));
    void* dev;
    CK(hipMalloc(&dev, N));
    CK(hipDeviceSynchronize());
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[128] = "?";
    if (f) { if (!fgets(line, sizeof(line), f)) line[0] = 0; fclose(f); }
    printf("transparent_hugepage/enabled: %s", line);
    auto h2d = [&](std::vector<void*>& v) {
        const double t0 = now();
        for (void* p : v) CK(hipMemcpyAsync(dev, p, N, hipMemcpyHostToDevice, 0));
        CK(hipStreamSynchronize(0));
        return (double)N * v.size() / (now() - t0) / 1e9;
    };
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<void*> v(K);
        double t_map = 0, t_touch = 0, t_pin = 0;
        const double t0 = now();
        for (int k = 0; k < K; ++k) {
            if (variant == 0) {
                const double a = now();
                CK(hipHostMalloc(&v[k], N, hipHostMallocPortable));
                t_pin += now() - a;
            } else {
                double a = now();
                uint8_t* p = (uint8_t*)mmap(nullptr, N + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                p = (uint8_t*)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
                if (variant == 2) madvise(p, N, MADV_HUGEPAGE);
                t_map += now() - a;
                a = now();
                for (size_t o = 0; o < N; o += 4096) p[o] = 1;
                t_touch += now() - a;
                a = now();
                CK(hipHostRegister(p, N, hipHostRegisterPortable));
                t_pin += now() - a;
                v[k] = p;
            }
        }
        const double total = now() - t0;
        const double bw = h2d(v), bw2 = h2d(v);
        printf("%c: %d x %zu MiB in %.3f s (%.2f GB/s; map %.3f, touch %.3f, pin %.3f)   H2D %.1f then %.1f GB/s\n", "ABC"[variant], K, N >> 20, total,
               (double)N * K / total / 1e9, t_map, t_touch, t_pin, bw, bw2);
    }
    // the same with 4 threads doing C on 4 buffers each (the pipe's workers page-lock their own buffers side by side)
    {
        std::vector<void*> v(K);
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                CK(hipSetDevice(0));
                for (int k = t; k < K; k += 4) {
                    uint8_t* p = (uint8_t*)mmap(nullptr, N + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    p = (uint8_t*)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
                    madvise(p, N, MADV_HUGEPAGE);
                    for (size_t o = 0; o < N; o += 4096) p[o] = 1;
                    CK(hipHostRegister(p, N, hipHostRegisterPortable));
                    v[k] = p;
                }
            });
        for (auto& t : th) t.join();
        const double total = now() - t0;
        printf("C x 4 threads: %.3f s (%.2f GB/s)   H2D %.1f GB/s\n", total, (double)N * K / total / 1e9, h2d(v));
    }
    {
        std::vector<void*> v(K);
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                CK(hipSetDevice(0));
                for (int k = t; k < K; k += 4) CK(hipHostMalloc(&v[k], N, hipHostMallocPortable));
            });
        for (auto& t : th) t.join();
        const double total = now() - t0;
        printf("A x 4 threads: %.3f s (%.2f GB/s)   H2D %.1f GB/s\n", total, (double)N * K / total / 1e9, h2d(v));
    }
    return 0;
}
