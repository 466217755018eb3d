// pcie_rate.hip — what does the host link of this box give the pipe?  H2D alone, D2H alone, both at once, for transfers of 16 / 64 /
// 256 MiB between HBM and page-locked host memory allocated the way the pipe allocates it (anonymous memory on transparent huge
// pages, touched, hipHostRegister'ed — aqc_host_alloc), through the copy engines (hipMemcpyAsync) and through a copy KERNEL that
// reads / writes the mapped host memory itself.  The pipe's pinned -> pinned figure (bench.py) moves 3.47 GB up and 3.44 GB down per
// 10 M reads: 100 - 105 Mreads/s are 36 GB/s each way at once.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_rate.hip -o /tmp/pcie_rate && /tmp/pcie_rate
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void* pinned(size_t n) {
    const size_t H = 2u << 20, len = (n + H - 1) & ~(H - 1);
    uint8_t* base = (uint8_t*)mmap(nullptr, len + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    uint8_t* p = (uint8_t*)(((uintptr_t)base + H - 1) & ~(uintptr_t)(H - 1));
    madvise(p, len, MADV_HUGEPAGE);
    memset(p, 7, len);
    if (hipHostRegister(p, len, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { printf("hipHostRegister failed\n"); return nullptr; }
    return p;
}
__global__ void blit(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
    const size_t MAXB = 256u << 20;
    uint8_t *h_up = (uint8_t*)pinned(MAXB), *h_dn = (uint8_t*)pinned(MAXB), *d_up, *d_dn;
    if (!h_up || !h_dn) return 1;
    hipMalloc(&d_up, MAXB); hipMalloc(&d_dn, MAXB); hipMemset(d_dn, 3, MAXB);
    uint8_t *m_up = nullptr, *m_dn = nullptr;
    hipHostGetDevicePointer((void**)&m_up, h_up, 0); hipHostGetDevicePointer((void**)&m_dn, h_dn, 0);
    hipStream_t su, sd;
    hipStreamCreateWithFlags(&su, hipStreamNonBlocking); hipStreamCreateWithFlags(&sd, hipStreamNonBlocking);
    printf("%-10s %-8s %12s %12s %24s\n", "transfer", "engine", "H2D alone", "D2H alone", "both at once (up / down)");
    for (size_t mb : {16, 64, 256}) {
        const size_t n = mb << 20;
        const int reps = (int)(2048 / mb);
        for (int kernel = 0; kernel < 2; ++kernel) {
            auto up = [&] { if (kernel) hipLaunchKernelGGL(blit, dim3(512), dim3(256), 0, su, (const uint4*)m_up, (uint4*)d_up, n / 16); else hipMemcpyAsync(d_up, h_up, n, hipMemcpyHostToDevice, su); };
            auto dn = [&] { if (kernel) hipLaunchKernelGGL(blit, dim3(512), dim3(256), 0, sd, (const uint4*)d_dn, (uint4*)m_dn, n / 16); else hipMemcpyAsync(h_dn, d_dn, n, hipMemcpyDeviceToHost, sd); };
            up(); dn(); hipDeviceSynchronize();
            double t = now(); for (int i = 0; i < reps; ++i) up(); hipStreamSynchronize(su); const double a = (double)n * reps / (now() - t) / 1e9;
            t = now(); for (int i = 0; i < reps; ++i) dn(); hipStreamSynchronize(sd); const double b = (double)n * reps / (now() - t) / 1e9;
            t = now(); for (int i = 0; i < reps; ++i) { up(); dn(); } hipStreamSynchronize(su); hipStreamSynchronize(sd); const double c = (double)n * reps / (now() - t) / 1e9;
            printf("%4zu MiB   %-8s %9.1f GB/s %9.1f GB/s %14.1f / %.1f GB/s\n", mb, kernel ? "kernel" : "copy", a, b, c, c);
        }
    }
    return 0;
}
