// copy_rate.hip — what does the memory system give a record-structured copy?  (fmt_copy_kernel's access pattern)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/copy_rate.hip -o /tmp/copy_rate && /tmp/copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint4 ld16(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16(uint8_t* p, uint4 v) { __builtin_memcpy(p, &v, 16); }
// V0: flat aligned copy, 16 B per lane, grid-stride
__global__ void flat(const uint4* s, uint4* d, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// V1: LPR lanes per record of REC bytes at r * REC (unaligned on both sides), windows min(16 l, REC - 16); U records in flight
template <int LPR, int U>
__global__ void recs(const uint8_t* s, uint8_t* d, size_t nrec, int REC, int dshift) {
    const int l = threadIdx.x % LPR;
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    uint4 v[U]; int off[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t r = g * U + u;
        const int items = (REC + 15) / 16;
        on[u] = r < nrec && l < items;
        off[u] = min(16 * l, REC - 16);
        if (on[u]) v[u] = ld16(s + r * REC + off[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t r = g * U + u;
        if (on[u]) st16(d + dshift + r * REC + off[u], v[u]);
    }
}
// V1b: the same with a 48-byte PLAN per record read first (dependent load) and two interleaved streams (file 0 / file 1)
template <int U>
__global__ void recs_plan(const uint8_t* s0, const uint8_t* s1, uint8_t* d0, uint8_t* d1, const uint4* plan, size_t ntask) {
    const int l = threadIdx.x & 31;
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    uint4 pa[U], pb[U], pc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t t = g * U + u;
        pa[u] = make_uint4(0, 0xffffffffu, 0, 0);
        if (t < ntask) { pa[u] = plan[3 * t]; pb[u] = plan[3 * t + 1]; pc[u] = plan[3 * t + 2]; }
    }
    uint4 v[U]; const uint8_t* sp[U]; uint8_t* dp[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t t = g * U + u;
        const int len = (int)(pb[u].z & 0xffffu);
        const int items = (len + 15) >> 4;
        on[u] = pa[u].y != 0xffffffffu && l < items;
        const int off = min(16 * l, len - 16);
        sp[u] = ((t & 1) ? s1 : s0) + pa[u].z + off + (pc[u].x & 0);
        dp[u] = ((t & 1) ? d1 : d0) + pa[u].x + off;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (on[u]) v[u] = ld16(sp[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) if (on[u]) st16(dp[u], v[u]);
}
__global__ void mkplan(uint4* plan, size_t ntask, int REC) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const uint32_t o = (uint32_t)((t >> 1) * REC);
    plan[3 * t] = make_uint4(o, 0x100, o, 0); plan[3 * t + 1] = make_uint4(0, 0, (uint32_t)REC, 0); plan[3 * t + 2] = make_uint4(0, 0, 0, 0);
}
// V2: the tile of T records is one contiguous run: lanes stride over it (what merging pieces across records would give)
__global__ void runs(const uint8_t* s, uint8_t* d, size_t bytes, size_t run) {
    const size_t r0 = (size_t)blockIdx.x * run;
    const size_t len = r0 < bytes ? (bytes - r0 < run ? bytes - r0 : run) : 0;
    for (size_t o = (size_t)threadIdx.x * 16; o + 16 <= len; o += (size_t)blockDim.x * 16) st16(d + 5 + r0 + o, ld16(s + 3 + r0 + o));
}
int main() {
    const int REC = 347; const size_t nrec = 10000000; const size_t bytes = nrec * REC;
    uint8_t *s, *d; hipMalloc(&s, bytes + 4096); hipMalloc(&d, bytes + 4096); hipMemset(s, 1, bytes + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-44s %.3f ms  %.2f TB/s (read+write)\n", name, ms, 2.0 * bytes / ms / 1e9);
    };
    run("flat aligned copy", [&] { hipLaunchKernelGGL(flat, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, bytes / 16); });
    run("32 lanes/record, 4 in flight, unaligned", [&] { hipLaunchKernelGGL((recs<32, 4>), dim3((unsigned)((nrec / 4 * 32 + 255) / 256)), dim3(256), 0, 0, s, d, nrec, REC, 0); });
    run("32 lanes/record, 1 in flight", [&] { hipLaunchKernelGGL((recs<32, 1>), dim3((unsigned)((nrec * 32 + 255) / 256)), dim3(256), 0, 0, s, d, nrec, REC, 0); });
    run("32 lanes/record, 2 in flight", [&] { hipLaunchKernelGGL((recs<32, 2>), dim3((unsigned)((nrec / 2 * 32 + 255) / 256)), dim3(256), 0, 0, s, d, nrec, REC, 0); });
    run("64 lanes/record, 4 in flight", [&] { hipLaunchKernelGGL((recs<64, 4>), dim3((unsigned)((nrec / 4 * 64 + 255) / 256)), dim3(256), 0, 0, s, d, nrec, REC, 0); });
    {
        const size_t ntask = nrec;             // 5 M records x 2 files
        uint4* plan; hipMalloc(&plan, 48 * ntask);
        hipLaunchKernelGGL(mkplan, dim3((unsigned)((ntask + 255) / 256)), dim3(256), 0, 0, plan, ntask, REC);
        uint8_t* s1 = s + (nrec / 2) * REC; uint8_t* d1 = d + (nrec / 2) * REC;
        run("48-byte plan first, two files interleaved, 4 in flight", [&] { hipLaunchKernelGGL((recs_plan<4>), dim3((unsigned)((ntask / 4 * 32 + 255) / 256)), dim3(256), 0, 0, s, s1, d, d1, plan, ntask); });
        run("48-byte plan first, two files interleaved, 2 in flight", [&] { hipLaunchKernelGGL((recs_plan<2>), dim3((unsigned)((ntask / 2 * 32 + 255) / 256)), dim3(256), 0, 0, s, s1, d, d1, plan, ntask); });
    }
    run("32 lanes/record, record = 352 B (aligned)", [&] { hipLaunchKernelGGL((recs<32, 4>), dim3((unsigned)((9800000 / 4 * 32 + 255) / 256)), dim3(256), 0, 0, s, d, (size_t)9800000, 352, 0); });
    run("runs of 44 KB (128 records), unaligned", [&] { hipLaunchKernelGGL(runs, dim3((unsigned)((bytes + 44415) / 44416)), dim3(256), 0, 0, s, d, bytes, (size_t)44416); });
    run("runs of 11 KB (32 records), unaligned", [&] { hipLaunchKernelGGL(runs, dim3((unsigned)((bytes + 11103) / 11104)), dim3(256), 0, 0, s, d, bytes, (size_t)11104); });
    return 0;
}
