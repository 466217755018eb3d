// micro-benchmark: issue rate of the integer VALU ops the hot kernel is made of (gfx950)
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] ^ (a[(i + 1) & 7]);
            if (OP == 1) a[i] = __builtin_amdgcn_alignbit(a[(i + 1) & 7], a[i], 6);
            if (OP == 2) a[i] = __popc(a[i]) + a[(i + 1) & 7];      // v_bcnt_u32_b32 with accumulate
            if (OP == 3) a[i] = __builtin_amdgcn_udot4(a[i], 0x40100401u, a[(i + 1) & 7], false);
            if (OP == 4) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x07050301u);
            if (OP == 5) a[i] = (a[i] << 1) | a[(i + 1) & 7];        // v_lshl_or_b32
            if (OP == 6) a[i] = min(a[i], a[(i + 1) & 7]);
            if (OP == 7) a[i] = __builtin_bitreverse32(a[i]) ^ a[(i + 1) & 7];
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char* name, int wpc) {
    uint32_t* d;
    const int blocks = 256 * wpc / 4, iters = 4096;
    hipMalloc(&d, sizeof(uint32_t) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 16, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 3);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 8 * (OP == 2 || OP == 7 ? 2 : 1);   // wave-instructions
    printf("%-22s waves/CU=%2d  %.1f G wave-instr/s  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, wpc, ops / ms / 1e6,
           1024 * 2.4e9 / (ops / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    for (int wpc : {8, 16, 32}) {
        run<0>("v_xor", wpc); run<1>("v_alignbit", wpc); run<2>("v_bcnt+v_add", wpc); run<3>("v_dot4_u32_u8", wpc);
        run<4>("v_perm", wpc); run<5>("v_lshl_or", wpc); run<6>("v_min_u32", wpc); run<7>("v_bfrev+v_xor", wpc);
    }
    return 0;
}
