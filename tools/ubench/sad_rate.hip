// micro-benchmark: issue rate of the SAD family on gfx950 — could a byte-wise sliding compare (V_QSAD_PK_U16_U8: four
// 4-byte windows of an 8-byte source against one 4-byte reference per instruction) replace the 2-bit
// alignbit / xor / popcount test of the diagonal scan?   
//   hipcc --offload-arch=gfx950 -O3 sad_rate.hip -o sad_rate && ./sad_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void k(uint64_t* out, int iters, uint32_t seed) {
    uint64_t a[8];
    uint32_t b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (uint64_t)seed * (threadIdx.x + i + 1) * 0x9e3779b97f4a7c15ull; b[i] = seed * (i + 3); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_qsad_pk_u16_u8(a[(i + 1) & 7], b[i], a[i]);
            if (OP == 1) a[i] = __builtin_amdgcn_mqsad_pk_u16_u8(a[(i + 1) & 7], b[i], a[i]);
            if (OP == 2) b[i] = __builtin_amdgcn_sad_u8(b[(i + 1) & 7], b[i], b[(i + 2) & 7]);
            if (OP == 3) b[i] = __builtin_amdgcn_msad_u8(b[(i + 1) & 7], b[i], b[(i + 2) & 7]);
            if (OP == 4) b[i] = __popc(__builtin_amdgcn_alignbit(b[(i + 1) & 7], b[i], 6) ^ b[(i + 2) & 7]) + b[i];   // the scan's triple (+add)
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char* name, int wpc, int instr_per_op) {
    uint64_t* d;
    const int blocks = 256 * wpc / 4, iters = 4096;
    hipMalloc(&d, sizeof(uint64_t) * blocks * 256);
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
    const double ops = (double)blocks * 4 * iters * 8;   // wave-level "ops" (one per inner statement)
    printf("%-34s waves/CU=%2d  %.1f G ops/s  -> %.2f cycles/op/SIMD @2.4GHz (%d instr/op)\n", name, wpc, ops / ms / 1e6,
           1024 * 2.4e9 / (ops / (ms * 1e-3)), instr_per_op);
    hipFree(d);
}
int main() {
    for (int wpc : {8, 16}) {
        run<0>("v_qsad_pk_u16_u8 (4 windows)", wpc, 1);
        run<1>("v_mqsad_pk_u16_u8 (4 windows)", wpc, 1);
        run<2>("v_sad_u8", wpc, 1);
        run<3>("v_msad_u8", wpc, 1);
        run<4>("alignbit+xor+bcnt(+add) (1 diag)", wpc, 3);
    }
    return 0;
}
