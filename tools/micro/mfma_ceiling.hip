// Calibration: how fast can gfx950 issue v_mfma_f32_32x32x16_bf16 from registers (no LDS, no memory)
// with 2 waves per SIMD, for zero and for random operands?  (DVFS: the sustained clock depends on data.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256, 2) void k(const short* src, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = src[(threadIdx.x * 64 + i * 8 + e) & 4095]; b[i][e] = src[(threadIdx.x * 64 + 32 + i * 8 + e) & 4095]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    short* src; float* out;
    hipMalloc(&src, 8192); hipMalloc(&out, 512 * 256 * 4);
    short h[4096];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 4096; ++i) h[i] = mode ? (short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;   // ~[-2,2] bf16 or zeros
        hipMemcpy(src, h, 8192, hipMemcpyHostToDevice);
        const int iters = 4000;   // 16 MFMAs per iteration
        k<<<512, 256>>>(src, out, 100);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k<<<512, 256>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double flop = 512.0 * 4 * iters * 16 * 32768.0;
        printf("%s operands: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", mode ? "random" : "zero", ms, flop / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (2.0 * iters * 16));
    }
    return 0;
}
