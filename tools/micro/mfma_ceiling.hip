// Calibration: how fast can gfx950 issue v_mfma_f32_32x32x16_bf16 from registers (no LDS, no memory)
// with 2 waves per SIMD, for zero and for random operands?  (DVFS: the sustained clock depends on data.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// ORDER 0: both operands change with every MFMA; 1: the A operand stays for 4 consecutive MFMAs (B and the accumulator change) -- does operand
// re-use lower the power?  (no: 1 767 against 1 760 TFLOP/s)
template <int ORDER>
__global__ __launch_bounds__(256, 2) void k(const short* src, float* out, int iters, unsigned long long* cyc) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = src[(threadIdx.x * 64 + i * 8 + e) & 4095]; b[i][e] = src[(threadIdx.x * 64 + 32 + i * 8 + e) & 4095]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ORDER == 0 ? (i + u) & 3 : u], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;      // shader cycles this workgroup saw
}
static void launch(int order, int grid, const short* src, float* out, int iters, unsigned long long* cyc) {
    if (order == 0) k<0><<<grid, 256>>>(src, out, iters, cyc); else k<1><<<grid, 256>>>(src, out, iters, cyc);
}
int main(int argc, char** argv) {
    const bool quick = argc > 1;          // any argument: only the line bench.py reads (whole chip, random operands)
    short* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 8192); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    short h[4096];
    for (int grid = 512; grid >= 2; grid = grid == 512 ? 2 : 0)         // the whole chip (2 workgroups per CU), then ONE CU alone
    for (int order = 0; order < (grid == 512 ? 2 : 1); ++order)
    for (int mode = 0; mode < 2; ++mode) {
        if (quick && !(grid == 512 && order == 0 && mode == 1)) continue;
        for (int i = 0; i < 4096; ++i) h[i] = mode ? (short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;   // ~[-2,2] bf16 or zeros
        hipMemcpy(src, h, 8192, hipMemcpyHostToDevice);
        const int iters = 40000;   // 16 MFMAs per iteration: ~10 ms per launch, 5 launches (the clock governor settles within the first)
        launch(order, grid, src, out, 100, nullptr);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch(order, grid, src, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double flop = (double)grid * 4 * iters * 16 * 32768.0;
        unsigned long long hc[512]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        double mc = 0; for (int i = 0; i < grid; ++i) mc += (double)hc[i]; mc /= grid;
        // s_memtime counts shader cycles as delivered: cycles / time = the clock the SIMDs actually ran at; cycles per MFMA and SIMD from the same count
        printf("%3d workgroups, order %d, %s operands: %.3f ms  %.1f TFLOP/s = %.1f %% of 2.5 PF; %.0f shader cycles per launch = %.2f GHz delivered, %.1f cycles per MFMA and SIMD (32 = back to back)\n",
               grid, order, mode ? "random" : "zero", ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0, mc, mc / (ms * 1e-3) / 1e9, mc / ((grid >= 512 ? 2.0 : 1.0) * iters * 16));        // (2 workgroups land on two CUs: one wave per SIMD)
    }
    return 0;
}
