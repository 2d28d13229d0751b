// Calibration: how many bytes per second can ONE CU stream (loads / stores / copy / LDS-DMA) when few or all CUs are
// active, from HBM (footprint > Infinity Cache), from the Infinity Cache and from L2?  One 512-thread workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int U = 8;
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* src, char* dst, long bytes_per_wg, int reps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* s = src + (long)blockIdx.x * bytes_per_wg;
    char* d = dst + (long)blockIdx.x * bytes_per_wg;
    const long step = 512L * 16 * U;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (long off = 0; off < bytes_per_wg; off += step) {
            if constexpr (MODE == 0 || MODE == 2) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *(const u32x4*)(s + off + (long)u * 8192 + threadIdx.x * 16);
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) acc ^= v[u];
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) *(u32x4*)(d + off + (long)u * 8192 + threadIdx.x * 16) = v[u];
                }
            } else if constexpr (MODE == 1) {
                const u32x4 v = {(unsigned)off, 1, 2, 3};
#pragma unroll
                for (int u = 0; u < U; ++u) *(u32x4*)(d + off + (long)u * 8192 + threadIdx.x * 16) = v;
            } else {   // LDS-DMA into a 64 KiB ring, never read
                const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
                const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned l = lds0 + u * 8192 + wave * 1024;
                    const char* p = s + off + (long)u * 8192 + threadIdx.x * 16;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(p) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        }
    }
    if constexpr (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
}
template <int MODE>
static float run(int wgs, const char* src, char* dst, long bpw, int reps, unsigned* sink) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<MODE><<<wgs, 512, 65536>>>(src, dst, bpw, 1, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<wgs, 512, 65536>>>(src, dst, bpw, reps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    const long total = 2L << 30;
    char *src, *dst; unsigned* sink;
    hipMalloc(&src, total); hipMalloc(&dst, total); hipMalloc(&sink, 64);
    hipMemset(src, 1, total); hipMemset(dst, 0, total);
    const char* names[4] = {"load", "store", "copy", "ldsdma"};
    // (footprint per workgroup, repetitions): HBM-cold = 8 MiB x 1 rep at 256 WGs (2 GiB total); cache-warm = small regions re-read
    struct { const char* what; long bpw; int reps; } cases[] = {{"hbm 8MiB/wg", 8L << 20, 1}, {"mall 512KiB/wg x16", 512L << 10, 16}, {"l2 64KiB/wg x128", 64L << 10, 128}};
    for (auto& c : cases)
        for (int wgs : {16, 64, 256})
            for (int mode = 0; mode < 4; ++mode) {
                float ms = 0;
                if (mode == 0) ms = run<0>(wgs, src, dst, c.bpw, c.reps, sink);
                if (mode == 1) ms = run<1>(wgs, src, dst, c.bpw, c.reps, sink);
                if (mode == 2) ms = run<2>(wgs, src, dst, c.bpw, c.reps, sink);
                if (mode == 3) ms = run<3>(wgs, src, dst, c.bpw, c.reps, sink);
                const double bytes = (double)c.bpw * c.reps * (mode == 2 ? 2 : 1);
                printf("%-20s wgs %3d %-7s %8.3f ms  per-CU %6.1f GB/s (%5.1f B/clk at 2.1 GHz)  chip %7.1f GB/s\n", c.what, wgs, names[mode], ms,
                       bytes / ms / 1e6, bytes / ms / 1e6 / 2.1, bytes * wgs / ms / 1e6);
            }
    return 0;
}
