// rc4_loop_test.hip -- stand-alone check + timing of the generated main loop (csrc/efts_rc4_loop.inc) of efts_resconv5's one-wave-per-SIMD
// kernel: one (32 h) x 256 tile per workgroup, accumulators dumped raw and compared with a host reference; then the same loop timed
// over all CUs with several tiles per workgroup, with the ablation variants of the generator.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I efficient_tts_amd/csrc tools/micro/rc4_loop_test.hip -o lab/rc4_loop_test && lab/rc4_loop_test
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "efts_rc4_loop.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Args {
    const char* a;        // [rows][ldA bytes] bf16 plane, row 0 = window row 0 of tile 0
    const char* w;        // [5 taps][256 cols][ldW bytes]
    float* c;             // [tiles][32 h][256] raw accumulators
    int lda, ldw, wts, nchunk, rows_total, h, reps, variant, tile_rows;
    unsigned long long* cyc;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

template <int H, int VAR>
__global__ __launch_bounds__(256, 1) void k(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int m0 = blockIdx.x * p.tile_rows;                           // this workgroup's first window row
    const char* abase = p.a + (long)m0 * p.lda;
    // kernel prologue: window chunk 0 -> buffer 0, weight tile (tap 0, chunk 0) -> slot 0
    for (int q = 0; q < H; ++q) {
        const int P = 4 * q + wave, r = 8 * P + (lane >> 3);
        const int sl = (lane & 7) ^ ((r >> 1) & 7);
        const u32x4 d = *(const u32x4*)(abase + (long)r * p.lda + sl * 16);
        *(u32x4*)(smem + P * 1024 + lane * 16) = d;
    }
    for (int g = 0; g < 8; ++g) {
        const int P = 4 * g + wave, r = 8 * P + (lane >> 3);
        const int sl = (lane & 7) ^ ((r >> 1) & 7);
        const u32x4 d = *(const u32x4*)(p.w + (long)r * p.ldw + sl * 16);
        *(u32x4*)(smem + 65536 + P * 1024 + lane * 16) = d;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t ars = rsrc(abase, 0x7fffffff), wrs = rsrc(p.w, 0x7fffffff);
    const int lda = __builtin_amdgcn_readfirstlane(p.lda), ldw = __builtin_amdgcn_readfirstlane(p.ldw), wts = __builtin_amdgcn_readfirstlane(p.wts);
    const int nch = __builtin_amdgcn_readfirstlane(p.nchunk);
    const int rmax = __builtin_amdgcn_readfirstlane(p.rows_total - 1 - m0);
    int state = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < p.reps; ++rep) {
        // the "next tile" is this same tile again: same window rows, same weights
        if constexpr (VAR == 0) {
            if constexpr (H == 2) RC4_LOOP_H2(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
            if constexpr (H == 3) RC4_LOOP_H3(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
            if constexpr (H == 4) RC4_LOOP_H4(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
            if constexpr (H == 5) RC4_LOOP_H5(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
            if constexpr (H == 6) RC4_LOOP_H6(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
            if constexpr (H == 7) RC4_LOOP_H7(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
            if constexpr (H == 8) RC4_LOOP_H8(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
        }
        if constexpr (VAR == 1) RC4_LOOP_H8_NOSTAGING(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
        if constexpr (VAR == 2) RC4_LOOP_H8_WRITES_ONLY(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
        if constexpr (VAR == 3) RC4_LOOP_H8_LOADS_ONLY(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
        if constexpr (VAR == 4) RC4_LOOP_H8_BUNCHED(ars, ars, wrs, wrs, lda, rmax, lda, rmax, ldw, wts, wts, nch, wave, state, lds0);
        state ^= ((5 * nch) & 1) | ((nch & 1) << 1);
        state = __builtin_amdgcn_readfirstlane(state);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (p.cyc && tid == 0) p.cyc[blockIdx.x] = t1 - t0;
    // dump the accumulators (held transposed): block (i, j) register r -> row i 32 + (lane & 31), col wave 64 + j 32 + 8 (r >> 2) + (r & 3) + 4 (lane >> 5)
    if (p.c) {
        float* cb = p.c + (long)blockIdx.x * (32 * H) * 256;
#define DUMP1(N) { float v_; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v_) : "n"(N)); \
        const int blk = (N) >> 4, r = (N) & 15, i = blk >> 1, j = blk & 1; \
        if (i < H) cb[(long)(i * 32 + (lane & 31)) * 256 + wave * 64 + j * 32 + 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5)] = v_; }
#define DUMP4(N) DUMP1(N) DUMP1(N + 1) DUMP1(N + 2) DUMP1(N + 3)
#define DUMP16(N) DUMP4(N) DUMP4(N + 4) DUMP4(N + 8) DUMP4(N + 12)
#define DUMP64(N) DUMP16(N) DUMP16(N + 16) DUMP16(N + 32) DUMP16(N + 48)
        DUMP64(0) DUMP64(64) DUMP64(128) DUMP64(192)
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int H, int VAR> static void launch(const Args& a, int grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)k<H, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); attr = true; }
    hipLaunchKernelGGL((k<H, VAR>), dim3(grid), dim3(256), 163840, st, a);
}

template <int H> static int check(int nchunk) {
    const int K = nchunk * 64, rows = 32 * H, ntile = 3, tile_rows = 40;
    const int rows_total = tile_rows * (ntile - 1) + rows + 8;
    const int lda = K * 2, ldw = K * 2, wts = 256 * ldw;
    std::vector<uint16_t> A((size_t)rows_total * K), W((size_t)5 * 256 * K);
    srand(1 + H);
    for (auto& x : A) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& x : W) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
    char *dA, *dW; float* dC;
    CK(hipMalloc(&dA, A.size() * 2 + 65536)); CK(hipMalloc(&dW, W.size() * 2 + 65536)); CK(hipMalloc(&dC, (size_t)ntile * rows * 256 * 4));
    CK(hipMemset(dA, 0, A.size() * 2 + 65536));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, (size_t)ntile * rows * 256 * 4));
    Args a{dA, dW, dC, lda, ldw, wts, nchunk, rows_total, H, 2, 0, tile_rows, nullptr};       // reps = 2: the second pass runs on what the first left in LDS
    launch<H, 0>(a, ntile, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> C((size_t)ntile * rows * 256);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; long bad = 0;
    for (int t = 0; t < ntile; ++t)
        for (int r = 0; r < rows - 4; ++r)
            for (int c = 0; c < 256; c += 7) {
                double ref = 0;
                for (int tap = 0; tap < 5; ++tap) {
                    const int ar = t * tile_rows + r + tap;
                    const int arc = ar < rows_total - 1 ? ar : rows_total - 1;
                    for (int kk = 0; kk < K; ++kk) ref += (double)bf2f(A[(size_t)arc * K + kk]) * bf2f(W[((size_t)tap * 256 + c) * K + kk]);
                }
                const double d = fabs(ref - C[((size_t)t * rows + r) * 256 + c]);
                if (!(d <= worst)) worst = d;
                if (!(d <= 2e-3 * (1 + fabs(ref)))) ++bad;
            }
    printf("check h=%d nchunk=%d: worst abs diff %.3e, bad %ld\n", H, nchunk, worst, bad);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC));
    return bad != 0;
}

template <int H, int VAR> static void bench(const char* name, int nchunk, int grid, int reps) {
    const int K = nchunk * 64, rows = 32 * H, tile_rows = rows - 4;
    const int rows_total = tile_rows * grid + 64;
    const int lda = K * 2, ldw = K * 2, wts = 256 * ldw;
    char *dA, *dW; unsigned long long* dcyc;
    CK(hipMalloc(&dA, (size_t)rows_total * lda + 65536)); CK(hipMalloc(&dW, (size_t)5 * wts + 65536)); CK(hipMalloc(&dcyc, grid * 8));
    std::vector<uint16_t> A((size_t)rows_total * K), W((size_t)5 * 256 * K);
    for (auto& x : A) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& x : W) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
    Args a{dA, dW, nullptr, lda, ldw, wts, nchunk, rows_total, H, reps, VAR, tile_rows, dcyc};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch<H, VAR>(a, grid, 0);
    CK(hipEventRecord(e0, 0));
    const int iters = 10;
    for (int i = 0; i < iters; ++i) launch<H, VAR>(a, grid, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(grid);
    CK(hipMemcpy(cyc.data(), dcyc, grid * 8, hipMemcpyDeviceToHost));
    double mc = 0; for (auto c : cyc) mc += c; mc /= grid;
    const double us_tile = ms * 1e3 / iters / reps;
    const double steps = 5.0 * nchunk;
    const double flop = 2.0 * rows * 256 * K * 5 * grid;
    printf("%-10s h=%d grid=%d reps=%d: %.2f us per tile, %.0f cycles per step (MFMA issue %d), %.0f TFLOP/s (%.1f %% of 2.5 PF)\n", name, H, grid, reps, us_tile,
           mc / reps / steps, 8 * H * 32, flop / (us_tile * 1e-6) / 1e12, flop / (us_tile * 1e-6) / 2.5e15 * 100);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dcyc));
}

int main(int argc, char** argv) {
    int bad = 0;
    bad |= check<8>(8); bad |= check<7>(8); bad |= check<6>(3); bad |= check<5>(2); bad |= check<4>(8); bad |= check<3>(4); bad |= check<2>(5);
    if (bad) { printf("CHECK FAILED\n"); }
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        bench<8, 0>("full", 8, cus, 4);
        bench<8, 1>("no-staging", 8, cus, 4);
        bench<8, 2>("writes-only", 8, cus, 4);
        bench<8, 3>("loads-only", 8, cus, 4);
        bench<8, 4>("bunched", 8, cus, 4);
        bench<7, 0>("full", 8, cus, 4);
        bench<6, 0>("full", 8, cus, 4);
        bench<4, 0>("full", 8, cus, 4);
    }
    bench<8, 0>("full-1cu", 8, 1, 4);
    return bad;
}
