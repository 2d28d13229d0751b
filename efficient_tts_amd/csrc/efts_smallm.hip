// efts_smallm.hip -- the MFMA contraction of efts_gemm for SHORT row spaces (one utterance: 50 .. 1 000 rows), where a launch
// is bound by one workgroup's chain of K steps, not by throughput: the free-running B = 1 `inference()` of the reference
// (nntts/models/efficient_tts.py:230-285, called per utterance by nntts/bin/inference.py:108) runs eleven k5 convolutions,
// two k3 convolutions and three Linears on such row spaces, one after the other.
//
// The ring kernels (efts_gemm.hip, efts_gemm_narrow.hip) walk 40 (chunk, tap) steps per tile, each one an LDS-DMA, a counted
// wait, a barrier and a short MFMA burst: ~0.6 us per step with one wave per SIMD and nothing to overlap = 25 us per layer
// whatever the row count.  Here
//   * a tile is 64 rows x 32 columns, so even one utterance spreads over 16+ workgroups (CUs) and each streams only its
//     32 weight rows;
//   * the K dimension is SPLIT ACROSS THE FOUR WAVES of the workgroup (wave w takes the K chunks w, w + 4, ...): four
//     independent short chains instead of one 40-step chain, partial sums combined through LDS in wave order;
//   * every wave is its own pipeline, no barrier inside the loop: the A window (64 + taps - 1 rows x one 128-byte chunk) and the
//     weight tiles of all taps of its NEXT chunk are requested into registers with plain coalesced loads (8 lanes per 128-byte
//     row) while the current chunk is multiplied out of the wave's private LDS region, then dropped into that region (LDS
//     operations of one wave execute in order).  (A first version took the MFMA fragments straight from global memory: every
//     such load touches 32 rows = 32 cache lines for 1 KiB, 120 of them per wave -- 17 us per layer, no better than the ring.)
// Same operand rounding as every other tiling; the summation order differs (K split), so results agree with the ring kernels
// to fp32 rounding, not bit for bit -- which is why EFTS_TILING_AUTO never picks this tiling: the caller asks for it
// (efficient_tts_amd/model.py does for free-running inference).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_gemm_kernels.h"

namespace efts {

constexpr int SM_BM = 64, SM_BN = 32;
constexpr int SM_WIN_PIECES = 9;                               // 8-row pieces of the A window: 72 rows >= 64 + 10
constexpr int SM_WIN_BYTES = SM_WIN_PIECES * 8 * 128;          // 9216
template <int TAPS> constexpr int sm_wave_bytes() { return SM_WIN_BYTES + TAPS * SM_BN * 128; }

template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256) void smallm_kernel(GemmKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = SPLIT == 1 ? 4 : 2;                                     // 16-k slices per 128-byte chunk
    constexpr int WB = sm_wave_bytes<TAPS>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int m0 = blockIdx.x * SM_BM, n0 = blockIdx.y * SM_BN;
    char* const aw = smem + wave * WB;                                         // this wave's A window ...
    char* const bw = aw + SM_WIN_BYTES;                                        // ... and its TAPS weight tiles of 32 rows

    // piece q = lane + 64 j covers window / tile row q >> 3, 16-byte slot q & 7 (8 lanes per 128-byte row: coalesced)
    const int prow = lane >> 3, pslot = lane & 7;
    const char* ag = p.a + (long)(m0 - p.pad + prow) * p.lda + pslot * 16;
    const char* bg = p.b + (long)(n0 + prow) * p.ldb + pslot * 16;
    u32x4 ra[SM_WIN_PIECES], rb[TAPS][4];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < SM_WIN_PIECES; ++j) ra[j] = *(const u32x4*)(ag + (long)j * 8 * p.lda + (long)chunk * 128);
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[k][j] = *(const u32x4*)(bg + (long)k * p.b_tap_stride + (long)j * 8 * p.ldb + (long)chunk * 128);
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < SM_WIN_PIECES; ++j) *(u32x4*)(aw + lds_off(j * 8 + prow, pslot)) = ra[j];
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(u32x4*)(bw + k * SM_BN * 128 + lds_off(j * 8 + prow, pslot)) = rb[k][j];
    };
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto multiply = [&]() {
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const char* bt = bw + k * SM_BN * 128;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int slot = 2 * s + lhalf;
                const bf16x8 fb = *(const bf16x8*)(bt + lds_off(lrow, slot));
                bf16x8 fb2;
                if constexpr (SPLIT == 2) fb2 = *(const bf16x8*)(bt + lds_off(lrow, slot + 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int arow = i * 32 + lrow + k;                         // window row of output row r at tap k is r + k
                    const bf16x8 fa = *(const bf16x8*)(aw + lds_off(arow, slot));
                    if constexpr (SPLIT == 2) {                                 // (the ring kernels' order inside a k-slice: lo*hi, hi*lo, hi*hi)
                        const bf16x8 fa2 = *(const bf16x8*)(aw + lds_off(arow, slot + 4));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2, fb, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb2, acc[i], 0, 0, 0);
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
                }
            }
        }
    };
    if (wave < p.nchunk) {
        fetch(wave);
        stash();
    }
    for (int chunk = wave; chunk < p.nchunk; chunk += 4) {
        const bool more = chunk + 4 < p.nchunk;
        if (more) fetch(chunk + 4);                         // in flight under this chunk's MFMAs
        multiply();
        if (more) stash();                                  // behind the fragment reads of this chunk (one wave: LDS operations in order)
    }
    // ---- partial tiles -> LDS (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); the operand
    // regions are dead once every wave is here
    __syncthreads();
    float (*part)[SM_BM][SM_BN] = (float (*)[SM_BM][SM_BN])smem;               // [4][64][32] fp32 = 32 KiB
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf][lrow] = acc[i][r];
    __syncthreads();
    // ---- combine in wave order + the fused epilogue of efts_gemm: alpha, bias, activation, residual, row mask; fp32 and / or
    // operand-plane rows of 8 columns per thread
    const int row = tid >> 2, c8 = (tid & 3) * 8;
    const int grow = m0 + row;
    if (grow >= p.m) return;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ((part[0][row][c8 + u] + part[1][row][c8 + u]) + part[2][row][c8 + u]) + part[3][row][c8 + u];
    const int col = n0 + c8;
    const float rm = p.rowmask ? p.rowmask[grow] : 1.f;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.resid) {
        const float4 q0 = *(const float4*)(p.resid + (long)grow * p.ldr + col), q1 = *(const float4*)(p.resid + (long)grow * p.ldr + col + 4);
        x[0] = q0.x; x[1] = q0.y; x[2] = q0.z; x[3] = q0.w; x[4] = q1.x; x[5] = q1.y; x[6] = q1.z; x[7] = q1.w;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float t = v[u] * p.alpha + (p.bias ? p.bias[col + u] : 0.f);
        if (p.act == EFTS_ACT_LEAKY) t = t > 0.f ? t : t * p.slope;
        else if (p.act == EFTS_ACT_RELU) t = t > 0.f ? t : 0.f;
        v[u] = (t + x[u]) * rm;
    }
    if (p.out_f32) {
        *(float4*)(p.out_f32 + (long)grow * p.ldo + col) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p.out_f32 + (long)grow * p.ldo + col + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (p.out_bf16) {
        float r8[8], d0, d1;
        const u32x4 hi = {pack_bf16x2(v[0], v[1], &r8[0], &r8[1]), pack_bf16x2(v[2], v[3], &r8[2], &r8[3]),
                          pack_bf16x2(v[4], v[5], &r8[4], &r8[5]), pack_bf16x2(v[6], v[7], &r8[6], &r8[7])};
        char* d = p.out_bf16 + (long)grow * p.ldob + plane_off_hi(col, p.out_split);
        *(u32x4*)d = hi;
        if (p.out_split == 2) {
            const u32x4 lo = {pack_bf16x2(r8[0], r8[1], &d0, &d1), pack_bf16x2(r8[2], r8[3], &d0, &d1),
                              pack_bf16x2(r8[4], r8[5], &d0, &d1), pack_bf16x2(r8[6], r8[7], &d0, &d1)};
            *(u32x4*)(d + 64) = lo;
        }
    }
}

template <int T, int S>
static void sm_launch(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    constexpr int lds = 4 * sm_wave_bytes<T>() > 32768 ? 4 * sm_wave_bytes<T>() : 32768;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)smallm_kernel<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((smallm_kernel<T, S>), grid, dim3(256), lds, st, k);
}

// false = no instantiation for this tap count
bool launch_smallm_any(int split, int taps, hipStream_t st, const GemmKernelArgs& k) {
    const dim3 grid((k.m + SM_BM - 1) / SM_BM, k.n / SM_BN);
    if (split == 1) {
        switch (taps) { case 5: sm_launch<5, 1>(grid, st, k); return true; case 3: sm_launch<3, 1>(grid, st, k); return true; case 1: sm_launch<1, 1>(grid, st, k); return true; default: return false; }
    }
    switch (taps) { case 5: sm_launch<5, 2>(grid, st, k); return true; case 3: sm_launch<3, 2>(grid, st, k); return true; case 1: sm_launch<1, 2>(grid, st, k); return true; default: return false; }
}

}  // namespace efts
