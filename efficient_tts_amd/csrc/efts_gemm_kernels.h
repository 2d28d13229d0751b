// efts_gemm_kernels.h -- what the kernels of efts_gemm share: the argument block, tile constants, and the launchers each
// translation unit exports (efts_gemm.hip: generic 124-row ring kernel + dispatch; efts_gemm_narrow.hip: 64- / 32-column tiles and
// the LDS-resident kernel; efts_conv5.hip: the 256-row k5 kernel).
#pragma once
#include "efts_mma.h"

// cache-policy bits of the epilogue's buffer accesses (bit 0 sc0, bit 1 nt, bit 4 sc1); lab builds only
#ifndef EFTS_AUX_LD
#define EFTS_AUX_LD 0
#endif
#ifndef EFTS_AUX_STF
#define EFTS_AUX_STF 0
#endif
#ifndef EFTS_AUX_STP
#define EFTS_AUX_STP 0
#endif

namespace efts {

struct GemmKernelArgs {
    const char* a;
    const char* b;
    const float* bias;
    const float* resid;
    const float* rowmask;
    float* out_f32;
    char* out_bf16;
    char* out_lo;               // out_split 1 only: separate plane for the bf16 remainder (generic gemm_kernel only)
    char* sign;                 // sign words of the activated outputs (efts_abi.h `sign_mask`), row stride n / 8 bytes, or null
    float* sidx;                // soft index of the softmax over each output row (efts_abi.h `soft_index`), [batch][m], or null
    const int* klen;            // valid columns per batch item
    const int* qlen;            // valid rows per batch item
    float* sqerr;               // squared-error partial sums against a target that travels in `resid` / ldr / r_bs (efts_abi.h `sqerr_part`), or null
    unsigned drop_thresh, drop_seed_h;   // train-mode dropout of the activated value (efts_abi.h drop_p): keep iff hash >= thresh; 0 = off
    float drop_inv_keep;
    long lda, ldb, b_tap_stride, ldr, ldo, ldob;
    long a_bs, b_bs, r_bs, m_bs, o_bs, ob_bs;
    long a_bs2, b_bs2, o_bs2;   // outer batch (blockIdx.z)
    int m;                      // rows per batch item
    int dil, bm;                // tap dilation (rows between taps); output rows per tile = 128 - (taps - 1) * dil
    int plane_act;              // 1: the operand plane receives act(out) (pre-activation consumers), slope = plane_slope
    float plane_slope;
    int n, nchunk, pad;
    int mtiles, ntiles;
    float alpha, slope;
    int act, out_split;
    int vec_ok;   // all fp32 row strides / pointers allow float4 access
    unsigned long long* prof;   // DBG instantiation: per-phase cycle sums
    int dbg;      // DBG instantiation (EFTS_GEMM_DBG): 1 = no epilogue memory traffic, 2 = no DMA, 4 = no MFMA
};

constexpr int C5_WIN = 256;
constexpr int C5_BM = C5_WIN - 4;
constexpr int C5_A_BYTES = C5_WIN * 128;                       // 32768
constexpr int C5_LDS = C5_A_BYTES + NST * TILE_BYTES;          // 81920
constexpr long C5_DEFAULT_MIN_TILES = 400;                     // bf16 planes only by default (measured +1.5 % there, -2.5 % on bf16x3)
constexpr int R32_WIN = 256;
constexpr int C8_BN = 256;
constexpr int C8_W_BYTES = C8_BN * 128;                          // 32768
constexpr int C8_LDS = C5_A_BYTES + NST * C8_W_BYTES;            // 131072

// efts_gemm_narrow.hip: false = no instantiation for this tap count
bool launch_narrow_any(int split, int bnt, int taps, dim3 grid, hipStream_t st, const GemmKernelArgs& k);
bool launch_resident32_any(int split, int taps, dim3 grid, hipStream_t st, const GemmKernelArgs& k);
// efts_smallm.hip: false = no instantiation for this tap count
bool launch_smallm_any(int split, int taps, hipStream_t st, const GemmKernelArgs& k);
// efts_conv5.hip
void launch_conv5_any(int split, dim3 grid, hipStream_t st, const GemmKernelArgs& k);
void conv5_set_lds_attr();
#ifdef EFTS_LAB
void launch_conv8(dim3 grid, hipStream_t st, const GemmKernelArgs& k);
#endif

}  // namespace efts
