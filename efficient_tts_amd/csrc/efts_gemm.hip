// efts_gemm.hip -- the MFMA contraction of the EFTS-CNN path on gfx950 (CDNA4).
//
//   out[row, col] = epi( alpha * sum_{tap} sum_{k} A[row + tap - pad, k] * Bw[tap][col, k] )
//
// One kernel serves the residual Conv1d stacks (taps 5; reference
// nntts/layers/efts_modules.py:48-51), the duration-predictor convs (taps 3;
// nntts/layers/duration_predictor.py:57), the Linears (taps 1; efficient_tts.py:149-153,161,198)
// and the two batched matrix products of the alignment block (efficient_tts.py:190,390).
//
// Design (MI355X-first):
//   * activations are channel-last in a padded row space, so a k-tap convolution is a GEMM
//     whose A tile is ONE (128+4)-row LDS window read at `taps` row shifts: the window is
//     staged once per K-chunk and re-used by every tap (LDS-staged conv window).
//   * 128x128 output tile per 256-thread workgroup, 2x2 waves of 64x64, 32x32x16 bf16 MFMA,
//     fp32 accumulators (64 VGPR/lane), 2 workgroups per CU (66 KiB LDS each).
//   * operands arrive by LDS-DMA (global_load_lds, 16 B/lane).  The LDS image is lane-linear,
//     so the bank-conflict swizzle is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 address (same involution on both sides).
//   * K is consumed in 128-byte chunks: 64 bf16 ("bf16"), or 32 hi + 32 lo bf16 ("bf16x3":
//     x = hi + lo, product = hi*hi + hi*lo + lo*hi, fp32 accumulate -> fp32-class accuracy at
//     3 MFMAs per product instead of the 16x slower f32 MFMA).
//   * epilogue fused: bias, LeakyReLU/ReLU, residual add, row mask (gap rows / padded
//     positions), fp32 store and bf16 operand planes for the next contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "efts_internal.h"

namespace efts {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BN = 128;
constexpr int W_BYTES = BN * 128;         // 16384

// Swizzled LDS byte offset of (row, 16-byte slot) inside a [rows][128 B] tile.  A ds_read_b128
// lane group holds 16 different rows at one logical slot; rows r and r+2 share banks, so the
// physical slot is XORed with (r >> 1) & 7 (conflict-free for every tap shift).
__device__ __forceinline__ int lds_off(int row, int slot) {
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

struct GemmKernelArgs {
    const char* a;
    const char* b;
    const float* bias;
    const float* resid;
    const float* rowmask;
    float* out_f32;
    char* out_bf16;
    long lda, ldb, b_tap_stride, ldr, ldo, ldob;
    long a_bs, b_bs, r_bs, m_bs, o_bs, ob_bs;
    long a_bs2, b_bs2, o_bs2;   // outer batch (blockIdx.z)
    int m_base, m_end;   // this launch covers rows [m_base, m_end) (tail launches use a smaller tile)
    int n, nchunk, pad;
    int mtiles, ntiles;
    float alpha, slope;
    int act, out_split;
    int vec_ok;   // all fp32 row strides / pointers allow float4 access
    int spread;   // EFTS_GEMM_SPREAD: spread the A-window DMA over the tap steps with counted waits
    int stagger;  // persistent mode: cycles the second half of the grid idles before its first tile
    int dbg;   // ablation switches (EFTS_GEMM_DBG): 1 = skip epilogue stores, 2 = no DMA in loop, 4 = no MFMA
};

// One DMA piece = one wave instruction = 8 tile rows x 128 B.  Lane l lands at LDS
// piece_base + 16*l, i.e. tile row 8*piece + l/8, physical slot l%8.
__device__ __forceinline__ void dma_piece(const char* rowbase0, long ld, int first_row, int max_row,
                                          int piece, int lane, char* lds_tile) {
    const int r = piece * 8 + (lane >> 3);
    const int ps = lane & 7;
    const int s = ps ^ ((r >> 1) & 7);
    int gr = first_row + r;
    gr = gr > max_row ? max_row : gr;
    const char* src = rowbase0 + (long)gr * ld + (s << 4);
    __builtin_amdgcn_global_load_lds((const void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024),
                                     16, 0, 0);
}

// Tile configuration.  WM = number of 64-row wave rows.
//   WM = 2: 128x128 tile, 4 waves, 2-stage operand ring, 2 workgroups per CU (66 KiB LDS each)
//   WM = 4: 256x128 tile, 8 waves, 3-stage weight ring with COUNTED vmcnt (LDS-DMA stays in flight
//           across the step barrier), 1 workgroup per CU (114 KiB LDS): half the weight traffic per
//           FLOP and two steps of latency budget per DMA.
template <int TAPS, int WM>
struct Cfg {
    static constexpr int NW = WM * 2;
    static constexpr int THREADS = NW * 64;
    static constexpr int BM = WM * 64;
    static constexpr int A_PIECES = BM / 8 + (TAPS == 1 ? 0 : 1);
    static constexpr int A_BYTES = A_PIECES * 1024;
    static constexpr int NST = (WM == 4) ? 3 : 2;
    static constexpr int WPW = 16 / NW;                       // weight DMA pieces per wave per step
    static constexpr int LDS = 2 * A_BYTES + NST * W_BYTES;
};

__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
}

template <int TAPS, int SPLIT, int WM>
__global__ __launch_bounds__(WM * 128, 2) void gemm_kernel(GemmKernelArgs p) {
    using C = Cfg<TAPS, WM>;
    constexpr int BM = C::BM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // all LDS in one array, buffers addressed arithmetically (no runtime-indexed pointer arrays)
#define EFTS_ABUF(i) (smem + ((i) & 1) * C::A_BYTES)
#define EFTS_WBUF(i) (smem + 2 * C::A_BYTES + ((i) % C::NST) * W_BYTES)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int z = blockIdx.y;
    const int lrow = lane & 31;
    const int lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;
    const int nA = (C::A_PIECES - wave + C::NW - 1) / C::NW;     // A window pieces this wave issues
    const int c4 = (tid & 31) << 2;
    constexpr int RPP = C::THREADS / 32;                          // tile rows per epilogue sweep
    constexpr int HROWS = (WM >= 2) ? 128 : 64;                   // rows staged per epilogue pass
    constexpr int NPS = HROWS / RPP;                              // sweeps per pass
    const int z2 = blockIdx.z;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;

    // Persistent workgroups: each walks the tile list with stride gridDim.x, so the epilogue's
    // stores of tile t drain from the memory queues while tile t+1's main loop already runs, and
    // the residual rows of a tile are prefetched under its last MFMA step.
    // XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous range of
    // tiles (n fastest) so the workgroups sharing an A window hit the same L2.
    const int ntot = p.mtiles * p.ntiles;
    if (p.stagger > 0 && blockIdx.x >= (gridDim.x >> 1) && gridDim.x < (unsigned)ntot) {
        // de-phase the two workgroups that share a CU so one's HBM-bound epilogue overlaps the
        // other's MFMA-bound main loop instead of all epilogues bursting together
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < p.stagger) __builtin_amdgcn_s_sleep(32);
    }
  for (int vt = blockIdx.x; vt < ntot; vt += gridDim.x) {
    int bid = vt;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = p.m_base + mt * BM, n0 = nt * BN;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);

    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;
    const int a_first = m0 - p.pad;            // may be negative: guard rows exist
    const int a_max = 0x7fffffff;              // A rows are never clamped (guards)
    const int b_max = p.n - 1 - n0;            // clamp B rows to the last real row

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_w = [&](int sn) {     // weights of step sn -> ring slot sn % NST
        const int cn = sn / TAPS, kn = sn - cn * TAPS;
        const char* wb = Bw + (long)kn * p.b_tap_stride + (long)n0 * p.ldb + (long)cn * 128;
#pragma unroll
        for (int pc = 0; pc < C::WPW; ++pc) dma_piece(wb, p.ldb, 0, b_max, wave * C::WPW + pc, lane, EFTS_WBUF(sn));
    };
    auto issue_a = [&](int cn) {     // A window of chunk cn -> window buffer cn & 1
        const char* ab = A + (long)cn * 128;
        for (int pc = wave; pc < C::A_PIECES; pc += C::NW) dma_piece(ab, p.lda, a_first, a_max, pc, lane, EFTS_ABUF(cn));
    };

    // prologue: window 0 + the first NST-1 weight tiles
    issue_a(0);
    issue_w(0);
    if constexpr (C::NST == 3) {
        if (nsteps > 1) issue_w(1);
        wait_vmcnt(nsteps > 1 ? C::WPW : 0);
        __builtin_amdgcn_s_barrier();
    } else {
        __syncthreads();
    }

    int s = 0;
    for (int c = 0; c < p.nchunk; ++c) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k, ++s) {
            // ---- stage operands NST-1 steps ahead (they land while this and the next step compute)
            const bool do_w = (s + C::NST - 1 < nsteps) && !(p.dbg & 2);
            const bool do_a = (k == 0) && (c + 1 < p.nchunk) && !(p.dbg & 2);
            // spread mode (2-stage ring, taps > 1): the next window is issued in TAPS-1 small groups, one
            // per tap step, AFTER that step's weight tile, and the step barrier uses a counted vmcnt that
            // lets the group just issued stay in flight: no 17-piece burst and two steps of latency budget
            // per window piece instead of a vmcnt(0) in the step that issued it.
            const bool spread = p.spread && TAPS > 1 && C::NST == 2;
            int n_inflight = 0;
            if (TAPS == 1 && do_a) issue_a(c + 1);          // taps 1: the window changes every step
            if (do_w) issue_w(s + C::NST - 1);
            if (spread) {
                if (c + 1 < p.nchunk && k < TAPS - 1) {
                    const char* ab = A + (long)(c + 1) * 128;
                    for (int pc = k * C::NW + wave; pc < C::A_PIECES; pc += C::NW * (TAPS - 1)) {
                        dma_piece(ab, p.lda, a_first, a_max, pc, lane, EFTS_ABUF(c + 1));
                        ++n_inflight;
                    }
                }
            } else if (TAPS != 1 && do_a) issue_a(c + 1);
            // ---- MFMAs of this (chunk, tap)
            const char* at = EFTS_ABUF(c);
            const char* wt = EFTS_WBUF(s);
            const int arow = wm * 64 + lrow + k;   // tile row of output row r at tap k is r + k
            const int brow = wn * 64 + lrow;
            if (p.dbg & 4) {
            } else if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 af[2], bfr[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                        al[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bh[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
                        bl[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot + 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
            if constexpr (C::NST == 3) {
                // Counted wait: everything this wave issued up to and including the weights of step
                // s+1 must have landed; younger DMAs (weights of s+2, a window issued this or the
                // previous step) stay in flight across the barrier.  taps 1: the window of the next
                // step is older than the weights issued this step, so it is covered too.
                int n = do_w ? C::WPW : 0;
                if (TAPS != 1) {
                    if (do_a) n += nA;
                    if (k == 1 && c + 1 < p.nchunk && !(p.dbg & 2)) n += nA;   // window issued one step ago
                }
                wait_vmcnt(n);
                __builtin_amdgcn_s_barrier();
            } else if (spread) {
                wait_vmcnt(n_inflight);
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();   // next operands landed (vmcnt(0)) and this step's reads are done
            }
        }
    }
    if constexpr (C::NST == 3) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- fused epilogue, staged through LDS so that every global access is a full-row vector.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); per
    // 128-row half, each wave drops its 64x64 block (bias + activation applied) into a [128][128]
    // fp32 LDS tile, then 32 consecutive threads sweep one 512-byte tile row: float4 residual
    // load, float4 store, and 8-byte bf16 (hi / lo) operand-plane stores.
    float* cs = (float*)smem;   // 64 KiB; the main loop's last barrier has retired all LDS reads
#pragma unroll
    for (int half = 0; half < (WM >= 2 ? WM / 2 : 1); ++half) {
        if ((wm >> 1) == half) {
            const float* bias = p.bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + lrow;
                const float bv = (bias && n0 + cl < p.n) ? bias[n0 + cl] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (wm & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[i][j][r] * p.alpha + bv;
                        if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                        cs[rl * 128 + cl] = v;
                    }
                }
            }
        }
        __syncthreads();
        if (col < p.n && !(p.dbg & 1)) {
#pragma unroll 4
            for (int ps = 0; ps < NPS; ++ps) {
                const int rl = ps * RPP + (tid >> 5);
                const int row = m0 + half * 128 + rl;
                if (row >= p.m_end) break;
                float4 v = *(const float4*)(cs + rl * 128 + c4);
                const float rm = rowmask ? rowmask[row] : 1.f;
                if (vec) {
                    if (resid) {
                        const float4 x = *(const float4*)(resid + (long)row * p.ldr + col);
                        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                    }
                    v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                    if (of) *(float4*)(of + (long)row * p.ldo + col) = v;
                    if (ob) plane_store4(ob + (long)row * p.ldob, col, v.x, v.y, v.z, v.w, p.out_split);
                } else {
                    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (col + u >= p.n) break;
                        float t = vv[u];
                        if (resid) t += resid[(long)row * p.ldr + col + u];
                        t *= rm;
                        if (of) of[(long)row * p.ldo + col + u] = t;
                        if (ob) {
                            const unsigned short hi = f32_to_bf16(t);
                            char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                            *(unsigned short*)d = hi;
                            if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                        }
                    }
                }
            }
        }
        __syncthreads();   // the LDS tile is re-used by the next half / the next tile's operand ring
    }
  }   // persistent tile loop
}


// =============================================================================================
// gemm_kernel_v3: 256x128 tile, 4 waves of 128x64 (8 accumulator blocks, 128 VGPR), 64-byte LDS rows
// (32 bf16 / 16 hi + 16 lo per step), 4-stage weight ring with COUNTED vmcnt so LDS-DMA stays in
// flight across three steps, still 2 workgroups per CU (66 KiB LDS each).  Per FLOP it moves 0.59x
// the DMA bytes (the weight tile is amortised over 256 rows) and issues 0.75x the ds_read_b128 of
// the 128x128 kernel.  A 128-byte global chunk is consumed as two 64-byte half-chunks.
// =============================================================================================
constexpr int V3_BM = 256;
constexpr int V3_NST = 4;
constexpr int V3_W_BYTES = BN * 64;     // 8192

// 64-byte rows: four 16-byte slots per row; rows r and r+4 share banks -> slot ^= (r >> 2) & 3
__device__ __forceinline__ int lds_off64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// one DMA piece = 16 tile rows x 64 B; `sub` selects the half-chunk, SPLIT the byte mapping
template <int SPLIT>
__device__ __forceinline__ void dma_piece64(const char* rowbase0, long ld, int first_row, int max_row, int piece, int lane,
                                            int sub, char* lds_tile) {
    const int r = piece * 16 + (lane >> 2);
    const int ps = lane & 3;
    const int s = ps ^ ((r >> 2) & 3);
    int gr = first_row + r;
    gr = gr > max_row ? max_row : gr;
    const int boff = (SPLIT == 1) ? sub * 64 + s * 16 : ((s & 2) ? 64 : 0) + sub * 32 + (s & 1) * 16;
    const char* src = rowbase0 + (long)gr * ld + boff;
    __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
}

template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel_v3(GemmKernelArgs p) {
    constexpr int A_PIECES = V3_BM / 16 + (TAPS == 1 ? 0 : 1);     // 16 or 17 pieces of 16 rows
    constexpr int A_BYTES = A_PIECES * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#define V3_ABUF(i) (smem + ((i) & 1) * A_BYTES)
#define V3_WBUF(i) (smem + 2 * A_BYTES + ((i) & (V3_NST - 1)) * V3_W_BYTES)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.y, z2 = blockIdx.z;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nhc = p.nchunk * 2;                      // half-chunks
    const int nsteps = nhc * TAPS;
    const int nA = (A_PIECES - wave + 3) / 4;

    const int ntot = p.mtiles * p.ntiles;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = p.m_base + mt * V3_BM, n0 = nt * BN;
    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;
    const int a_first = m0 - p.pad;
    const int b_max = p.n - 1 - n0;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_w = [&](int sn) {
        const int hc = sn / TAPS, kn = sn - hc * TAPS;
        const char* wb = Bw + (long)kn * p.b_tap_stride + (long)n0 * p.ldb + (long)(hc >> 1) * 128;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dma_piece64<SPLIT>(wb, p.ldb, 0, b_max, wave * 2 + pc, lane, hc & 1, V3_WBUF(sn));
    };
    auto issue_a = [&](int hc) {
        const char* ab = A + (long)(hc >> 1) * 128;
        for (int pc = wave; pc < A_PIECES; pc += 4) dma_piece64<SPLIT>(ab, p.lda, a_first, 0x7fffffff, pc, lane, hc & 1, V3_ABUF(hc));
    };

    // prologue: window 0, weights of steps 0..2; wait for window 0 + W(0)
    issue_a(0);
    issue_w(0);
    int pro = 0;
    if (nsteps > 1) { issue_w(1); pro += 2; }
    if (nsteps > 2) { issue_w(2); pro += 2; }
    wait_vmcnt(pro);
    __builtin_amdgcn_s_barrier();

    int s = 0;
    for (int hc = 0; hc < nhc; ++hc) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k, ++s) {
            const bool do_w = (s + 3 < nsteps);
            const bool do_a = (k == 0) && (hc + 1 < nhc);
            if (TAPS == 1 && do_a) issue_a(hc + 1);
            if (do_w) issue_w(s + 3);
            if (TAPS != 1 && do_a) issue_a(hc + 1);

            const char* at = V3_ABUF(hc);
            const char* wt = V3_WBUF(s);
            const int arow = wm * 128 + lrow + k;
            const int brow = wn * 64 + lrow;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 af[4], bfr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(wt + lds_off64(brow + j * 32, slot));
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(at + lds_off64(arow + i * 32, slot));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            } else {
                bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *(const bf16x8*)(wt + lds_off64(brow + j * 32, lhalf));
                    bl[j] = *(const bf16x8*)(wt + lds_off64(brow + j * 32, 2 + lhalf));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *(const bf16x8*)(at + lds_off64(arow + i * 32, lhalf));
                    al[i] = *(const bf16x8*)(at + lds_off64(arow + i * 32, 2 + lhalf));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            // Counted wait.  DMAs retire in issue order; after this step the weights of step s+1 (and
            // for taps 1 the next window) must have landed, while the weights of s+2 / s+3 and a
            // window issued this or the previous step stay in flight across the barrier.
            int n = 0;
            if (TAPS == 1) {
                n = do_w ? 2 : 0;
            } else {
                if (s + 2 < nsteps) n += 2;
                if (do_w) n += 2;
                if (do_a) n += nA;
                if (k == 1 && hc + 1 < nhc) n += nA;
            }
            wait_vmcnt(n);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: two 128-row halves through the [128][128] fp32 LDS tile (wave row wm owns half wm)
    float* cs = (float*)smem;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const int c4 = (tid & 31) << 2;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
            const float* bias = p.bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + lrow;
                const float bv = (bias && n0 + cl < p.n) ? bias[n0 + cl] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[i][j][r] * p.alpha + bv;
                        if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                        cs[rl * 128 + cl] = v;
                    }
                }
            }
        }
        __syncthreads();
        if (col < p.n && !(p.dbg & 1)) {
#pragma unroll 4
            for (int ps = 0; ps < 16; ++ps) {
                const int rl = ps * 8 + (tid >> 5);
                const int row = m0 + half * 128 + rl;
                if (row >= p.m_end) break;
                float4 v = *(const float4*)(cs + rl * 128 + c4);
                const float rm = rowmask ? rowmask[row] : 1.f;
                if (vec) {
                    if (resid) {
                        const float4 x = *(const float4*)(resid + (long)row * p.ldr + col);
                        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                    }
                    v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                    if (of) *(float4*)(of + (long)row * p.ldo + col) = v;
                    if (ob) plane_store4(ob + (long)row * p.ldob, col, v.x, v.y, v.z, v.w, p.out_split);
                } else {
                    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (col + u >= p.n) break;
                        float t = vv[u];
                        if (resid) t += resid[(long)row * p.ldr + col + u];
                        t *= rm;
                        if (of) of[(long)row * p.ldo + col + u] = t;
                        if (ob) {
                            const unsigned short hi = f32_to_bf16(t);
                            char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                            *(unsigned short*)d = hi;
                            if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}


// =============================================================================================
// gemm_kernel_v4: 128x128 tile, 4 waves of 64x64, 64-byte LDS rows, 3-stage ring, 42 KiB LDS -> THREE workgroups
// per CU (occupancy experiment); derived from gemm_kernel_v3 below:
// gemm_kernel_v3: 256x128 tile, 4 waves of 128x64 (8 accumulator blocks, 128 VGPR), 64-byte LDS rows
// (32 bf16 / 16 hi + 16 lo per step), 4-stage weight ring with COUNTED vmcnt so LDS-DMA stays in
// flight across three steps, still 2 workgroups per CU (66 KiB LDS each).  Per FLOP it moves 0.59x
// the DMA bytes (the weight tile is amortised over 256 rows) and issues 0.75x the ds_read_b128 of
// the 128x128 kernel.  A 128-byte global chunk is consumed as two 64-byte half-chunks.
// =============================================================================================


template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256, 3) void gemm_kernel_v4(GemmKernelArgs p) {
    constexpr int A_PIECES = 128 / 16 + (TAPS == 1 ? 0 : 1);       // 8 or 9 pieces of 16 rows
    constexpr int V4_NST = 3;
    constexpr int A_BYTES = A_PIECES * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#define V4_ABUF(i) (smem + ((i) & 1) * A_BYTES)
#define V4_WBUF(i) (smem + 2 * A_BYTES + ((i) % V4_NST) * V3_W_BYTES)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.y, z2 = blockIdx.z;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nhc = p.nchunk * 2;                      // half-chunks
    const int nsteps = nhc * TAPS;
    const int nA = (A_PIECES - wave + 3) / 4;

    const int ntot = p.mtiles * p.ntiles;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = p.m_base + mt * 128, n0 = nt * BN;
    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;
    const int a_first = m0 - p.pad;
    const int b_max = p.n - 1 - n0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_w = [&](int sn) {
        const int hc = sn / TAPS, kn = sn - hc * TAPS;
        const char* wb = Bw + (long)kn * p.b_tap_stride + (long)n0 * p.ldb + (long)(hc >> 1) * 128;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dma_piece64<SPLIT>(wb, p.ldb, 0, b_max, wave * 2 + pc, lane, hc & 1, V4_WBUF(sn));
    };
    auto issue_a = [&](int hc) {
        const char* ab = A + (long)(hc >> 1) * 128;
        for (int pc = wave; pc < A_PIECES; pc += 4) dma_piece64<SPLIT>(ab, p.lda, a_first, 0x7fffffff, pc, lane, hc & 1, V4_ABUF(hc));
    };

    // prologue: window 0, weights of steps 0..2; wait for window 0 + W(0)
    issue_a(0);
    issue_w(0);
    int pro = 0;
    if (nsteps > 1) { issue_w(1); pro += 2; }
    wait_vmcnt(pro);
    __builtin_amdgcn_s_barrier();

    int s = 0;
    for (int hc = 0; hc < nhc; ++hc) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k, ++s) {
            const bool do_w = (s + 2 < nsteps);
            const bool do_a = (k == 0) && (hc + 1 < nhc);
            if (TAPS == 1 && do_a) issue_a(hc + 1);
            if (do_w) issue_w(s + 2);
            if (TAPS != 1 && do_a) issue_a(hc + 1);

            const char* at = V4_ABUF(hc);
            const char* wt = V4_WBUF(s);
            const int arow = wm * 64 + lrow + k;
            const int brow = wn * 64 + lrow;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 af[2], bfr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(wt + lds_off64(brow + j * 32, slot));
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(at + lds_off64(arow + i * 32, slot));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            } else {
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *(const bf16x8*)(wt + lds_off64(brow + j * 32, lhalf));
                    bl[j] = *(const bf16x8*)(wt + lds_off64(brow + j * 32, 2 + lhalf));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *(const bf16x8*)(at + lds_off64(arow + i * 32, lhalf));
                    al[i] = *(const bf16x8*)(at + lds_off64(arow + i * 32, 2 + lhalf));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            // Counted wait.  DMAs retire in issue order; after this step the weights of step s+1 (and
            // for taps 1 the next window) must have landed, while the weights of s+2 / s+3 and a
            // window issued this or the previous step stay in flight across the barrier.
            int n = 0;
            if (TAPS == 1) {
                n = do_w ? 2 : 0;
            } else {
                if (do_w) n += 2;
                if (do_a) n += nA;
                if (k == 1 && hc + 1 < nhc) n += nA;
            }
            wait_vmcnt(n);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: two 64-row halves through a [64][128] fp32 LDS tile (wave row wm owns half wm)
    float* cs = (float*)smem;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const int c4 = (tid & 31) << 2;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
            const float* bias = p.bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + lrow;
                const float bv = (bias && n0 + cl < p.n) ? bias[n0 + cl] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[i][j][r] * p.alpha + bv;
                        if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                        cs[rl * 128 + cl] = v;
                    }
                }
            }
        }
        __syncthreads();
        if (col < p.n && !(p.dbg & 1)) {
#pragma unroll 4
            for (int ps = 0; ps < 8; ++ps) {
                const int rl = ps * 8 + (tid >> 5);
                const int row = m0 + half * 64 + rl;
                if (row >= p.m_end) break;
                float4 v = *(const float4*)(cs + rl * 128 + c4);
                const float rm = rowmask ? rowmask[row] : 1.f;
                if (vec) {
                    if (resid) {
                        const float4 x = *(const float4*)(resid + (long)row * p.ldr + col);
                        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                    }
                    v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                    if (of) *(float4*)(of + (long)row * p.ldo + col) = v;
                    if (ob) plane_store4(ob + (long)row * p.ldob, col, v.x, v.y, v.z, v.w, p.out_split);
                } else {
                    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (col + u >= p.n) break;
                        float t = vv[u];
                        if (resid) t += resid[(long)row * p.ldr + col + u];
                        t *= rm;
                        if (of) of[(long)row * p.ldo + col + u] = t;
                        if (ob) {
                            const unsigned short hi = f32_to_bf16(t);
                            char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                            *(unsigned short*)d = hi;
                            if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}


// =============================================================================================
// gemm_kernel_p3: the 128x128 / 4-wave / 128-byte-row kernel with a 3-stage weight ring and COUNTED
// vmcnt (weights are prefetched two steps ahead and stay in flight across the step barrier), paid for
// by single-buffering the A window (17 KiB + 3 x 16 KiB = 65 KiB -> still 2 workgroups per CU).  The
// window of the next chunk is issued at the top of its first tap step and waited for there (one
// exposed DMA round trip per chunk instead of one per step); the co-resident workgroup covers it.
// Ablation that motivated it (k5 conv, B=64): DMA-only loop 98 us, MFMA+LDS-only 109 us, both 168 us:
// the 2-stage ring serialises a ~1500-cycle DMA round trip into every step.  taps > 1 only.
// =============================================================================================
template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel_p3(GemmKernelArgs p) {
    constexpr int A_PIECES = 17, A_BYTES = A_PIECES * 1024, NST = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#define P3_WBUF(i) (smem + A_BYTES + ((i) % NST) * W_BYTES)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.y, z2 = blockIdx.z;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;

    const int ntot = p.mtiles * p.ntiles;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = p.m_base + mt * 128, n0 = nt * BN;
    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;
    const int a_first = m0 - p.pad;
    const int b_max = p.n - 1 - n0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_w = [&](int sn) {
        const int cn = sn / TAPS, kn = sn - cn * TAPS;
        const char* wb = Bw + (long)kn * p.b_tap_stride + (long)n0 * p.ldb + (long)cn * 128;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dma_piece(wb, p.ldb, 0, b_max, wave * 4 + pc, lane, P3_WBUF(sn));
    };
    auto issue_a = [&](int cn) {
        const char* ab = A + (long)cn * 128;
        for (int pc = wave; pc < A_PIECES; pc += 4) dma_piece(ab, p.lda, a_first, 0x7fffffff, pc, lane, smem);
    };

    issue_a(0);
    issue_w(0);
    if (nsteps > 1) issue_w(1);
    wait_vmcnt(nsteps > 1 ? 4 : 0);
    __builtin_amdgcn_s_barrier();

    int s = 0;
    for (int c = 0; c < p.nchunk; ++c) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k, ++s) {
            const bool do_w = (s + 2 < nsteps);
            if (k == 0 && c > 0) {
                // all waves have left the previous chunk (barrier below): reload the single window
                issue_a(c);
                if (do_w) issue_w(s + 2);
                wait_vmcnt(do_w ? 4 : 0);
                __builtin_amdgcn_s_barrier();
            } else if (do_w) {
                issue_w(s + 2);
            }
            const char* at = smem;
            const char* wt = P3_WBUF(s);
            const int arow = wm * 64 + lrow + k;
            const int brow = wn * 64 + lrow;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 af[2], bfr[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                        al[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bh[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
                        bl[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot + 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
            // weights of step s+1 must have landed; the tile issued this step (s+2) stays in flight
            wait_vmcnt((do_w && !(k == 0 && c > 0)) ? 4 : (do_w ? 4 : 0));
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float* cs = (float*)smem;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const int c4 = (tid & 31) << 2;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);
    {
        const float* bias = p.bias;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + j * 32 + lrow;
            const float bv = (bias && n0 + cl < p.n) ? bias[n0 + cl] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    float v = acc[i][j][r] * p.alpha + bv;
                    if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                    else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                    cs[rl * 128 + cl] = v;
                }
            }
        }
    }
    __syncthreads();
    if (col < p.n && !(p.dbg & 1)) {
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int rl = ps * 8 + (tid >> 5);
            const int row = m0 + rl;
            if (row >= p.m_end) break;
            float4 v = *(const float4*)(cs + rl * 128 + c4);
            const float rm = rowmask ? rowmask[row] : 1.f;
            if (vec) {
                if (resid) {
                    const float4 x = *(const float4*)(resid + (long)row * p.ldr + col);
                    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                }
                v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                if (of) *(float4*)(of + (long)row * p.ldo + col) = v;
                if (ob) plane_store4(ob + (long)row * p.ldob, col, v.x, v.y, v.z, v.w, p.out_split);
            } else {
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (col + u >= p.n) break;
                    float t = vv[u];
                    if (resid) t += resid[(long)row * p.ldr + col + u];
                    t *= rm;
                    if (of) of[(long)row * p.ldo + col + u] = t;
                    if (ob) {
                        const unsigned short hi = f32_to_bf16(t);
                        char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                        *(unsigned short*)d = hi;
                        if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                    }
                }
            }
        }
    }
}


// =============================================================================================
// gemm_kernel_w4: "wave-private weights".  128x128 tile, 4 waves, but each wave owns a 128x32 column
// slab (4 accumulator blocks) and streams ITS OWN 32 weight rows (4 KiB per step) into a private
// 2-slot LDS ring, waiting only on its own counted vmcnt.  The shared A window is still loaded
// cooperatively, so the workgroup needs ONE s_barrier per K-chunk (8 per tile) instead of one per
// step (40 per tile): between chunk boundaries the four waves run free and drift apart, which
// overlaps one wave's DMA wait with another's MFMAs inside the workgroup.  Same 66 KiB LDS -> 2
// workgroups per CU.  Costs 1.25 ds_read_b128 per MFMA instead of 1.0.
// =============================================================================================
template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel_w4(GemmKernelArgs p) {
    constexpr int A_PIECES = 16 + (TAPS == 1 ? 0 : 1);
    constexpr int A_BYTES = A_PIECES * 1024;
    constexpr int WP_BYTES = 32 * 128;                  // one private weight slot: 32 rows x 128 B
    extern __shared__ __attribute__((aligned(16))) char smem[];
#define W4_ABUF(i) (smem + ((i) & 1) * A_BYTES)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const wring = smem + 2 * A_BYTES + wave * (2 * WP_BYTES);
    const int z = blockIdx.y, z2 = blockIdx.z;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;
    const int nA = (A_PIECES - wave + 3) / 4;

    const int ntot = p.mtiles * p.ntiles;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = p.m_base + mt * 128, n0 = nt * BN;
    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;
    const int a_first = m0 - p.pad;
    const int b_last = p.n - 1 - n0;                    // last valid weight row of this n-tile (>= 0)
    const int wrow0 = min(wave * 32, b_last);           // this wave's first weight row (clamped into range)
    const int wmax = b_last - wrow0;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto issue_w = [&](int sn) {                        // this wave's 32 weight rows of step sn -> private slot sn & 1
        const int cn = sn / TAPS, kn = sn - cn * TAPS;
        const char* wb = Bw + (long)kn * p.b_tap_stride + (long)(n0 + wrow0) * p.ldb + (long)cn * 128;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dma_piece(wb, p.ldb, 0, wmax, pc, lane, wring + (sn & 1) * WP_BYTES);
    };
    auto issue_a = [&](int cn) {
        const char* ab = A + (long)cn * 128;
        for (int pc = wave; pc < A_PIECES; pc += 4) dma_piece(ab, p.lda, a_first, 0x7fffffff, pc, lane, W4_ABUF(cn));
    };

    issue_a(0);
    issue_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int s = 0;
    bool a_prev = false;                                // a window was issued in the previous step
    for (int c = 0; c < p.nchunk; ++c) {
        if (c > 0) {
            // chunk boundary: every wave has finished reading window c-1 and its pieces of window c
            // landed two steps ago (counted waits below) -> one barrier publishes window c
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int k = 0; k < TAPS; ++k, ++s) {
            const bool do_w = (s + 1 < nsteps);
            const bool do_a = (k == 0) && (c + 1 < p.nchunk);
            if (TAPS == 1 && do_a) issue_a(c + 1);       // taps 1: the next window is needed at the very next
            if (do_w) issue_w(s + 1);                     //         barrier, so it goes first and is waited for
            if (TAPS != 1 && do_a) issue_a(c + 1);
            // own weights of step s must have landed; younger DMAs stay in flight
            {
                int n = do_w ? 4 : 0;
                if (TAPS != 1) n += (do_a ? nA : 0) + (a_prev ? nA : 0);
                wait_vmcnt(n);
            }
            a_prev = do_a;
            const char* at = W4_ABUF(c);
            const char* wt = wring + (s & 1) * WP_BYTES;
            const int arow = lrow + k;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    const bf16x8 bfr = *(const bf16x8*)(wt + lds_off(lrow, slot));
                    bf16x8 af[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr, acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    const bf16x8 bh = *(const bf16x8*)(wt + lds_off(lrow, slot));
                    const bf16x8 bl = *(const bf16x8*)(wt + lds_off(lrow, slot + 4));
                    bf16x8 ah[4], al[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ah[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                        al[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i], 0, 0, 0);
                    }
                }
            }
            // the private slot read in this step is overwritten by the DMA issued at the top of step
            // s+1 by THIS wave: its ds_reads have retired (their data fed the MFMAs above)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float* cs = (float*)smem;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const int c4 = (tid & 31) << 2;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);
    {
        const int cl = wave * 32 + lrow;
        const float bv = (p.bias && n0 + cl < p.n) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                float v = acc[i][r] * p.alpha + bv;
                if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                cs[rl * 128 + cl] = v;
            }
        }
    }
    __syncthreads();
    if (col < p.n && !(p.dbg & 1)) {
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
            const int rl = ps * 8 + (tid >> 5);
            const int row = m0 + rl;
            if (row >= p.m_end) break;
            float4 v = *(const float4*)(cs + rl * 128 + c4);
            const float rm = rowmask ? rowmask[row] : 1.f;
            if (vec) {
                if (resid) {
                    const float4 x = *(const float4*)(resid + (long)row * p.ldr + col);
                    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                }
                v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                if (of) *(float4*)(of + (long)row * p.ldo + col) = v;
                if (ob) plane_store4(ob + (long)row * p.ldob, col, v.x, v.y, v.z, v.w, p.out_split);
            } else {
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (col + u >= p.n) break;
                    float t = vv[u];
                    if (resid) t += resid[(long)row * p.ldr + col + u];
                    t *= rm;
                    if (of) of[(long)row * p.ldo + col + u] = t;
                    if (ob) {
                        const unsigned short hi = f32_to_bf16(t);
                        char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                        *(unsigned short*)d = hi;
                        if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                    }
                }
            }
        }
    }
}


// =============================================================================================
// gemm_kernel_t256: 256x128 tile, 4 waves of 128x64 (acc[4][2]), full 128-byte LDS rows, SINGLE A window
// (33 KiB) + 2-stage weight ring (2 x 16 KiB) = 65 KiB -> 2 workgroups per CU.  Per FLOP it issues
// 0.57x the LDS-DMA pieces of the 128x128 kernel (weights amortised over 256 rows) as FULL cache
// lines, which is what the round-1 ablations point at: every 128-row variant saturates at the same
// main-loop time, i.e. at the CU's LDS-DMA throughput.  The window of the next chunk is loaded at the
// chunk boundary (exposed once per 5 steps, covered by the co-resident workgroup).  taps > 1 only.
// =============================================================================================
template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel_t256(GemmKernelArgs p) {
    constexpr int A_PIECES = 33, A_BYTES = A_PIECES * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#define T256_WBUF(i) (smem + A_BYTES + ((i) & 1) * W_BYTES)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.y, z2 = blockIdx.z;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;

    const int ntot = p.mtiles * p.ntiles;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = p.m_base + mt * 256, n0 = nt * BN;
    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;
    const int a_first = m0 - p.pad;
    const int b_max = p.n - 1 - n0;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto issue_w = [&](int sn) {
        const int cn = sn / TAPS, kn = sn - cn * TAPS;
        const char* wb = Bw + (long)kn * p.b_tap_stride + (long)n0 * p.ldb + (long)cn * 128;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dma_piece(wb, p.ldb, 0, b_max, wave * 4 + pc, lane, T256_WBUF(sn));
    };
    auto issue_a = [&](int cn) {
        const char* ab = A + (long)cn * 128;
        for (int pc = wave; pc < A_PIECES; pc += 4) dma_piece(ab, p.lda, a_first, 0x7fffffff, pc, lane, smem);
    };

    issue_a(0);
    issue_w(0);
    __syncthreads();

    int s = 0;
    for (int c = 0; c < p.nchunk; ++c) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k, ++s) {
            if (k == 0 && c > 0) {
                issue_a(c);                       // previous step's barrier retired every read of window c-1
                if (s + 1 < nsteps) issue_w(s + 1);
                __syncthreads();
            } else if (s + 1 < nsteps) {
                issue_w(s + 1);
            }
            const char* at = smem;
            const char* wt = T256_WBUF(s);
            const int arow = wm * 128 + lrow + k;
            const int brow = wn * 64 + lrow;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 af[4], bfr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bh[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
                        bl[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot + 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ah[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                        al[i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
            __syncthreads();
        }
    }

    float* cs = (float*)smem;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const int c4 = (tid & 31) << 2;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + lrow;
                const float bv = (p.bias && n0 + cl < p.n) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[i][j][r] * p.alpha + bv;
                        if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                        cs[rl * 128 + cl] = v;
                    }
                }
            }
        }
        __syncthreads();
        if (col < p.n && !(p.dbg & 1)) {
#pragma unroll 4
            for (int ps = 0; ps < 16; ++ps) {
                const int rl = ps * 8 + (tid >> 5);
                const int row = m0 + half * 128 + rl;
                if (row >= p.m_end) break;
                float4 v = *(const float4*)(cs + rl * 128 + c4);
                const float rm = rowmask ? rowmask[row] : 1.f;
                if (vec) {
                    if (resid) {
                        const float4 x = *(const float4*)(resid + (long)row * p.ldr + col);
                        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                    }
                    v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                    if (of) *(float4*)(of + (long)row * p.ldo + col) = v;
                    if (ob) plane_store4(ob + (long)row * p.ldob, col, v.x, v.y, v.z, v.w, p.out_split);
                } else {
                    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (col + u >= p.n) break;
                        float t = vv[u];
                        if (resid) t += resid[(long)row * p.ldr + col + u];
                        t *= rm;
                        if (of) of[(long)row * p.ldo + col + u] = t;
                        if (ob) {
                            const unsigned short hi = f32_to_bf16(t);
                            char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                            *(unsigned short*)d = hi;
                            if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace efts

using namespace efts;

template <int T, int S, int W>
static void launch_gemm(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    int lds = Cfg<T, W>::LDS;
    if (W == 2) { const char* e = getenv("EFTS_GEMM_LDS_PAD"); if (e) lds += atoi(e); }
    hipLaunchKernelGGL((gemm_kernel<T, S, W>), grid, dim3(W * 128), lds, st, k);
}
template <int T, int S>
static void launch_gemm_v3(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    constexpr int lds = 2 * (V3_BM / 16 + (T == 1 ? 0 : 1)) * 1024 + V3_NST * V3_W_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_kernel_v3<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((gemm_kernel_v3<T, S>), grid, dim3(256), lds, st, k);
}
template <int T, int S>
static void launch_gemm_v4(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    constexpr int lds = 2 * (128 / 16 + (T == 1 ? 0 : 1)) * 1024 + 3 * V3_W_BYTES;
    hipLaunchKernelGGL((gemm_kernel_v4<T, S>), grid, dim3(256), lds, st, k);
}
template <int T, int S>
static void launch_gemm_t256(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    constexpr int lds = 33 * 1024 + 2 * W_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_kernel_t256<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((gemm_kernel_t256<T, S>), grid, dim3(256), lds, st, k);
}
template <int T, int S>
static void launch_gemm_w4(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    constexpr int lds = 2 * (16 + (T == 1 ? 0 : 1)) * 1024 + 4 * 2 * 4096;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_kernel_w4<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((gemm_kernel_w4<T, S>), grid, dim3(256), lds, st, k);
}
template <int T, int S>
static void launch_gemm_p3(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    constexpr int lds = 17 * 1024 + 3 * W_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_kernel_p3<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((gemm_kernel_p3<T, S>), grid, dim3(256), lds, st, k);
}
template <int T, int S>
static void set_lds_attr() {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<T, S, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemm_kernel<T, S, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

extern "C" int efts_gemm(const efts_gemm_args* a, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_gemm: null args");
    if (!(a->split == 1 || a->split == 2)) return efts_fail(EFTS_EINVAL, "efts_gemm: split must be 1 or 2");
    if (!(a->taps == 1 || a->taps == 3 || a->taps == 5)) return efts_fail(EFTS_EINVAL, "efts_gemm: taps must be 1, 3 or 5");
    if (a->m <= 0 || a->n <= 0 || a->nchunk <= 0 || a->batch <= 0) return efts_fail(EFTS_ESHAPE, "efts_gemm: m, n, nchunk, batch must be positive");
    if (!a->a || !a->b) return efts_fail(EFTS_EINVAL, "efts_gemm: null operand");
    if (((uintptr_t)a->a & 15) || ((uintptr_t)a->b & 15) || (a->lda & 15) || (a->ldb & 15) || (a->b_tap_stride & 15) ||
        (a->a_batch_stride & 15) || (a->b_batch_stride & 15))
        return efts_fail(EFTS_EALIGN, "efts_gemm: operand planes must be 16-byte aligned (pointer, row, tap and batch strides)");
    if (a->lda < (int64_t)a->nchunk * 128 || a->ldb < (int64_t)a->nchunk * 128)
        return efts_fail(EFTS_ESHAPE, "efts_gemm: row stride smaller than nchunk*128 bytes");
    if (a->out_bf16 && !(a->out_split == 1 || a->out_split == 2)) return efts_fail(EFTS_EINVAL, "efts_gemm: out_split must be 1 or 2");
    if (!a->out_f32 && !a->out_bf16) return efts_fail(EFTS_EINVAL, "efts_gemm: no output");

    GemmKernelArgs k;
    k.a = (const char*)a->a; k.b = (const char*)a->b;
    k.bias = a->bias; k.resid = a->resid; k.rowmask = a->rowmask;
    k.out_f32 = a->out_f32; k.out_bf16 = (char*)a->out_bf16;
    k.lda = a->lda; k.ldb = a->ldb; k.b_tap_stride = a->b_tap_stride; k.ldr = a->ldr; k.ldo = a->ldo; k.ldob = a->ldob;
    k.a_bs = a->a_batch_stride; k.b_bs = a->b_batch_stride; k.r_bs = a->resid_batch_stride;
    k.m_bs = a->rowmask_batch_stride; k.o_bs = a->out_batch_stride; k.ob_bs = a->outb_batch_stride;
    k.a_bs2 = a->a_batch2_stride; k.b_bs2 = a->b_batch2_stride; k.o_bs2 = a->out_batch2_stride;
    const int nb2 = a->batch2 > 1 ? a->batch2 : 1;
    if (nb2 > 1 && (a->out_bf16 || a->resid || a->rowmask)) return efts_fail(EFTS_EINVAL, "efts_gemm: batch2 supports fp32 output only");
    k.n = a->n; k.nchunk = a->nchunk; k.pad = (a->taps - 1) / 2;
    k.ntiles = (a->n + BN - 1) / BN;
    k.alpha = a->alpha; k.slope = a->slope; k.act = a->act; k.out_split = a->out_split;
    k.vec_ok = (!a->out_f32 || ((a->ldo & 3) == 0 && ((uintptr_t)a->out_f32 & 15) == 0 && (a->out_batch_stride & 3) == 0)) &&
               (!a->resid || ((a->ldr & 3) == 0 && ((uintptr_t)a->resid & 15) == 0 && (a->resid_batch_stride & 3) == 0)) &&
               (!a->out_bf16 || ((a->ldob & 7) == 0 && ((uintptr_t)a->out_bf16 & 7) == 0 && (a->outb_batch_stride & 7) == 0));
    { const char* e = getenv("EFTS_GEMM_DBG"); k.dbg = e ? atoi(e) : 0; }
    { const char* e = getenv("EFTS_GEMM_STAGGER"); k.stagger = e ? atoi(e) : 0; }
    { const char* e = getenv("EFTS_GEMM_SPREAD"); k.spread = e ? atoi(e) : 0; }
    hipStream_t st = (hipStream_t)stream;

    // Tile plan.  Default: the 128x128 / 2-stage kernel everywhere.  EFTS_GEMM_TILE=256 routes
    // non-batched launches to gemm_kernel_v3 (256x128 tile, 64-byte LDS rows, 4-stage ring, counted
    // vmcnt): correct, and measured 5-8 % SLOWER on MI355X this round, kept for
    // the next round's work on the staging path.
    int big_rows = 0, t256 = 0;
    {
        const char* e = getenv("EFTS_GEMM_TILE");
        if (e && (atoi(e) == 256 || (atoi(e) == 2560 && a->taps > 1)) && a->batch == 1 && nb2 == 1) big_rows = a->m;
        t256 = e && atoi(e) == 2560;
    }
#define EFTS_LAUNCH_TS(W)                                                                                   \
    do {                                                                                                    \
        if (a->split == 1) {                                                                                \
            if (a->taps == 5) launch_gemm<5, 1, W>(grid, st, k); else if (a->taps == 3) launch_gemm<3, 1, W>(grid, st, k); else launch_gemm<1, 1, W>(grid, st, k); \
        } else {                                                                                            \
            if (a->taps == 5) launch_gemm<5, 2, W>(grid, st, k); else if (a->taps == 3) launch_gemm<3, 2, W>(grid, st, k); else launch_gemm<1, 2, W>(grid, st, k); \
        }                                                                                                   \
    } while (0)
    if (big_rows > 0) {
        k.m_base = 0; k.m_end = big_rows < a->m ? big_rows : a->m;
        k.mtiles = (k.m_end - k.m_base + 255) / 256;
        dim3 grid(k.mtiles * k.ntiles, a->batch, nb2);
        if (t256) {
            if (a->split == 1) { if (a->taps == 5) launch_gemm_t256<5, 1>(grid, st, k); else launch_gemm_t256<3, 1>(grid, st, k); }
            else { if (a->taps == 5) launch_gemm_t256<5, 2>(grid, st, k); else launch_gemm_t256<3, 2>(grid, st, k); }
        } else if (a->split == 1) {
            if (a->taps == 5) launch_gemm_v3<5, 1>(grid, st, k); else if (a->taps == 3) launch_gemm_v3<3, 1>(grid, st, k); else launch_gemm_v3<1, 1>(grid, st, k);
        } else {
            if (a->taps == 5) launch_gemm_v3<5, 2>(grid, st, k); else if (a->taps == 3) launch_gemm_v3<3, 2>(grid, st, k); else launch_gemm_v3<1, 2>(grid, st, k);
        }
    }
    if (big_rows < a->m) {
        // 128x128 tiles, one workgroup per tile, 2 resident per CU.  If the tile count leaves a
        // thinly filled last round (e.g. 1604 tiles on 512 slots = 3.13 rounds), the rows of that
        // partial round go to 64x128 tiles (2-wave workgroups) in a second launch: the tail then costs
        // about half a round instead of a whole one.  Measured SLOWER (244 vs 231 us: freed slots already let
        // the last workgroups run alone at full speed), so it is opt-in: EFTS_GEMM_TAIL=1.
        int main_end = a->m;
        {
            const char* e = getenv("EFTS_GEMM_TAIL");
            const int tail_on = e ? atoi(e) : 0;
            const long slots = 2L * efts_num_cus();
            const long mt128 = (a->m - big_rows + 127) / 128;
            const long tiles = mt128 * k.ntiles;
            const long full = tiles / slots, rem = tiles - full * slots;
            if (tail_on && a->batch == 1 && nb2 == 1 && full >= 1 && rem > 0 && rem * 10 <= slots * 6) {
                const long main_mt = (full * slots) / k.ntiles;
                main_end = big_rows + (int)(main_mt * 128);
            }
        }
        k.m_base = big_rows; k.m_end = main_end;
        k.mtiles = (k.m_end - k.m_base + 127) / 128;
        const int nt_all = k.mtiles * k.ntiles;
        // one workgroup per tile by default; EFTS_GEMM_PERSIST=1 runs 2 persistent workgroups per CU
        // instead (measured neutral on MI355X)
        int cap = nt_all;
        { const char* e = getenv("EFTS_GEMM_PERSIST"); if (e && atoi(e) == 1) cap = 2 * efts_num_cus(); }
        if (a->batch > 1) cap = nt_all;
        dim3 grid(nt_all < cap ? nt_all : cap, a->batch, nb2);
        int p3 = 0, v4 = 0;
        { const char* e = getenv("EFTS_GEMM_P3"); p3 = e ? atoi(e) : 0; }
        { const char* e = getenv("EFTS_GEMM_V4"); v4 = e ? atoi(e) : 0; }
        int w4 = 0;
        { const char* e = getenv("EFTS_GEMM_W4"); w4 = e ? atoi(e) : 0; }
        if (w4 && cap == nt_all) {
            if (a->split == 1) { if (a->taps == 5) launch_gemm_w4<5, 1>(grid, st, k); else if (a->taps == 3) launch_gemm_w4<3, 1>(grid, st, k); else launch_gemm_w4<1, 1>(grid, st, k); }
            else { if (a->taps == 5) launch_gemm_w4<5, 2>(grid, st, k); else if (a->taps == 3) launch_gemm_w4<3, 2>(grid, st, k); else launch_gemm_w4<1, 2>(grid, st, k); }
        } else if (v4 && cap == nt_all) {
            if (a->split == 1) { if (a->taps == 5) launch_gemm_v4<5, 1>(grid, st, k); else if (a->taps == 3) launch_gemm_v4<3, 1>(grid, st, k); else launch_gemm_v4<1, 1>(grid, st, k); }
            else { if (a->taps == 5) launch_gemm_v4<5, 2>(grid, st, k); else if (a->taps == 3) launch_gemm_v4<3, 2>(grid, st, k); else launch_gemm_v4<1, 2>(grid, st, k); }
        } else if (p3 && a->taps > 1 && cap == nt_all) {
            if (a->split == 1) { if (a->taps == 5) launch_gemm_p3<5, 1>(grid, st, k); else launch_gemm_p3<3, 1>(grid, st, k); }
            else { if (a->taps == 5) launch_gemm_p3<5, 2>(grid, st, k); else launch_gemm_p3<3, 2>(grid, st, k); }
        } else {
            EFTS_LAUNCH_TS(2);
        }
        if (main_end < a->m) {
            k.m_base = main_end; k.m_end = a->m;
            k.mtiles = (k.m_end - k.m_base + 63) / 64;
            dim3 grid(k.mtiles * k.ntiles, a->batch, nb2);
            EFTS_LAUNCH_TS(1);
        }
    }
#undef EFTS_LAUNCH_TS
    return efts_check_launch("efts_gemm");
}

// Opt every instantiation into > 64 KiB of dynamic LDS once, at library load.
namespace {
struct GemmInit {
    GemmInit() {
        set_lds_attr<5, 1>(); set_lds_attr<3, 1>(); set_lds_attr<1, 1>(); set_lds_attr<5, 2>(); set_lds_attr<3, 2>(); set_lds_attr<1, 2>();
    }
};
}  // namespace
extern "C" void efts_gemm_init(void) {
    static GemmInit once;
    if (getenv("EFTS_DEBUG")) {
        int nb = -1;
        constexpr int l2 = Cfg<5, 2>::LDS, l4 = Cfg<5, 4>::LDS;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm_kernel<5, 1, 2>, 256, l2);
        fprintf(stderr, "[efts] gemm_kernel<5,1,2>: %d workgroups/CU at %d B LDS\n", nb, l2);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm_kernel<5, 1, 4>, 512, l4);
        fprintf(stderr, "[efts] gemm_kernel<5,1,4>: %d workgroups/CU at %d B LDS\n", nb, l4);
    }
}
