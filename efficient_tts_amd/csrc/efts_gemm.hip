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
//     whose A tile is ONE 128-row LDS window read at `taps` row shifts: the window is staged
//     once per K-chunk and re-used by every tap (LDS-staged conv window).  A tile therefore
//     produces 128 - (taps - 1) output rows (124 for k5): the window is exactly 16 KiB, and
//     2 windows + a 3-stage weight ring are exactly 80 KiB = two workgroups per CU.
//   * 256-thread workgroup, 2x2 waves of 64x64, 32x32x16 bf16 MFMA, fp32 accumulators.
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4, 16 B/lane), issued from inline asm with
//     a scalar base + a per-lane 32-bit offset computed once per tile.  (Issued through the clang
//     builtin, hipcc cannot prove that a window read does not alias an in-flight DMA and puts an
//     s_waitcnt vmcnt(0) in front of the ds_reads of every step, which serialises the pipeline.)
//     The weight tile of step s+2 is in flight while step s computes; steps end with a COUNTED
//     s_waitcnt vmcnt(N) + s_barrier.  The LDS image is lane-linear, so the bank-conflict swizzle is
//     applied on the per-lane SOURCE address and again on the ds_read_b128 address.
//   * K is consumed in 128-byte chunks: 64 bf16 ("bf16"), or 32 hi + 32 lo bf16 ("bf16x3":
//     x = hi + lo, product = hi*hi + hi*lo + lo*hi, fp32 accumulate -> fp32-class accuracy at
//     3 MFMAs per product instead of the 16x slower f32 MFMA).
//   * epilogue fused: bias, LeakyReLU/ReLU, residual add, row mask (gap rows / padded
//     positions), fp32 store and bf16 operand planes for the next contraction.  The residual and
//     mask values are requested at the start of the LAST K step, so their HBM latency hides under
//     that step and the LDS staging of the accumulators; stores are never waited for.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "efts_gemm_kernels.h"

namespace efts {

template <int TAPS, int SPLIT, int DBG>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmKernelArgs p) {
    const int BM = p.bm;                      // output rows per tile: WIN - (TAPS - 1) * dilation
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tq = 0;
#define EFTS_STAMP(i) do { if constexpr (DBG == 2) { const unsigned long long tn = __builtin_readcyclecounter(); pt[i] += tn - tq; tq = tn; } } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // all LDS in one array, buffers addressed arithmetically
#define EFTS_ABUF(i) (smem + ((i) & 1) * TILE_BYTES)
#define EFTS_WBUF(i) (smem + 2 * TILE_BYTES + ((i) % NST) * TILE_BYTES)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int dbg = DBG ? p.dbg : 0;

    const int z = blockIdx.y;
    const int z2 = blockIdx.z;
    const int lrow = lane & 31;
    const int lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;
    const int c4 = (tid & 31) << 2;
    constexpr int RPP = 8;                    // tile rows per epilogue sweep (256 threads / 32)
    constexpr int NPS = WIN / RPP;            // 16 sweeps
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    char* obl = (DBG == 0 && p.out_lo) ? p.out_lo + (long)z * p.ob_bs : nullptr;
    const char* A = p.a + (long)z * p.a_bs + (long)z2 * p.a_bs2;
    const char* Bw = p.b + (long)z * p.b_bs + (long)z2 * p.b_bs2;

    // Workgroups walk the tile list with stride gridDim.x (one tile each by default).
    // XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous range of
    // tiles (n fastest) so the workgroups sharing an A window hit the same L2.
    const int ntot = p.mtiles * p.ntiles;
  for (int vt = blockIdx.x; vt < ntot; vt += gridDim.x) {
    int bid = vt;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);

    // per-lane DMA offsets of the 4 window pieces and 4 weight pieces this wave issues per tile:
    // piece pc covers tile rows 8*pc .. 8*pc+7; lane l -> row 8*pc + l/8, physical slot l%8
    unsigned voa[4], vow[4];
    {
        const int b_max = p.n - 1 - n0;        // clamp B rows to the last real row
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            voa[q] = (unsigned)(r * (int)p.lda + (sl << 4));
            vow[q] = (unsigned)((r < b_max ? r : b_max) * (int)p.ldb + (sl << 4));
        }
    }
    // debug bits 8 / 16 (timing experiments): every workgroup stages the SAME weight tile / window (always L1-resident)
    const bool alias_w = DBG && (dbg & 8), alias_a = DBG && (dbg & 16);
    const char* a_base = A + (long)((alias_a ? 0 : m0) - p.pad * p.dil) * p.lda;     // window row 0 (may start in the guard rows)
    const char* w_base = Bw + (long)(alias_w ? 0 : n0) * p.ldb;
    const unsigned lds_piece = lds0 + wave * 4096;            // this wave's first piece inside a tile

    auto issue_w = [&](int cn, int kn, int slot) {     // weights of step (chunk cn, tap kn) -> ring slot
        if (alias_w) { cn = 0; kn = 0; }
        const char* sb = w_base + (long)kn * p.b_tap_stride + (long)cn * 128;
        const unsigned l = lds_piece + 2 * TILE_BYTES + slot * TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(l + q * 1024, vow[q], sb);
    };
    auto issue_a = [&](int cn) {     // A window of chunk cn -> window buffer cn & 1
        const char* sb = a_base + (long)(alias_a ? 0 : cn) * 128;
        const unsigned l = lds_piece + (cn & 1) * TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(l + q * 1024, voa[q], sb);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // MFMAs of one (chunk, tap) step: window buffer `ab`, weight ring slot `ws`, tap k
    auto compute = [&](int ab, int ws, int k) {
        const char* at = smem + ab * TILE_BYTES;
        const char* wt = smem + 2 * TILE_BYTES + ws * TILE_BYTES;
        const int arow = wm * 64 + lrow + k * p.dil;   // tile row of output row r at tap k is r + k * dilation
        const int brow = wn * 64 + lrow;
        if (DBG && (dbg & 4)) return;
        // Operand fragments are double-buffered in registers: the ds_reads of k-slice kk+1 are issued
        // before the MFMAs of slice kk (scheduler fenced), so only the first slice's LDS latency is
        // exposed per step and the waits are counted lgkmcnt(N).
        if constexpr (SPLIT == 1) {
            bf16x8 af[2][2], bfr[2][2];
            auto ld = [&](int kk, int b) {
                const int slot = kk * 2 + lhalf;
#pragma unroll
                for (int i = 0; i < 2; ++i) af[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[b][j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
            };
            ld(0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) ld(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
            auto ld = [&](int kk, int b) {
                const int slot = kk * 2 + lhalf;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                    al[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[b][j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
                    bl[b][j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot + 4));
                }
            };
            ld(0, 0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk + 1 < 2) ld(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk & 1][i], bh[kk & 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk & 1][i], bl[kk & 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk & 1][i], bh[kk & 1][j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // prologue: window 0, weights of steps 0 and 1; wait for window 0 + weights 0
    const bool dma_on = !(dbg & 2);
    if (dma_on) {
        issue_a(0);
        issue_w(0, 0, 0);
        if (nsteps > 1) issue_w(TAPS == 1 ? 1 : 0, TAPS == 1 ? 0 : 1, 1);
        wait_vmcnt(nsteps > 1 ? 4 : 0);
    }
    __builtin_amdgcn_s_barrier();

    // (c, k): this step; (c2, k2): the step whose weights are issued now (two ahead); ws: ring slot of this step
    int c = 0, k = 0, ws = 0;
    int c2 = (TAPS == 1) ? 2 : (TAPS == 2 ? 1 : 0), k2 = (TAPS == 1) ? 0 : 2 % TAPS;
    for (int s = 0; s + 1 < nsteps; ++s) {
        if constexpr (DBG == 2) tq = __builtin_readcyclecounter();
        // ---- issue: weights two steps ahead; the next window at the first tap of a chunk.
        // taps 1: the window is needed one step later, so it goes out BEFORE the weights.
        const bool do_w = (s + 2 < nsteps) && dma_on;
        const bool do_a = (k == 0) && (c + 1 < p.nchunk) && dma_on;
        if (TAPS == 1 && do_a) issue_a(c + 1);
        if (do_w) issue_w(c2, k2, ws == 0 ? 2 : ws - 1);      // slot (s + 2) % 3
        if (TAPS != 1 && do_a) issue_a(c + 1);
        EFTS_STAMP(0);
        compute(c & 1, ws, k);
        EFTS_STAMP(2);
        // ---- step end: the operands of step s+1 must have landed.  LDS-DMA completes in issue
        // order, so it is enough to bound what may still be in flight: everything issued AFTER the
        // weights of s+1, i.e. this step's issues and (taps > 1) a window issued one step ago.
        int n = do_w ? 4 : 0;
        if (TAPS != 1) {
            if (do_a) n += 4;
            if (k == 1 && c + 1 < p.nchunk && dma_on) n += 4;
        }
        wait_vmcnt(n);
        EFTS_STAMP(3);
        lds_barrier();
        EFTS_STAMP(4);
        if (++k == TAPS) { k = 0; ++c; }
        if (++k2 == TAPS) { k2 = 0; ++c2; }
        ws = (ws == 2) ? 0 : ws + 1;
    }

    // ---- last step: nothing left to stage.  The epilogue operands of this thread (16 residual
    // float4 + 16 row-mask values) are requested first so that their HBM latency hides under the
    // step's MFMAs and the LDS staging of the accumulators.
    // Addressing: raw buffer descriptors per tile, one per-thread byte offset, the sweep index in the
    // scalar offset; rows past the end of the matrix (or of this tile's 124 rows, for the stores) fall
    // outside the descriptor, so loads return 0 and stores are dropped without a per-row predicate.
    u32x4 rres[NPS];
    float rmv[NPS];
    const bool pre = vec && col < p.n && !(dbg & 1);
    const int rows_in = p.m - m0 < WIN ? p.m - m0 : WIN;      // readable rows of this tile
    const int rows_out = p.m - m0 < BM ? p.m - m0 : BM;       // rows this tile owns
    const unsigned trow = tid >> 5;
    if constexpr (DBG == 2) tq = __builtin_readcyclecounter();
    if (pre) {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(resid ? resid + (long)m0 * p.ldr : nullptr, resid ? (long)rows_in * p.ldr * 4 : 0);
        const __amdgpu_buffer_rsrc_t rk = make_rsrc(rowmask ? rowmask + m0 : nullptr, rowmask ? (long)rows_in * 4 : 0);
        const unsigned vr = trow * (unsigned)p.ldr * 4 + col * 4, sr = RPP * (unsigned)p.ldr * 4;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            rres[ps] = __builtin_amdgcn_raw_buffer_load_b128(rr, vr, ps * sr, EFTS_AUX_LD);
            rmv[ps] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rk, trow * 4, ps * RPP * 4, 0));
        }
    }
    EFTS_STAMP(0);
    compute((p.nchunk - 1) & 1, ws, TAPS - 1);
    EFTS_STAMP(2);
    lds_barrier();
    EFTS_STAMP(4);

    // ---- fused epilogue, staged through LDS so that every global access is a full-row vector.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); each
    // wave drops its 64x64 block (bias + activation applied) into a [128][128] fp32 LDS tile, then
    // 32 consecutive threads sweep one 512-byte tile row: residual add (prefetched), row mask,
    // float4 store and 8-byte bf16 (hi / lo) operand-plane stores.
    float* cs = (float*)smem;   // 64 KiB; the main loop's last barrier has retired all LDS reads
    if constexpr (DBG == 2) tq = __builtin_readcyclecounter();
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
                    else if (p.act == EFTS_ACT_TANH) v = tanhf(v);
                    cs[rl * 128 + cl] = v;
                }
            }
        }
    }
    lds_barrier();
    if (p.sidx) {
        // softmax over the valid columns of every row + its expected column index, from the staged tile (32 threads per row, 4
        // columns each; the three reductions run over the 32 lanes of a half wave).  ntiles == 1: the tile holds whole rows.
        const int kl = min(p.klen[z], p.n), ql = min(p.qlen[z], p.m);
        float* so = p.sidx + (long)z * p.m;
#pragma unroll 4
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * RPP + (int)trow;
            const int row = m0 + rl;
            const float4 v = *(const float4*)(cs + rl * 128 + c4);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = (c4 + u < kl) ? fmaxf(mx, vv[u]) : mx;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 32));
            float se = 0.f, si = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float e = (c4 + u < kl) ? __expf(vv[u] - mx) : 0.f;
                se += e;
                si += e * (float)(c4 + u);
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) { se += __shfl_xor(se, off, 32); si += __shfl_xor(si, off, 32); }
            if ((tid & 31) == 0 && rl < BM && row < p.m) so[row] = row < ql ? si / se : 0.f;
        }
    }
    float part_sq = 0.f;
    if (pre) {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(of ? of + (long)m0 * p.ldo : nullptr, of ? (long)rows_out * p.ldo * 4 : 0);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(ob ? ob + (long)m0 * p.ldob : nullptr, ob ? (long)rows_out * p.ldob : 0);
        const __amdgpu_buffer_rsrc_t rbl = make_rsrc(obl ? obl + (long)m0 * p.ldob : nullptr, obl ? (long)rows_out * p.ldob : 0);
        const unsigned vo = trow * (unsigned)p.ldo * 4 + col * 4, so = RPP * (unsigned)p.ldo * 4;
        const unsigned vb = trow * (unsigned)p.ldob + (unsigned)plane_off_hi(col, p.out_split), sb = RPP * (unsigned)p.ldob;
        const bool has_mask = rowmask != nullptr;
        float sq = 0.f;                                      // p.sqerr: this thread's share of sum (out - target)^2, sweeps in ascending order
        const unsigned ld_sg = (unsigned)(p.n >> 3);
        const __amdgpu_buffer_rsrc_t rsg = make_rsrc(p.sign ? p.sign + (long)m0 * ld_sg : nullptr, p.sign ? (long)rows_out * ld_sg : 0);
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * RPP + trow;
            float4 v = *(const float4*)(cs + rl * 128 + c4);
            if (p.sign) {                                    // (n % 128 == 0: `pre` is uniform, every lane of the wave is here)
                const unsigned long long b0 = __ballot(v.x > 0.f), b1 = __ballot(v.y > 0.f), b2 = __ballot(v.z > 0.f), b3 = __ballot(v.w > 0.f);
                if ((tid & 31) == 0) {                       // lanes 0 / 32: the two rows this wave sweeps
                    const int sh = tid & 32;
                    const u32x4 w = {(unsigned)(b0 >> sh), (unsigned)(b1 >> sh), (unsigned)(b2 >> sh), (unsigned)(b3 >> sh)};
                    store_b128(w, rsg, trow * ld_sg + (unsigned)(n0 >> 7) * 16, ps * RPP * ld_sg);
                }
            }
            if (p.drop_thresh) {                             // Dropout on the activated value, in front of the residual add
                const unsigned e0 = (unsigned)(m0 + rl) * (unsigned)p.n + col;
                v.x *= drop_scale(p.drop_seed_h, e0, p.drop_thresh, p.drop_inv_keep); v.y *= drop_scale(p.drop_seed_h, e0 + 1, p.drop_thresh, p.drop_inv_keep);
                v.z *= drop_scale(p.drop_seed_h, e0 + 2, p.drop_thresh, p.drop_inv_keep); v.w *= drop_scale(p.drop_seed_h, e0 + 3, p.drop_thresh, p.drop_inv_keep);
            }
            const u32x4 x = rres[ps];
            const float rm = has_mask ? rmv[ps] : 1.f;
            if (p.sqerr) {                                   // `resid` carries the loss target, not a residual (rows past the matrix: rm = 0)
                v.x *= rm; v.y *= rm; v.z *= rm; v.w *= rm;
                if (rm != 0.f) {
                    const float d0 = v.x - __uint_as_float(x.x), d1 = v.y - __uint_as_float(x.y), d2 = v.z - __uint_as_float(x.z), d3 = v.w - __uint_as_float(x.w);
                    sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            } else {
                v.x = (v.x + __uint_as_float(x.x)) * rm; v.y = (v.y + __uint_as_float(x.y)) * rm;
                v.z = (v.z + __uint_as_float(x.z)) * rm; v.w = (v.w + __uint_as_float(x.w)) * rm;
            }
            if (of) {
                const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                { __builtin_amdgcn_raw_buffer_store_b128(o, ro, vo, ps * so, EFTS_AUX_STF); asm volatile("s_nop 4" ::"v"(o)); }
            }
            if (ob) {
                if (p.plane_act) {
                    v.x = v.x > 0.f ? v.x : v.x * p.plane_slope; v.y = v.y > 0.f ? v.y : v.y * p.plane_slope;
                    v.z = v.z > 0.f ? v.z : v.z * p.plane_slope; v.w = v.w > 0.f ? v.w : v.w * p.plane_slope;
                }
                float r0, r1, r2, r3;
                const u32x2 hi = {pack_bf16x2(v.x, v.y, &r0, &r1), pack_bf16x2(v.z, v.w, &r2, &r3)};
                __builtin_amdgcn_raw_buffer_store_b64(hi, rb, vb, ps * sb, EFTS_AUX_STP);
                if (p.out_split == 2) {
                    float d0, d1;
                    const u32x2 lo = {pack_bf16x2(r0, r1, &d0, &d1), pack_bf16x2(r2, r3, &d0, &d1)};
                    __builtin_amdgcn_raw_buffer_store_b64(lo, rb, vb + 64, ps * sb, EFTS_AUX_STP);
                } else if (obl) {
                    float d0, d1;
                    const u32x2 lo = {pack_bf16x2(r0, r1, &d0, &d1), pack_bf16x2(r2, r3, &d0, &d1)};
                    __builtin_amdgcn_raw_buffer_store_b64(lo, rbl, vb, ps * sb, EFTS_AUX_STP);
                }
            }
        }
        if (p.sqerr) part_sq = sq;
    } else if (col < p.n && !(dbg & 1)) {
#pragma unroll 4
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * RPP + (tid >> 5);
            const int row = m0 + rl;
            if (rl >= BM || row >= p.m) break;
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
                if (obl) {
                    plane_store4(obl + (long)row * p.ldob, col, v.x - bf16_to_f32(f32_to_bf16(v.x)), v.y - bf16_to_f32(f32_to_bf16(v.y)),
                                 v.z - bf16_to_f32(f32_to_bf16(v.z)), v.w - bf16_to_f32(f32_to_bf16(v.w)), 1);
                }
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
                        if (p.plane_act) t = t > 0.f ? t : t * p.plane_slope;
                        const unsigned short hi = f32_to_bf16(t);
                        char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                        *(unsigned short*)d = hi;
                        if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                        else if (obl) *(unsigned short*)(obl + (long)row * p.ldob + (col + u) * 2) = f32_to_bf16(t - bf16_to_f32(hi));
                    }
                }
            }
        }
    }
    if (p.sqerr) {   // every wave adds its 64 shares up the same way (xor butterfly) and owns one slot: no atomics, no run-to-run difference
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part_sq += __shfl_xor(part_sq, off, 64);
        if (lane == 0) p.sqerr[((long)z * ntot + bid) * 4 + wave] = part_sq;
    }
    lds_barrier();   // the LDS tile is re-used by the next tile's operand ring; stores drain on their own
    if constexpr (DBG == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EFTS_STAMP(5);
    }
  }   // tile loop
    if constexpr (DBG == 2) {
        if (lane == 0 && p.prof) {
            for (int i = 0; i < 6; ++i) atomicAdd(p.prof + i, pt[i]);
            atomicAdd(p.prof + 6, 1ull);
        }
    }
#undef EFTS_STAMP
#undef EFTS_ABUF
#undef EFTS_WBUF
}


}  // namespace efts

using namespace efts;

template <int T, int S, int D>
static void launch_one(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    if (D) (void)hipFuncSetAttribute((const void*)gemm_kernel<T, S, D>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    hipLaunchKernelGGL((gemm_kernel<T, S, D>), grid, dim3(256), GEMM_LDS, st, k);
}

#ifdef EFTS_LAB
// Lab builds: debug instantiation (k5 only) honouring EFTS_GEMM_DBG ablation bits and, with EFTS_GEMM_PROF=1, printing
// per-phase s_memtime sums of every wave.  Synchronises the stream.
template <int S>
static void launch_debug(dim3 grid, hipStream_t st, GemmKernelArgs k, int prof) {
    static unsigned long long* buf = nullptr;
    if (prof) {
        if (!buf) (void)hipMalloc((void**)&buf, 64);
        (void)hipMemsetAsync(buf, 0, 64, st);
        k.prof = buf;
    }
    if (prof) launch_one<5, S, 2>(grid, st, k); else launch_one<5, S, 1>(grid, st, k);
    if (prof) {
        unsigned long long h[8];
        (void)hipMemcpyAsync(h, buf, 56, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        const double w = (double)h[6];
        fprintf(stderr, "[efts prof] waves %.0f  per-wave cycles: issue %.0f  mfma+reads %.0f  dma-wait %.0f  barrier %.0f  epilogue %.0f\n",
                w, h[0] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w);
    }
}
#endif

extern "C" int efts_gemm(const efts_gemm_args* a, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_gemm: null args");
    if (!(a->split == 1 || a->split == 2)) return efts_fail(EFTS_EINVAL, "efts_gemm: split must be 1 or 2");
    if (!(a->taps == 1 || a->taps == 3 || a->taps == 5 || a->taps == 7 || a->taps == 9 || a->taps == 11)) return efts_fail(EFTS_EINVAL, "efts_gemm: taps must be 1, 3, 5, 7, 9 or 11");
    const int dil = a->dilation > 0 ? a->dilation : 1;
    if ((a->taps - 1) * dil > 64) return efts_fail(EFTS_ESHAPE, "efts_gemm: (taps - 1) * dilation must not exceed 64 rows");
    if (a->m <= 0 || a->n <= 0 || a->nchunk <= 0 || a->batch <= 0) return efts_fail(EFTS_ESHAPE, "efts_gemm: m, n, nchunk, batch must be positive");
    if (!a->a || !a->b) return efts_fail(EFTS_EINVAL, "efts_gemm: null operand");
    if (((uintptr_t)a->a & 15) || ((uintptr_t)a->b & 15) || (a->lda & 15) || (a->ldb & 15) || (a->b_tap_stride & 15) ||
        (a->a_batch_stride & 15) || (a->b_batch_stride & 15))
        return efts_fail(EFTS_EALIGN, "efts_gemm: operand planes must be 16-byte aligned (pointer, row, tap and batch strides)");
    if (a->lda < (int64_t)a->nchunk * 128 || a->ldb < (int64_t)a->nchunk * 128)
        return efts_fail(EFTS_ESHAPE, "efts_gemm: row stride smaller than nchunk*128 bytes");
    if (a->lda > (1 << 23) || a->ldb > (1 << 23)) return efts_fail(EFTS_ESHAPE, "efts_gemm: row stride above 8 MiB");
    if (a->out_bf16 && !(a->out_split == 1 || a->out_split == 2)) return efts_fail(EFTS_EINVAL, "efts_gemm: out_split must be 1 or 2");
    if (!a->out_f32 && !a->out_bf16 && !a->soft_index) return efts_fail(EFTS_EINVAL, "efts_gemm: no output");

    GemmKernelArgs k;
    k.a = (const char*)a->a; k.b = (const char*)a->b;
    k.bias = a->bias; k.resid = a->resid; k.rowmask = a->rowmask;
    k.out_f32 = a->out_f32; k.out_bf16 = (char*)a->out_bf16; k.out_lo = (char*)a->out_bf16_lo; k.sign = (char*)a->sign_mask;
    k.sidx = a->soft_index; k.klen = a->key_len; k.qlen = a->query_len;
    k.sqerr = a->sqerr_part;
    k.drop_thresh = 0; k.drop_seed_h = 0; k.drop_inv_keep = 1.f;
    if (a->drop_p > 0.f) {
        if (!(a->drop_p < 1.f)) return efts_fail(EFTS_EINVAL, "efts_gemm: drop_p must be in [0, 1)");
        k.drop_thresh = (unsigned)((double)a->drop_p * 4294967296.0);
        k.drop_seed_h = hash_u32(a->drop_seed);
        k.drop_inv_keep = 1.f / (1.f - a->drop_p);
    }
    if (a->out_bf16_lo && (!a->out_bf16 || a->out_split != 1 || a->plane_act || ((uintptr_t)a->out_bf16_lo & 7)))
        return efts_fail(EFTS_EINVAL, "efts_gemm: out_bf16_lo goes with an un-activated split-1 out_bf16 plane (8-byte aligned)");
    k.lda = a->lda; k.ldb = a->ldb; k.b_tap_stride = a->b_tap_stride; k.ldr = a->ldr; k.ldo = a->ldo; k.ldob = a->ldob;
    k.a_bs = a->a_batch_stride; k.b_bs = a->b_batch_stride; k.r_bs = a->resid_batch_stride;
    k.m_bs = a->rowmask_batch_stride; k.o_bs = a->out_batch_stride; k.ob_bs = a->outb_batch_stride;
    k.a_bs2 = a->a_batch2_stride; k.b_bs2 = a->b_batch2_stride; k.o_bs2 = a->out_batch2_stride;
    const int nb2 = a->batch2 > 1 ? a->batch2 : 1;
    if (nb2 > 1 && (a->out_bf16 || a->resid || a->rowmask)) return efts_fail(EFTS_EINVAL, "efts_gemm: batch2 supports fp32 output only");
    k.m = a->m; k.n = a->n; k.nchunk = a->nchunk; k.pad = (a->taps - 1) / 2;
    const int bm = WIN - (a->taps - 1) * dil;
    k.dil = dil; k.bm = bm; k.plane_act = a->plane_act; k.plane_slope = a->plane_slope;
    k.mtiles = (a->m + bm - 1) / bm;
    k.ntiles = (a->n + BN - 1) / BN;
    k.alpha = a->alpha; k.slope = a->slope; k.act = a->act; k.out_split = a->out_split;
    k.vec_ok = (!a->out_f32 || ((a->ldo & 3) == 0 && ((uintptr_t)a->out_f32 & 15) == 0 && (a->out_batch_stride & 3) == 0)) &&
               (!a->resid || ((a->ldr & 3) == 0 && ((uintptr_t)a->resid & 15) == 0 && (a->resid_batch_stride & 3) == 0)) &&
               (!a->out_bf16 || ((a->ldob & 7) == 0 && ((uintptr_t)a->out_bf16 & 7) == 0 && (a->outb_batch_stride & 7) == 0));
    k.prof = nullptr; k.dbg = 0;
    if (a->soft_index && (!a->key_len || !a->query_len || a->n > BN || nb2 > 1 || a->resid || a->out_bf16 || a->taps != 1 || a->act != EFTS_ACT_NONE))
        return efts_fail(EFTS_EINVAL, "efts_gemm: soft_index needs key_len / query_len, n <= 128, one tap, no activation, residual, plane output or batch2");
    if (k.drop_thresh && (a->batch > 1 || nb2 > 1 || !k.vec_ok || a->n % 4 || (long)a->m * a->n > 0xffffffffL))
        return efts_fail(EFTS_EINVAL, "efts_gemm: dropout needs batch 1, 16-byte aligned fp32 / plane rows, n % 4 == 0 and m * n < 2^32");
    if (a->sign_mask && (a->n % 128 || a->batch > 1 || nb2 > 1 || !k.vec_ok || ((uintptr_t)a->sign_mask & 15)))
        return efts_fail(EFTS_EINVAL, "efts_gemm: sign_mask needs n %% 128 == 0, batch 1, 16-byte aligned output rows and mask");
    if (a->sqerr_part) {
        if (!a->sqerr_target || !a->rowmask || a->resid || a->taps != 1 || a->n > BN || (a->n & 3) || nb2 > 1 || k.drop_thresh || a->out_bf16 || a->soft_index ||
            (a->ld_target & 3) || (a->target_batch_stride & 3) || ((uintptr_t)a->sqerr_target & 15) || !k.vec_ok)
            return efts_fail(EFTS_EINVAL, "efts_gemm: sqerr_part needs sqerr_target (16-byte aligned rows), rowmask, one tap, n <= 128, n %% 4 == 0, fp32 output only, no residual / dropout / batch2");
        if (a->tiling > EFTS_TILING_GENERIC) return efts_fail(EFTS_EINVAL, "efts_gemm: sqerr_part needs the generic tiling");
        k.resid = a->sqerr_target; k.ldr = a->ld_target; k.r_bs = a->target_batch_stride;      // the target travels where the residual would
    }
    hipStream_t st = (hipStream_t)stream;

    // ---- which kernel.  `tiling` AUTO (0) applies the measured rules below; the explicit values exist for A/B runs and
    // for the bit-equality tests between the kernels (they all compute identical results).
    const int tiling = a->tiling;
    if (tiling < EFTS_TILING_AUTO || tiling > EFTS_TILING_SMALLM) return efts_fail(EFTS_EINVAL, "efts_gemm: unknown tiling %d", tiling);
    // (0) short row spaces on request (never AUTO: the K dimension is split across the waves, so the summation order -- not the
    //     operand rounding -- differs from the ring kernels): 64 x 32 tiles, fragments straight from global memory
    if (tiling == EFTS_TILING_SMALLM) {
        if (a->batch != 1 || nb2 != 1 || dil != 1 || !(a->taps == 1 || a->taps == 3 || a->taps == 5) || a->n % 32 || a->plane_act || a->act == EFTS_ACT_TANH ||
            a->out_bf16_lo || a->sign_mask || a->soft_index || k.drop_thresh || !k.vec_ok || (a->out_bf16 && ((a->ldob & 15) || ((uintptr_t)a->out_bf16 & 15))) ||
            (a->ldo & 3) || (a->ldr & 3) || a->n % 8)
            return efts_fail(EFTS_ESHAPE, "efts_gemm: the small-M tiling takes one dense 1 / 3 / 5-tap launch, n %% 32 == 0, 16-byte aligned rows, plain outputs");
        if (launch_smallm_any(a->split, a->taps, st, k)) return efts_check_launch("efts_gemm");
        return efts_fail(EFTS_ESHAPE, "efts_gemm: no small-M instantiation for taps %d", a->taps);
    }
    const bool generic_only = a->out_bf16_lo != nullptr || nb2 > 1 || a->soft_index != nullptr || a->sqerr_part != nullptr;      // the remainder plane / the outer batch / the soft index: gemm_kernel only
    const bool no_narrow = generic_only || a->sign_mask != nullptr || k.drop_thresh != 0;      // sign words / dropout: gemm_kernel and conv5_kernel only
    if (generic_only && tiling > EFTS_TILING_GENERIC) return efts_fail(EFTS_EINVAL, "efts_gemm: out_bf16_lo / batch2 need the generic tiling");
    const dim3 grid(k.mtiles * k.ntiles, a->batch, nb2);                  // one workgroup per 124 x 128 tile, 2 resident per CU

    // (1) one K chunk and at most 64 columns over many rows (the 32- and 64-channel stages of the vocoder): window + all taps
    //     resident in LDS, 32-column tiles (2.11 -> 2.03 ms per utterance, 11.1 -> 10.5 ms per batch of 8)
    const bool resident_ok = a->n <= 64 && a->nchunk == 1 && (a->taps - 1) * dil <= 64 && (a->taps == 3 || a->taps == 7 || a->taps == 11);
    if (tiling == EFTS_TILING_RESIDENT && !resident_ok) return efts_fail(EFTS_ESHAPE, "efts_gemm: the resident tiling needs n <= 64, one K chunk, taps 3 / 7 / 11");
    if (no_narrow && !generic_only && (tiling == EFTS_TILING_NARROW || tiling == EFTS_TILING_RESIDENT))
        return efts_fail(EFTS_EINVAL, "efts_gemm: sign_mask needs the generic or wide tiling");
    if (!no_narrow && resident_ok && (tiling == EFTS_TILING_RESIDENT || (tiling == EFTS_TILING_AUTO && a->m >= 8 * R32_WIN))) {
        GemmKernelArgs kr = k;
        kr.bm = R32_WIN - (a->taps - 1) * dil;
        kr.mtiles = (a->m + kr.bm - 1) / kr.bm;
        kr.ntiles = (a->n + 31) / 32;
        if (launch_resident32_any(a->split, a->taps, dim3(kr.mtiles * kr.ntiles, a->batch, 1), st, kr)) return efts_check_launch("efts_gemm");
    }
    // (2) outputs of at most 64 columns: column tile 64 / 32 -- and launches whose 128-column tiling cannot give every CU one
    //     workgroup (the text side of the acoustic model: 64 x 130 rows = 272 workgroups; the 256-channel stage of the vocoder
    //     at one utterance: 110): such a launch is bound by the per-workgroup step latency, and 64-column tiles double the
    //     number of workgroups that overlap (one per CU is the measured threshold; two was slower for the text side)
    const bool few = (long)k.mtiles * k.ntiles * a->batch < (long)efts_num_cus() && a->n > 64;
    if (!no_narrow && (tiling == EFTS_TILING_NARROW || (tiling == EFTS_TILING_AUTO && (a->n <= 64 || few)))) {
        GemmKernelArgs kn = k;
        kn.ntiles = a->n <= 32 ? 1 : (a->n + 63) / 64;
        if (launch_narrow_any(a->split, a->n <= 32 ? 32 : 64, a->taps, dim3(k.mtiles * kn.ntiles, a->batch, 1), st, kn)) return efts_check_launch("efts_gemm");
        if (tiling == EFTS_TILING_NARROW) return efts_fail(EFTS_ESHAPE, "efts_gemm: no narrow instantiation for taps %d", a->taps);
    }
    // (3) k5 convolutions with at least 400 (252-row tile x column tile) workgroups on bf16 planes: the 256-row kernel
    //     (+1.5 % on the training step's forward / dgrad launches; -2.5 % on bf16x3 planes, hence bf16 only in AUTO).  Its last
    //     window may reach mt5 * 252 + 2 - m rows past the matrix: only inside the 144 guard rows of the ABI.
    const bool wide_ok = a->taps == 5 && dil == 1 && !a->plane_act && a->act != EFTS_ACT_TANH;
    const int mt5 = (a->m + C5_BM - 1) / C5_BM;
    const bool wide_fits = (long)mt5 * C5_BM + 2 - a->m <= 144;
    if (tiling == EFTS_TILING_WIDE && !(wide_ok && wide_fits)) return efts_fail(EFTS_ESHAPE, "efts_gemm: the wide tiling is for dense k5 launches whose last 256-row window stays inside the guard rows");
    if (!generic_only && wide_ok && wide_fits &&
        (tiling == EFTS_TILING_WIDE || (tiling == EFTS_TILING_AUTO && a->split == 1 && (long)mt5 * k.ntiles * a->batch >= C5_DEFAULT_MIN_TILES))) {
        GemmKernelArgs k5 = k;
        k5.mtiles = mt5;
        launch_conv5_any(a->split, dim3(mt5 * k.ntiles, a->batch, 1), st, k5);
        return efts_check_launch("efts_gemm");
    }
#ifdef EFTS_LAB
    {   // lab builds: the 8-wave experiment and the ablation / cycle-stamp instantiations of the generic kernel
        const char* e8 = getenv("EFTS_CONV8");
        if (e8 && atoi(e8) == 1 && wide_ok && wide_fits && a->split == 1 && a->n % C8_BN == 0 && k.vec_ok && !generic_only) {
            GemmKernelArgs k8 = k;
            k8.mtiles = mt5;
            k8.ntiles = a->n / C8_BN;
            launch_conv8(dim3(mt5 * k8.ntiles, a->batch, 1), st, k8);
            return efts_check_launch("efts_gemm");
        }
        const char* ed = getenv("EFTS_GEMM_DBG");
        const char* ep = getenv("EFTS_GEMM_PROF");
        const int dbg = ed ? atoi(ed) : 0, prof = ep ? atoi(ep) : 0;
        if ((dbg || prof) && a->taps == 5) {
            k.dbg = dbg;
            if (a->split == 1) launch_debug<1>(grid, st, k, prof); else launch_debug<2>(grid, st, k, prof);
            return efts_check_launch("efts_gemm");
        }
    }
#endif
    if (a->split == 1) {
        switch (a->taps) {
            case 11: launch_one<11, 1, 0>(grid, st, k); break;
            case 9: launch_one<9, 1, 0>(grid, st, k); break;
            case 7: launch_one<7, 1, 0>(grid, st, k); break;
            case 5: launch_one<5, 1, 0>(grid, st, k); break;
            case 3: launch_one<3, 1, 0>(grid, st, k); break;
            default: launch_one<1, 1, 0>(grid, st, k);
        }
    } else {
        switch (a->taps) {
            case 11: launch_one<11, 2, 0>(grid, st, k); break;
            case 9: launch_one<9, 2, 0>(grid, st, k); break;
            case 7: launch_one<7, 2, 0>(grid, st, k); break;
            case 5: launch_one<5, 2, 0>(grid, st, k); break;
            case 3: launch_one<3, 2, 0>(grid, st, k); break;
            default: launch_one<1, 2, 0>(grid, st, k);
        }
    }
    return efts_check_launch("efts_gemm");
}

// Opt every product instantiation into 80 KiB of dynamic LDS once, at library load.
template <int T, int S>
static void set_lds_attr() {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<T, S, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
}
__attribute__((visibility("hidden"))) void efts_gemm_init(void) {
    static bool once = false;
    if (!once) {
        conv5_set_lds_attr();
        set_lds_attr<5, 1>(); set_lds_attr<3, 1>(); set_lds_attr<1, 1>(); set_lds_attr<5, 2>(); set_lds_attr<3, 2>(); set_lds_attr<1, 2>();
        set_lds_attr<7, 1>(); set_lds_attr<11, 1>(); set_lds_attr<7, 2>(); set_lds_attr<11, 2>(); set_lds_attr<9, 1>(); set_lds_attr<9, 2>();
        once = true;
    }
}
