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

#include "efts_internal.h"

// cache-policy bits of the epilogue's buffer accesses (bit 0 sc0, bit 1 nt, bit 4 sc1); experiments only
#ifndef EFTS_AUX_LD
#define EFTS_AUX_LD 0
#endif
#ifndef EFTS_AUX_STF
#define EFTS_AUX_STF 0
#endif
#ifndef EFTS_AUX_STP
#define EFTS_AUX_STP 0
#endif

#include "efts_mma.h"

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
    if (pre) {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(of ? of + (long)m0 * p.ldo : nullptr, of ? (long)rows_out * p.ldo * 4 : 0);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(ob ? ob + (long)m0 * p.ldob : nullptr, ob ? (long)rows_out * p.ldob : 0);
        const __amdgpu_buffer_rsrc_t rbl = make_rsrc(obl ? obl + (long)m0 * p.ldob : nullptr, obl ? (long)rows_out * p.ldob : 0);
        const unsigned vo = trow * (unsigned)p.ldo * 4 + col * 4, so = RPP * (unsigned)p.ldo * 4;
        const unsigned vb = trow * (unsigned)p.ldob + (unsigned)plane_off_hi(col, p.out_split), sb = RPP * (unsigned)p.ldob;
        const bool has_mask = rowmask != nullptr;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * RPP + trow;
            float4 v = *(const float4*)(cs + rl * 128 + c4);
            const u32x4 x = rres[ps];
            const float rm = has_mask ? rmv[ps] : 1.f;
            v.x = (v.x + __uint_as_float(x.x)) * rm; v.y = (v.y + __uint_as_float(x.y)) * rm;
            v.z = (v.z + __uint_as_float(x.z)) * rm; v.w = (v.w + __uint_as_float(x.w)) * rm;
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


// =============================================================================================
// narrow_kernel: gemm_kernel for outputs of at most 64 / 32 columns (BNT): the same window, ring and
// epilogue, but the column tile is BNT wide instead of 128, so the 32- and 64-channel stages of the
// vocoder do not spend 4x / 2x of their MFMAs on clamped duplicate columns.  BNT = 64: 2x2 waves of
// 64x32; BNT = 32: 4x1 waves of 32x32.  The weight tile shrinks with it (BNT / 8 DMA pieces per step).
// =============================================================================================
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
}

template <int TAPS, int SPLIT, int BNT>
__global__ __launch_bounds__(256, 2) void narrow_kernel(GemmKernelArgs p) {
    constexpr int DBG = 0;
    constexpr int NI = BNT == 32 ? 1 : 2;         // 32-row accumulator blocks per wave
    constexpr int WP = BNT / 32;                  // weight DMA pieces per wave and step
    constexpr int TPR = BNT / 4;                  // epilogue threads per tile row
    const int BM = p.bm;                      // output rows per tile: WIN - (TAPS - 1) * dilation
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tq = 0;
#define EFTS_STAMP(i) do { if constexpr (false) { const unsigned long long tn = __builtin_readcyclecounter(); pt[i] += tn - tq; tq = tn; } } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // all LDS in one array, buffers addressed arithmetically
#define EFTS_ABUF(i) (smem + ((i) & 1) * TILE_BYTES)
#define EFTS_WBUF(i) (smem + 2 * TILE_BYTES + ((i) % NST) * TILE_BYTES)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BNT == 32 ? wave : wave >> 1, wn = BNT == 32 ? 0 : wave & 1;
    const int dbg = DBG ? p.dbg : 0;

    const int z = blockIdx.y;
    const int z2 = blockIdx.z;
    const int lrow = lane & 31;
    const int lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;
    const int c4 = (tid % TPR) << 2;
    constexpr int RPP = 256 / TPR;            // tile rows per epilogue sweep
    constexpr int NPS = WIN / RPP;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs + (long)z2 * p.o_bs2 : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
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
    const int m0 = mt * BM, n0 = nt * BNT;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);

    // per-lane DMA offsets of the 4 window pieces and 4 weight pieces this wave issues per tile:
    // piece pc covers tile rows 8*pc .. 8*pc+7; lane l -> row 8*pc + l/8, physical slot l%8
    unsigned voa[4], vow[WP];
    {
        const int b_max = p.n - 1 - n0;        // clamp B rows to the last real row
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            voa[q] = (unsigned)(r * (int)p.lda + (sl << 4));
        }
#pragma unroll
        for (int q = 0; q < WP; ++q) {
            const int r = (wave * WP + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            vow[q] = (unsigned)((r < b_max ? r : b_max) * (int)p.ldb + (sl << 4));
        }
    }
    const char* a_base = A + (long)(m0 - p.pad * p.dil) * p.lda;     // window row 0 (may start in the guard rows)
    const char* w_base = Bw + (long)n0 * p.ldb;
    const unsigned lds_piece = lds0 + wave * 4096;            // this wave's first piece inside a tile

    auto issue_w = [&](int cn, int kn, int slot) {     // weights of step (chunk cn, tap kn) -> ring slot
        const char* sb = w_base + (long)kn * p.b_tap_stride + (long)cn * 128;
        const unsigned l = lds0 + 2 * TILE_BYTES + slot * TILE_BYTES + wave * WP * 1024;
#pragma unroll
        for (int q = 0; q < WP; ++q) dma16(l + q * 1024, vow[q], sb);
    };
    auto issue_a = [&](int cn) {     // A window of chunk cn -> window buffer cn & 1
        const char* sb = a_base + (long)cn * 128;
        const unsigned l = lds_piece + (cn & 1) * TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(l + q * 1024, voa[q], sb);
    };

    f32x16 acc[NI][1];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // MFMAs of one (chunk, tap) step: window buffer `ab`, weight ring slot `ws`, tap k
    auto compute = [&](int ab, int ws, int k) {
        const char* at = smem + ab * TILE_BYTES;
        const char* wt = smem + 2 * TILE_BYTES + ws * TILE_BYTES;
        const int arow = wm * (NI * 32) + lrow + k * p.dil;   // tile row of output row r at tap k is r + k * dilation
        const int brow = wn * 32 + lrow;
        if (DBG && (dbg & 4)) return;
        // Operand fragments are double-buffered in registers: the ds_reads of k-slice kk+1 are issued
        // before the MFMAs of slice kk (scheduler fenced), so only the first slice's LDS latency is
        // exposed per step and the waits are counted lgkmcnt(N).
        if constexpr (SPLIT == 1) {
            bf16x8 af[2][NI], bfr[2][1];
            auto ld = [&](int kk, int b) {
                const int slot = kk * 2 + lhalf;
#pragma unroll
                for (int i = 0; i < NI; ++i) af[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
#pragma unroll
                for (int j = 0; j < 1; ++j) bfr[b][j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
            };
            ld(0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) ld(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 1; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            bf16x8 ah[2][NI], al[2][NI], bh[2][1], bl[2][1];
            auto ld = [&](int kk, int b) {
                const int slot = kk * 2 + lhalf;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    ah[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                    al[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                }
#pragma unroll
                for (int j = 0; j < 1; ++j) {
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
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 1; ++j) {
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
        wait_vmcnt_n(nsteps > 1 ? WP : 0);
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
        int n = do_w ? WP : 0;
        if (TAPS != 1) {
            if (do_a) n += 4;
            if (k == 1 && c + 1 < p.nchunk && dma_on) n += 4;
        }
        wait_vmcnt_n(n);
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
    const unsigned trow = tid / TPR;
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
        for (int j = 0; j < 1; ++j) {
            const int cl = wn * 32 + j * 32 + lrow;
            const float bv = (bias && n0 + cl < p.n) ? bias[n0 + cl] : 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * (NI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    float v = acc[i][j][r] * p.alpha + bv;
                    if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                    else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (p.act == EFTS_ACT_TANH) v = tanhf(v);
                    cs[rl * BNT + cl] = v;
                }
            }
        }
    }
    lds_barrier();
    if (pre) {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(of ? of + (long)m0 * p.ldo : nullptr, of ? (long)rows_out * p.ldo * 4 : 0);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(ob ? ob + (long)m0 * p.ldob : nullptr, ob ? (long)rows_out * p.ldob : 0);
        const unsigned vo = trow * (unsigned)p.ldo * 4 + col * 4, so = RPP * (unsigned)p.ldo * 4;
        const unsigned vb = trow * (unsigned)p.ldob + (unsigned)plane_off_hi(col, p.out_split), sb = RPP * (unsigned)p.ldob;
        const bool has_mask = rowmask != nullptr;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * RPP + trow;
            float4 v = *(const float4*)(cs + rl * BNT + c4);
            const u32x4 x = rres[ps];
            const float rm = has_mask ? rmv[ps] : 1.f;
            v.x = (v.x + __uint_as_float(x.x)) * rm; v.y = (v.y + __uint_as_float(x.y)) * rm;
            v.z = (v.z + __uint_as_float(x.z)) * rm; v.w = (v.w + __uint_as_float(x.w)) * rm;
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
                }
            }
        }
    } else if (col < p.n && !(dbg & 1)) {
#pragma unroll 4
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * RPP + trow;
            const int row = m0 + rl;
            if (rl >= BM || row >= p.m) break;
            float4 v = *(const float4*)(cs + rl * BNT + c4);
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
                        if (p.plane_act) t = t > 0.f ? t : t * p.plane_slope;
                        const unsigned short hi = f32_to_bf16(t);
                        char* d = ob + (long)row * p.ldob + plane_off_hi(col + u, p.out_split);
                        *(unsigned short*)d = hi;
                        if (p.out_split == 2) *(unsigned short*)(d + 64) = f32_to_bf16(t - bf16_to_f32(hi));
                    }
                }
            }
        }
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


// =============================================================================================
// conv5_kernel: the k5 residual convolution at large M.  Same machinery as gemm_kernel, tiled like
// the direct wgrad kernel: a 256-row window (252 output rows) x 128 columns per workgroup, 2x2 waves
// of 128x64 (8 accumulator blocks = 128 VGPRs), FULL 128-byte rows.  Every 16 KiB weight tile now
// feeds 32 MFMAs per wave instead of 16, i.e. 0.58x the LDS-DMA line requests per FLOP.
// LDS: ONE 32 KiB window + the 3-stage weight ring = exactly 80 KiB (two workgroups per CU); the
// window of the next chunk can only be requested after the last tap has read the current one, so
// that latency is exposed once per chunk (every 5th step) and covered by the co-resident workgroup.
// The epilogue runs in two passes through the 64 KiB staging tile, each pass taking accumulator
// blocks i = 2*ep, 2*ep+1 of EVERY wave (all waves stage equally, half the accumulators die early).
// =============================================================================================
constexpr int C5_WIN = 256;
constexpr int C5_BM = C5_WIN - 4;
constexpr int C5_A_BYTES = C5_WIN * 128;                       // 32768
constexpr int C5_LDS = C5_A_BYTES + NST * TILE_BYTES;          // 81920
constexpr long C5_DEFAULT_MIN_TILES = 400;                     // bf16 planes only by default (measured +1.5 % there, -2.5 % on bf16x3)

template <int SPLIT>
__global__ __launch_bounds__(256, 2) void conv5_kernel(GemmKernelArgs p) {
    constexpr int TAPS = 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.y;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;
    const int c4 = (tid & 31) << 2;
    constexpr int RPP = 8, NPS = 16, NRING = 8;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const char* A = p.a + (long)z * p.a_bs;
    const char* Bw = p.b + (long)z * p.b_bs;
    const int ntot = p.mtiles * p.ntiles;
  for (int vt = blockIdx.x; vt < ntot; vt += gridDim.x) {
    int bid = vt;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = mt * C5_BM, n0 = nt * BN;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);

    unsigned voa[8], vow[4];      // per-lane DMA offsets: 8 window pieces + 4 weight pieces per wave
    {
        const int b_max = p.n - 1 - n0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (wave * 8 + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            voa[q] = (unsigned)(r * (int)p.lda + (sl << 4));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            vow[q] = (unsigned)((r < b_max ? r : b_max) * (int)p.ldb + (sl << 4));
        }
    }
    const char* a_base = A + (long)(m0 - 2) * p.lda;
    const char* w_base = Bw + (long)n0 * p.ldb;
    auto issue_w = [&](int cn, int kn, int slot) {
        const char* sb = w_base + (long)kn * p.b_tap_stride + (long)cn * 128;
        const unsigned l = lds0 + C5_A_BYTES + slot * TILE_BYTES + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(l + q * 1024, vow[q], sb);
    };
    auto issue_a = [&](int cn) {
        const char* sb = a_base + (long)cn * 128;
        const unsigned l = lds0 + wave * 8192;
#pragma unroll
        for (int q = 0; q < 8; ++q) dma16(l + q * 1024, voa[q], sb);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int ws, int k) {
        const char* at = smem;
        const char* wt = smem + C5_A_BYTES + ws * TILE_BYTES;
        const int arow = wm * 128 + lrow + k;
        const int brow = wn * 64 + lrow;
        if constexpr (SPLIT == 1) {
            bf16x8 af[2][4], bfr[2][2];
            auto ld = [&](int kk, int b) {
                const int slot = kk * 2 + lhalf;
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[b][j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
#pragma unroll
                for (int i = 0; i < 4; ++i) af[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
            };
            ld(0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) ld(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot = kk * 2 + lhalf;
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
                    bl[j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot + 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 ah = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                    const bf16x8 al = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    issue_a(0);
    issue_w(0, 0, 0);
    if (nsteps > 1) issue_w(0, 1, 1);
    wait_vmcnt(nsteps > 1 ? 4 : 0);
    __builtin_amdgcn_s_barrier();

    int c = 0, k = 0, ws = 0, c2 = 0, k2 = 2;
    for (int s = 0; s + 1 < nsteps; ++s) {
        const bool do_w = s + 2 < nsteps;
        if (do_w) issue_w(c2, k2, ws == 0 ? 2 : ws - 1);        // weights two steps ahead, slot (s + 2) % 3
        compute(ws, k);
        if (k == TAPS - 1) {
            lds_barrier();                                      // every wave has finished reading the window of chunk c
            issue_a(c + 1);
            wait_vmcnt(0);
        } else {
            wait_vmcnt(do_w ? 4 : 0);
        }
        lds_barrier();
        if (++k == TAPS) { k = 0; ++c; }
        if (++k2 == TAPS) { k2 = 0; ++c2; }
        ws = (ws == 2) ? 0 : ws + 1;
    }

    // ---- last step + epilogue (two passes; operands of the first 8 sweeps of each pass prefetched, ring of 8)
    u32x4 rres[NRING];
    float rmv[NRING];
    const bool pre = vec && col < p.n;
    const int rows_in = p.m - m0 < C5_WIN ? p.m - m0 : C5_WIN;
    const int rows_out = p.m - m0 < C5_BM ? p.m - m0 : C5_BM;
    const unsigned trow = tid >> 5;
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(resid ? resid + (long)m0 * p.ldr : nullptr, resid ? (long)rows_in * p.ldr * 4 : 0);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(rowmask ? rowmask + m0 : nullptr, rowmask ? (long)rows_in * 4 : 0);
    const unsigned vr = trow * (unsigned)p.ldr * 4 + col * 4;
    // sweep ps of pass ep covers tile rows row_of(ep, ps) + (0..7)
    auto row_of = [&](int ep, int ps) { return ps * RPP + 64 * ep + (ps >= 8 ? 64 : 0); };
    auto request = [&](int ep, int ps) {
        rres[ps % NRING] = __builtin_amdgcn_raw_buffer_load_b128(rr, vr, row_of(ep, ps) * (unsigned)p.ldr * 4, EFTS_AUX_LD);
        rmv[ps % NRING] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rk, trow * 4, row_of(ep, ps) * 4, 0));
    };
    compute(ws, TAPS - 1);
    lds_barrier();

    float* cs = (float*)smem;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(of ? of + (long)m0 * p.ldo : nullptr, of ? (long)rows_out * p.ldo * 4 : 0);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(ob ? ob + (long)m0 * p.ldob : nullptr, ob ? (long)rows_out * p.ldob : 0);
    const unsigned vo = trow * (unsigned)p.ldo * 4 + col * 4;
    const unsigned vb = trow * (unsigned)p.ldob + (unsigned)plane_off_hi(col, p.out_split);
    const bool has_mask = rowmask != nullptr;
#pragma unroll
    for (int ep = 0; ep < 2; ++ep) {
        if (pre) {
#pragma unroll
            for (int ps = 0; ps < NRING; ++ps) request(ep, ps);     // in flight while the accumulators are staged
        }
        {
            const float* bias = p.bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + lrow;
                const float bv = (bias && n0 + cl < p.n) ? bias[n0 + cl] : 0.f;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * 64 + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[2 * ep + ii][j][r] * p.alpha + bv;
                        if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                        cs[rl * 128 + cl] = v;
                    }
                }
            }
        }
        lds_barrier();
        if (pre) {
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int rl = ps * RPP + trow;
                float4 v = *(const float4*)(cs + rl * 128 + c4);
                const u32x4 x = rres[ps % NRING];
                const float rm = has_mask ? rmv[ps % NRING] : 1.f;
                if (ps + NRING < NPS) request(ep, ps + NRING);
                v.x = (v.x + __uint_as_float(x.x)) * rm; v.y = (v.y + __uint_as_float(x.y)) * rm;
                v.z = (v.z + __uint_as_float(x.z)) * rm; v.w = (v.w + __uint_as_float(x.w)) * rm;
                if (of) {
                    const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    { __builtin_amdgcn_raw_buffer_store_b128(o, ro, vo, row_of(ep, ps) * (unsigned)p.ldo * 4, EFTS_AUX_STF); asm volatile("s_nop 4" ::"v"(o)); }
                }
                if (ob) {
                    float r0, r1, r2, r3;
                    const u32x2 hi = {pack_bf16x2(v.x, v.y, &r0, &r1), pack_bf16x2(v.z, v.w, &r2, &r3)};
                    __builtin_amdgcn_raw_buffer_store_b64(hi, rb, vb, row_of(ep, ps) * (unsigned)p.ldob, EFTS_AUX_STP);
                    if (p.out_split == 2) {
                        float d0, d1;
                        const u32x2 lo = {pack_bf16x2(r0, r1, &d0, &d1), pack_bf16x2(r2, r3, &d0, &d1)};
                        __builtin_amdgcn_raw_buffer_store_b64(lo, rb, vb + 64, row_of(ep, ps) * (unsigned)p.ldob, EFTS_AUX_STP);
                    }
                }
            }
        } else if (col < p.n) {
            for (int ps = 0; ps < NPS; ++ps) {
                const int rl = ps * RPP + trow;
                const int trl = row_of(ep, ps) + trow;
                const int row = m0 + trl;
                if (trl >= C5_BM || row >= p.m) continue;
                const float4 v = *(const float4*)(cs + rl * 128 + c4);
                const float rm = rowmask ? rowmask[row] : 1.f;
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
        lds_barrier();
    }
  }   // tile loop
}


// =============================================================================================
// resident32_kernel: convolutions with ONE K chunk (cin <= 64 bf16 / 32 bf16x3) and at most 32 output columns -- the
// 32-channel stage of the vocoder, 1.6 M rows at a batch of 8.  There the ring kernels are all overhead: 124 rows per
// workgroup, a window wait, one barrier per tap, an epilogue, for 44 MFMAs per wave.  Here a workgroup keeps a 256-row
// window AND the weights of every tap in LDS (32 KiB + taps x 4 KiB <= 76 KiB: two workgroups per CU): everything is
// requested up front, one wait, one barrier, then all taps back to back (wave = 64 rows x 32 columns) and the usual
// staged epilogue.  Same K order per output element as gemm_kernel / narrow_kernel (bit-compatible).
// =============================================================================================
constexpr int R32_WIN = 256;

template <int TAPS, int SPLIT>
__global__ __launch_bounds__(256, 2) void resident32_kernel(GemmKernelArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.y;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int BM = p.bm;                                       // R32_WIN - (TAPS - 1) * dilation
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const char* A = p.a + (long)z * p.a_bs;
    const char* Bw = p.b + (long)z * p.b_bs;
    const int mt = blockIdx.x / p.ntiles, nt = blockIdx.x - mt * p.ntiles;    // column tiles of 32 (n fastest: they share the window in L2)
    const int m0 = mt * BM, n0 = nt * 32;
    const int w0 = m0 - p.pad * p.dil;                         // first row of the window (may lie in the guard rows)
    const int row_max = p.m + 143;                             // last row the ABI lets us read (144 zero guard rows)

    // ---- request everything: 8 window pieces per wave, then this wave's piece (8 weight rows) of every tap
    {
        const char* a_base = A + (long)w0 * p.lda;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (wave * 8 + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            const int rc = w0 + r > row_max ? row_max - w0 : r;   // rows past the guard read the last (zero) guard row
            dma16(lds0 + (wave * 8 + q) * 1024, (unsigned)(rc * (int)p.lda + (sl << 4)), a_base);
        }
        const int r = wave * 8 + (lane >> 3);
        const int sl = (lane & 7) ^ ((r >> 1) & 7);
        const int b_max = p.n - 1 - n0;
        const unsigned vw = (unsigned)((r < b_max ? r : b_max) * (int)p.ldb + (sl << 4));
        const char* w_base = Bw + (long)n0 * p.ldb;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) dma16(lds0 + R32_WIN * 128 + (k * 4 + wave) * 1024, vw, w_base + (long)k * p.b_tap_stride);
    }
    // epilogue operands of this thread (8 sweeps of 32 rows, 8 threads per 128-byte row): in flight under the DMA wait
    constexpr int NPS = 8;
    const int c4 = (tid & 7) << 2;
    const int col = n0 + c4;
    const bool vec = p.vec_ok && (col + 3 < p.n);
    const bool pre = vec;
    const unsigned trow = tid >> 3;
    const int rows_in = p.m - m0 < R32_WIN ? p.m - m0 : R32_WIN;
    const int rows_out = p.m - m0 < BM ? p.m - m0 : BM;
    u32x4 rres[NPS];
    float rmv[NPS];
    if (pre) {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(resid ? resid + (long)m0 * p.ldr : nullptr, resid ? (long)rows_in * p.ldr * 4 : 0);
        const __amdgpu_buffer_rsrc_t rk = make_rsrc(rowmask ? rowmask + m0 : nullptr, rowmask ? (long)rows_in * 4 : 0);
        const unsigned vr = trow * (unsigned)p.ldr * 4 + col * 4, sr = 32 * (unsigned)p.ldr * 4;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            rres[ps] = __builtin_amdgcn_raw_buffer_load_b128(rr, vr, ps * sr, EFTS_AUX_LD);
            rmv[ps] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rk, trow * 4, ps * 32 * 4, 0));
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    {
        const char* at = smem;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const char* wt = smem + R32_WIN * 128 + k * 4096;
            const int arow = wave * 64 + lrow + k * p.dil;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    const bf16x8 b = *(const bf16x8*)(wt + lds_off(lrow, slot));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const bf16x8 a = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int slot = kk * 2 + lhalf;
                    const bf16x8 bh = *(const bf16x8*)(wt + lds_off(lrow, slot));
                    const bf16x8 bl = *(const bf16x8*)(wt + lds_off(lrow, slot + 4));
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const bf16x8 ah = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
                        const bf16x8 al = *(const bf16x8*)(at + lds_off(arow + i * 32, slot + 4));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    lds_barrier();                                             // every wave is done with the window: it becomes the staging tile

    float* cs = (float*)smem;                                  // [256][32] fp32 = 32 KiB
    {
        const float bv = (p.bias && n0 + lrow < p.n) ? p.bias[n0 + lrow] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                float v = acc[i][r] * p.alpha + bv;
                if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (p.act == EFTS_ACT_TANH) v = tanhf(v);
                cs[rl * 32 + lrow] = v;
            }
        }
    }
    lds_barrier();
    if (pre) {
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(of ? of + (long)m0 * p.ldo : nullptr, of ? (long)rows_out * p.ldo * 4 : 0);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(ob ? ob + (long)m0 * p.ldob : nullptr, ob ? (long)rows_out * p.ldob : 0);
        const unsigned vo = trow * (unsigned)p.ldo * 4 + col * 4, so = 32 * (unsigned)p.ldo * 4;
        const unsigned vb = trow * (unsigned)p.ldob + (unsigned)plane_off_hi(col, p.out_split), sb = 32 * (unsigned)p.ldob;
        const bool has_mask = rowmask != nullptr;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * 32 + trow;
            float4 v = *(const float4*)(cs + rl * 32 + c4);
            const u32x4 x = rres[ps];
            const float rm = has_mask ? rmv[ps] : 1.f;
            v.x = (v.x + __uint_as_float(x.x)) * rm; v.y = (v.y + __uint_as_float(x.y)) * rm;
            v.z = (v.z + __uint_as_float(x.z)) * rm; v.w = (v.w + __uint_as_float(x.w)) * rm;
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
                }
            }
        }
    } else if (col < p.n) {
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * 32 + trow;
            const int row = m0 + rl;
            if (rl >= BM || row >= p.m) break;
            const float4 v = *(const float4*)(cs + rl * 32 + c4);
            const float rm = rowmask ? rowmask[row] : 1.f;
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
                }
            }
        }
    }
}


// =============================================================================================
// conv8_kernel (experiment, EFTS_CONV8=1): the k5 convolution with ONE 8-wave workgroup per CU on a 256-row x 256-column
// tile: 2 x 4 waves of 128 x 64 (conv5_kernel's wave tile), one 32 KiB window + a 3-stage ring of 32 KiB weight tiles =
// 128 KiB.  Half the weight DMA per MFMA of conv5_kernel, one barrier domain of 8 waves, two waves of the same
// workgroup per SIMD.  The epilogue goes through the 128 KiB in two column halves.  bf16 planes, n % 256 == 0.
// =============================================================================================
constexpr int C8_BN = 256;
constexpr int C8_W_BYTES = C8_BN * 128;                          // 32768
constexpr int C8_LDS = C5_A_BYTES + NST * C8_W_BYTES;            // 131072

__global__ __launch_bounds__(512, 2) void conv8_kernel(GemmKernelArgs p) {
    constexpr int TAPS = 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int z = blockIdx.y;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nsteps = p.nchunk * TAPS;
    const float* resid = p.resid ? p.resid + (long)z * p.r_bs : nullptr;
    const float* rowmask = p.rowmask ? p.rowmask + (long)z * p.m_bs : nullptr;
    float* of = p.out_f32 ? p.out_f32 + (long)z * p.o_bs : nullptr;
    char* ob = p.out_bf16 ? p.out_bf16 + (long)z * p.ob_bs : nullptr;
    const char* A = p.a + (long)z * p.a_bs;
    const char* Bw = p.b + (long)z * p.b_bs;
    const int ntot = p.mtiles * p.ntiles;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
    const int m0 = mt * C5_BM, n0 = nt * C8_BN;

    unsigned voa[4], vow[4];      // per-lane DMA offsets: 4 window pieces + 4 weight pieces per wave
    {
        const int b_max = p.n - 1 - n0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 8 + (lane >> 3);
            const int sl = (lane & 7) ^ ((r >> 1) & 7);
            voa[q] = (unsigned)(r * (int)p.lda + (sl << 4));
            vow[q] = (unsigned)((r < b_max ? r : b_max) * (int)p.ldb + (sl << 4));
        }
    }
    const char* a_base = A + (long)(m0 - 2) * p.lda;
    const char* w_base = Bw + (long)n0 * p.ldb;
    auto issue_w = [&](int cn, int kn, int slot) {
        const char* sb = w_base + (long)kn * p.b_tap_stride + (long)cn * 128;
        const unsigned l = lds0 + C5_A_BYTES + slot * C8_W_BYTES + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(l + q * 1024, vow[q], sb);
    };
    auto issue_a = [&](int cn) {
        const char* sb = a_base + (long)cn * 128;
        const unsigned l = lds0 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(l + q * 1024, voa[q], sb);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int ws, int k) {
        const char* at = smem;
        const char* wt = smem + C5_A_BYTES + ws * C8_W_BYTES;
        const int arow = wm * 128 + lrow + k;
        const int brow = wn * 64 + lrow;
        bf16x8 af[2][4], bfr[2][2];
        auto ld = [&](int kk, int b) {
            const int slot = kk * 2 + lhalf;
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[b][j] = *(const bf16x8*)(wt + lds_off(brow + j * 32, slot));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[b][i] = *(const bf16x8*)(at + lds_off(arow + i * 32, slot));
        };
        ld(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) ld(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_a(0);
    issue_w(0, 0, 0);
    if (nsteps > 1) issue_w(0, 1, 1);
    wait_vmcnt(nsteps > 1 ? 4 : 0);
    __builtin_amdgcn_s_barrier();

    int c = 0, k = 0, ws = 0, c2 = 0, k2 = 2;
    for (int s = 0; s < nsteps; ++s) {
        const bool do_w = s + 2 < nsteps;
#ifndef C8_EXP
#define C8_EXP 0
#endif
        if (do_w && !(C8_EXP & 1)) issue_w(c2, k2, ws == 0 ? 2 : ws - 1);
        if (!(C8_EXP & 4)) compute(ws, k);
        if (k == TAPS - 1 && c + 1 < p.nchunk) {
            lds_barrier();                                      // every wave has finished reading the window of chunk c
            if (!(C8_EXP & 1)) issue_a(c + 1);
            if (!(C8_EXP & 2)) wait_vmcnt(0);
        } else {
            if (!(C8_EXP & 2)) wait_vmcnt(do_w ? 4 : 0);
        }
        if (!(C8_EXP & 8)) lds_barrier();
        if (++k == TAPS) { k = 0; ++c; }
        if (++k2 == TAPS) { k2 = 0; ++c2; }
        ws = (ws == 2) ? 0 : ws + 1;
    }

    // ---- epilogue: two column halves of 128 through [256][128] fp32 = 128 KiB; 32 threads per 512-byte row, 16 rows a sweep
    constexpr int NPS = 16, NRING = 4;
    float* cs = (float*)smem;
    const int c4 = (tid & 31) << 2;
    const unsigned trow = tid >> 5;
    const int rows_in = p.m - m0 < C5_WIN ? p.m - m0 : C5_WIN;
    const int rows_out = p.m - m0 < C5_BM ? p.m - m0 : C5_BM;
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(resid ? resid + (long)m0 * p.ldr : nullptr, resid ? (long)rows_in * p.ldr * 4 : 0);
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(rowmask ? rowmask + m0 : nullptr, rowmask ? (long)rows_in * 4 : 0);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(of ? of + (long)m0 * p.ldo : nullptr, of ? (long)rows_out * p.ldo * 4 : 0);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(ob ? ob + (long)m0 * p.ldob : nullptr, ob ? (long)rows_out * p.ldob : 0);
    const bool has_mask = rowmask != nullptr;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int col = n0 + h * 128 + c4;
        const unsigned vr = trow * (unsigned)p.ldr * 4 + col * 4;
        const unsigned vo = trow * (unsigned)p.ldo * 4 + col * 4;
        const unsigned vb = trow * (unsigned)p.ldob + (unsigned)plane_off_hi(col, p.out_split);
        u32x4 rres[NRING];
        float rmv[NRING];
        auto request = [&](int ps) {
            rres[ps % NRING] = __builtin_amdgcn_raw_buffer_load_b128(rr, vr, ps * 16 * (unsigned)p.ldr * 4, EFTS_AUX_LD);
            rmv[ps % NRING] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rk, trow * 4, ps * 16 * 4, 0));
        };
#pragma unroll
        for (int ps = 0; ps < NRING; ++ps) request(ps);
        if ((wn >> 1) == h) {
            const float* bias = p.bias;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = (wn & 1) * 64 + j * 32 + lrow;
                const float bv = bias ? bias[n0 + h * 128 + cl] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        float v = acc[i][j][r] * p.alpha + bv;
                        if (p.act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
                        cs[rl * 128 + cl] = v;
                    }
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int rl = ps * 16 + trow;
            float4 v = *(const float4*)(cs + rl * 128 + c4);
            const u32x4 x = rres[ps % NRING];
            const float rm = has_mask ? rmv[ps % NRING] : 1.f;
            if (ps + NRING < NPS) request(ps + NRING);
            v.x = (v.x + __uint_as_float(x.x)) * rm; v.y = (v.y + __uint_as_float(x.y)) * rm;
            v.z = (v.z + __uint_as_float(x.z)) * rm; v.w = (v.w + __uint_as_float(x.w)) * rm;
            if (of) {
                const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                { __builtin_amdgcn_raw_buffer_store_b128(o, ro, vo, ps * 16 * (unsigned)p.ldo * 4, EFTS_AUX_STF); asm volatile("s_nop 4" ::"v"(o)); }
            }
            if (ob) {
                float r0, r1, r2, r3;
                const u32x2 hi = {pack_bf16x2(v.x, v.y, &r0, &r1), pack_bf16x2(v.z, v.w, &r2, &r3)};
                __builtin_amdgcn_raw_buffer_store_b64(hi, rb, vb, ps * 16 * (unsigned)p.ldob, EFTS_AUX_STP);
                if (p.out_split == 2) {
                    float d0, d1;
                    const u32x2 lo = {pack_bf16x2(r0, r1, &d0, &d1), pack_bf16x2(r2, r3, &d0, &d1)};
                    __builtin_amdgcn_raw_buffer_store_b64(lo, rb, vb + 64, ps * 16 * (unsigned)p.ldob, EFTS_AUX_STP);
                }
            }
        }
        lds_barrier();
    }
}

}  // namespace efts

using namespace efts;

template <int T, int S, int D>
static void launch_one(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    if (D) (void)hipFuncSetAttribute((const void*)gemm_kernel<T, S, D>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    hipLaunchKernelGGL((gemm_kernel<T, S, D>), grid, dim3(256), GEMM_LDS, st, k);
}

static void launch_conv8(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS); attr = true; }
    hipLaunchKernelGGL(conv8_kernel, grid, dim3(512), C8_LDS, st, k);
}

template <int S>
static void launch_conv5(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    hipLaunchKernelGGL((conv5_kernel<S>), grid, dim3(256), C5_LDS, st, k);
}

template <int T, int S, int B>
static void launch_narrow(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)narrow_kernel<T, S, B>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr = true;
    }
    hipLaunchKernelGGL((narrow_kernel<T, S, B>), grid, dim3(256), GEMM_LDS, st, k);
}
template <int T, int S>
static void launch_resident32(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    static bool attr = false;
    constexpr int lds = R32_WIN * 128 + T * 4096;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)resident32_kernel<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((resident32_kernel<T, S>), grid, dim3(256), lds, st, k);
}
template <int S>
static bool launch_resident32_taps(int taps, dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    switch (taps) {
        case 3: launch_resident32<3, S>(grid, st, k); return true;
        case 7: launch_resident32<7, S>(grid, st, k); return true;
        case 11: launch_resident32<11, S>(grid, st, k); return true;
        default: return false;
    }
}

template <int S, int B>
static bool launch_narrow_taps(int taps, dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    switch (taps) {
        case 1: launch_narrow<1, S, B>(grid, st, k); return true;
        case 3: launch_narrow<3, S, B>(grid, st, k); return true;
        case 5: launch_narrow<5, S, B>(grid, st, k); return true;
        case 7: launch_narrow<7, S, B>(grid, st, k); return true;
        case 11: launch_narrow<11, S, B>(grid, st, k); return true;
        default: return false;
    }
}

// Debug instantiation (k5 only): honours EFTS_GEMM_DBG ablation bits and, with EFTS_GEMM_PROF=1,
// prints per-phase s_memtime sums of every wave.  Synchronises the stream; never used by default.
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

extern "C" int efts_gemm(const efts_gemm_args* a, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_gemm: null args");
    if (!(a->split == 1 || a->split == 2)) return efts_fail(EFTS_EINVAL, "efts_gemm: split must be 1 or 2");
    if (!(a->taps == 1 || a->taps == 3 || a->taps == 5 || a->taps == 7 || a->taps == 11)) return efts_fail(EFTS_EINVAL, "efts_gemm: taps must be 1, 3, 5, 7 or 11");
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
    if (!a->out_f32 && !a->out_bf16) return efts_fail(EFTS_EINVAL, "efts_gemm: no output");

    GemmKernelArgs k;
    k.a = (const char*)a->a; k.b = (const char*)a->b;
    k.bias = a->bias; k.resid = a->resid; k.rowmask = a->rowmask;
    k.out_f32 = a->out_f32; k.out_bf16 = (char*)a->out_bf16; k.out_lo = (char*)a->out_bf16_lo;
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
    hipStream_t st = (hipStream_t)stream;

    // one workgroup per tile, 2 resident per CU; EFTS_GEMM_PERSIST=1 caps the grid at 2 workgroups per
    // CU that walk the tile list instead
    const int nt_all = k.mtiles * k.ntiles;
    int cap = nt_all;
    { const char* e = getenv("EFTS_GEMM_PERSIST"); if (e && atoi(e) == 1 && a->batch == 1 && nb2 == 1) cap = 2 * efts_num_cus(); }
    dim3 grid(nt_all < cap ? nt_all : cap, a->batch, nb2);

    // outputs of at most 64 columns (the 64- and 32-channel stages of the vocoder): column tile 64 / 32
    // ... and launches whose 128-column tiling cannot give every CU its two resident workgroups (the text side of
    // the acoustic model: 64 x 130 rows = 272 workgroups; the 256-channel stage of the vocoder at one utterance:
    // 110): such a launch is bound by the per-workgroup step latency, and 64-column tiles double the number of
    // workgroups that overlap.  EFTS_NARROW_FEW = the threshold in workgroups per CU (default 1, 0 disables; 2 measured slower for the 272-workgroup text side).
    // one K chunk and at most 64 columns over many rows (the 32- and 64-channel stages of the vocoder): window + all taps
    // resident, 32-column tiles (64 columns = two workgroups per window: 2.11 -> 2.03 ms per utterance, 11.1 -> 10.5 ms per 8)
    static const int resident_nmax = [] { const char* e = getenv("EFTS_RESIDENT_NMAX"); return e ? atoi(e) : 64; }();
    const bool generic_only = a->out_bf16_lo != nullptr;      // the remainder plane is written by gemm_kernel only
    if (!generic_only && a->n <= resident_nmax && a->nchunk == 1 && nb2 == 1 && (a->taps - 1) * dil <= 64 && a->m >= 8 * R32_WIN && !getenv("EFTS_NO_RESIDENT")) {
        GemmKernelArgs kr = k;
        kr.bm = R32_WIN - (a->taps - 1) * dil;
        kr.mtiles = (a->m + kr.bm - 1) / kr.bm;
        kr.ntiles = (a->n + 31) / 32;
        dim3 gr(kr.mtiles * kr.ntiles, a->batch, 1);
        const bool done = a->split == 1 ? launch_resident32_taps<1>(a->taps, gr, st, kr) : launch_resident32_taps<2>(a->taps, gr, st, kr);
        if (done) return efts_check_launch("efts_gemm");
    }
    int few_per_cu = 1;
    { const char* e = getenv("EFTS_NARROW_FEW"); if (e) few_per_cu = atoi(e); }
    const bool few = (long)k.mtiles * k.ntiles * a->batch < (long)few_per_cu * efts_num_cus() && a->n > 64;
    if (!generic_only && (a->n <= 64 || few) && nb2 == 1 && !getenv("EFTS_NO_NARROW")) {
        GemmKernelArgs kn = k;
        kn.ntiles = a->n <= 32 ? 1 : (a->n + 63) / 64;
        dim3 gn(k.mtiles * kn.ntiles, a->batch, 1);
        bool done;
        if (a->n <= 32) done = a->split == 1 ? launch_narrow_taps<1, 32>(a->taps, gn, st, kn) : launch_narrow_taps<2, 32>(a->taps, gn, st, kn);
        else done = a->split == 1 ? launch_narrow_taps<1, 64>(a->taps, gn, st, kn) : launch_narrow_taps<2, 64>(a->taps, gn, st, kn);
        if (done) return efts_check_launch("efts_gemm");
    }
    int dbg = 0, prof = 0;
    { const char* e = getenv("EFTS_GEMM_DBG"); dbg = e ? atoi(e) : 0; }
    { const char* e = getenv("EFTS_GEMM_PROF"); prof = e ? atoi(e) : 0; }

    // k5 convolutions with enough rows take the 256-row kernel: EFTS_CONV5 = minimum number of
    // (252-row tile x column tile x batch) workgroups, 0 disables it
    if (!generic_only && a->taps == 5 && dil == 1 && !a->plane_act && a->act != EFTS_ACT_TANH && nb2 == 1 && !dbg && !prof) {
        long min_tiles = a->split == 1 ? C5_DEFAULT_MIN_TILES : 0x7fffffffL;
        { const char* e = getenv("EFTS_CONV5"); if (e) min_tiles = atol(e) > 0 ? atol(e) : 0x7fffffffL; }
        const int mt5 = (a->m + C5_BM - 1) / C5_BM;
        if (a->split == 1 && a->n % C8_BN == 0 && k.vec_ok && (long)mt5 * C5_BM + 2 - a->m <= 144 && getenv("EFTS_CONV8") && atoi(getenv("EFTS_CONV8")) == 1) {
            GemmKernelArgs k8 = k;
            k8.mtiles = mt5;
            k8.ntiles = a->n / C8_BN;
            dim3 g8(mt5 * k8.ntiles, a->batch, 1);
            launch_conv8(g8, st, k8);
            return efts_check_launch("efts_gemm");
        }
        // the last 256-row window may reach mt5 * 252 + 2 - m rows past the matrix: only inside the 144 guard rows of the ABI
        if ((long)mt5 * k.ntiles * a->batch >= min_tiles && (long)mt5 * C5_BM + 2 - a->m <= 144) {
            GemmKernelArgs k5 = k;
            k5.mtiles = mt5;
            dim3 g5(mt5 * k.ntiles, a->batch, 1);
            if (a->split == 1) launch_conv5<1>(g5, st, k5); else launch_conv5<2>(g5, st, k5);
            return efts_check_launch("efts_gemm");
        }
    }
    if ((dbg || prof) && a->taps == 5) {
        k.dbg = dbg;
        if (a->split == 1) launch_debug<1>(grid, st, k, prof); else launch_debug<2>(grid, st, k, prof);
    } else if (a->split == 1) {
        switch (a->taps) {
            case 11: launch_one<11, 1, 0>(grid, st, k); break;
            case 7: launch_one<7, 1, 0>(grid, st, k); break;
            case 5: launch_one<5, 1, 0>(grid, st, k); break;
            case 3: launch_one<3, 1, 0>(grid, st, k); break;
            default: launch_one<1, 1, 0>(grid, st, k);
        }
    } else {
        switch (a->taps) {
            case 11: launch_one<11, 2, 0>(grid, st, k); break;
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
extern "C" void efts_gemm_init(void) {
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)conv5_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C5_LDS);
        (void)hipFuncSetAttribute((const void*)conv5_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, C5_LDS);
        set_lds_attr<5, 1>(); set_lds_attr<3, 1>(); set_lds_attr<1, 1>(); set_lds_attr<5, 2>(); set_lds_attr<3, 2>(); set_lds_attr<1, 2>();
        set_lds_attr<7, 1>(); set_lds_attr<11, 1>(); set_lds_attr<7, 2>(); set_lds_attr<11, 2>();
        once = true;
    }
    if (getenv("EFTS_DEBUG")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm_kernel<5, 1, 0>, 256, GEMM_LDS);
        fprintf(stderr, "[efts] gemm_kernel<5,1>: %d workgroups/CU at %d B LDS\n", nb, GEMM_LDS);
    }
}
