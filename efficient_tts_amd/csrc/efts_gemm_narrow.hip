// efts_gemm_narrow.hip -- the MFMA contraction of efts_gemm.hip on narrow column tiles (see that file for the design):
// narrow_kernel (64- / 32-column tiles: outputs of at most 64 columns, and launches that would leave most CUs idle) and
// resident32_kernel (one K chunk, <= 64 columns, many rows: window and all taps resident in LDS).  Both are bit-identical
// to gemm_kernel on the same operands.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_gemm_kernels.h"

namespace efts {

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
// resident32_kernel: convolutions with ONE K chunk (cin <= 64 bf16 / 32 bf16x3) and at most 32 output columns -- the
// 32-channel stage of the vocoder, 1.6 M rows at a batch of 8.  There the ring kernels are all overhead: 124 rows per
// workgroup, a window wait, one barrier per tap, an epilogue, for 44 MFMAs per wave.  Here a workgroup keeps a 256-row
// window AND the weights of every tap in LDS (32 KiB + taps x 4 KiB <= 76 KiB: two workgroups per CU): everything is
// requested up front, one wait, one barrier, then all taps back to back (wave = 64 rows x 32 columns) and the usual
// staged epilogue.  Same K order per output element as gemm_kernel / narrow_kernel (bit-compatible).
// =============================================================================================

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

bool launch_narrow_any(int split, int bnt, int taps, dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    if (bnt == 32) return split == 1 ? launch_narrow_taps<1, 32>(taps, grid, st, k) : launch_narrow_taps<2, 32>(taps, grid, st, k);
    return split == 1 ? launch_narrow_taps<1, 64>(taps, grid, st, k) : launch_narrow_taps<2, 64>(taps, grid, st, k);
}
bool launch_resident32_any(int split, int taps, dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    return split == 1 ? launch_resident32_taps<1>(taps, grid, st, k) : launch_resident32_taps<2>(taps, grid, st, k);
}

}  // namespace efts
