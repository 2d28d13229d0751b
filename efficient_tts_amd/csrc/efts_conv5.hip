// efts_conv5.hip -- the k5 convolution of efts_gemm.hip on 256-row windows (two 4-wave workgroups per CU, 252 x 128 tiles):
// used for large bf16 k5 launches that keep an fp32 residual stream (the training step's forward and dgrad convolutions);
// the inference stacks run on efts_resconv5 (efts_resconv.hip).  Bit-identical to gemm_kernel on the same operands.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_gemm_kernels.h"

namespace efts {

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
    const unsigned ld_sg = (unsigned)(p.n >> 3);
    const __amdgpu_buffer_rsrc_t rsg = make_rsrc(p.sign ? p.sign + (long)m0 * ld_sg : nullptr, p.sign ? (long)rows_out * ld_sg : 0);
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
                if (p.sign) {                                // sign words of efts_abi.h `sign_mask` (see gemm_kernel)
                    const unsigned long long b0 = __ballot(v.x > 0.f), b1 = __ballot(v.y > 0.f), b2 = __ballot(v.z > 0.f), b3 = __ballot(v.w > 0.f);
                    if ((tid & 31) == 0) {
                        const int sh = tid & 32;
                        const u32x4 w = {(unsigned)(b0 >> sh), (unsigned)(b1 >> sh), (unsigned)(b2 >> sh), (unsigned)(b3 >> sh)};
                        store_b128(w, rsg, trow * ld_sg + (unsigned)(n0 >> 7) * 16, row_of(ep, ps) * ld_sg);
                    }
                }
                if (p.drop_thresh) {                         // Dropout on the activated value (see gemm_kernel)
                    const unsigned e0 = (unsigned)(m0 + row_of(ep, ps) + (int)trow) * (unsigned)p.n + col;
                    v.x *= drop_scale(p.drop_seed_h, e0, p.drop_thresh, p.drop_inv_keep); v.y *= drop_scale(p.drop_seed_h, e0 + 1, p.drop_thresh, p.drop_inv_keep);
                    v.z *= drop_scale(p.drop_seed_h, e0 + 2, p.drop_thresh, p.drop_inv_keep); v.w *= drop_scale(p.drop_seed_h, e0 + 3, p.drop_thresh, p.drop_inv_keep);
                }
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


#ifdef EFTS_LAB
// =============================================================================================
// conv8_kernel (experiment, EFTS_CONV8=1): the k5 convolution with ONE 8-wave workgroup per CU on a 256-row x 256-column
// tile: 2 x 4 waves of 128 x 64 (conv5_kernel's wave tile), one 32 KiB window + a 3-stage ring of 32 KiB weight tiles =
// 128 KiB.  Half the weight DMA per MFMA of conv5_kernel, one barrier domain of 8 waves, two waves of the same
// workgroup per SIMD.  The epilogue goes through the 128 KiB in two column halves.  bf16 planes, n % 256 == 0.
// =============================================================================================

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

void launch_conv8(dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS); attr = true; }
    hipLaunchKernelGGL(conv8_kernel, grid, dim3(512), C8_LDS, st, k);
}
#endif

void launch_conv5_any(int split, dim3 grid, hipStream_t st, const GemmKernelArgs& k) {
    if (split == 1) hipLaunchKernelGGL((conv5_kernel<1>), grid, dim3(256), C5_LDS, st, k);
    else hipLaunchKernelGGL((conv5_kernel<2>), grid, dim3(256), C5_LDS, st, k);
}
void conv5_set_lds_attr() {
    (void)hipFuncSetAttribute((const void*)conv5_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C5_LDS);
    (void)hipFuncSetAttribute((const void*)conv5_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, C5_LDS);
}

}  // namespace efts
