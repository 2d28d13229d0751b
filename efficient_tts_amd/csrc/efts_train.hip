// efts_train.hip -- backward-pass and optimizer kernels of the EFTS-CNN training step on gfx950
// (reference: loss.backward() / clip_grad_norm_ / Adam(amsgrad) in
// nntts/trainers/efficient_tts_trainer.py:152-160; the backward itself is torch autograd of the ops
// in nntts/models/efficient_tts.py:120-228).  The MFMA-shaped gradients (dgrad, wgrad, the four
// alignment-block products) reuse efts_gemm on (transposed) operand planes built here; everything
// else is fp32 VALU, HBM-bound, one pass per tensor.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

__device__ __forceinline__ float block_sum256t(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------
// d(loss)/d(mel_pred), d(loss)/d(dur_pred) of FastSpeechLoss(use_masking) (fastspeech_loss.py:54-67)
// ---------------------------------------------------------------------------------------------
__global__ void loss_bwd_kernel(const float* __restrict__ mp, long ldm, const float* __restrict__ sp,
                                const int* __restrict__ mlen, const float* __restrict__ dp, const float* __restrict__ lde,
                                const int* __restrict__ tlen, const float* __restrict__ gscale, float* __restrict__ dmel,
                                char* __restrict__ dmel_plane, long ldp, int split, float* __restrict__ ddur, int B, int T1,
                                int T1p, int T2, int T2p, int odim, int kp) {
    __shared__ float nm_s, nt_s;
    if (threadIdx.x == 0) {
        float nm = 0.f, nt = 0.f;
        for (int b = 0; b < B; ++b) { nm += (float)min(mlen[b], T2); nt += (float)min(tlen[b], T1); }
        nm_s = nm * (float)odim; nt_s = nt;
    }
    __syncthreads();
    const float gs = gscale ? gscale[0] : 1.f;
    const int row = blockIdx.x;                       // row space 2, then row space 1
    if (row < B * T2p) {
        const int b = row / T2p, j = row - b * T2p;
        const bool live = j < T2 && j < mlen[b];
        const int c4 = threadIdx.x << 2;
        if (c4 < kp) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = c4 + u;
                v[u] = (live && o < odim) ? 2.f * (mp[(long)row * ldm + o] - sp[((long)b * T2 + j) * odim + o]) / nm_s * gs : 0.f;
                if (dmel && o < odim) dmel[(long)row * odim + o] = v[u];
            }
            if (dmel_plane) plane_store4(dmel_plane + (long)row * ldp, c4, v[0], v[1], v[2], v[3], split);
        }
    } else {
        const int r1 = (row - B * T2p) * blockDim.x + threadIdx.x;
        if (r1 < B * T1p) {
            const int b = r1 / T1p, i = r1 - b * T1p;
            float g = 0.f;
            if (i < T1 && i < tlen[b]) {
                const float d = dp[r1] - lde[(long)b * T1 + i];
                g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / nt_s * gs;
            }
            ddur[r1] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dZ = G * act'(.) * rowmask ; bias grad += column sums of dZ
//   mode 1 (LeakyReLU residual layer y = x + leaky(z)): z > 0  <=>  y - x > 0
//   mode 2 (ReLU, h = relu(z))                        : h > 0
//   mode 3 (LeakyReLU, no residual, out = leaky(z))   : out > 0
//   mode 4 (LeakyReLU, sign words from the forward)    : y points at efts_gemm's `sign_mask` words (row stride c / 8 bytes)
//   mode 5 (LeakyReLU, sign bits from the forward)     : y points at efts_resconv5's `sign_bits` rows (row stride c / 8 bytes)
//   mode 0 : identity
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                      const float* __restrict__ x, const float* __restrict__ rowmask,
                                                      float slope, int mode, float* __restrict__ dz, char* __restrict__ plane,
                                                      long ldp, int split, float* __restrict__ dbias, int bias_parts, int rows, int c,
                                                      unsigned drop_thresh, unsigned drop_seed_h, float drop_inv_keep) {
    // block: 64 rows x 128 columns (blockIdx.y = column group); 256 threads = 32 column quads x 8
    // rows in flight.  Column sums are combined in LDS first: ONE global atomic per column per
    // block (same-address atomics serialise at L2 and were the bottleneck of the first version).
    __shared__ float colsum[8][128];
    const int q = threadIdx.x & 31, rsub = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 64, r1 = min(r0 + 64, rows);
    const int cbase = blockIdx.y * 128;
    constexpr int rpar = 8;
    {
        const int c4 = cbase + (q << 2);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < c)
        for (int r = r0 + rsub; r < r1; r += rpar) {
            const long o = (long)r * c + c4;
            float4 gv = *(const float4*)(g + o);
            const float rm = rowmask ? rowmask[r] : 1.f;
            float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
            if (mode == 1) {
                const float4 yv = *(const float4*)(y + o), xv = *(const float4*)(x + o);
                m.x = (yv.x - xv.x) > 0.f ? 1.f : slope; m.y = (yv.y - xv.y) > 0.f ? 1.f : slope;
                m.z = (yv.z - xv.z) > 0.f ? 1.f : slope; m.w = (yv.w - xv.w) > 0.f ? 1.f : slope;
            } else if (mode == 4) {
                // 16 bytes per row and 128-column group: word u, bit q = column 4 q + u of the group (this thread: q)
                const uint4 w = *(const uint4*)((const char*)y + (long)r * (c >> 3) + blockIdx.y * 16);
                m.x = (w.x >> q) & 1u ? 1.f : slope; m.y = (w.y >> q) & 1u ? 1.f : slope;
                m.z = (w.z >> q) & 1u ? 1.f : slope; m.w = (w.w >> q) & 1u ? 1.f : slope;
            } else if (mode == 5) {
                // plain bit rows (efts_resconv5 `sign_bits`): bit j of byte cc = column 8 cc + j
                const unsigned b = ((const unsigned char*)y)[(long)r * (c >> 3) + (c4 >> 3)] >> (c4 & 4);
                m.x = b & 1u ? 1.f : slope; m.y = b & 2u ? 1.f : slope; m.z = b & 4u ? 1.f : slope; m.w = b & 8u ? 1.f : slope;
            } else if (mode == 2 || mode == 3) {
                const float4 yv = *(const float4*)(y + o);
                const float neg = mode == 2 ? 0.f : slope;
                m.x = yv.x > 0.f ? 1.f : neg; m.y = yv.y > 0.f ? 1.f : neg;
                m.z = yv.z > 0.f ? 1.f : neg; m.w = yv.w > 0.f ? 1.f : neg;
            }
            if (drop_thresh) {                               // the forward launch's Dropout mask, regenerated from (seed, element index)
                const unsigned e0 = (unsigned)o;
                m.x *= drop_scale(drop_seed_h, e0, drop_thresh, drop_inv_keep); m.y *= drop_scale(drop_seed_h, e0 + 1, drop_thresh, drop_inv_keep);
                m.z *= drop_scale(drop_seed_h, e0 + 2, drop_thresh, drop_inv_keep); m.w *= drop_scale(drop_seed_h, e0 + 3, drop_thresh, drop_inv_keep);
            }
            gv.x *= m.x * rm; gv.y *= m.y * rm; gv.z *= m.z * rm; gv.w *= m.w * rm;
            if (dz) *(float4*)(dz + o) = gv;
            if (plane) plane_store4(plane + (long)r * ldp, c4, gv.x, gv.y, gv.z, gv.w, split);
            acc.x += gv.x; acc.y += gv.y; acc.z += gv.z; acc.w += gv.w;
        }
        if (dbias) {
            colsum[rsub][q * 4] = acc.x; colsum[rsub][q * 4 + 1] = acc.y;
            colsum[rsub][q * 4 + 2] = acc.z; colsum[rsub][q * 4 + 3] = acc.w;
            __syncthreads();
            if (threadIdx.x < 128 && cbase + threadIdx.x < c) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += colsum[k][threadIdx.x];
                // bias_parts: dbias is a [row blocks][c] workspace of per-block sums (plain stores, summed by efts_wgrad_reduce_grouped):
                // at mel length the 400 same-address atomics per column cost ~8 us of a 22 us launch
                if (bias_parts) dbias[(long)blockIdx.x * c + cbase + threadIdx.x] = t;
                else atomicAdd(dbias + cbase + threadIdx.x, t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Transposed operand plane for wgrad: out[ch][t] = x[t + shift][ch] (0 outside [0, rows)), K = t.
// One block = 64 t x 32 channels through LDS; each thread stores 2 consecutive t (4-byte stores,
// 128 contiguous bytes per 32 lanes).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_t_kernel(const float* __restrict__ x, long ldx, char* __restrict__ plane,
                                                     long ldp, long plane_stride, int split, int rows, int c, int shift0,
                                                     int nshift, int kpad) {
    // nshift planes (shift0, shift0+1, ...) from ONE staged (64 + nshift - 1)-row tile
    __shared__ float tile[64 + 10][33];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 64 + nshift - 1; r += 8) {
        const int t = t0 + r + shift0;
        tile[r][tx] = (t >= 0 && t < rows && c0 + tx < c) ? x[(long)t * ldx + c0 + tx] : 0.f;
    }
    __syncthreads();
    const int tt = t0 + 2 * tx;                        // this thread's first t
    if (tt >= kpad) return;
    for (int sft = 0; sft < nshift; ++sft) {
        char* pl = plane + (long)sft * plane_stride;
        for (int r = ty; r < 32; r += 8) {
            if (c0 + r >= c) continue;
            const float a = tile[2 * tx + sft][r], b = tile[2 * tx + 1 + sft][r];
            const unsigned short ha = f32_to_bf16(a), hb = f32_to_bf16(b);
            char* d = pl + (long)(c0 + r) * ldp + plane_off_hi(tt, split);
            *(unsigned*)d = (unsigned)ha | ((unsigned)hb << 16);
            if (split == 2) {
                const unsigned short la = f32_to_bf16(a - bf16_to_f32(ha)), lb = f32_to_bf16(b - bf16_to_f32(hb));
                *(unsigned*)(d + 64) = (unsigned)la | ((unsigned)lb << 16);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad reduction: dW[co][ci][k] = sum_s part[k][s][co][ci]; optional weight-norm backward
// (w = g v / ||v||:  dg = <dW, v> / ||v||,  dv = g/||v|| (dW - v <dW, v> / ||v||^2)).
// One block per output channel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ v,
                                                           const float* __restrict__ g, float* __restrict__ dw_or_dv,
                                                           float* __restrict__ dg, int cout, int cin, int taps,
                                                           const float* __restrict__ bias_part, int nparts, float* __restrict__ dbias) {
    extern __shared__ float dw_s[];                    // [cin * taps]
    __shared__ float sh[4];
    const int co = blockIdx.x;
    const int n = cin * taps;
    float dot = 0.f, nn = 0.f;
    if (bias_part) {                                   // bias gradient of this output channel: the row-block sums efts_act_bwd left, fixed order
        float b = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) b += bias_part[(long)i * cout + co];
        b = block_sum256t(b, sh);
        if (threadIdx.x == 0) dbias[co] += b;
        __syncthreads();
    }
    if ((cin & 3) == 0 && (((uintptr_t)part) & 15) == 0) {
        // float4 along ci, one (tap, 4 ci) item per thread and pass, the K-split sum unrolled by 4:
        // 4 independent 16-byte loads in flight per thread (the scalar version was latency-bound at ~2 TB/s)
        const int q = cin >> 2;
        for (int it = threadIdx.x; it < taps * q; it += 256) {
            const int k = it / q, c4 = (it - k * q) << 2;
            const float* src = part + ((long)k * nsplit * cout + co) * cin + c4;
            const long sstride = (long)cout * cin;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int sp = 0;
            for (; sp + 4 <= nsplit; sp += 4) {
                const float4 a = *(const float4*)(src + (sp + 0) * sstride), b = *(const float4*)(src + (sp + 1) * sstride);
                const float4 c = *(const float4*)(src + (sp + 2) * sstride), d = *(const float4*)(src + (sp + 3) * sstride);
                s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
                s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
            }
            for (; sp < nsplit; ++sp) {
                const float4 a = *(const float4*)(src + sp * sstride);
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            dw_s[(c4 + 0) * taps + k] = s.x; dw_s[(c4 + 1) * taps + k] = s.y;
            dw_s[(c4 + 2) * taps + k] = s.z; dw_s[(c4 + 3) * taps + k] = s.w;
        }
    } else {
        for (int k = 0; k < taps; ++k)
            for (int ci = threadIdx.x; ci < cin; ci += 256) {      // threads along ci: coalesced partial reads
                float s = 0.f;
                for (int sp = 0; sp < nsplit; ++sp) s += part[(((long)k * nsplit + sp) * cout + co) * cin + ci];
                dw_s[ci * taps + k] = s;
            }
    }
    __syncthreads();
    if (g)
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const float vv = v[(long)co * n + idx];
            dot += dw_s[idx] * vv;
            nn += vv * vv;
        }
    if (!g) {
        for (int idx = threadIdx.x; idx < n; idx += 256) dw_or_dv[(long)co * n + idx] = dw_s[idx];
        return;
    }
    dot = block_sum256t(dot, sh);
    nn = block_sum256t(nn, sh);
    const float nrm = sqrtf(nn), gg = g[co];
    if (threadIdx.x == 0) dg[co] = dot / nrm;
    const float a = gg / nrm, bcoef = dot / nn;
    for (int idx = threadIdx.x; idx < n; idx += 256) dw_or_dv[(long)co * n + idx] = a * (dw_s[idx] - v[(long)co * n + idx] * bcoef);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward over channels (+ the ReLU in front of it, + optionally the Linear(c,1) behind
// it), one wave per row:
//   y = xhat * gamma + beta, xhat = (x - mean) * rstd, x = relu(z)
//   dy = dy_in[row]            (mode 0)      or  ddur[row] * w      (mode 1, Linear(c,1) tail)
//   dx = rstd (dyg - mean(dyg) - xhat mean(dyg xhat)), dyg = dy gamma ; dz = dx * [x > 0] * rowmask
//   dgamma += dy xhat, dbeta += dy, dbias_conv += dz, (mode 1) dw += ddur * y, db += ddur
// ---------------------------------------------------------------------------------------------
#ifndef EFTS_LNB_ROWS
#define EFTS_LNB_ROWS 8
#endif
constexpr int LNB_ROWS = EFTS_LNB_ROWS;    // default rows per block (EFTS_LNB_ROWS env overrides): the duration predictor only has B*T1 (~4k) rows
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ dy_in, const float* __restrict__ ddur,
                                                            const float* __restrict__ w, const float* __restrict__ rowmask,
                                                            float* __restrict__ dz, char* __restrict__ plane, long ldp, int split,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dbias, float* __restrict__ dw, float* __restrict__ db,
                                                            int rows, int c, float drop_p, unsigned drop_seed, const unsigned* __restrict__ seed_add,
                                                            int ROWS) {
    extern __shared__ float acc_s[];                   // [4 waves][4][c]: dgamma, dbeta, dbias, dw partials of each wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // per-lane accumulators of the channels this lane owns in every row (u * 256 + lane * 4 + e): the column sums
    // stay in registers across the wave's rows; LDS only combines the 4 waves, global atomics only the blocks
    float ag[8][4], ab[8][4], ac[8][4], aw[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) ag[u][e] = ab[u][e] = ac[u][e] = aw[u][e] = 0.f;
    const int nv = c >> 8;
    const bool drop = drop_p > 0.f;
    const unsigned thresh = drop ? (unsigned)(drop_p * 4294967296.0) : 0u, seed_h = hash_u32(drop_seed + (seed_add ? *seed_add : 0u));
    const float inv_keep = drop ? 1.f / (1.f - drop_p) : 1.f;
    float db_loc = 0.f;
    // each wave walks rows blockIdx.x*ROWS + wv, +4, ...
    for (int rr = wv; rr < ROWS; rr += 4) {
        const int row = blockIdx.x * ROWS + rr;
        if (row >= rows) break;
        const float* xr = x + (long)row * c;
        float4 xv[8];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nv) { xv[u] = *(const float4*)(xr + u * 256 + lane * 4); s += xv[u].x + xv[u].y + xv[u].z + xv[u].w; }
        const float mean = wave_sum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nv) {
                const float a0 = xv[u].x - mean, a1 = xv[u].y - mean, a2 = xv[u].z - mean, a3 = xv[u].w - mean;
                q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)c + eps);
        const float dd = ddur ? ddur[row] : 0.f;
        const float rm = rowmask ? rowmask[row] : 1.f;
        float s1 = 0.f, s2 = 0.f;
        float4 dyv[8], dmv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nv) {
                const int c4 = u * 256 + lane * 4;
                const float4 gm = *(const float4*)(gamma + c4);
                float4 dy;
                if (ddur) { const float4 ww = *(const float4*)(w + c4); dy = make_float4(dd * ww.x, dd * ww.y, dd * ww.z, dd * ww.w); }
                else { dy = *(const float4*)(dy_in + (long)row * c + c4); dy.x *= rm; dy.y *= rm; dy.z *= rm; dy.w *= rm; }
                float4 dm = make_float4(1.f, 1.f, 1.f, 1.f);
                if (drop) {
                    const unsigned e0 = (unsigned)row * (unsigned)c + c4;
                    dm = make_float4(drop_scale(seed_h, e0, thresh, inv_keep), drop_scale(seed_h, e0 + 1, thresh, inv_keep),
                                     drop_scale(seed_h, e0 + 2, thresh, inv_keep), drop_scale(seed_h, e0 + 3, thresh, inv_keep));
                    dy.x *= dm.x; dy.y *= dm.y; dy.z *= dm.z; dy.w *= dm.w;
                }
                dmv[u] = dm;
                dyv[u] = dy;
                const float h0 = (xv[u].x - mean) * rstd, h1 = (xv[u].y - mean) * rstd, h2 = (xv[u].z - mean) * rstd, h3 = (xv[u].w - mean) * rstd;
                s1 += dy.x * gm.x + dy.y * gm.y + dy.z * gm.z + dy.w * gm.w;
                s2 += dy.x * gm.x * h0 + dy.y * gm.y * h1 + dy.z * gm.z * h2 + dy.w * gm.w * h3;
            }
        s1 = wave_sum(s1) / (float)c;
        s2 = wave_sum(s2) / (float)c;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nv) {
                const int c4 = u * 256 + lane * 4;
                const float4 gm = *(const float4*)(gamma + c4), bt = *(const float4*)(beta + c4);
                const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                const float dys[4] = {dyv[u].x, dyv[u].y, dyv[u].z, dyv[u].w};
                const float gms[4] = {gm.x, gm.y, gm.z, gm.w}, bts[4] = {bt.x, bt.y, bt.z, bt.w};
                const float dms[4] = {dmv[u].x, dmv[u].y, dmv[u].z, dmv[u].w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (xs[e] - mean) * rstd;
                    const float dx = rstd * (dys[e] * gms[e] - s1 - h * s2);
                    o[e] = (xs[e] > 0.f ? dx : 0.f) * rm;
                    ag[u][e] += dys[e] * h;
                    ab[u][e] += dys[e];
                    ac[u][e] += o[e];
                    if (ddur) aw[u][e] += dd * (h * gms[e] + bts[e]) * dms[e];
                }
                if (dz) *(float4*)(dz + (long)row * c + c4) = make_float4(o[0], o[1], o[2], o[3]);
                if (plane) plane_store4(plane + (long)row * ldp, c4, o[0], o[1], o[2], o[3], split);
            }
        if (lane == 0) db_loc += dd;
    }
    float* mine = acc_s + (long)wv * 4 * c;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nv) {
            const int c4 = u * 256 + lane * 4;
            *(float4*)(mine + c4) = make_float4(ag[u][0], ag[u][1], ag[u][2], ag[u][3]);
            *(float4*)(mine + c + c4) = make_float4(ab[u][0], ab[u][1], ab[u][2], ab[u][3]);
            *(float4*)(mine + 2 * c + c4) = make_float4(ac[u][0], ac[u][1], ac[u][2], ac[u][3]);
            *(float4*)(mine + 3 * c + c4) = make_float4(aw[u][0], aw[u][1], aw[u][2], aw[u][3]);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) {
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = acc_s[q * c + i] + acc_s[4 * c + q * c + i] + acc_s[8 * c + q * c + i] + acc_s[12 * c + q * c + i];
        atomicAdd(dgamma + i, t[0]);
        atomicAdd(dbeta + i, t[1]);
        if (dbias) atomicAdd(dbias + i, t[2]);
        if (dw) atomicAdd(dw + i, t[3]);
    }
    if (db && lane == 0 && db_loc != 0.f) atomicAdd(db, db_loc);
}

// ---------------------------------------------------------------------------------------------
// Alignment-block backward (notation of DESIGN.md section 9)
// ---------------------------------------------------------------------------------------------
// (2a) r[b][j] = sum_i alpha'[i][j] * dA[i][j]
__global__ __launch_bounds__(256) void alpha_bwd_r_kernel(const float* __restrict__ ra, const float* __restrict__ dA, float* __restrict__ r,
                                                          int T1, int T2) {
    // 64 columns j per block, the T1 rows split over the block's 4 waves (one thread per (b, j) with a 128-long
    // dependent loop left most of the chip idle: 55 us for 26 MB)
    __shared__ float part[4][64];
    const int b = blockIdx.y, jj = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jj;
    float s0 = 0.f, s1 = 0.f;
    if (j < T2) {
        const float* a = ra + (long)b * T1 * T2 + j;
        const float* d = dA + (long)b * T1 * T2 + j;
        int i = w;
        for (; i + 4 < T1; i += 8) {
            s0 += a[(long)i * T2] * d[(long)i * T2];
            s1 += a[(long)(i + 4) * T2] * d[(long)(i + 4) * T2];
        }
        if (i < T1) s0 += a[(long)i * T2] * d[(long)i * T2];
    }
    part[w][jj] = s0 + s1;
    __syncthreads();
    if (w == 0 && j < T2) r[(long)b * T2 + j] = part[0][jj] + part[1][jj] + part[2][jj] + part[3][jj];
}
// (2b) de[b][i] = sum_j alpha'[i][j] (dA[i][j] - r[j]) * 2 sigma (q_j - e_i); one wave per (b, i)
__global__ __launch_bounds__(256) void alpha_bwd_e_kernel(const float* __restrict__ ra, const float* __restrict__ dA,
                                                          const float* __restrict__ r, const float* __restrict__ e,
                                                          const int* __restrict__ tlen, const int* __restrict__ mlen,
                                                          float sigma, float* __restrict__ de, int T1, int T2) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= T1) return;
    float s = 0.f;
    if (i < tlen[b]) {
        const int ml = min(mlen[b], T2);
        const float ei = e[(long)b * T1 + i];
        const float* a = ra + ((long)b * T1 + i) * T2;
        const float* d = dA + ((long)b * T1 + i) * T2;
        for (int j = lane; j < ml; j += 64) s += a[j] * (d[j] - r[(long)b * T2 + j]) * 2.f * sigma * ((float)j - ei);
        s = wave_sum(s);
    }
    if (lane == 0) de[(long)b * T1 + i] = s;
}
// (3a) softmax statistics of beta rows: mx[b][i], se[b][i]
__global__ __launch_bounds__(256) void beta_stats_kernel(const float* __restrict__ imv, const int* __restrict__ tlen,
                                                         const int* __restrict__ mlen, float sigma_e, float* __restrict__ mxo,
                                                         float* __restrict__ seo, int T1, int T2) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= T1) return;
    const int ml = min(mlen[b], T2);
    float mx = 0.f, se = 1.f;
    if (i < tlen[b] && ml > 0) {
        const float* pi = imv + (long)b * T2;
        mx = -INFINITY;
        for (int j = lane; j < ml; j += 64) { const float d = pi[j] - (float)i; mx = fmaxf(mx, -sigma_e * d * d); }
        mx = wave_max(mx);
        se = 0.f;
        for (int j = lane; j < ml; j += 64) { const float d = pi[j] - (float)i; se += __expf(-sigma_e * d * d - mx); }
        se = wave_sum(se);
    }
    if (lane == 0) { mxo[(long)b * T1 + i] = mx; seo[(long)b * T1 + i] = se; }
}
// (3b) dpi[b][j] = sum_i beta_ij de_i (j - e_i) (-2 sigma_e)(pi_j - i); one thread per (b, j)
__global__ __launch_bounds__(128) void e_bwd_kernel(const float* __restrict__ imv, const float* __restrict__ e,
                                                    const float* __restrict__ de, const float* __restrict__ mx,
                                                    const float* __restrict__ se, const int* __restrict__ tlen,
                                                    const int* __restrict__ mlen, float sigma_e, float* __restrict__ dpi,
                                                    int T1, int T2) {
    extern __shared__ float sm[];                      // e, de, mx, 1/se  [4][T1]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < T1; i += blockDim.x) {
        sm[i] = e[(long)b * T1 + i];
        sm[T1 + i] = de[(long)b * T1 + i];
        sm[2 * T1 + i] = mx[(long)b * T1 + i];
        sm[3 * T1 + i] = 1.f / se[(long)b * T1 + i];
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T2) return;
    float out = 0.f;
    if (j < mlen[b]) {
        const int tl = min(tlen[b], T1);
        const float pj = imv[(long)b * T2 + j];
        for (int i = 0; i < tl; ++i) {
            const float d = pj - (float)i;
            const float beta = __expf(-sigma_e * d * d - sm[2 * T1 + i]) * sm[3 * T1 + i];
            out += beta * sm[T1 + i] * ((float)j - sm[i]) * (-2.f * sigma_e) * d;
        }
    }
    dpi[(long)b * T2 + j] = out;
}
// (4) backward of normalise / mask / cumsum / relu-diff: one wavefront per item
__global__ __launch_bounds__(64) void imv_bwd_kernel(const float* __restrict__ sidx, const float* __restrict__ imv,
                                                     const float* __restrict__ dpi, const int* __restrict__ tlen,
                                                     const int* __restrict__ mlen, float* __restrict__ ds, int T2) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* s = sidx + (long)b * T2;
    const float* pi = imv + (long)b * T2;
    const float* dp = dpi + (long)b * T2;
    float* o = ds + (long)b * T2;
    const int ml = min(mlen[b], T2);
    const float scale = (float)tlen[b] - 1.f;
    // pi = u / mx * scale.  Recover the arg-max position: u is non-decreasing, so max u <=> max pi;
    // torch.max returns the FIRST index attaining it.
    float pmax = 0.f;
    for (int j = lane; j < ml; j += 64) pmax = fmaxf(pmax, pi[j]);
    pmax = wave_max(pmax);
    int jstar = 0x7fffffff;
    for (int j = lane; j < ml; j += 64) if (pi[j] == pmax) jstar = min(jstar, j);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) jstar = min(jstar, __shfl_xor(jstar, off));
    // sum_j dpi_j pi_j (for d mx): dmx = -sum_j dpi_j u_j scale / mx^2 = -(sum_j dpi_j pi_j) / mx
    float dot = 0.f;
    for (int j = lane; j < ml; j += 64) dot += dp[j] * pi[j];
    dot = wave_sum(dot);
    // re-run the forward scan for u_max = max_j cs_j [j < ml] (= mx before the 1e-8 clamp)
    const int chunk = (T2 + 63) / 64;
    const int j0 = lane * chunk, j1 = min(j0 + chunk, T2);
    float loc = 0.f;
    for (int j = j0; j < j1; ++j) loc += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f;
    const float incl = wave_scan_incl(loc);
    float run = incl - loc, umax = 0.f;
    for (int j = j0; j < j1; ++j) { run += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f; if (j < ml) umax = fmaxf(umax, run); }
    umax = wave_max(umax);
    const bool clamped = umax < 1e-8f;
    const float mx = clamped ? 1e-8f : umax;
    const float dmx = clamped ? 0.f : -dot / mx;
    // du_j = dpi_j * scale / mx (+ dmx at j*), j < ml;  dDelta_j = suffix sum of du
    float locs = 0.f;
    for (int j = j0; j < j1; ++j) if (j < ml) locs += dp[j] * scale / mx + (j == jstar ? dmx : 0.f);
    float suf = locs;                                  // inclusive suffix sum over lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const float t = __shfl_down(suf, off); if (lane + off < 64) suf += t; }
    const float carry = suf - locs;                    // sum of all later lanes
    // g_j = dDelta_j * [s_j - s_{j-1} > 0] (j >= 1, g_0 = 0);  ds_j = g_j - g_{j+1}
    // g at this lane's first element needs no loop: dDelta_{j0} = carry + locs
    float gfirst = (j0 < T2 && j0 > 0 && (s[j0] - s[j0 - 1]) > 0.f) ? carry + locs : 0.f;
    float prev = __shfl_down(gfirst, 1);
    if (lane == 63) prev = 0.f;
    float runb = carry;
    for (int j = j1 - 1; j >= j0; --j) {
        if (j < ml) runb += dp[j] * scale / mx + (j == jstar ? dmx : 0.f);
        const float gj = (j > 0 && (s[j] - s[j - 1]) > 0.f) ? runb : 0.f;
        o[j] = gj - prev;
        prev = gj;
    }
}
// (5) dS[j][i] = alpha_ij * ds_j * (i - s_j): one wave per (b, j); recomputes alpha from the scores
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ sc, long ld, const float* __restrict__ sidx,
                                                       const float* __restrict__ ds, const int* __restrict__ tlen,
                                                       const int* __restrict__ mlen, float* __restrict__ dS, long ldd,
                                                       char* __restrict__ plane, long ldp, int B, int T1, int T2, int T2p) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)B * T2) return;
    const int b = (int)(r / T2), j = (int)(r - (long)b * T2);
    const int tl = min(tlen[b], T1);
    const bool live = j < mlen[b];
    const float* row = sc + r * ld;
    float mx = -INFINITY;
    for (int i = lane; i < tl; i += 64) mx = fmaxf(mx, row[i]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int i = lane; i < tl; i += 64) se += __expf(row[i] - mx);
    se = wave_sum(se);
    const float g = live ? ds[r] / se : 0.f, sj = sidx[r];
    const int kp = (T1 + 31) & ~31;
    char* prow = plane + ((long)b * T2p + j) * ldp;
    for (int i0 = lane * 4; i0 < kp; i0 += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            v[u] = (i < tl) ? __expf(row[i] - mx) * g * ((float)i - sj) : 0.f;
            if (i < T1) dS[r * ldd + i] = v[u];
        }
        plane_store4(prow, i0, v[0], v[1], v[2], v[3], 2);
    }
}

// embedding backward: dtable[id] += g[row]   (rows of the padded row space; gap rows skipped)
__global__ void embed_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ g, float* __restrict__ dtable,
                                 int T, int Tp, int c, int nsym) {
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    if (t >= T) return;
    long id = ids[(long)b * T + t];
    id = id < 0 ? 0 : (id >= nsym ? nsym - 1 : id);
    for (int cc = threadIdx.x; cc < c; cc += blockDim.x) atomicAdd(dtable + id * c + cc, g[(long)row * c + cc]);
}

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam(amsgrad, coupled L2) on flat fp32 buffers
// (trainer.py:154-158; egs/lj/conf/efficient_tts_cnn_phnseq_noDropout.v1.yaml:34-40)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) { const float4 v = *(const float4*)(g + i); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
        else for (long k = i; k < n; ++k) s += g[k] * g[k];
    }
    s = block_sum256t(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// second stage: ONE block adds the per-block partials in a fixed order, so data-parallel replicas that
// hold identical gradients compute bit-identical norms (and stay in lock-step without parameter syncs)
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum256t(s, sh);
    if (threadIdx.x == 0) out[0] += s;
}
// x *= *s unless *s == 1 (the d(loss) factor autograd hands to backward(): 1 in the reference loop, where the pass over the
// 82 MB gradient buffer is skipped on the device without a host round trip)
__global__ __launch_bounds__(256) void scale_unless_one_kernel(float* __restrict__ x, long n, const float* __restrict__ s) {
    const float sv = *s;
    if (sv == 1.f) return;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) { float4 v = *(float4*)(x + i); v.x *= sv; v.y *= sv; v.z *= sv; v.w *= sv; *(float4*)(x + i) = v; }
        else for (long k = i; k < n; ++k) x[k] *= sv;
    }
}
// hyper (optional): {lr, 1 - beta1^t, sqrt(1 - beta2^t)} in device memory instead of the by-value arguments -- the step as a hipGraph
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ vmax, long n,
                                                   const float* __restrict__ sumsq, float max_norm, float gscale, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }
    float coef = gscale;
    if (sumsq && max_norm > 0.f) {
        const float nrm = sqrtf(sumsq[0]) * gscale;    // norm of the (already averaged) gradient
        const float cc = max_norm / (nrm + 1e-6f);
        coef *= cc < 1.f ? cc : 1.f;
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pp = p[i];
        const float gg = g[i] * coef + wd * pp;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        const float vm = fmaxf(vmax[i], vv);
        m[i] = mm; v[i] = vv; vmax[i] = vm;
        p[i] = pp - (lr / bc1) * mm / (sqrtf(vm) / bc2_sqrt + eps);
    }
}

// transposed + tap-flipped weight plane for dgrad: plane[taps-1-k][ci][co] = w[co][ci][k]
__global__ __launch_bounds__(256) void pack_weight_t_kernel(const float* __restrict__ w, char* __restrict__ plane, long ldb,
                                                            int cout, int cin, int taps, int kp, int split) {
    const int ci = blockIdx.x;
    for (int q = threadIdx.x; q < (kp >> 2) * taps; q += 256) {
        const int k = q / (kp >> 2), c4 = (q - k * (kp >> 2)) << 2;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (c4 + u < cout) ? w[((long)(c4 + u) * cin + ci) * taps + k] : 0.f;
        plane_store4(plane + ((long)(taps - 1 - k) * cin + ci) * ldb, c4, v[0], v[1], v[2], v[3], split);
    }
}

// ---- grouped weight preparation: every item (same shape) in one launch per plane kind.  A training step repacks
// ~20 weights twice (forward plane + folded fp32 copy, transposed dgrad plane); as separate 8-11 us launches that
// was 0.4 ms of a 5 ms step, almost all of it launch latency.
struct PackItem { const float* w; const float* g; float* wout; char* plane; char* plane_t; };

__global__ __launch_bounds__(256) void pack_weight_group_kernel(const PackItem* __restrict__ items, long ldb, int cout, int cin,
                                                                int taps, int kp, int split) {
    __shared__ float sh[4];
    const PackItem it = items[blockIdx.y];
    const int co = blockIdx.x;
    const float* wr = it.w + (long)co * cin * taps;
    float scale = 1.f;
    if (it.g) {
        float ss = 0.f;
        for (int i = threadIdx.x; i < cin * taps; i += 256) ss += wr[i] * wr[i];
        ss = block_sum256t(ss, sh);
        scale = it.g[co] / sqrtf(ss);
    }
    if (it.wout)
        for (int i = threadIdx.x; i < cin * taps; i += 256) it.wout[(long)co * cin * taps + i] = wr[i] * scale;
    if (!it.plane) return;
    for (int q = threadIdx.x; q < (kp >> 2) * taps; q += 256) {
        const int k = q / (kp >> 2), c4 = (q - k * (kp >> 2)) << 2;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (c4 + u < cin) ? wr[(long)(c4 + u) * taps + k] * scale : 0.f;
        plane_store4(it.plane + ((long)k * cout + co) * ldb, c4, v[0], v[1], v[2], v[3], split);
    }
}

// transposed planes of the group; reads the folded copy when the item has one (written by the kernel above)
__global__ __launch_bounds__(256) void pack_weight_t_group_kernel(const PackItem* __restrict__ items, long ldb, int cout, int cin,
                                                                  int taps, int kp, int split) {
    const PackItem it = items[blockIdx.y];
    if (!it.plane_t) return;
    const float* w = (it.g && it.wout) ? it.wout : it.w;
    const int ci = blockIdx.x;
    for (int q = threadIdx.x; q < (kp >> 2) * taps; q += 256) {
        const int k = q / (kp >> 2), c4 = (q - k * (kp >> 2)) << 2;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (c4 + u < cout) ? w[((long)(c4 + u) * cin + ci) * taps + k] : 0.f;
        plane_store4(it.plane_t + ((long)(taps - 1 - k) * cin + ci) * ldb, c4, v[0], v[1], v[2], v[3], split);
    }
}

// ---- fast path (cout % 64 == 0, cin % 64 == 0, taps <= 5): weight-norm scales first, then one pass over the
// weights in [32 co][64 ci][taps] tiles through LDS that writes BOTH planes (and the folded copy if asked for)
// with full-sector accesses.  14 k5 layers: 82 + 199 us with the row kernels above -> see DESIGN.md.
__global__ __launch_bounds__(256) void weight_scale_kernel(const PackItem* __restrict__ items, float* __restrict__ scale, int cout,
                                                           int row_len) {
    __shared__ float sh[4];
    const PackItem it = items[blockIdx.y];
    const int co = blockIdx.x;
    float sc = 1.f;
    if (it.g) {
        const float* wr = it.w + (long)co * row_len;
        float ss = 0.f;
        for (int i = threadIdx.x; i < row_len; i += 256) ss += wr[i] * wr[i];
        ss = block_sum256t(ss, sh);            // same reduction order as pack_weight_kernel: identical scales
        sc = it.g[co] / sqrtf(ss);
    }
    if (threadIdx.x == 0) scale[(long)blockIdx.y * cout + co] = sc;
}

constexpr int PT_CO = 32, PT_CI = 64, PT_MAXT = 5;
__global__ __launch_bounds__(256) void pack_weight_tile_kernel(const PackItem* __restrict__ items, const float* __restrict__ scale,
                                                               long ldb, long ldb_t, int cout, int cin, int taps, int split) {
    __shared__ float tile[PT_CO][PT_CI * PT_MAXT + 1];
    __shared__ float sc[PT_CO];
    const PackItem it = items[blockIdx.z];
    const int co0 = blockIdx.x * PT_CO, ci0 = blockIdx.y * PT_CI;
    const int seg = PT_CI * taps;              // contiguous floats per co row of this tile
    if (threadIdx.x < PT_CO) sc[threadIdx.x] = scale[(long)blockIdx.z * cout + co0 + threadIdx.x];
    for (int i = threadIdx.x; i < PT_CO * seg; i += 256) {
        const int c = i / seg, r = i - c * seg;
        tile[c][r] = it.w[((long)(co0 + c) * cin + ci0) * taps + r];
    }
    __syncthreads();
    if (it.wout)
        for (int i = threadIdx.x; i < PT_CO * seg; i += 256) {
            const int c = i / seg, r = i - c * seg;
            it.wout[((long)(co0 + c) * cin + ci0) * taps + r] = tile[c][r] * sc[c];
        }
    if (it.plane)                               // [k][co][Kp(ci)]: 16 threads per 64-ci row piece
        for (int i = threadIdx.x; i < taps * PT_CO * (PT_CI / 4); i += 256) {
            const int q = i & 15, c = (i >> 4) & (PT_CO - 1), k = i >> 9;
            const float s = sc[c];
            const float* t = &tile[c][(4 * q) * taps + k];
            plane_store4(it.plane + ((long)k * cout + co0 + c) * ldb, ci0 + 4 * q, t[0] * s, t[taps] * s, t[2 * taps] * s, t[3 * taps] * s, split);
        }
    if (it.plane_t)                             // [taps-1-k][ci][Kp(co)]: 8 threads per 32-co row piece
        for (int i = threadIdx.x; i < taps * PT_CI * (PT_CO / 4); i += 256) {
            const int q = i & 7, c = (i >> 3) & (PT_CI - 1), k = i >> 9;
            const int r = c * taps + k;
            plane_store4(it.plane_t + ((long)(taps - 1 - k) * cin + ci0 + c) * ldb_t, co0 + 4 * q, tile[4 * q][r] * sc[4 * q],
                         tile[4 * q + 1][r] * sc[4 * q + 1], tile[4 * q + 2][r] * sc[4 * q + 2], tile[4 * q + 3][r] * sc[4 * q + 3], split);
        }
}

}  // namespace efts

using namespace efts;
#define ST ((hipStream_t)stream)

extern "C" int efts_pack_weights_grouped(const efts_pack_item* items, int32_t n_items, float* scale_ws, int64_t ldb, int64_t ldb_t,
                                         int32_t cout, int32_t cin, int32_t taps, int32_t split, int32_t with_t, void* stream) {
    static_assert(sizeof(PackItem) == sizeof(efts_pack_item), "efts_pack_item layout");
    if (!items || n_items <= 0) return efts_fail(EFTS_EINVAL, "efts_pack_weights_grouped: no items");
    if (!(split == 1 || split == 2) || cout <= 0 || cin <= 0 || taps <= 0) return efts_fail(EFTS_ESHAPE, "efts_pack_weights_grouped: bad shape/split");
    const int kp = split == 1 ? (cin + 63) & ~63 : (cin + 31) & ~31;
    const int kpt = split == 1 ? (cout + 63) & ~63 : (cout + 31) & ~31;
    if (ldb < (split == 1 ? kp * 2 : kp * 4) || (ldb & 15)) return efts_fail(EFTS_EALIGN, "efts_pack_weights_grouped: ldb too small or not 16-byte aligned");
    if (with_t && (ldb_t < (split == 1 ? kpt * 2 : kpt * 4) || (ldb_t & 15)))
        return efts_fail(EFTS_EALIGN, "efts_pack_weights_grouped: ldb_t too small or not 16-byte aligned");
    if (scale_ws && cout % 64 == 0 && cin % PT_CI == 0 && taps <= PT_MAXT) {
        hipLaunchKernelGGL(weight_scale_kernel, dim3(cout, n_items), dim3(256), 0, ST, (const PackItem*)items, scale_ws, cout, cin * taps);
        hipLaunchKernelGGL(pack_weight_tile_kernel, dim3(cout / PT_CO, cin / PT_CI, n_items), dim3(256), 0, ST, (const PackItem*)items,
                           (const float*)scale_ws, (long)ldb, (long)ldb_t, cout, cin, taps, split);
        return efts_check_launch("efts_pack_weights_grouped");
    }
    hipLaunchKernelGGL(pack_weight_group_kernel, dim3(cout, n_items), dim3(256), 0, ST, (const PackItem*)items, (long)ldb, cout, cin, taps, kp, split);
    if (with_t)
        hipLaunchKernelGGL(pack_weight_t_group_kernel, dim3(cin, n_items), dim3(256), 0, ST, (const PackItem*)items, (long)ldb_t, cout, cin, taps, kpt, split);
    return efts_check_launch("efts_pack_weights_grouped");
}

extern "C" int efts_pack_weight_t(const float* w, void* plane, int64_t ldb, int32_t cout, int32_t cin, int32_t taps, int32_t split, void* stream) {
    if (!w || !plane) return efts_fail(EFTS_EINVAL, "efts_pack_weight_t: null pointer");
    const int kp = split == 1 ? (cout + 63) & ~63 : (cout + 31) & ~31;
    if (ldb < (split == 1 ? kp * 2 : kp * 4) || (ldb & 15)) return efts_fail(EFTS_EALIGN, "efts_pack_weight_t: ldb too small or not 16-byte aligned");
    hipLaunchKernelGGL(pack_weight_t_kernel, dim3(cin), dim3(256), 0, ST, w, (char*)plane, (long)ldb, cout, cin, taps, kp, split);
    return efts_check_launch("efts_pack_weight_t");
}

extern "C" int efts_loss_bwd(const float* mel_pred, int64_t ldm, const float* speech, const int32_t* mel_len, const float* dur_pred,
                             const float* log_delta_e, const int32_t* text_len, const float* gscale, float* dmel, void* dmel_plane,
                             int64_t ld_plane, int32_t split, float* ddur, int32_t B, int32_t T1, int32_t T1p, int32_t T2, int32_t T2p,
                             int32_t odim, void* stream) {
    if (!mel_pred || !speech || !mel_len || !dur_pred || !log_delta_e || !text_len || !ddur) return efts_fail(EFTS_EINVAL, "efts_loss_bwd: null pointer");
    const int kp = split == 1 ? (odim + 63) & ~63 : (odim + 31) & ~31;
    const int th = ((kp / 4) + 63) & ~63;
    const int nb2 = (B * T1p + th - 1) / th;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(B * T2p + nb2), dim3(th), 0, ST, mel_pred, (long)ldm, speech, mel_len, dur_pred, log_delta_e,
                       text_len, gscale, dmel, (char*)dmel_plane, (long)ld_plane, split, ddur, B, T1, T1p, T2, T2p, odim, kp);
    return efts_check_launch("efts_loss_bwd");
}

extern "C" int efts_act_bwd_dropout(const float* g, const float* y, const float* x, const float* rowmask, float slope, int32_t mode, float* dz,
                                    void* plane, int64_t ld_plane, int32_t split, float* dbias, int32_t rows, int32_t c, float drop_p,
                                    uint32_t drop_seed, void* stream) {
    if (!g || (!dz && !plane)) return efts_fail(EFTS_EINVAL, "efts_act_bwd: null pointer");
    const int parts = mode & EFTS_ACT_BWD_BIAS_PARTS ? 1 : 0;
    mode &= ~EFTS_ACT_BWD_BIAS_PARTS;
    if (c % 4 || (mode == 1 && (!x || !y)) || ((mode == 2 || mode == 3) && !y) || (mode == 4 && (!y || c % 128 || ((uintptr_t)y & 15))) || (mode == 5 && (!y || c % 8)) || mode < 0 || mode > 5 ||
        (parts && !dbias))
        return efts_fail(EFTS_EINVAL, "efts_act_bwd: bad mode/shape");
    unsigned thresh = 0, seed_h = 0;
    float inv_keep = 1.f;
    if (drop_p > 0.f) {
        if (!(drop_p < 1.f) || (long)rows * c > 0xffffffffL) return efts_fail(EFTS_EINVAL, "efts_act_bwd_dropout: drop_p must be in [0, 1) and rows * c < 2^32");
        thresh = (unsigned)((double)drop_p * 4294967296.0);
        seed_h = hash_u32(drop_seed);
        inv_keep = 1.f / (1.f - drop_p);
    }
    hipLaunchKernelGGL(act_bwd_kernel, dim3((rows + 63) / 64, (c + 127) / 128), dim3(256), 0, ST, g, y, x, rowmask, slope, mode, dz, (char*)plane, (long)ld_plane,
                       split, dbias, parts, rows, c, thresh, seed_h, inv_keep);
    return efts_check_launch("efts_act_bwd");
}

extern "C" int efts_act_bwd(const float* g, const float* y, const float* x, const float* rowmask, float slope, int32_t mode, float* dz,
                            void* plane, int64_t ld_plane, int32_t split, float* dbias, int32_t rows, int32_t c, void* stream) {
    return efts_act_bwd_dropout(g, y, x, rowmask, slope, mode, dz, plane, ld_plane, split, dbias, rows, c, 0.f, 0u, stream);
}

extern "C" int efts_pack_t(const float* x, int64_t ldx, void* plane, int64_t ld_plane, int64_t plane_stride, int32_t split, int32_t rows,
                           int32_t c, int32_t shift0, int32_t nshift, int32_t kpad, void* stream) {
    if (!x || !plane) return efts_fail(EFTS_EINVAL, "efts_pack_t: null pointer");
    if (kpad % 64 || ld_plane < (split == 1 ? kpad * 2 : kpad * 4)) return efts_fail(EFTS_ESHAPE, "efts_pack_t: kpad must be a multiple of 64 and fit ld_plane");
    if (nshift < 1 || nshift > 11) return efts_fail(EFTS_ESHAPE, "efts_pack_t: nshift must be in 1..11");
    hipLaunchKernelGGL(pack_t_kernel, dim3(kpad / 64, (c + 31) / 32), dim3(256), 0, ST, x, (long)ldx, (char*)plane, (long)ld_plane, (long)plane_stride,
                       split, rows, c, shift0, nshift, kpad);
    return efts_check_launch("efts_pack_t");
}

extern "C" int efts_wgrad_reduce(const float* part, int32_t nsplit, const float* v, const float* g, float* dw_or_dv, float* dg, int32_t cout,
                                 int32_t cin, int32_t taps, void* stream) {
    if (!part || !dw_or_dv || (g && (!v || !dg))) return efts_fail(EFTS_EINVAL, "efts_wgrad_reduce: null pointer");
    if ((size_t)cin * taps * 4 > 60000) return efts_fail(EFTS_ESHAPE, "efts_wgrad_reduce: cin*taps too large for LDS");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cout), dim3(256), (size_t)cin * taps * sizeof(float), ST, part, nsplit, v, g, dw_or_dv, dg, cout, cin, taps,
                       (const float*)nullptr, 0, (float*)nullptr);
    return efts_check_launch("efts_wgrad_reduce");
}

// ---------------------------------------------------------------------------------------------
// Reduction of a grouped (stream-K) wgrad launch (efts_wgrad_tn_grouped): one block per (item, output channel).  The slabs
// that hold a tile's partial sums follow from the launch geometry -- list x = T % nx, position j = T / nx, workgroups
// floor(j N / q) .. floor(((j + 1) N - 1) / q) of that list, N = steps per tile, q = steps per workgroup -- and are added
// in that (fixed) order; then, as in wgrad_reduce_kernel, the weight-norm backward and the bias gradient.
// ---------------------------------------------------------------------------------------------
struct WrItem { const float* v; const float* g; float* dw_or_dv; float* dg; const float* bias_part; float* dbias; int nparts; };
struct WrSkArgs {
    WrItem it[EFTS_WGRAD_MAX_ITEMS];
    const float* part;
    int cout, cin, taps;
    int tiles_item, ntn, nx, steps_tile, q, maxseg;
    int stamp[8];
    const int* stamp_src;
};

__global__ __launch_bounds__(256) void wgrad_reduce_sk_kernel(WrSkArgs p) {
    extern __shared__ float dw_s[];                    // [cin * taps]
    __shared__ float sh[4];
    const int item = blockIdx.x / p.cout, co = blockIdx.x - item * p.cout;
    const WrItem& q = p.it[item];
    const int cin = p.cin, taps = p.taps, n = cin * taps;
    bool stale = false;                                // `part` was filled by a launch of another geometry (or by none): NaN, loudly, instead of a wrong sum
#pragma unroll
    for (int i = 0; i < 8; ++i) stale |= p.stamp_src[i] != p.stamp[i];
    if (stale) {
        float* o = q.dw_or_dv + (long)co * n;
        for (int idx = threadIdx.x; idx < n; idx += 256) o[idx] = __builtin_nanf("");
        if (threadIdx.x == 0) { if (q.dg) q.dg[co] = __builtin_nanf(""); if (q.bias_part) q.dbias[co] = __builtin_nanf(""); }
        return;
    }
    if (q.bias_part) {
        float b = 0.f;
        for (int i = threadIdx.x; i < q.nparts; i += 256) b += q.bias_part[(long)i * p.cout + co];
        b = block_sum256t(b, sh);
        if (threadIdx.x == 0) q.dbias[co] += b;
        __syncthreads();
    }
    const int mt = co >> 7, col = co & 127;
    const int q4 = cin >> 2;
    const long slab = (long)taps * 128 * 64;
    for (int it = threadIdx.x; it < taps * q4; it += 256) {
        const int k = it / q4, c4 = (it - k * q4) << 2;
        const int nt = c4 >> 6, cil = c4 & 63;
        const int T = item * p.tiles_item + mt * p.ntn + nt;
        const int x = T % p.nx, j = T / p.nx;
        const int l0 = (int)(((long)j * p.steps_tile) / p.q), l1 = (int)(((long)(j + 1) * p.steps_tile - 1) / p.q);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = l0; l <= l1; ++l) {
            const int seg = j - (int)(((long)l * p.q) / p.steps_tile);
            const float4 a = *(const float4*)(p.part + (long)((l * p.nx + x) * p.maxseg + seg) * slab + (k * 128 + col) * 64 + cil);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        dw_s[(c4 + 0) * taps + k] = s.x; dw_s[(c4 + 1) * taps + k] = s.y;
        dw_s[(c4 + 2) * taps + k] = s.z; dw_s[(c4 + 3) * taps + k] = s.w;
    }
    __syncthreads();
    float* out = q.dw_or_dv + (long)co * n;
    if (!q.g) {
        for (int idx = threadIdx.x; idx < n; idx += 256) out[idx] = dw_s[idx];
        return;
    }
    const float* vr = q.v + (long)co * n;
    float dot = 0.f, nn = 0.f;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const float vv = vr[idx];
        dot += dw_s[idx] * vv;
        nn += vv * vv;
    }
    dot = block_sum256t(dot, sh);
    nn = block_sum256t(nn, sh);
    const float nrm = sqrtf(nn), gg = q.g[co];
    if (threadIdx.x == 0) q.dg[co] = dot / nrm;
    const float a = gg / nrm, bcoef = dot / nn;
    for (int idx = threadIdx.x; idx < n; idx += 256) out[idx] = a * (dw_s[idx] - vr[idx] * bcoef);
}

extern "C" int efts_wgrad_reduce_grouped(const efts_wgrad_item* items, int32_t count, const float* part, int32_t rows, int32_t cout, int32_t cin,
                                         int32_t taps, int32_t split, int32_t workgroups, void* stream) {
    if (!items || !part) return efts_fail(EFTS_EINVAL, "efts_wgrad_reduce_grouped: null pointer");
    if (!(taps == 1 || taps == 3 || taps == 5)) return efts_fail(EFTS_EINVAL, "efts_wgrad_reduce_grouped: implemented for taps 1, 3, 5 (as efts_wgrad_tn_grouped)");
    if ((size_t)cin * taps * 4 > 60000) return efts_fail(EFTS_ESHAPE, "efts_wgrad_reduce_grouped: cin*taps too large for LDS");
    efts_wgrad_sk_geom gm;
    const int rc = efts_wgrad_sk_geometry(count, rows, cout, cin, split, workgroups, &gm);
    if (rc) return rc;
    WrSkArgs k;
    for (int i = 0; i < EFTS_WGRAD_MAX_ITEMS; ++i) {
        const efts_wgrad_item* q = items + (i < count ? i : 0);
        if (!q->dw_or_dv || (q->g && (!q->v || !q->dg))) return efts_fail(EFTS_EINVAL, "efts_wgrad_reduce_grouped: null pointer in item %d", i);
        if (q->bias_part && (!q->dbias || q->nparts < 1)) return efts_fail(EFTS_EINVAL, "efts_wgrad_reduce_grouped: bias_part needs dbias and nparts >= 1");
        k.it[i] = WrItem{q->v, q->g, q->dw_or_dv, q->dg, q->bias_part, q->dbias, q->nparts};
    }
    k.part = part; k.cout = cout; k.cin = cin; k.taps = taps;
    k.tiles_item = gm.tiles_item; k.ntn = gm.ntn; k.nx = gm.nx; k.steps_tile = gm.steps_tile; k.q = gm.q; k.maxseg = gm.maxseg;
    efts_wgrad_stamp(k.stamp, count, rows, cout, cin, taps, split, gm);
    k.stamp_src = (const int*)(part + (size_t)gm.workgroups * gm.maxseg * taps * 128 * 64);
    hipLaunchKernelGGL(wgrad_reduce_sk_kernel, dim3(count * cout), dim3(256), (size_t)cin * taps * sizeof(float), ST, k);
    return efts_check_launch("efts_wgrad_reduce_grouped");
}


extern "C" int efts_layernorm_bwd(const float* x, const float* gamma, const float* beta, float eps, const float* dy, const float* ddur,
                                  const float* w, const float* rowmask, float* dz, void* plane, int64_t ld_plane, int32_t split,
                                  float* dgamma, float* dbeta, float* dbias, float* dw, float* db, int32_t rows, int32_t c, float drop_p,
                                  uint32_t drop_seed, const uint32_t* drop_seed_add, void* stream) {
    if (!x || !gamma || !beta || (!dy && !ddur) || (ddur && !w) || !dgamma || !dbeta) return efts_fail(EFTS_EINVAL, "efts_layernorm_bwd: null pointer");
    if (c % 256 || c > 2048) return efts_fail(EFTS_ESHAPE, "efts_layernorm_bwd: c must be a multiple of 256, <= 2048");
    int rpb = LNB_ROWS;                        // rows per block
    if (c > 768) (void)hipFuncSetAttribute((const void*)layernorm_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * c * (int)sizeof(float));
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + rpb - 1) / rpb), dim3(256), (size_t)16 * c * sizeof(float), ST, x, gamma, beta, eps, dy, ddur, w,
                       rowmask, dz, (char*)plane, (long)ld_plane, split, dgamma, dbeta, dbias, dw, db, rows, c, drop_p, drop_seed, drop_seed_add, rpb);
    return efts_check_launch("efts_layernorm_bwd");
}

extern "C" int efts_alpha_bwd(const float* ralpha, const float* dalpha, const float* e, const int32_t* text_len, const int32_t* mel_len,
                              float sigma, float* r_ws, float* de, int32_t B, int32_t T1, int32_t T2, void* stream) {
    if (!ralpha || !dalpha || !e || !text_len || !mel_len || !r_ws || !de) return efts_fail(EFTS_EINVAL, "efts_alpha_bwd: null pointer");
    hipLaunchKernelGGL(alpha_bwd_r_kernel, dim3((T2 + 63) / 64, B), dim3(256), 0, ST, ralpha, dalpha, r_ws, T1, T2);
    hipLaunchKernelGGL(alpha_bwd_e_kernel, dim3((T1 + 3) / 4, B), dim3(256), 0, ST, ralpha, dalpha, (const float*)r_ws, e, text_len, mel_len, sigma, de, T1, T2);
    return efts_check_launch("efts_alpha_bwd");
}

extern "C" int efts_e_bwd(const float* imv, const float* e, const float* de, const int32_t* text_len, const int32_t* mel_len, float sigma_e,
                          float* stats_ws, float* dpi, int32_t B, int32_t T1, int32_t T2, void* stream) {
    if (!imv || !e || !de || !text_len || !mel_len || !stats_ws || !dpi) return efts_fail(EFTS_EINVAL, "efts_e_bwd: null pointer");
    float* mx = stats_ws; float* se = stats_ws + (long)B * T1;
    hipLaunchKernelGGL(beta_stats_kernel, dim3((T1 + 3) / 4, B), dim3(256), 0, ST, imv, text_len, mel_len, sigma_e, mx, se, T1, T2);
    hipLaunchKernelGGL(e_bwd_kernel, dim3((T2 + 127) / 128, B), dim3(128), (size_t)4 * T1 * sizeof(float), ST, imv, e, de, (const float*)mx, (const float*)se,
                       text_len, mel_len, sigma_e, dpi, T1, T2);
    return efts_check_launch("efts_e_bwd");
}

extern "C" int efts_imv_bwd(const float* soft_idx, const float* imv, const float* dpi, const int32_t* text_len, const int32_t* mel_len, float* ds,
                            int32_t B, int32_t T2, void* stream) {
    if (!soft_idx || !imv || !dpi || !text_len || !mel_len || !ds) return efts_fail(EFTS_EINVAL, "efts_imv_bwd: null pointer");
    hipLaunchKernelGGL(imv_bwd_kernel, dim3(B), dim3(64), 0, ST, soft_idx, imv, dpi, text_len, mel_len, ds, T2);
    return efts_check_launch("efts_imv_bwd");
}

extern "C" int efts_attn_bwd(const float* scores, int64_t ld, const float* soft_idx, const float* ds, const int32_t* text_len,
                             const int32_t* mel_len, float* dscores, int64_t ldd, void* plane, int64_t ld_plane, int32_t B, int32_t T1,
                             int32_t T2, int32_t T2p, void* stream) {
    if (!scores || !soft_idx || !ds || !text_len || !mel_len || !dscores || !plane) return efts_fail(EFTS_EINVAL, "efts_attn_bwd: null pointer");
    const long rows = (long)B * T2;
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, scores, (long)ld, soft_idx, ds, text_len, mel_len, dscores,
                       (long)ldd, (char*)plane, (long)ld_plane, B, T1, T2, T2p);
    return efts_check_launch("efts_attn_bwd");
}

extern "C" int efts_embed_bwd(const int64_t* ids, const float* g, float* dtable, int32_t B, int32_t T, int32_t Tp, int32_t c, int32_t num_symbols,
                              void* stream) {
    if (!ids || !g || !dtable) return efts_fail(EFTS_EINVAL, "efts_embed_bwd: null pointer");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(B * Tp), dim3(256), 0, ST, (const long*)ids, g, dtable, T, Tp, c, num_symbols);
    return efts_check_launch("efts_embed_bwd");
}

extern "C" size_t efts_sumsq_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int efts_sumsq(const float* g, int64_t n, float* out1, void* workspace, void* stream) {
    if (!g || !out1 || !workspace || n <= 0) return efts_fail(EFTS_EINVAL, "efts_sumsq: bad arguments");
    if ((uintptr_t)g & 15) return efts_fail(EFTS_EALIGN, "efts_sumsq: buffer must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, ST, g, (long)n, (float*)workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, ST, (const float*)workspace, 1024, out1);
    return efts_check_launch("efts_sumsq");
}

extern "C" int efts_scale_unless_one(float* x, int64_t n, const float* scale, void* stream) {
    if (!x || !scale || n <= 0) return efts_fail(EFTS_EINVAL, "efts_scale_unless_one: bad arguments");
    if ((uintptr_t)x & 15) return efts_fail(EFTS_EALIGN, "efts_scale_unless_one: buffer must be 16-byte aligned");
    hipLaunchKernelGGL(scale_unless_one_kernel, dim3(2048), dim3(256), 0, ST, x, (long)n, scale);
    return efts_check_launch("efts_scale_unless_one");
}

extern "C" int efts_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, const float* sumsq, float max_norm,
                                 float gscale, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    if (!p || !g || !m || !v || !vmax || n <= 0 || step < 1) return efts_fail(EFTS_EINVAL, "efts_adam_amsgrad: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, ST, p, g, m, v, vmax, (long)n, sumsq, max_norm, gscale, lr, beta1, beta2, eps, weight_decay,
                       bc1, sqrtf(bc2), (const float*)nullptr);
    return efts_check_launch("efts_adam_amsgrad");
}

extern "C" int efts_adam_hyper(float lr, float beta1, float beta2, int32_t step, float* out3) {
    if (!out3 || step < 1) return efts_fail(EFTS_EINVAL, "efts_adam_hyper: bad arguments");
    out3[0] = lr; out3[1] = 1.f - powf(beta1, (float)step); out3[2] = sqrtf(1.f - powf(beta2, (float)step));
    return EFTS_OK;
}

extern "C" int efts_adam_amsgrad_dev(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, const float* sumsq, float max_norm,
                                     float gscale, const float* hyper, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (!p || !g || !m || !v || !vmax || !hyper || n <= 0) return efts_fail(EFTS_EINVAL, "efts_adam_amsgrad_dev: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, ST, p, g, m, v, vmax, (long)n, sumsq, max_norm, gscale, 0.f, beta1, beta2, eps, weight_decay,
                       1.f, 1.f, hyper);
    return efts_check_launch("efts_adam_amsgrad_dev");
}

struct Words8 { uint32_t w[8]; };
__global__ void store_words_kernel(uint32_t* __restrict__ dst, Words8 v, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.w[threadIdx.x];
}
extern "C" int efts_store_words(uint32_t* dst, const uint32_t* words, int32_t n, void* stream) {
    if (!dst || !words || n < 1 || n > 8) return efts_fail(EFTS_EINVAL, "efts_store_words: 1..8 words");
    Words8 v;
    for (int i = 0; i < 8; ++i) v.w[i] = i < n ? words[i] : 0u;
    hipLaunchKernelGGL(store_words_kernel, dim3(1), dim3(64), 0, ST, dst, v, n);
    return efts_check_launch("efts_store_words");
}
