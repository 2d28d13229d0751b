// efts_ops.hip -- the HBM-bound kernels of the EFTS-CNN path on gfx950: parameter packing,
// row-space producers, the alignment (IMV) block, LayerNorm tails and the masked losses.
// All fp32 VALU work on 64-lane wavefronts; every kernel reads/writes coalesced rows and
// derives masks from the int32 length vectors in-kernel (the reference builds them on the
// host from lengths.tolist(): nntts/utils/nets_utils.py:145-166).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

// ----------------------------------------------------------------------------------------
// block-level helpers (256 threads = 4 waves)
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max256(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
// exclusive prefix of one value per thread over a 256-thread block; *total = block sum
__device__ __forceinline__ float block_scan_excl256(float v, float* sh /* >= 4 */, float* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float incl = wave_scan_incl(v);
    __syncthreads();
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    float base = 0.f;
    for (int i = 0; i < w; ++i) base += sh[i];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + incl - v;
}

// ----------------------------------------------------------------------------------------
// efts_pack_weight
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                          float* __restrict__ wout, char* __restrict__ plane,
                                                          long ldb, int cout, int cin, int taps, int kp, int split) {
    __shared__ float sh[4];
    const int co = blockIdx.x;
    const float* wr = w + (long)co * cin * taps;
    float scale = 1.f;
    if (g) {
        float ss = 0.f;
        for (int i = threadIdx.x; i < cin * taps; i += 256) ss += wr[i] * wr[i];
        ss = block_sum256(ss, sh);
        scale = g[co] / sqrtf(ss);
    }
    if (wout)
        for (int i = threadIdx.x; i < cin * taps; i += 256) wout[(long)co * cin * taps + i] = wr[i] * scale;
    for (int q = threadIdx.x; q < (kp >> 2) * taps; q += 256) {
        const int k = q / (kp >> 2), c4 = (q - k * (kp >> 2)) << 2;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (c4 + u < cin) ? wr[(long)(c4 + u) * taps + k] * scale : 0.f;
        plane_store4(plane + ((long)k * cout + co) * ldb, c4, v[0], v[1], v[2], v[3], split);
    }
}

// ----------------------------------------------------------------------------------------
// row-space producers
// ----------------------------------------------------------------------------------------
__global__ void row_masks_kernel(const int* __restrict__ len, float* gap, float* lm, int B, int T, int Tp) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B * Tp) return;
    const int b = row / Tp, t = row - b * Tp;
    if (gap) gap[row] = t < T ? 1.f : 0.f;
    if (lm) lm[row] = (t < T && t < len[b]) ? 1.f : 0.f;
}

// both row spaces of a teacher-forced pass in one launch, straight from the caller's length tensors (int64 as torch makes them, or
// int32): their int32 copies (what every later launch reads), gap masks and length masks
__global__ void row_masks_pair_kernel(const void* __restrict__ l1, const void* __restrict__ l2, int is64, int* __restrict__ o1, int* __restrict__ o2,
                                      float* gap1, float* lm1, float* gap2, float* lm2, int B, int T1, int Tp1, int T2, int Tp2) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int n1 = B * Tp1, n2 = B * Tp2;
    auto len_of = [&](const void* l, int b) { return is64 ? (int)((const long*)l)[b] : ((const int*)l)[b]; };
    if (row < B) { o1[row] = len_of(l1, row); o2[row] = len_of(l2, row); }
    if (row < n1) {
        const int b = row / Tp1, t = row - b * Tp1;
        if (gap1) gap1[row] = t < T1 ? 1.f : 0.f;
        if (lm1) lm1[row] = (t < T1 && t < len_of(l1, b)) ? 1.f : 0.f;
    }
    if (row < n2) {
        const int b = row / Tp2, t = row - b * Tp2;
        if (gap2) gap2[row] = t < T2 ? 1.f : 0.f;
        if (lm2) lm2[row] = (t < T2 && t < len_of(l2, b)) ? 1.f : 0.f;
    }
}

__global__ void embed_kernel(const long* __restrict__ ids, const float* __restrict__ table, float* __restrict__ f32o,
                             char* __restrict__ plane, long ldp, int T, int Tp, int c, int nsym, int split) {
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    const int c4 = threadIdx.x << 2;
    if (c4 >= c) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T) {
        long id = ids[(long)b * T + t];
        id = id < 0 ? 0 : (id >= nsym ? nsym - 1 : id);
        v = *(const float4*)(table + id * c + c4);
    }
    if (f32o) *(float4*)(f32o + (long)row * c + c4) = v;
    if (plane) plane_store4(plane + (long)row * ldp, c4, v.x, v.y, v.z, v.w, split);
}

// Embedding + the FIRST residual convolution of the text encoder as table look-ups.  The input of that layer takes only
// num_symbols distinct values per position, and the convolution is linear, so tap k's contribution W_k . emb[v] can be tabulated
// once per weight set for every symbol v (tap_table [taps][V][c], built by the caller with `taps` one-tap efts_gemm launches on the
// embedding rows); the layer is then
//   y[b, t] = emb[id_t] + LeakyReLU( bias + sum_k tap_table[k][ id_{t + k - pad} ] )     (positions outside [0, lim) contribute 0)
// -- 21 GFLOP of MFMA work at 64 x 128 tokens become 8 320 x (taps + 1) row gathers from a 0.9 MB table.
// lim = T (lens == NULL: the teacher-forced forward, where padded ids are real symbols and leak: efficient_tts.py:144-148) or the
// item's length (batched free-running inference: zero embedding and zero output beyond it).
__global__ void embed_conv_kernel(const long* __restrict__ ids, const int* __restrict__ lens, const float* __restrict__ table,
                                  const float* __restrict__ taptab, const float* __restrict__ bias, float slope, float* __restrict__ f32o,
                                  char* __restrict__ plane, long ldp, int T, int Tp, int c, int nsym, int taps, int split) {
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    const int c4 = threadIdx.x << 2;
    if (c4 >= c) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int lim = lens ? min(lens[b], T) : T;
    if (t < lim) {
        const long* idr = ids + (long)b * T;
        auto sym = [&](int tt) { long id = idr[tt]; return (int)(id < 0 ? 0 : (id >= nsym ? nsym - 1 : id)); };
        // (bias and the embedding table are parameters: inside a fused optimizer's flat buffer they are only 4-byte aligned)
        float4 acc = bias ? make_float4(bias[c4], bias[c4 + 1], bias[c4 + 2], bias[c4 + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int pad = (taps - 1) >> 1;
        for (int k = 0; k < taps; ++k) {
            const int tt = t + k - pad;
            if (tt < 0 || tt >= lim) continue;
            const float4 p = *(const float4*)(taptab + ((long)k * nsym + sym(tt)) * c + c4);
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        const float* er = table + (long)sym(t) * c + c4;
        const float4 e = make_float4(er[0], er[1], er[2], er[3]);
        v.x = e.x + (acc.x > 0.f ? acc.x : acc.x * slope);
        v.y = e.y + (acc.y > 0.f ? acc.y : acc.y * slope);
        v.z = e.z + (acc.z > 0.f ? acc.z : acc.z * slope);
        v.w = e.w + (acc.w > 0.f ? acc.w : acc.w * slope);
    }
    if (f32o) *(float4*)(f32o + (long)row * c + c4) = v;
    if (plane) plane_store4(plane + (long)row * ldp, c4, v.x, v.y, v.z, v.w, split);
}

__global__ void pack_rows_kernel(const float* __restrict__ x, float* __restrict__ f32o, char* __restrict__ plane,
                                 long ldp, int T, int Tp, int c, int kp, int split) {
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    const int c4 = threadIdx.x << 2;
    if (c4 >= kp) return;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < T) {
        const float* xr = x + ((long)b * T + t) * c;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c4 + u < c) v[u] = xr[c4 + u];
    }
    if (f32o)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c4 + u < c) f32o[(long)row * c + c4 + u] = v[u];
    if (plane) plane_store4(plane + (long)row * ldp, c4, v[0], v[1], v[2], v[3], split);
}

// ----------------------------------------------------------------------------------------
// alignment block
// ----------------------------------------------------------------------------------------
// one wave per (b, j): softmax over the valid keys + expected key index
__global__ __launch_bounds__(256) void attn_soft_index_kernel(const float* __restrict__ sc, long ld,
                                                              const int* __restrict__ tlen, const int* __restrict__ mlen,
                                                              float* __restrict__ sidx, float* __restrict__ alpha,
                                                              int B, int T1, int T2) {
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
    float se = 0.f, si = 0.f;
    for (int i = lane; i < tl; i += 64) {
        const float e = __expf(row[i] - mx);
        se += e;
        si += e * (float)i;
    }
    se = wave_sum(se);
    si = wave_sum(si);
    if (lane == 0) sidx[r] = live ? si / se : 0.f;
    if (alpha) {
        const float inv = live ? 1.f / se : 0.f;
        for (int i = lane; i < T1; i += 64)
            alpha[((long)b * T1 + i) * T2 + j] = (i < tl) ? __expf(row[i] - mx) * inv : 0.f;
    }
}

// One wavefront per item: relu-diff, prefix scan over T2, mask, max, normalise
// (imv_generator, efficient_tts.py:314-323).  Lane l owns the contiguous segment
// [l*chunk, (l+1)*chunk): serial sums inside the lane, a 64-lane shuffle scan of the lane totals
// for the carry.  The parallel association differs from a serial cumsum by rounding only; a
// prefix-max (exact in any association) restores the monotonicity a serial cumsum of
// non-negative terms guarantees.
__global__ __launch_bounds__(64) void imv_scan_kernel(const float* __restrict__ sidx, const int* __restrict__ tlen,
                                                      const int* __restrict__ mlen, float* __restrict__ imv, int T2) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* s = sidx + (long)b * T2;
    float* o = imv + (long)b * T2;
    const int ml = min(mlen[b], T2);
    const int chunk = (T2 + 63) / 64;
    const int j0 = lane * chunk, j1 = min(j0 + chunk, T2);
    float loc = 0.f;
    for (int j = j0; j < j1; ++j) loc += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f;
    const float incl = wave_scan_incl(loc);
    const float excl = incl - loc;                // exclusive carry of this lane
    float fin = excl;                             // the value this lane will actually write last
    for (int j = j0; j < j1; ++j) fin += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f;
    // prefix-max carry: everything in this lane must be >= what the previous lanes wrote
    float carry = __shfl_up(fin, 1);
    if (lane == 0) carry = 0.f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(carry, off);
        if (lane >= off) carry = fmaxf(carry, t);
    }
    float run = excl, last = carry;
    for (int j = j0; j < j1; ++j) {
        run += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f;
        last = fmaxf(last, run);
        o[j] = j < ml ? last : 0.f;
    }
    // max over the masked prefix = value at the last valid frame (monotone), taken by reduction
    float mx = 0.f;
    for (int j = j0; j < j1; ++j) mx = fmaxf(mx, o[j]);
    mx = fmaxf(wave_max(mx), 1e-8f);
    const float scale = (float)tlen[b] - 1.f;
    for (int j = j0; j < j1; ++j) o[j] = o[j] / mx * scale;
}

// one wave per (b, i): e_i = sum_j softmax_j(-sigma_e (pi_j - i)^2) * j over valid frames
__global__ __launch_bounds__(256) void aligned_pos_kernel(const float* __restrict__ imv, const int* __restrict__ tlen,
                                                          const int* __restrict__ mlen, float sigma_e,
                                                          float* __restrict__ e, int T1, int T2) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= T1) return;
    const int tl = tlen[b], ml = min(mlen[b], T2);
    float out = 0.f;
    if (i < tl && ml > 0) {
        const float* pi = imv + (long)b * T2;
        const float p = (float)i;
        float mx = -INFINITY;
        for (int j = lane; j < ml; j += 64) {
            const float d = pi[j] - p;
            mx = fmaxf(mx, -sigma_e * d * d);
        }
        mx = wave_max(mx);
        float se = 0.f, sj = 0.f;
        for (int j = lane; j < ml; j += 64) {
            const float d = pi[j] - p;
            const float w = __expf(-sigma_e * d * d - mx);
            se += w;
            sj += w * (float)j;
        }
        se = wave_sum(se);
        sj = wave_sum(sj);
        out = sj / se;
    }
    if (lane == 0) e[(long)b * T1 + i] = out;
}

__global__ void dur_target_kernel(const float* __restrict__ e, const int* __restrict__ tlen, const int* __restrict__ mlen, float offset,
                                  int method1, float* __restrict__ lde, int B, int T1) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T1) return;
    const int b = idx / T1, i = idx - b * T1;
    float v = 0.f;
    if (i < tlen[b]) {
        // method 1 (efficient_tts.py:204): e_i - e_{i-1}, e_{-1} = 0.  Otherwise (:205-213): e_{i+1} - e_i, e_{len} = mel length
        const float d = method1 ? e[idx] - (i > 0 ? e[idx - 1] : 0.f)
                                : (i + 1 < tlen[b] ? e[idx + 1] : (float)mlen[b]) - e[idx];
        v = logf(d + offset);
    }
    lde[idx] = v;
}

// softmax over the T1 keys of -sigma (q_j - e_i)^2 for 64 columns j per block; the keys are split over the block's 4
// waves (max and sum combined through LDS), the split-2 plane rows leave through a [64 j][32 i] LDS tile so that every
// 128-byte K chunk of a row is written by 4 adjacent threads (one thread per column with three T1-long dependent loops
// and 8-byte stores scattered over 64 rows: 35 us at B=64)
__global__ __launch_bounds__(256) void reconst_alpha_kernel(const float* __restrict__ e, const int* __restrict__ tlen,
                                                            const int* __restrict__ mlen, float sigma,
                                                            float* __restrict__ alpha, char* __restrict__ plane,
                                                            long ldp, int T1, int T2, int T2p) {
    extern __shared__ float es[];   // e[b, :]
    __shared__ float red[4][64];
    __shared__ float tile[64][33];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < T1; i += 256) es[i] = e[(long)b * T1 + i];
    __syncthreads();
    const int jj = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jj;
    const int tl = tlen ? min(tlen[b], T1) : T1;
    const bool live = (j < T2) && (mlen ? (j < mlen[b]) : true);
    const float q = live ? (float)j : 0.f;
    float mx = -INFINITY;
    for (int i = w; i < tl; i += 4) {
        const float d = q - es[i];
        mx = fmaxf(mx, -sigma * d * d);
    }
    red[w][jj] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][jj], red[1][jj]), fmaxf(red[2][jj], red[3][jj]));
    __syncthreads();
    float se = 0.f;
    for (int i = w; i < tl; i += 4) {
        const float d = q - es[i];
        se += __expf(-sigma * d * d - mx);
    }
    red[w][jj] = se;
    __syncthreads();
    se = red[0][jj] + red[1][jj] + red[2][jj] + red[3][jj];
    const float inv = (live && tl > 0) ? 1.f / se : 0.f;
    const int kp = (T1 + 31) & ~31;
    const int prow_l = threadIdx.x >> 2, pq = threadIdx.x & 3;          // plane pass: row (column j) and 8-key piece
    const int pj = blockIdx.x * 64 + prow_l;
    for (int i0 = 0; i0 < kp; i0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + w * 8 + u;
            float a = 0.f;
            if (i < tl) {
                const float d = q - es[i];
                a = __expf(-sigma * d * d - mx) * inv;
            }
            if (alpha && i < T1 && j < T2) alpha[((long)b * T1 + i) * T2 + j] = a;
            tile[jj][w * 8 + u] = a;
        }
        if (plane) {
            __syncthreads();
            if (pj < T2) {
                char* prow = plane + ((long)b * T2p + pj) * ldp;
                const float* t = &tile[prow_l][pq * 8];
                plane_store4(prow, i0 + pq * 8, t[0], t[1], t[2], t[3], 2);
                plane_store4(prow, i0 + pq * 8 + 4, t[4], t[5], t[6], t[7], 2);
            }
            __syncthreads();
        }
    }
}

// V [B*T1p][c] fp32 -> V^T split-2 planes [B][c][K = i]; one block = 32 i x 32 c
__global__ __launch_bounds__(256) void pack_vt_kernel(const float* __restrict__ v, long ldv, char* __restrict__ plane,
                                                      long ldp, int T1, int T1p, int c) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r;
        tile[r][tx] = (i < T1 && c0 + tx < c) ? v[((long)b * T1p + i) * ldv + c0 + tx] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        if (c0 + r >= c) continue;
        const float x = tile[tx][r];
        const unsigned short hi = f32_to_bf16(x);
        const unsigned short lo = f32_to_bf16(x - bf16_to_f32(hi));
        char* d = plane + ((long)b * c + c0 + r) * ldp + (long)blockIdx.x * 128 + tx * 2;
        *(unsigned short*)d = hi;
        *(unsigned short*)(d + 64) = lo;
    }
}

// The same for c % 64 == 0 and 16-byte aligned rows: one block = 64 i x 64 c.  Rows come in as 256-byte segments (float4 per thread), and every
// thread stores 16 bytes = eight consecutive K positions of the hi or of the lo half of a 128-byte chunk (the 32 x 32 form above moves 2 bytes
// per store and 128-byte row segments: 1.8 TB/s on the training step's two mel-length transposes, 87-100 us each on its critical chain).
// Bit-identical to pack_vt_kernel (same rounding of x and of its remainder).
__global__ __launch_bounds__(256) void pack_vt64_kernel(const float* __restrict__ v, long ldv, char* __restrict__ plane,
                                                        long ldp, int T1, int T1p, int c, int nchunk) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, i0 = blockIdx.x * 64;
    {
        const int cx = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;                  // 16 threads x float4 per row, 16 rows per pass
#pragma unroll
        for (int r = ry; r < 64; r += 16) {
            const int i = i0 + r;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < T1) x = *(const float4*)(v + ((long)b * T1p + i) * ldv + c0 + cx);
            tile[r][cx] = x.x; tile[r][cx + 1] = x.y; tile[r][cx + 2] = x.z; tile[r][cx + 3] = x.w;
        }
    }
    __syncthreads();
    {
        // plane row (b, c0 + r): 256 bytes = chunks 2 bx, 2 bx + 1; thread slot s (0 .. 15): chunk s / 8, half (s / 4) & 1 (0 hi, 1 lo), positions 8 (s & 3) ..
        const int s = threadIdx.x & 15, ry = threadIdx.x >> 4;
        const int ch = s >> 3, lo = (s >> 2) & 1, k0 = 32 * ch + 8 * (s & 3);
        if (2 * (int)blockIdx.x + ch < nchunk) {
#pragma unroll
            for (int r = ry; r < 64; r += 16) {
                unsigned w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float x0 = tile[k0 + 2 * u][r], x1 = tile[k0 + 2 * u + 1][r];
                    const unsigned h = cvt_pk_bf16(x0, x1);
                    w[u] = lo ? cvt_pk_bf16(x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u)) : h;
                }
                *(uint4*)(plane + ((long)b * c + c0 + r) * ldp + (long)(2 * blockIdx.x + ch) * 128 + lo * 64 + (s & 3) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cumsum_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int T) {
    __shared__ float sh[4];
    const float* xr = x + (long)blockIdx.x * T;
    float* yr = y + (long)blockIdx.x * T;
    const int chunk = (T + 255) / 256;
    const int j0 = threadIdx.x * chunk, j1 = min(j0 + chunk, T);
    float loc = 0.f;
    for (int j = j0; j < j1; ++j) loc += xr[j];
    float total;
    float run = block_scan_excl256(loc, sh, &total);
    for (int j = j0; j < j1; ++j) {
        run += xr[j];
        yr[j] = run;
    }
}

// ----------------------------------------------------------------------------------------
// LayerNorm over channels, one wave per row (c <= 2048, c % 4 == 0)
// ----------------------------------------------------------------------------------------
template <bool DOT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ rowmask, float* __restrict__ f32o,
                                                        char* __restrict__ plane, long ldp, int rows, int c, int split,
                                                        const float* __restrict__ w, const float* __restrict__ bptr,
                                                        int mode, float offset, float* __restrict__ out, float drop_p,
                                                        unsigned drop_seed, const unsigned* __restrict__ seed_add) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * c;
    float4 v[8];
    const int nv = c >> 8;   // float4 per lane (c = 512 -> 2)
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nv) {
            v[u] = *(const float4*)(xr + u * 256 + lane * 4);
            s += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nv) {
            v[u].x -= mean; v[u].y -= mean; v[u].z -= mean; v[u].w -= mean;
            q += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
        }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)c + eps);
    const float rm = rowmask ? rowmask[row] : 1.f;
    const bool drop = drop_p > 0.f;
    // (seed_add: a device word added to the seed -- the step counter of a training step replayed as a hipGraph, whose kernel
    //  arguments are frozen at capture time)
    const unsigned thresh = drop ? (unsigned)(drop_p * 4294967296.0) : 0u, seed_h = hash_u32(drop_seed + (seed_add ? *seed_add : 0u));
    const float inv_keep = drop ? 1.f / (1.f - drop_p) : 1.f;
    float dot = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nv) {
            const int c4 = u * 256 + lane * 4;
            const float4 g = *(const float4*)(gamma + c4), bb = *(const float4*)(beta + c4);
            float4 y;
            y.x = v[u].x * rstd * g.x + bb.x; y.y = v[u].y * rstd * g.y + bb.y;
            y.z = v[u].z * rstd * g.z + bb.z; y.w = v[u].w * rstd * g.w + bb.w;
            if (drop) {   // Dropout after LayerNorm (duration_predictor.py:61), train mode only
                const unsigned e0 = (unsigned)row * (unsigned)c + c4;
                y.x *= drop_scale(seed_h, e0, thresh, inv_keep); y.y *= drop_scale(seed_h, e0 + 1, thresh, inv_keep);
                y.z *= drop_scale(seed_h, e0 + 2, thresh, inv_keep); y.w *= drop_scale(seed_h, e0 + 3, thresh, inv_keep);
            }
            if constexpr (DOT) {
                const float4 ww = *(const float4*)(w + c4);
                dot += y.x * ww.x + y.y * ww.y + y.z * ww.z + y.w * ww.w;
            } else {
                y.x *= rm; y.y *= rm; y.z *= rm; y.w *= rm;
                if (f32o) *(float4*)(f32o + (long)row * c + c4) = y;
                if (plane) plane_store4(plane + (long)row * ldp, c4, y.x, y.y, y.z, y.w, split);
            }
        }
    if constexpr (DOT) {
        dot = wave_sum(dot) + bptr[0];
        if (mode == 1) dot = fmaxf(expf(dot) - offset, 0.f);
        if (lane == 0) out[row] = dot * rm;
    }
}

// ----------------------------------------------------------------------------------------
// masked losses: stage 1 partial sums per block, stage 2 one block finalises (deterministic)
// ----------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 512;

__global__ __launch_bounds__(256) void losses_stage1(const float* __restrict__ mp, long ldm, const float* __restrict__ sp,
                                                     const int* __restrict__ mlen, const float* __restrict__ dp,
                                                     const float* __restrict__ lde, const int* __restrict__ tlen,
                                                     float* __restrict__ part, int B, int T1, int T1p, int T2, int T2p,
                                                     int odim) {
    __shared__ float sh[4];
    const long nmel = (long)B * T2 * odim;
    float sm = 0.f;
    if ((odim & 3) == 0 && (ldm & 3) == 0 && nmel < 0x7fffffffL && (((uintptr_t)mp | (uintptr_t)sp) & 15) == 0) {
        // float4 path, 32-bit index math (the 64-bit divisions of the scalar path made this kernel ALU-bound)
        const unsigned q = (unsigned)odim >> 2, nv = (unsigned)(nmel >> 2);
        for (unsigned v = blockIdx.x * 256 + threadIdx.x; v < nv; v += LOSS_BLOCKS * 256) {
            const unsigned fr = v / q, o4 = (v - fr * q) << 2;
            const unsigned b = fr / (unsigned)T2, j = fr - b * (unsigned)T2;
            if ((int)j < mlen[b]) {
                const float4 a = *(const float4*)(mp + ((long)b * T2p + j) * ldm + o4);
                const float4 t = *(const float4*)(sp + (long)fr * odim + o4);
                const float d0 = a.x - t.x, d1 = a.y - t.y, d2 = a.z - t.z, d3 = a.w - t.w;
                sm += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
    } else {
        for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nmel; idx += (long)LOSS_BLOCKS * 256) {
            const long fr = idx / odim;
            const int o = (int)(idx - fr * odim);
            const int b = (int)(fr / T2), j = (int)(fr - (long)b * T2);
            if (j < mlen[b]) {
                const float d = mp[((long)b * T2p + j) * ldm + o] - sp[idx];
                sm += d * d;
            }
        }
    }
    float sd = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)B * T1; idx += (long)LOSS_BLOCKS * 256) {
        const int b = (int)(idx / T1), i = (int)(idx - (long)b * T1);
        if (i < tlen[b]) sd += fabsf(dp[(long)b * T1p + i] - lde[idx]);
    }
    sm = block_sum256(sm, sh);
    sd = block_sum256(sd, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = sm;
        part[blockIdx.x * 2 + 1] = sd;
    }
}

__global__ __launch_bounds__(256) void losses_stage2(const float* __restrict__ part, const int* __restrict__ mlen,
                                                     const int* __restrict__ tlen, float* __restrict__ out3, int B,
                                                     int T1, int T2, int odim) {
    __shared__ float sh[4];
    float sm = 0.f, sd = 0.f, nm = 0.f, nt = 0.f;
    for (int i = threadIdx.x; i < LOSS_BLOCKS; i += 256) {
        sm += part[i * 2];
        sd += part[i * 2 + 1];
    }
    for (int b = threadIdx.x; b < B; b += 256) {
        nm += (float)min(mlen[b], T2);
        nt += (float)min(tlen[b], T1);
    }
    sm = block_sum256(sm, sh);
    sd = block_sum256(sd, sh);
    nm = block_sum256(nm, sh);
    nt = block_sum256(nt, sh);
    if (threadIdx.x == 0) {
        const float ml = sm / (nm * (float)odim), dl = sd / nt;
        out3[0] = ml + dl;
        out3[1] = ml;
        out3[2] = dl;
    }
}

// the losses from the squared-error partial sums of the mel head's launch (efts_gemm_args.sqerr_part): one workgroup of 1024 threads, fixed
// order.  The duration term's loads are all issued before the first is used (a 256-thread loop of dependent loads took 32 memory latencies:
// the fused form was 15 us SLOWER than the two loss launches until this was fixed)
__global__ __launch_bounds__(1024) void losses_from_parts(const float* __restrict__ part, int n_part, const int* __restrict__ mlen,
                                                          const float* __restrict__ dp, const float* __restrict__ lde,
                                                          const int* __restrict__ tlen, float* __restrict__ out3, int B, int T1, int T1p,
                                                          int T2, int odim) {
    __shared__ float sh[4][16];
    const int tid = threadIdx.x;
    float sm = 0.f, sd = 0.f, nm = 0.f, nt = 0.f;
    constexpr int U = 8;
    const int nd = B * T1;
    for (int base = 0; base < nd; base += 1024 * U) {
        float a[U], l[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 1024 + tid;
            const int b = idx < nd ? idx / T1 : 0, i = idx - b * T1;
            ok[u] = idx < nd && i < tlen[b];
            a[u] = ok[u] ? dp[(long)b * T1p + i] : 0.f;
            l[u] = ok[u] ? lde[idx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) sd += fabsf(a[u] - l[u]);
    }
    for (int i = tid; i < n_part; i += 1024) sm += part[i];
    for (int b = tid; b < B; b += 1024) {
        nm += (float)min(mlen[b], T2);
        nt += (float)min(tlen[b], T1);
    }
    sm = wave_sum(sm); sd = wave_sum(sd); nm = wave_sum(nm); nt = wave_sum(nt);
    if ((tid & 63) == 0) { sh[0][tid >> 6] = sm; sh[1][tid >> 6] = sd; sh[2][tid >> 6] = nm; sh[3][tid >> 6] = nt; }
    __syncthreads();
    if (tid == 0) {
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) acc += sh[q][w];
            t[q] = acc;
        }
        const float ml = t[0] / (t[2] * (float)odim), dl = t[1] / t[3];
        out3[0] = ml + dl;
        out3[1] = ml;
        out3[2] = dl;
    }
}

}  // namespace efts

using namespace efts;

#define ST ((hipStream_t)stream)

extern "C" int efts_pack_weight(const float* w, const float* g, float* w_f32_out, void* plane, int64_t ldb,
                                int32_t cout, int32_t cin, int32_t taps, int32_t split, void* stream) {
    if (!w || !plane) return efts_fail(EFTS_EINVAL, "efts_pack_weight: null pointer");
    if (!(split == 1 || split == 2) || cout <= 0 || cin <= 0 || taps <= 0) return efts_fail(EFTS_ESHAPE, "efts_pack_weight: bad shape/split");
    const int kp = split == 1 ? (cin + 63) & ~63 : (cin + 31) & ~31;
    if (ldb < (split == 1 ? kp * 2 : kp * 4) || (ldb & 15)) return efts_fail(EFTS_EALIGN, "efts_pack_weight: ldb too small or not 16-byte aligned");
    hipLaunchKernelGGL(pack_weight_kernel, dim3(cout), dim3(256), 0, ST, w, g, w_f32_out, (char*)plane, (long)ldb, cout, cin, taps, kp, split);
    return efts_check_launch("efts_pack_weight");
}

extern "C" int efts_row_masks(const int32_t* lengths, float* gapmask, float* lenmask, int32_t B, int32_t T, int32_t Tp, void* stream) {
    if (!lengths || B <= 0 || T <= 0 || Tp < T) return efts_fail(EFTS_EINVAL, "efts_row_masks: bad arguments");
    hipLaunchKernelGGL(row_masks_kernel, dim3((B * Tp + 255) / 256), dim3(256), 0, ST, lengths, gapmask, lenmask, B, T, Tp);
    return efts_check_launch("efts_row_masks");
}

extern "C" int efts_row_masks_pair(const void* len1, const void* len2, int32_t is_int64, int32_t* len1_i32, int32_t* len2_i32, float* gap1,
                                   float* lenmask1, float* gap2, float* lenmask2, int32_t B, int32_t T1, int32_t Tp1, int32_t T2, int32_t Tp2, void* stream) {
    if (!len1 || !len2 || !len1_i32 || !len2_i32 || B <= 0 || T1 <= 0 || T2 <= 0 || Tp1 < T1 || Tp2 < T2)
        return efts_fail(EFTS_EINVAL, "efts_row_masks_pair: bad arguments");
    const int n = B * (Tp1 > Tp2 ? Tp1 : Tp2);
    hipLaunchKernelGGL(row_masks_pair_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, len1, len2, is_int64 ? 1 : 0, len1_i32, len2_i32, gap1, lenmask1, gap2,
                       lenmask2, B, T1, Tp1, T2, Tp2);
    return efts_check_launch("efts_row_masks_pair");
}

extern "C" int efts_embed(const int64_t* ids, const float* table, float* f32_out, void* plane, int64_t ld_plane, int32_t B,
                          int32_t T, int32_t Tp, int32_t c, int32_t num_symbols, int32_t split, void* stream) {
    if (!ids || !table || (!f32_out && !plane)) return efts_fail(EFTS_EINVAL, "efts_embed: null pointer");
    if (c % 4 || c > 4096 || B <= 0 || T <= 0 || Tp < T) return efts_fail(EFTS_ESHAPE, "efts_embed: c must be a multiple of 4 (<= 4096)");
    hipLaunchKernelGGL(embed_kernel, dim3(B * Tp), dim3(((c / 4) + 63) & ~63), 0, ST, (const long*)ids, table, f32_out, (char*)plane,
                       (long)ld_plane, T, Tp, c, num_symbols, split);
    return efts_check_launch("efts_embed");
}

extern "C" int efts_embed_conv(const int64_t* ids, const int32_t* lengths, const float* table, const float* tap_table, const float* bias, float slope,
                               float* f32_out, void* plane, int64_t ld_plane, int32_t B, int32_t T, int32_t Tp, int32_t c, int32_t num_symbols,
                               int32_t taps, int32_t split, void* stream) {
    if (!ids || !table || !tap_table || (!f32_out && !plane)) return efts_fail(EFTS_EINVAL, "efts_embed_conv: null pointer");
    if (c % 4 || c > 4096 || B <= 0 || T <= 0 || Tp < T || num_symbols <= 0 || !(taps == 1 || taps == 3 || taps == 5))
        return efts_fail(EFTS_ESHAPE, "efts_embed_conv: c must be a multiple of 4 (<= 4096), taps 1 / 3 / 5");
    if (((uintptr_t)tap_table | (uintptr_t)f32_out) & 15) return efts_fail(EFTS_EALIGN, "efts_embed_conv: 16-byte aligned tap table and output");
    hipLaunchKernelGGL(embed_conv_kernel, dim3(B * Tp), dim3(((c / 4) + 63) & ~63), 0, ST, (const long*)ids, lengths, table, tap_table, bias, slope, f32_out,
                       (char*)plane, (long)ld_plane, T, Tp, c, num_symbols, taps, split);
    return efts_check_launch("efts_embed_conv");
}

extern "C" int efts_pack_rows(const float* x, float* f32_out, void* plane, int64_t ld_plane, int32_t B, int32_t T, int32_t Tp,
                              int32_t c, int32_t kp, int32_t split, void* stream) {
    if (!x || (!f32_out && !plane)) return efts_fail(EFTS_EINVAL, "efts_pack_rows: null pointer");
    if (kp % 4 || kp < c || kp > 4096) return efts_fail(EFTS_ESHAPE, "efts_pack_rows: kp must be a multiple of 4, >= c");
    hipLaunchKernelGGL(pack_rows_kernel, dim3(B * Tp), dim3(((kp / 4) + 63) & ~63), 0, ST, x, f32_out, (char*)plane, (long)ld_plane, T, Tp, c, kp, split);
    return efts_check_launch("efts_pack_rows");
}

extern "C" int efts_attn_soft_index(const float* scores, int64_t ld, const int32_t* text_len, const int32_t* mel_len,
                                    float* soft_idx, float* alpha_out, int32_t B, int32_t T1, int32_t T2, void* stream) {
    if (!scores || !text_len || !mel_len || !soft_idx) return efts_fail(EFTS_EINVAL, "efts_attn_soft_index: null pointer");
    if (ld < T1) return efts_fail(EFTS_ESHAPE, "efts_attn_soft_index: ld < T1");
    const long rows = (long)B * T2;
    hipLaunchKernelGGL(attn_soft_index_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, scores, (long)ld, text_len, mel_len, soft_idx, alpha_out, B, T1, T2);
    return efts_check_launch("efts_attn_soft_index");
}

extern "C" int efts_imv_scan(const float* soft_idx, const int32_t* text_len, const int32_t* mel_len, float* imv, int32_t B, int32_t T2, void* stream) {
    if (!soft_idx || !text_len || !mel_len || !imv) return efts_fail(EFTS_EINVAL, "efts_imv_scan: null pointer");
    hipLaunchKernelGGL(imv_scan_kernel, dim3(B), dim3(64), 0, ST, soft_idx, text_len, mel_len, imv, T2);
    return efts_check_launch("efts_imv_scan");
}

extern "C" int efts_aligned_positions(const float* imv, const int32_t* text_len, const int32_t* mel_len, float sigma_e, float offset,
                                      float* e, float* log_delta_e, int32_t B, int32_t T1, int32_t T2, void* stream) {
    if (!imv || !text_len || !mel_len || !e) return efts_fail(EFTS_EINVAL, "efts_aligned_positions: null pointer");
    hipLaunchKernelGGL(aligned_pos_kernel, dim3((T1 + 3) / 4, B), dim3(256), 0, ST, imv, text_len, mel_len, sigma_e, e, T1, T2);
    if (log_delta_e)
        hipLaunchKernelGGL(dur_target_kernel, dim3((B * T1 + 255) / 256), dim3(256), 0, ST, (const float*)e, text_len, mel_len, offset, 1, log_delta_e, B, T1);
    return efts_check_launch("efts_aligned_positions");
}

extern "C" int efts_duration_target(const float* e, const int32_t* text_len, const int32_t* mel_len, float offset, int32_t method1,
                                    float* log_delta_e, int32_t B, int32_t T1, void* stream) {
    if (!e || !text_len || !mel_len || !log_delta_e) return efts_fail(EFTS_EINVAL, "efts_duration_target: null pointer");
    hipLaunchKernelGGL(dur_target_kernel, dim3((B * T1 + 255) / 256), dim3(256), 0, ST, e, text_len, mel_len, offset, method1 ? 1 : 0, log_delta_e, B, T1);
    return efts_check_launch("efts_duration_target");
}

extern "C" int efts_reconst_alpha(const float* e, const int32_t* text_len, const int32_t* mel_len, float sigma, float* alpha_out,
                                  void* plane, int64_t ld_plane, int32_t B, int32_t T1, int32_t T2, int32_t T2p, void* stream) {
    if (!e || (!alpha_out && !plane)) return efts_fail(EFTS_EINVAL, "efts_reconst_alpha: null pointer");
    if (T1 <= 0 || T2 <= 0 || T2p < T2 || T1 > 8192) return efts_fail(EFTS_ESHAPE, "efts_reconst_alpha: bad shape");
    if (plane && ld_plane < (int64_t)((T1 + 31) / 32) * 128) return efts_fail(EFTS_ESHAPE, "efts_reconst_alpha: ld_plane too small");
    hipLaunchKernelGGL(reconst_alpha_kernel, dim3((T2 + 63) / 64, B), dim3(256), T1 * sizeof(float), ST, e, text_len, mel_len, sigma,
                       alpha_out, (char*)plane, (long)ld_plane, T1, T2, T2p);
    return efts_check_launch("efts_reconst_alpha");
}

extern "C" int efts_pack_vt(const float* v, int64_t ldv, void* plane, int64_t ld_plane, int32_t B, int32_t T1, int32_t T1p, int32_t c, void* stream) {
    if (!v || !plane) return efts_fail(EFTS_EINVAL, "efts_pack_vt: null pointer");
    if (ld_plane < (int64_t)((T1 + 31) / 32) * 128) return efts_fail(EFTS_ESHAPE, "efts_pack_vt: ld_plane too small");
    if (c % 64 == 0 && (ldv & 3) == 0 && ((uintptr_t)v & 15) == 0 && (ld_plane & 15) == 0 && ((uintptr_t)plane & 15) == 0 && T1 >= 64) {
        hipLaunchKernelGGL(pack_vt64_kernel, dim3((T1 + 63) / 64, c / 64, B), dim3(256), 0, ST, v, (long)ldv, (char*)plane, (long)ld_plane, T1, T1p, c, (T1 + 31) / 32);
        return efts_check_launch("efts_pack_vt");
    }
    hipLaunchKernelGGL(pack_vt_kernel, dim3((T1 + 31) / 32, (c + 31) / 32, B), dim3(256), 0, ST, v, (long)ldv, (char*)plane, (long)ld_plane, T1, T1p, c);
    return efts_check_launch("efts_pack_vt");
}

extern "C" int efts_cumsum_rows(const float* x, float* y, int32_t B, int32_t T, void* stream) {
    if (!x || !y || B <= 0 || T <= 0) return efts_fail(EFTS_EINVAL, "efts_cumsum_rows: bad arguments");
    hipLaunchKernelGGL(cumsum_rows_kernel, dim3(B), dim3(256), 0, ST, x, y, T);
    return efts_check_launch("efts_cumsum_rows");
}

extern "C" int efts_layernorm_rows(const float* x, const float* gamma, const float* beta, float eps, const float* rowmask,
                                   float* f32_out, void* plane, int64_t ld_plane, int32_t rows, int32_t c, int32_t split, float drop_p,
                                   uint32_t drop_seed, const uint32_t* drop_seed_add, void* stream) {
    if (!x || !gamma || !beta || (!f32_out && !plane)) return efts_fail(EFTS_EINVAL, "efts_layernorm_rows: null pointer");
    if (c % 256 || c > 2048 || rows <= 0) return efts_fail(EFTS_ESHAPE, "efts_layernorm_rows: c must be a multiple of 256, <= 2048");
    hipLaunchKernelGGL((layernorm_kernel<false>), dim3((rows + 3) / 4), dim3(256), 0, ST, x, gamma, beta, eps, rowmask, f32_out, (char*)plane,
                       (long)ld_plane, rows, c, split, (const float*)nullptr, (const float*)nullptr, 0, 0.f, (float*)nullptr, drop_p, drop_seed, drop_seed_add);
    return efts_check_launch("efts_layernorm_rows");
}

extern "C" int efts_layernorm_dot(const float* x, const float* gamma, const float* beta, float eps, const float* w, const float* b,
                                  const float* rowmask, int32_t mode, float offset, float* out, int32_t rows, int32_t c, float drop_p,
                                  uint32_t drop_seed, const uint32_t* drop_seed_add, void* stream) {
    if (!x || !gamma || !beta || !w || !b || !out) return efts_fail(EFTS_EINVAL, "efts_layernorm_dot: null pointer");
    if (c % 256 || c > 2048 || rows <= 0) return efts_fail(EFTS_ESHAPE, "efts_layernorm_dot: c must be a multiple of 256, <= 2048");
    hipLaunchKernelGGL((layernorm_kernel<true>), dim3((rows + 3) / 4), dim3(256), 0, ST, x, gamma, beta, eps, rowmask, (float*)nullptr, (char*)nullptr,
                       0L, rows, c, 1, w, b, mode, offset, out, drop_p, drop_seed, drop_seed_add);
    return efts_check_launch("efts_layernorm_dot");
}

extern "C" size_t efts_losses_workspace_bytes(void) { return (size_t)LOSS_BLOCKS * 2 * sizeof(float); }

extern "C" int efts_masked_losses(const float* mel_pred, int64_t ldm, const float* speech, const int32_t* mel_len, const float* dur_pred,
                                  const float* log_delta_e, const int32_t* text_len, float* out3, void* workspace, int32_t B, int32_t T1,
                                  int32_t T1p, int32_t T2, int32_t T2p, int32_t odim, void* stream) {
    if (!mel_pred || !speech || !mel_len || !dur_pred || !log_delta_e || !text_len || !out3 || !workspace)
        return efts_fail(EFTS_EINVAL, "efts_masked_losses: null pointer");
    hipLaunchKernelGGL(losses_stage1, dim3(LOSS_BLOCKS), dim3(256), 0, ST, mel_pred, (long)ldm, speech, mel_len, dur_pred, log_delta_e, text_len,
                       (float*)workspace, B, T1, T1p, T2, T2p, odim);
    hipLaunchKernelGGL(losses_stage2, dim3(1), dim3(256), 0, ST, (const float*)workspace, mel_len, text_len, out3, B, T1, T2, odim);
    return efts_check_launch("efts_masked_losses");
}

extern "C" int efts_losses_from_parts(const float* sqerr_part, int32_t n_part, const int32_t* mel_len, const float* dur_pred,
                                      const float* log_delta_e, const int32_t* text_len, float* out3, int32_t B, int32_t T1,
                                      int32_t T1p, int32_t T2, int32_t odim, void* stream) {
    if (!sqerr_part || !mel_len || !dur_pred || !log_delta_e || !text_len || !out3)
        return efts_fail(EFTS_EINVAL, "efts_losses_from_parts: null pointer");
    if (n_part <= 0 || B <= 0 || T1 <= 0 || T2 <= 0 || odim <= 0 || (long)B * T1 > 0x7fffffffL)
        return efts_fail(EFTS_ESHAPE, "efts_losses_from_parts: bad shape");
    hipLaunchKernelGGL(losses_from_parts, dim3(1), dim3(1024), 0, ST, sqerr_part, n_part, mel_len, dur_pred, log_delta_e, text_len, out3, B, T1, T1p, T2, odim);
    return efts_check_launch("efts_losses_from_parts");
}
