// efts_align.hip -- the alignment block between the two conv stacks of EFTS-CNN on gfx950, as two launches:
//
//   efts_imv_align : soft index -> IMV (ReLU-diff, prefix scan over the mel frames, normalise), aligned positions e and the
//                    duration target, one workgroup per item, everything after the first load served from LDS
//                    (imv_generator nntts/models/efficient_tts.py:314-323, get_aligned_positions :326-345, duration target :203-216)
//   efts_expand    : alpha' = softmax_i(-sigma (q_j - e_i)^2) generated IN REGISTERS as the MFMA A operand and contracted
//                    with the value projection: H[b, j, :] = sum_i alpha'[b, i, j] V[b, i, :]
//                    (reconstruct_align_from_aligned_position :347-375 + mask :186, expand bmm :190-194)
//
// Why: the chain used to be six launches -- imv_scan (one wave per item walking global memory: 21 us),
// aligned_pos, dur_target, reconst_alpha (writes alpha' fp32 AND an alpha'^T operand plane), pack_vt (V^T operand planes) and a
// generic efts_gemm whose K is only T1 = 128 (108 us for 6.7 GFLOP at 64 x 800 frames).  The contraction is HBM-bound on its
// output (4 B per element and channel), so the kernel is built around the store stream: V of one item (its T1 x 128 channel
// slice, hi/lo bf16 MFMA B fragments) stays in LDS for the whole workgroup, every wave produces alpha' rows in registers,
// and the results leave as full 128-byte lines.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "efts_rowsweep.h"

namespace efts {

// ---------------------------------------------------------------------------------------------------------------
// efts_imv_align
// ---------------------------------------------------------------------------------------------------------------
constexpr int IA_THREADS = 1024;

// The scan is the algorithm of imv_scan_kernel (efts_ops.hip) word for word -- one wavefront, lane l owns the contiguous
// segment [l * chunk, (l + 1) * chunk), a 64-lane shuffle scan carries the lane totals, a prefix max restores monotonicity --
// so the two paths agree bit for bit; only the operands come from LDS instead of global memory.
__device__ __forceinline__ void imv_scan_wave(const float* s, float* o, int lane, int T2, int ml, float scale) {
    const int chunk = (T2 + 63) / 64;
    const int j0 = lane * chunk, j1 = min(j0 + chunk, T2);
    float loc = 0.f;
    for (int j = j0; j < j1; ++j) loc += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f;
    const float incl = wave_scan_incl(loc);
    const float excl = incl - loc;
    float fin = excl;
    for (int j = j0; j < j1; ++j) fin += (j > 0) ? fmaxf(s[j] - s[j - 1], 0.f) : 0.f;
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
    float mx = 0.f;
    for (int j = j0; j < j1; ++j) mx = fmaxf(mx, o[j]);
    mx = fmaxf(wave_max(mx), 1e-8f);
    for (int j = j0; j < j1; ++j) o[j] = o[j] / mx * scale;
}

// Workgroup (b, part): every part repeats the scan of item b (a few us on one wave, the rest of the chip is idle anyway), then
// takes T1 / nsplit keys -- plus the one neighbouring key its duration targets need -- two keys per wave at a time (two
// independent dependency chains per frame load).  Per key the arithmetic is aligned_pos_kernel's, so results agree bit for bit.
__global__ __launch_bounds__(IA_THREADS) void imv_align_kernel(const float* __restrict__ sidx, const int* __restrict__ tlen,
                                                               const int* __restrict__ mlen, float sigma_e, float offset, int method1,
                                                               float* __restrict__ imv, float* __restrict__ e, float* __restrict__ lde,
                                                               int T1, int T2, int T2r, int nsplit, int kper) {
    extern __shared__ float sm[];
    float* s = sm;                  // [T2r] soft index
    float* pi = sm + T2r;           // [T2r] IMV
    float* es = sm + 2 * T2r;       // [kper + 1] aligned positions of keys lo .. hi - 1
    const int b = blockIdx.x / nsplit, part = blockIdx.x - b * nsplit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = tlen[b], ml = min(mlen[b], T2);
    const int i0 = part * kper, i1 = min(i0 + kper, T1);
    if (i0 >= T1) return;
    const int lo = (lde && method1 && i0 > 0) ? i0 - 1 : i0;
    const int hi = (lde && !method1 && i1 < T1) ? i1 + 1 : i1;
    for (int j = tid; j < T2; j += IA_THREADS) s[j] = sidx[(long)b * T2 + j];
    __syncthreads();
    if (wave == 0) imv_scan_wave(s, pi, lane, T2, ml, (float)tl - 1.f);
    __syncthreads();
    if (part == 0)
        for (int j = tid; j < T2; j += IA_THREADS) imv[(long)b * T2 + j] = pi[j];
    // e_i = sum_j softmax_j(-sigma_e (pi_j - i)^2) * j over the valid frames
    for (int k = lo + 2 * wave; k < hi; k += 2 * (IA_THREADS / 64)) {
        const bool two = k + 1 < hi;
        const float p0 = (float)k, p1 = (float)(k + 1);
        float o0 = 0.f, o1 = 0.f;
        if (k < tl && ml > 0) {                          // (a padded key next to a valid one is computed and discarded)
            float mx0 = -INFINITY, mx1 = -INFINITY;
            for (int j = lane; j < ml; j += 64) {
                const float pj = pi[j];
                const float d0 = pj - p0, d1 = pj - p1;
                mx0 = fmaxf(mx0, -sigma_e * d0 * d0);
                mx1 = fmaxf(mx1, -sigma_e * d1 * d1);
            }
            mx0 = wave_max(mx0);
            mx1 = wave_max(mx1);
            float se0 = 0.f, sj0 = 0.f, se1 = 0.f, sj1 = 0.f;
            for (int j = lane; j < ml; j += 64) {
                const float pj = pi[j], fj = (float)j;
                const float d0 = pj - p0, d1 = pj - p1;
                const float w0 = __expf(-sigma_e * d0 * d0 - mx0), w1 = __expf(-sigma_e * d1 * d1 - mx1);
                se0 += w0; sj0 += w0 * fj;
                se1 += w1; sj1 += w1 * fj;
            }
            se0 = wave_sum(se0); sj0 = wave_sum(sj0);
            se1 = wave_sum(se1); sj1 = wave_sum(sj1);
            o0 = sj0 / se0;
            o1 = (k + 1 < tl) ? sj1 / se1 : 0.f;
        }
        if (lane == 0) {
            es[k - lo] = o0;
            if (k >= i0 && k < i1) e[(long)b * T1 + k] = o0;
            if (two) {
                es[k + 1 - lo] = o1;
                if (k + 1 >= i0 && k + 1 < i1) e[(long)b * T1 + k + 1] = o1;
            }
        }
    }
    if (!lde) return;
    __syncthreads();
    // duration target (dur_target_kernel): method 1 e_i - e_{i-1} (e_{-1} = 0), else e_{i+1} - e_i with e_{len} = the mel length
    for (int i = i0 + tid; i < i1; i += IA_THREADS) {
        float v = 0.f;
        if (i < tl) {
            const float d = method1 ? es[i - lo] - (i > 0 ? es[i - 1 - lo] : 0.f)
                                    : ((i + 1 < tl && i + 1 < T1) ? es[i + 1 - lo] : (float)mlen[b]) - es[i - lo];
            v = logf(d + offset);
        }
        lde[(long)b * T1 + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// efts_expand
// ---------------------------------------------------------------------------------------------------------------
struct ExArgs {
    const float* e;
    const int* tlen;
    const int* mlen;
    const float* v;
    float* alpha;
    float* y_f32;
    char* y;
    char* y_lo;
    long ldv, ldo, ldy;
    int T1, T1p, T2, T2p, n, y_split;
    int rsplit;            // workgroups per (item, channel slice): the 32-frame blocks are dealt round-robin over rsplit x 8 waves
    float sigma;
};

// One workgroup = one item b x one slice of 32 * NCB channels, 8 waves; wave w takes the 32-frame blocks w, w + 8, ...
// KS = 16-key slices held per frame (T1 <= 16 KS).  LDS: V fragments [KS][NCB][hi, lo][64 lanes][16 B] (the MFMA B operand
// in the order the lanes read it), 8 KiB of wave-private staging per wave, e[b, :].
template <int KS, int NCB>
__global__ __launch_bounds__(512, 2) void expand_kernel(ExArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VF_BYTES = KS * NCB * 2048;
    char* const vf = smem;
    float* const es = (float*)(smem + VF_BYTES + 8 * 8192);
    float* const km = es + 16 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int ncq = p.n / (32 * NCB);
    const int rpart = blockIdx.x % p.rsplit, bc = blockIdx.x / p.rsplit;
    const int b = bc / ncq, cq = bc - b * ncq;
    const int c0 = cq * 32 * NCB;
    const int tl = p.tlen ? min(p.tlen[b], p.T1) : p.T1;
    const int ml = p.mlen ? min(p.mlen[b], p.T2) : p.T2;

    for (int i = tid; i < 16 * KS; i += 512) {
        es[i] = i < p.T1 ? p.e[(long)b * p.T1 + i] : 0.f;
        km[i] = i < tl ? 0.f : -INFINITY;                       // keys beyond the text length leave the softmax (:370-372)
    }
    // V[b, i, c0 + c] -> hi / lo fragments: lane (c & 31) + 32 * ((i >> 3) & 1) of slice i >> 4 holds keys 8 (i >> 3) .. + 7
    for (int idx = tid; idx < 2 * KS * 32 * NCB; idx += 512) {
        const int c = idx % (32 * NCB), ig = idx / (32 * NCB);
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = ig * 8 + u;
            f[u] = i < p.T1 ? p.v[((long)b * p.T1p + i) * p.ldv + c0 + c] : 0.f;
        }
        u32x4 hi, lo;
        split8(f, &hi, &lo);
        char* d = vf + (((ig >> 1) * NCB + (c >> 5)) * 2) * 1024 + ((c & 31) + 32 * (ig & 1)) * 16;
        *(u32x4*)d = hi;
        *(u32x4*)(d + 1024) = lo;
    }
    __syncthreads();

    char* const st = smem + VF_BYTES + wave * 8192;
    const int nrb = (p.T2 + 31) >> 5;
    for (int rb = rpart * 8 + wave; rb < nrb; rb += 8 * p.rsplit) {
        const int j = rb * 32 + lrow;
        const bool live = j < ml;
        const float q = live ? (float)j : 0.f;                 // (:366-367) padded frames sit at position 0, then are zeroed
        // ---- alpha'[:, j] for this lane's 8 KS keys (keys 16 s + 8 lhalf + u); the other half of the keys is in lane ^ 32
        float pr[KS][8];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 e0 = *(const float4*)(es + 16 * s + 8 * lhalf), e1 = *(const float4*)(es + 16 * s + 8 * lhalf + 4);
            const float4 k0 = *(const float4*)(km + 16 * s + 8 * lhalf), k1 = *(const float4*)(km + 16 * s + 8 * lhalf + 4);
            const float ev[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            const float kv[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = q - ev[u];
                pr[s][u] = -p.sigma * d * d + kv[u];
                mx = fmaxf(mx, pr[s][u]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        mx = tl > 0 ? mx : 0.f;                                // no valid key: every term is exp(-inf) = 0, the row is zero
        float se = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pr[s][u] = __expf(pr[s][u] - mx);
                se += pr[s][u];
            }
        se += __shfl_xor(se, 32);
        const float inv = (live && tl > 0) ? 1.f / se : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 8; ++u) pr[s][u] *= inv;
        // the API tensor reconst_alpha[b, i, j]: every channel slice writes the key slices s % ncq == cq (balanced)
        if (p.alpha && j < p.T2) {
            int t1v = p.T1, t2v = p.T2;
            asm volatile("" : "+s"(t1v), "+s"(t2v));           // keeps the 8 KS row pointers out of the loop-invariant SGPR set
            const unsigned loff = (unsigned)(8 * lhalf * t2v + j);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (s % ncq == cq) {
                    float* rp = p.alpha + ((long)b * t1v + 16 * s) * t2v;        // wave-uniform
                    if (16 * s + 16 <= t1v) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) rp[(unsigned)(u * t2v) + loff] = pr[s][u];
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (16 * s + 8 * lhalf + u < t1v) rp[(unsigned)(u * t2v) + loff] = pr[s][u];
                    }
                }
        }
        // ---- H[32 frames, 32 NCB channels] = alpha'^T V in split-bf16: lo*hi + hi*lo + hi*hi per k-slice (efts_gemm's order)
        f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 h4, l4;
            split8(pr[s], &h4, &l4);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, h4), al = __builtin_bit_cast(bf16x8, l4);
            bf16x8 bh[NCB], bl[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const char* src = vf + ((s * NCB + cb) * 2) * 1024 + lane * 16;
                bh[cb] = *(const bf16x8*)src;
                bl[cb] = *(const bf16x8*)(src + 1024);
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[cb], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[cb], acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[cb], acc[cb], 0, 0, 0);
        }
        // ---- epilogue (efts_rowsweep.h): 64 channels at a time through the wave's 8 KiB, whole 128-byte lines per store
        // instruction.  Frames >= mel_len hold exact zeros (alpha' = 0), frames >= T2 are not stored.
        const RowOut o{p.y_f32, p.y, p.y_lo, p.ldo, p.ldy, p.y_split};
#pragma unroll
        for (int hp = 0; hp < NCB / 2; ++hp)
            sweep64<false>(acc[2 * hp], acc[2 * hp + 1], st, lane, o, (long)b * p.T2p + rb * 32, p.T2 - rb * 32, c0 + hp * 64, 0.f, 0.f, 0, 0.f);
    }
}

}  // namespace efts

using namespace efts;

extern "C" int efts_imv_align(const float* soft_idx, const int32_t* text_len, const int32_t* mel_len, float sigma_e, float offset,
                              int32_t method1, float* imv, float* e, float* log_delta_e, int32_t B, int32_t T1, int32_t T2, void* stream) {
    if (!soft_idx || !text_len || !mel_len || !imv || !e) return efts_fail(EFTS_EINVAL, "efts_imv_align: null pointer");
    if (B <= 0 || T1 <= 0 || T2 <= 0) return efts_fail(EFTS_ESHAPE, "efts_imv_align: B, T1, T2 must be positive");
    // parts per item: about two workgroups per CU over the batch, at least 8 keys each
    int nsplit = (2 * efts_num_cus()) / B;
    nsplit = nsplit < 1 ? 1 : nsplit;
    if (nsplit > (T1 + 7) / 8) nsplit = (T1 + 7) / 8;
    const int kper = (T1 + nsplit - 1) / nsplit;
    nsplit = (T1 + kper - 1) / kper;
    const int T2r = (T2 + 3) & ~3;
    const size_t lds = ((size_t)2 * T2r + kper + 1) * sizeof(float);
    if (lds > 160 * 1024) return efts_fail(EFTS_ESHAPE, "efts_imv_align: 2 * T2 floats must fit the 160 KiB of LDS (use efts_imv_scan + efts_aligned_positions)");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)imv_align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(imv_align_kernel, dim3(B * nsplit), dim3(IA_THREADS), lds, (hipStream_t)stream, soft_idx, text_len, mel_len, sigma_e, offset,
                       method1 ? 1 : 0, imv, e, log_delta_e, T1, T2, T2r, nsplit, kper);
    return efts_check_launch("efts_imv_align");
}

// ---------------------------------------------------------------------------------------------------------------
// efts_duration_positions: the glue between the two phases of free-running synthesis in ONE launch -- per item the
// cumulative sum of the predicted durations (the aligned positions e, efficient_tts.py:260), the mel length
// round(e[len - 1]) (:270 / :361, round-half-to-even like torch.round) and, for delta_e_method_1 = False, positions that
// start at 0 (:261-265).  Replaces a slice copy, efts_cumsum_rows, a gather, a round, two casts and a subtraction.  The scan
// is efts_cumsum_rows' (256 threads, contiguous chunks, wave scans combined in wave order): same sums, bit for bit.
// ---------------------------------------------------------------------------------------------------------------
namespace efts {
__global__ __launch_bounds__(256) void dur_positions_kernel(const float* __restrict__ dur, long ld, const int* __restrict__ tlen, float force,
                                                            int method1, float* __restrict__ e, int* __restrict__ mlen, int T) {
    __shared__ float sh[4];
    __shared__ float last;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tl = min(tlen[b], T);
    const float* xr = dur + (long)b * ld;
    float* yr = e + (long)b * T;
    auto x = [&](int j) { return force >= 0.f ? (j < tl ? force : 0.f) : xr[j]; };
    const int chunk = (T + 255) / 256;
    const int j0 = tid * chunk, j1 = min(j0 + chunk, T);
    float loc = 0.f;
    for (int j = j0; j < j1; ++j) loc += x(j);
    const float incl = wave_scan_incl(loc);
    if (lane == 63) sh[w] = incl;
    if (tid == 0) last = 0.f;
    __syncthreads();
    float base = 0.f;
    for (int i = 0; i < w; ++i) base += sh[i];
    float run = base + incl - loc;
    for (int j = j0; j < j1; ++j) {
        const float d = x(j);
        run += d;
        yr[j] = method1 ? run : run - d;
        if (j == tl - 1) last = run;
    }
    __syncthreads();
    if (tid == 0) mlen[b] = (int)rintf(last);
}
}  // namespace efts

extern "C" int efts_duration_positions(const float* dur, int64_t ld, const int32_t* text_len, float force_delta, int32_t method1, float* e,
                                       int32_t* mel_len, int32_t B, int32_t T1, void* stream) {
    if (!dur || !text_len || !e || !mel_len || B <= 0 || T1 <= 0 || ld < T1) return efts_fail(EFTS_EINVAL, "efts_duration_positions: bad arguments");
    hipLaunchKernelGGL(dur_positions_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dur, (long)ld, text_len, force_delta, method1 ? 1 : 0, e, mel_len, T1);
    return efts_check_launch("efts_duration_positions");
}

// ---------------------------------------------------------------------------------------------------------------
// efts_bf16_round: the fp32 -> bf16 rounding every operand-plane producer of this library applies (mode 0; gfx950's
// v_cvt_pk_bf16_f32) next to its integer reference form (mode 1), so that a test can sweep bit patterns through both.
// ---------------------------------------------------------------------------------------------------------------
namespace efts {
__global__ void bf16_round_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n, int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = mode ? f32_to_bf16_sw(x[i]) : f32_to_bf16(x[i]);
}
}  // namespace efts

extern "C" int efts_bf16_round(const float* x, uint16_t* y, int64_t n, int32_t mode, void* stream) {
    if (!x || !y || n <= 0) return efts_fail(EFTS_EINVAL, "efts_bf16_round: bad arguments");
    hipLaunchKernelGGL(bf16_round_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, mode);
    return efts_check_launch("efts_bf16_round");
}

extern "C" int efts_expand(const efts_expand_args* a, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_expand: null args");
    if (!a->e || !a->v || (!a->y && !a->y_f32)) return efts_fail(EFTS_EINVAL, "efts_expand: null operand / no output");
    if (a->B <= 0 || a->T1 <= 0 || a->T2 <= 0 || a->T1p < a->T1 || a->T2p < a->T2 || a->n <= 0 || a->n % 128)
        return efts_fail(EFTS_ESHAPE, "efts_expand: n must be a multiple of 128, T1 <= T1p, T2 <= T2p");
    if (a->T1 > 256) return efts_fail(EFTS_ESHAPE, "efts_expand: T1 <= 256 (longer texts: efts_reconst_alpha + efts_pack_vt + efts_gemm)");
    if (a->y && !(a->y_split == 1 || a->y_split == 2)) return efts_fail(EFTS_EINVAL, "efts_expand: y_split must be 1 or 2");
    if (a->y_split == 2 && a->y_lo) return efts_fail(EFTS_EINVAL, "efts_expand: y_lo is for split-1 output planes");
    if ((a->y && ((a->ldy & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->y_lo & 15))) || (a->y_f32 && ((a->ldo & 3) || ((uintptr_t)a->y_f32 & 15))))
        return efts_fail(EFTS_EALIGN, "efts_expand: output rows must be 16-byte aligned");
    if (a->ldv < a->n) return efts_fail(EFTS_ESHAPE, "efts_expand: ldv < n");
    ExArgs k;
    k.e = a->e; k.tlen = a->text_len; k.mlen = a->mel_len; k.v = a->v; k.alpha = a->alpha_out; k.y_f32 = a->y_f32;
    k.y = (char*)a->y; k.y_lo = (char*)a->y_lo; k.ldv = a->ldv; k.ldo = a->ldo; k.ldy = a->ldy;
    k.T1 = a->T1; k.T1p = a->T1p; k.T2 = a->T2; k.T2p = a->T2p; k.n = a->n; k.y_split = a->y ? a->y_split : 0; k.sigma = a->sigma;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)expand_kernel<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)expand_kernel<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    // small batches: several workgroups per (item, channel slice) share its 32-frame blocks, so that one utterance still covers
    // a few dozen CUs (each re-reads the item's V slice, 64 KiB: nothing next to the launch's latency)
    const int ncq = a->T1 <= 128 ? a->n / 128 : a->n / 64, nrb = (a->T2 + 31) / 32;
    int rs = efts_num_cus() / (a->B * ncq);                   // (one workgroup per CU: 128 KiB of LDS each)
    rs = rs < 1 ? 1 : rs;
    if (rs > (nrb + 7) / 8) rs = (nrb + 7) / 8;
    k.rsplit = rs;
    if (a->T1 <= 128) {
        const size_t lds = 8 * 4 * 2048 + 8 * 8192 + 2 * 16 * 8 * sizeof(float);
        hipLaunchKernelGGL((expand_kernel<8, 4>), dim3(a->B * ncq * rs), dim3(512), lds, (hipStream_t)stream, k);
    } else {
        const size_t lds = 16 * 2 * 2048 + 8 * 8192 + 2 * 16 * 16 * sizeof(float);
        hipLaunchKernelGGL((expand_kernel<16, 2>), dim3(a->B * ncq * rs), dim3(512), lds, (hipStream_t)stream, k);
    }
    return efts_check_launch("efts_expand");
}
