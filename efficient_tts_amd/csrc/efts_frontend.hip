// efts_frontend.hip -- on-device log-mel front-end: the step right
// before the acoustic-model path.  Reference: nntts/datasets/meldataset.py:49-82 (mel_spectrogram)
// and the per-item + collate plumbing of nntts/datasets/taco2_data.py:66-76,122-139.
//
//   audio [B][L] --frame_pack--> windowed frames as bf16x3 operand planes [B*Tp][1024]
//         --efts_gemm (taps 1) x DFT plane [1026][1024]--> re/im spectrum fp32 [B*Tp][1028]
//         --logmel--> log(clamp(mel_basis . sqrt(re^2+im^2+1e-9), 1e-5))  -> [B][T][80] (zero past each length)
//
// The STFT is a dense MFMA product on the same contraction kernel as the model (no rocFFT, no host
// round trip): 2.1 MFLOP per frame, less than one ResConv layer.  frame_pack and logmel are HBM-bound.
//
// Round 6: radix-R decimation in time (efts_frame_pack_dit / efts_logmel_dit).  The dense 1024-point product is 107.6 GFLOP per
// 64 x 800 frames where an FFT needs ~3; splitting the frame into R interleaved sub-sequences x_p[j] = x[R j + p] turns it into R
// real DFTs of N / R points -- ONE batched efts_gemm of R items against the same [N / R][N / R] plane: R times fewer FLOPs -- and
//     X[f] = sum_p W_N^(p f) Y_p[f mod N/R],   Y_p[N/R - g] = conj(Y_p[g])  (real input)
// is R complex multiply-adds per bin in front of the magnitude, in the logmel kernel.  A real M-point DFT has M independent real
// outputs (re[0 .. M/2], im[1 .. M/2 - 1]), so the batched product has exactly N output columns.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

// one block (256 threads) per row (b, t); each thread handles n_fft / 256 consecutive samples (n_fft = 1024 -> 4)
__global__ __launch_bounds__(256) void frame_pack_kernel(const float* __restrict__ audio, long ld_audio, const int* __restrict__ lengths,
                                                         const float* __restrict__ window, char* __restrict__ plane, long ld_plane,
                                                         int T, int Tp, int n_fft, int hop, int split) {
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    const int L = lengths[b];
    const int nfr = L / hop;                 // frames of this item: (L + 2*pad - n_fft) / hop + 1 with pad = (n_fft - hop) / 2
    const int pad = (n_fft - hop) / 2;
    const bool valid = t < T && t < nfr;
    const float* a = audio + (long)b * ld_audio;
    char* dst = plane + (long)row * ld_plane;
    for (int k = threadIdx.x * 4; k < n_fft; k += 1024) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x = 0.f;
            if (valid) {
                int s = t * hop + k + u - pad;       // index into the un-padded signal
                s = s < 0 ? -s : s;                  // reflect (torch 'reflect': no edge repeat), meldataset.py:69
                s = s >= L ? 2 * (L - 1) - s : s;
                x = a[s] * window[k + u];
            }
            v[u] = x;
        }
        plane_store4(dst, k, v[0], v[1], v[2], v[3], split);
    }
}

// the same frames with the samples of a frame de-interleaved: column (k mod R) * (N / R) + k / R holds sample k.  The frame goes through LDS:
// samples are read the way they lie in memory (4 consecutive ones per thread: whole cache lines per wave) and dropped at their de-interleaved
// position (LDS index u * N/R + j for the thread's samples R j' + u: consecutive threads, consecutive words), then every thread takes the 4
// consecutive COLUMNS of its plane store.  (Gathering the columns straight from memory used a quarter of every cache line per wave: 190 us per
// 64 x 800 frames against 100 for the natural order.)
__global__ __launch_bounds__(256) void frame_pack_dit_kernel(const float* __restrict__ audio, long ld_audio, const int* __restrict__ lengths,
                                                             const float* __restrict__ window, char* __restrict__ plane, long ld_plane,
                                                             int T, int Tp, int n_fft, int hop, int split, int radix) {
    extern __shared__ float fr[];            // [n_fft], de-interleaved
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    const int L = lengths[b];
    const int nfr = L / hop;
    const int pad = (n_fft - hop) / 2;
    const bool valid = t < T && t < nfr;
    const float* a = audio + (long)b * ld_audio;
    char* dst = plane + (long)row * ld_plane;
    const int sub = n_fft / radix;
    for (int k = threadIdx.x * 4; k < n_fft; k += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x = 0.f;
            if (valid) {
                int s = t * hop + k + u - pad;
                s = s < 0 ? -s : s;
                s = s >= L ? 2 * (L - 1) - s : s;
                x = a[s] * window[k + u];
            }
            const int kk = k + u;
            fr[(kk % radix) * sub + kk / radix] = x;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x * 4; c < n_fft; c += 1024) {
        const float4 v = *(const float4*)(fr + c);
        plane_store4(dst, c, v.x, v.y, v.z, v.w, split);
    }
}

// one wave per row: magnitudes into LDS, then lane m (and m + 64) accumulates its triangular filter
__global__ __launch_bounds__(64) void logmel_kernel(const float* __restrict__ spec, long ld_spec, const float* __restrict__ basis,
                                                    const int* __restrict__ ranges, const int* __restrict__ frames, float* __restrict__ out,
                                                    int T, int Tp, int n_bins, int n_mels) {
    extern __shared__ float mag[];
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    if (t >= T) return;
    const int lane = threadIdx.x;
    float* o = out + ((long)b * T + t) * n_mels;
    if (t >= frames[b]) {                        // TextMelCollate pads the time axis with zeros AFTER the log (taco2_data.py:122-134)
        for (int m = lane; m < n_mels; m += 64) o[m] = 0.f;
        return;
    }
    const float* re = spec + (long)row * ld_spec;
    const float* im = re + n_bins;
    for (int k = lane; k < n_bins; k += 64) {
        const float r = re[k], i = im[k];
        mag[k] = sqrtf(r * r + i * i + 1e-9f);   // meldataset.py:75
    }
    __syncthreads();
    for (int m = lane; m < n_mels; m += 64) {
        const int lo = ranges[2 * m], hi = ranges[2 * m + 1];
        const float* w = basis + (long)m * n_bins;
        float acc = 0.f;
        for (int k = lo; k < hi; ++k) acc += w[k] * mag[k];
        o[m] = logf(fmaxf(acc, 1e-5f));          // meldataset.py:27-28,78
    }
}

// radix-R form: spec row = R blocks of M = N / R floats, block p = [re Y_p[0 .. M/2] | im Y_p[1 .. M/2 - 1]]; twiddle[p][f] = (cos, sin)(2 pi p f / N)
__global__ __launch_bounds__(64) void logmel_dit_kernel(const float* __restrict__ spec, long ld_spec, const float* __restrict__ basis,
                                                        const int* __restrict__ ranges, const int* __restrict__ frames,
                                                        const float2* __restrict__ twiddle, float* __restrict__ out, int T, int Tp, int n_bins,
                                                        int n_mels, int radix) {
    extern __shared__ float mag[];
    const int row = blockIdx.x;
    const int b = row / Tp, t = row - b * Tp;
    if (t >= T) return;
    const int lane = threadIdx.x;
    float* o = out + ((long)b * T + t) * n_mels;
    if (t >= frames[b]) {
        for (int m = lane; m < n_mels; m += 64) o[m] = 0.f;
        return;
    }
    const int M = 2 * (n_bins - 1) / radix, half = M >> 1;
    const float* y = spec + (long)row * ld_spec;
    for (int f = lane; f < n_bins; f += 64) {
        int g = f % M;
        float sgn = 1.f;
        if (g > half) { g = M - g; sgn = -1.f; }                 // Y[M - g] = conj(Y[g])
        const bool real_only = g == 0 || g == half;
        float xr = 0.f, xi = 0.f;
        for (int p = 0; p < radix; ++p) {
            const float* yp = y + p * M;
            const float yr = yp[g], yi = real_only ? 0.f : sgn * yp[half + g];
            const float2 w = twiddle[p * n_bins + f];            // W^(p f) = cos - i sin
            xr += w.x * yr + w.y * yi;
            xi += w.x * yi - w.y * yr;
        }
        mag[f] = sqrtf(xr * xr + xi * xi + 1e-9f);                // meldataset.py:75
    }
    __syncthreads();
    for (int m = lane; m < n_mels; m += 64) {
        const int lo = ranges[2 * m], hi = ranges[2 * m + 1];
        const float* w = basis + (long)m * n_bins;
        float acc = 0.f;
        for (int k = lo; k < hi; ++k) acc += w[k] * mag[k];
        o[m] = logf(fmaxf(acc, 1e-5f));                          // meldataset.py:27-28,78
    }
}

// The same with the twiddles in REGISTERS and the spectrum row staged through LDS: lane l owns the bins f = l + 64 i (i < NB), loads its
// (R - 1) * NB twiddles once and then works through LM_ROWS rows (the table version above re-read 16 KiB of twiddles per row from L2: 226 us per
// 64 x 800 frames against 150 for the dense spectrum's kernel).  N = 1024 only (NB = 9 bins per lane).
constexpr int LM_WAVES = 4, LM_RPW = 4, LM_ROWS = LM_WAVES * LM_RPW, LM_NB = 9, LM_CB = 2048;
template <int R>
__global__ __launch_bounds__(64 * LM_WAVES) void logmel_dit_fast_kernel(const float* __restrict__ spec, long ld_spec, const float* __restrict__ basis,
                                                                        const int* __restrict__ ranges, const int* __restrict__ frames,
                                                                        const float2* __restrict__ twiddle, float* __restrict__ out, int T, int Tp,
                                                                        int rows_total, int n_mels) {
    constexpr int N = 1024, NBINS = 513, M = N / R, HALF = M / 2;
    __shared__ __attribute__((aligned(16))) float ys[LM_WAVES][N];
    __shared__ float mags[LM_WAVES][NBINS + 3];
    __shared__ float cb[LM_CB];              // the filterbank's non-zero spans, filter after filter (every bin lies in at most two triangles: ~1 030 weights)
    __shared__ int offs[129];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* y = ys[wave];
    float* mag = mags[wave];
    float2 tw[R - 1][LM_NB];
#pragma unroll
    for (int p = 1; p < R; ++p)
#pragma unroll
        for (int i = 0; i < LM_NB; ++i) {
            const int f = lane + 64 * i;
            tw[p - 1][i] = f < NBINS ? twiddle[p * NBINS + f] : make_float2(0.f, 0.f);
        }
    // this lane's filters: m = lane and m = lane + 64; the weights go to LDS once per block (the mel sums were a chain of up to 40 dependent
    // L2 reads per lane and row: the longest part of the first version of this kernel)
    int flo[2], fhi[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = lane + 64 * h;
        flo[h] = m < n_mels ? ranges[2 * m] : 0;
        fhi[h] = m < n_mels ? ranges[2 * m + 1] : 0;
    }
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int m = 0; m < n_mels; ++m) { offs[m] = acc; acc += ranges[2 * m + 1] - ranges[2 * m]; }
        offs[n_mels] = acc;
    }
    __syncthreads();
    const bool in_lds = offs[n_mels] <= LM_CB;
    if (in_lds && (int)threadIdx.x < n_mels) {
        const int m = threadIdx.x, lo = ranges[2 * m], hi = ranges[2 * m + 1];
        const float* w = basis + (long)m * NBINS;
        for (int k = lo; k < hi; ++k) cb[offs[m] + k - lo] = w[k];
    }
    // the rows of this wave: blockIdx.x * LM_ROWS + rr * LM_WAVES + wave; every wave passes every barrier (row-dependent work is predicated),
    // the next row's spectrum is requested into registers before the current one is worked on
    auto state = [&](int rr, int& row, bool& live, float*& o) {
        row = blockIdx.x * LM_ROWS + rr * LM_WAVES + wave;
        live = false; o = nullptr;
        if (rr >= LM_RPW || row >= rows_total) return;
        const int b = row / Tp, t = row - b * Tp;
        if (t >= T) return;
        o = out + ((long)b * T + t) * n_mels;
        live = t < frames[b];
    };
    float4 nx[N / 256];
    int row; bool live; float* o;
    state(0, row, live, o);
    if (live) {
        const float4* src = (const float4*)(spec + (long)row * ld_spec);
#pragma unroll
        for (int i = 0; i < N / 256; ++i) nx[i] = src[lane + 64 * i];
    }
    for (int rr = 0; rr < LM_RPW; ++rr) {
        __syncthreads();                                         // (the previous row's readers are done with y / mag; first pass: cb is written)
        if (live) {
#pragma unroll
            for (int i = 0; i < N / 256; ++i) ((float4*)y)[lane + 64 * i] = nx[i];
        }
        int row1; bool live1; float* o1;
        state(rr + 1, row1, live1, o1);
        if (live1) {
            const float4* src = (const float4*)(spec + (long)row1 * ld_spec);
#pragma unroll
            for (int i = 0; i < N / 256; ++i) nx[i] = src[lane + 64 * i];
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int i = 0; i < LM_NB; ++i) {
                const int f = lane + 64 * i;
                if (f < NBINS) {
                    int g = f & (M - 1);
                    float sgn = 1.f;
                    if (g > HALF) { g = M - g; sgn = -1.f; }
                    const bool real_only = g == 0 || g == HALF;
                    float xr = y[g], xi = real_only ? 0.f : sgn * y[HALF + g];          // p = 0: W^0 = 1
#pragma unroll
                    for (int p = 1; p < R; ++p) {
                        const float yr = y[p * M + g], yi = real_only ? 0.f : sgn * y[p * M + HALF + g];
                        const float2 w = tw[p - 1][i];
                        xr += w.x * yr + w.y * yi;
                        xi += w.x * yi - w.y * yr;
                    }
                    mag[f] = sqrtf(xr * xr + xi * xi + 1e-9f);                            // meldataset.py:75
                }
            }
        }
        __syncthreads();
        if (o != nullptr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = lane + 64 * h;
                if (m < n_mels) {
                    float acc = 0.f;
                    if (!live) {
                        o[m] = 0.f;                                                     // zero padding AFTER the log (taco2_data.py:122-134)
                        continue;
                    }
                    if (in_lds) {
                        const float* w = cb + offs[m] - flo[h];
                        for (int k = flo[h]; k < fhi[h]; ++k) acc += w[k] * mag[k];
                    } else {
                        const float* w = basis + (long)m * NBINS;
                        for (int k = flo[h]; k < fhi[h]; ++k) acc += w[k] * mag[k];
                    }
                    o[m] = logf(fmaxf(acc, 1e-5f));                                     // meldataset.py:27-28,78
                }
            }
        }
        row = row1; live = live1; o = o1;
    }
}

}  // namespace efts

using namespace efts;

extern "C" int efts_frame_pack(const float* audio, int64_t ld_audio, const int32_t* lengths, const float* window, void* plane,
                               int64_t ld_plane, int32_t B, int32_t T, int32_t Tp, int32_t n_fft, int32_t hop, int32_t split,
                               void* stream) {
    if (!audio || !lengths || !window || !plane) return efts_fail(EFTS_EINVAL, "efts_frame_pack: null pointer");
    if (B <= 0 || T <= 0 || Tp < T) return efts_fail(EFTS_ESHAPE, "efts_frame_pack: bad B / T / Tp");
    if (n_fft <= 0 || (n_fft & 3) || hop <= 0 || hop > n_fft || ((n_fft - hop) & 1))
        return efts_fail(EFTS_ESHAPE, "efts_frame_pack: n_fft must be a multiple of 4, 0 < hop <= n_fft, n_fft - hop even");
    if (!(split == 1 || split == 2)) return efts_fail(EFTS_EINVAL, "efts_frame_pack: split must be 1 or 2");
    const int64_t need = (int64_t)((n_fft + (split == 1 ? 63 : 31)) / (split == 1 ? 64 : 32)) * 128;
    if (ld_plane < need || (ld_plane & 15)) return efts_fail(EFTS_ESHAPE, "efts_frame_pack: plane row stride too small / unaligned");
    hipLaunchKernelGGL(frame_pack_kernel, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, audio, (long)ld_audio, lengths, window,
                       (char*)plane, (long)ld_plane, T, Tp, n_fft, hop, split);
    return efts_check_launch("efts_frame_pack");
}

extern "C" int efts_logmel(const float* spec, int64_t ld_spec, const float* basis, const int32_t* ranges, const int32_t* frames,
                           float* out, int32_t B, int32_t T, int32_t Tp, int32_t n_bins, int32_t n_mels, void* stream) {
    if (!spec || !basis || !ranges || !frames || !out) return efts_fail(EFTS_EINVAL, "efts_logmel: null pointer");
    if (B <= 0 || T <= 0 || Tp < T || n_bins <= 0 || n_mels <= 0) return efts_fail(EFTS_ESHAPE, "efts_logmel: bad shape");
    if (ld_spec < 2 * (int64_t)n_bins) return efts_fail(EFTS_ESHAPE, "efts_logmel: spectrum row stride smaller than 2*n_bins");
    if (n_bins * 4 > 64 * 1024) return efts_fail(EFTS_ESHAPE, "efts_logmel: n_bins too large for the LDS tile");
    hipLaunchKernelGGL(logmel_kernel, dim3(B * Tp), dim3(64), n_bins * sizeof(float), (hipStream_t)stream, spec, (long)ld_spec, basis,
                       ranges, frames, out, T, Tp, n_bins, n_mels);
    return efts_check_launch("efts_logmel");
}

extern "C" int efts_frame_pack_dit(const float* audio, int64_t ld_audio, const int32_t* lengths, const float* window, void* plane,
                                   int64_t ld_plane, int32_t B, int32_t T, int32_t Tp, int32_t n_fft, int32_t hop, int32_t split,
                                   int32_t radix, void* stream) {
    if (!audio || !lengths || !window || !plane) return efts_fail(EFTS_EINVAL, "efts_frame_pack_dit: null pointer");
    if (B <= 0 || T <= 0 || Tp < T) return efts_fail(EFTS_ESHAPE, "efts_frame_pack_dit: bad B / T / Tp");
    if (n_fft <= 0 || hop <= 0 || hop > n_fft || ((n_fft - hop) & 1)) return efts_fail(EFTS_ESHAPE, "efts_frame_pack_dit: 0 < hop <= n_fft, n_fft - hop even");
    if (radix < 1 || n_fft % radix || (n_fft / radix) % (split == 1 ? 64 : 32))
        return efts_fail(EFTS_ESHAPE, "efts_frame_pack_dit: n_fft / radix must be a whole number of 128-byte operand chunks");
    if (!(split == 1 || split == 2)) return efts_fail(EFTS_EINVAL, "efts_frame_pack_dit: split must be 1 or 2");
    const int64_t need = (int64_t)((n_fft + (split == 1 ? 63 : 31)) / (split == 1 ? 64 : 32)) * 128;
    if (ld_plane < need || (ld_plane & 15)) return efts_fail(EFTS_ESHAPE, "efts_frame_pack_dit: plane row stride too small / unaligned");
    if ((n_fft & 3) || n_fft * 4 > 64 * 1024) return efts_fail(EFTS_ESHAPE, "efts_frame_pack_dit: n_fft must be a multiple of 4 and fit the LDS tile");
    hipLaunchKernelGGL(frame_pack_dit_kernel, dim3(B * Tp), dim3(256), n_fft * sizeof(float), (hipStream_t)stream, audio, (long)ld_audio, lengths, window,
                       (char*)plane, (long)ld_plane, T, Tp, n_fft, hop, split, radix);
    return efts_check_launch("efts_frame_pack_dit");
}

extern "C" int efts_logmel_dit(const float* spec, int64_t ld_spec, const float* basis, const int32_t* ranges, const int32_t* frames,
                               const float* twiddle, float* out, int32_t B, int32_t T, int32_t Tp, int32_t n_bins, int32_t n_mels, int32_t radix,
                               void* stream) {
    if (!spec || !basis || !ranges || !frames || !out || !twiddle) return efts_fail(EFTS_EINVAL, "efts_logmel_dit: null pointer");
    if (B <= 0 || T <= 0 || Tp < T || n_bins <= 1 || n_mels <= 0) return efts_fail(EFTS_ESHAPE, "efts_logmel_dit: bad shape");
    const int n_fft = 2 * (n_bins - 1);
    if (radix < 1 || n_fft % radix || ((n_fft / radix) & 1)) return efts_fail(EFTS_ESHAPE, "efts_logmel_dit: (n_bins - 1) * 2 / radix must be an even whole number");
    if (ld_spec < (int64_t)n_fft) return efts_fail(EFTS_ESHAPE, "efts_logmel_dit: spectrum row stride smaller than n_fft");
    if (n_bins * 4 > 64 * 1024) return efts_fail(EFTS_ESHAPE, "efts_logmel_dit: n_bins too large for the LDS tile");
    if (n_fft == 1024 && n_mels <= 128 && (ld_spec & 3) == 0 && ((uintptr_t)spec & 15) == 0 && (radix == 2 || radix == 4 || radix == 8)) {
        const int rows = B * Tp;
        const dim3 grid((rows + LM_ROWS - 1) / LM_ROWS);
#define EFTS_LMF(R) hipLaunchKernelGGL(logmel_dit_fast_kernel<R>, grid, dim3(64 * LM_WAVES), 0, (hipStream_t)stream, spec, (long)ld_spec, basis, ranges, frames, \
                                       (const float2*)twiddle, out, T, Tp, rows, n_mels)
        if (radix == 2) EFTS_LMF(2); else if (radix == 4) EFTS_LMF(4); else EFTS_LMF(8);
#undef EFTS_LMF
        return efts_check_launch("efts_logmel_dit");
    }
    hipLaunchKernelGGL(logmel_dit_kernel, dim3(B * Tp), dim3(64), n_bins * sizeof(float), (hipStream_t)stream, spec, (long)ld_spec, basis,
                       ranges, frames, (const float2*)twiddle, out, T, Tp, n_bins, n_mels, radix);
    return efts_check_launch("efts_logmel_dit");
}
