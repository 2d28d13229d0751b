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


// ----------------------------------------------------------------------------------------------------------------------------
// The whole front-end as ONE launch for n_fft = 1024: audio -> log-mel, the STFT as a register / LDS FFT (no operand plane, no
// spectrum in memory: 52 MB of audio in, 16 MB of log-mels out per 64 x 800 frames, where the MFMA pipeline above moves 910 MB).
// One wave per PAIR of neighbouring frames: z = frame_a + i frame_b goes through one complex 1024-point FFT and the two real
// spectra come apart by conjugate symmetry, X_a[f] = (Z[f] + conj Z[N - f]) / 2, X_b[f] = (Z[f] - conj Z[N - f]) / 2i.
// The FFT is 1024 = 16 x (16 x 4), 16 complex values per lane:
//   n = 64 n1 + n2, k = k1 + 16 k2:  X[k] = sum_n2 W1024^(n2 k1) W64^(n2 k2) [sum_n1 z[64 n1 + n2] W16^(n1 k1)]
//   (1) lane n2 reads its 16 samples of both frames straight from the audio (for a fixed n1 the 64 lanes read 64 consecutive
//       samples; frame b is frame a moved by one hop = 4 values of n1, so 20 loads serve both), a 16-point DFT in registers,
//       the twiddles W1024^(n2 k1) (per-lane constants, computed once per wave);
//   (2) ONE transpose through LDS (rows of 68 complex: the column reads of step 3 then spread over all banks);
//   (3) lane (k1, q) holds n2 = 4 r + q: 16-point DFT over r, twiddles W64^(q s), and the last radix-4 across the four lanes of
//       a quad with DPP moves -- lane (k1, q) ends up with X[k1 + 16 s + 256 u(q)], s = 0 .. 15;
//   (4) Z in natural order through the same LDS region, magnitudes of both frames, the mel filterbank (its non-zero spans in LDS,
//       filter m and m + 64 per lane), log, store.
// fp32 throughout (the MFMA pipeline carried 16 mantissa bits).  meldataset.py:49-82, taco2_data.py:66-76, :122-139.
// ----------------------------------------------------------------------------------------------------------------------------
namespace fft {

// a complex value = one register pair: +, - and the two halves of a complex product are single packed instructions (v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 with op_sel picking the halves).  With a plain struct of two floats the vectoriser paired unrelated scalars
// and spent a fifth of the loop on v_mov to assemble the pairs.
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cmul(cf a, cf b) { const cf bp = {-b.y, b.x}; return a.xx * b + a.yy * bp; }
__device__ __forceinline__ cf mul_mi(cf a) { const cf r = {a.y, -a.x}; return r; }                 // a * (-i)

// forward 4-point DFT in place (W4 = -i)
__device__ __forceinline__ void dft4(cf& a, cf& b, cf& c, cf& d) {
    const cf t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_mi(b - d);
    a = t0 + t2; b = t1 + t3; c = t0 - t2; d = t1 - t3;
}

// forward 16-point DFT in place, natural order in and out: n = 4 a + b, k = c + 4 d
__device__ __forceinline__ void dft16(cf* v) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);          // over a: v[4 c + b] = y_b[c]
    // y_b[c] *= W16^(b c)
    v[4 + 1] = cmul(v[4 + 1], cf{C1, -S1});  v[4 + 2] = cmul(v[4 + 2], cf{H, -H});     v[4 + 3] = cmul(v[4 + 3], cf{S1, -C1});
    v[8 + 1] = cmul(v[8 + 1], cf{H, -H});    v[8 + 2] = mul_mi(v[8 + 2]);            v[8 + 3] = cmul(v[8 + 3], cf{-H, -H});
    v[12 + 1] = cmul(v[12 + 1], cf{S1, -C1}); v[12 + 2] = cmul(v[12 + 2], cf{-H, -H}); v[12 + 3] = cmul(v[12 + 3], cf{-C1, S1});
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);   // over b: v[4 c + d] = X[c + 4 d]
    // to natural order: X[k] sits at v[4 (k & 3) + (k >> 2)] -- a 4 x 4 transpose of the register names
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = c + 1; d < 4; ++d) { const cf t = v[4 * c + d]; v[4 * c + d] = v[4 * d + c]; v[4 * d + c] = t; }
}

template <int CTRL>
__device__ __forceinline__ cf quad(cf a) {       // the value of the quad's lane selected by the DPP quad_perm control
    cf r;
    r.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.x), CTRL, 0xf, 0xf, false));
    r.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.y), CTRL, 0xf, 0xf, false));
    return r;
}

}  // namespace fft

#ifndef FF_ABL
#define FF_ABL 0
#endif
constexpr int FF_WAVES = 4, FF_PITCH = 68, FF_CB = 2560, FF_WIDE = 64;      // filters FF_WIDE .. run on four lanes each (they are the widest)
// SAMPLE: float (audio in [-1, 1]) or short (int16 PCM as TextMelLoader.get_mel reads it, scaled by `pcm_scale` = 1 / max_wav_value at the load:
// taco2_data.py:70 -- exact in fp32 for a power of two, so both forms give the same bits; the int16 form saves the conversion pass over the batch)
template <typename SAMPLE>
__global__ __launch_bounds__(64 * FF_WAVES, 3) void logmel_fft_kernel(const SAMPLE* __restrict__ audio, long ld_audio, const int* __restrict__ lengths,
                                                                      const float* __restrict__ window, const float* __restrict__ basis,
                                                                      const int* __restrict__ ranges, float* __restrict__ out, int B, int T,
                                                                      int n_mels, float pcm_scale) {
    using namespace fft;
    constexpr int N = 1024, HOP = 256, NBINS = 513, PAD = (N - HOP) / 2;
    __shared__ __attribute__((aligned(16))) cf zs[FF_WAVES][16 * FF_PITCH];   // transpose buffer, then Z in natural order, then (|X_a|, |X_b|) per bin in place
    __shared__ __attribute__((aligned(16))) float cb[FF_CB];                  // the filterbank's non-zero spans, every span padded with zeros to whole 4-tap steps
    __shared__ cf tw1s[15][64];                             // W1024^(n2 k), k = 1 .. 15
    __shared__ cf tw2s[4][16];                              // W64^(q s)
    __shared__ int nst[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cf* zw = zs[wave];
    // ---- per block (a block works through many frame pairs): twiddle tables, filterbank
    for (int i = threadIdx.x; i < 15 * 64; i += 64 * FF_WAVES) {
        const int k = i / 64 + 1, n2 = i & 63;
        float sn, cs;
        if (FF_ABL & 64) { sn = 0.f; cs = 1.f; } else sincospif(-(float)((n2 * k) & 1023) / 512.f, &sn, &cs);
        tw1s[k - 1][n2] = cf{cs, sn};
    }
    if (threadIdx.x < 64) {
        const int qq = threadIdx.x >> 4, ss = threadIdx.x & 15;
        float sn, cs;
        sincospif(-(float)((qq * ss) & 63) / 32.f, &sn, &cs);
        tw2s[qq][ss] = cf{cs, sn};
    }
    // span of a filter in cb: filters below FF_WIDE run on one lane, the others on four (a quarter each); all spans of a kind have the length of
    // the widest one (w0 / w1 4-tap steps, per lane), shorter ones are padded with zero weights -- so offsets are closed-form and a lane may run
    // its own number of steps or the common one
    if (threadIdx.x < 2) nst[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < FF_CB; i += 64 * FF_WAVES) cb[i] = 0.f;
    __syncthreads();
    if ((int)threadIdx.x < n_mels) {
        const int m = threadIdx.x, n = ranges[2 * m + 1] - ranges[2 * m];
        atomicMax(&nst[m < FF_WIDE ? 0 : 1], m < FF_WIDE ? (n + 3) >> 2 : (n + 15) >> 4);
    }
    __syncthreads();
    const int w0 = nst[0], w1 = nst[1];
    // strides in cb: odd multiples of four floats, so that the 16-byte weight reads of 64 lanes (one filter / one quarter each) spread over all banks
    // (a stride of 16 floats put 32 lanes on the same four banks: two thirds of the kernel's LDS cycles were bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT)
    const int S0 = 4 * (w0 | 1), SQ = 4 * (w1 | 1);
    auto off_of = [&](int m) { return m < FF_WIDE ? S0 * m : S0 * min(n_mels, FF_WIDE) + 4 * SQ * (m - FF_WIDE); };
    // (the padding steps of a span read zw behind the filter's last bin: they must stay inside the wave's 16 x 65 buffer)
    const bool in_lds = off_of(n_mels) <= FF_CB && NBINS + 4 * w0 <= 16 * FF_PITCH && NBINS + 16 * w1 <= 16 * FF_PITCH;
    if (in_lds) {
        for (int m = wave; m < n_mels; m += FF_WAVES) {
            const int lo = ranges[2 * m], hi = ranges[2 * m + 1];
            for (int k = lo + lane; k < hi; k += 64) {
                const int t = k - lo;
                cb[off_of(m) + (m < FF_WIDE ? t : SQ * (t / (4 * w1)) + t % (4 * w1))] = basis[(long)m * NBINS + k];      // (wide filters: quarter after quarter)
            }
        }
    }
    __syncthreads();
    // this lane's filters: m0 = lane (all of it) and a quarter of filter m1 = FF_WIDE + lane / 4
    const int k1 = lane >> 2, q = lane & 3;
    const int m0 = lane, m1 = FF_WIDE + k1;
    // (per-lane step counts: a lane stops behind its own filter's last bin -- measured faster than the common count with two steps in flight)
    int lo0 = 0, lo1 = 0, o0 = 0, o1 = 0, ns0 = 0, ns1 = 0;
    if (in_lds && m0 < n_mels && m0 < FF_WIDE) { lo0 = ranges[2 * m0]; o0 = off_of(m0); ns0 = (ranges[2 * m0 + 1] - lo0 + 3) >> 2; }
    if (in_lds && m1 < n_mels) {
        const int lo = ranges[2 * m1], hi = ranges[2 * m1 + 1];
        lo1 = lo + 4 * w1 * q; o1 = off_of(m1) + SQ * q;
        ns1 = max(0, min(w1, (hi - lo1 + 3) >> 2));
    }
    const float s1 = (q & 2) ? -1.f : 1.f, s2 = (q & 1) ? -1.f : 1.f;     // signs of the quad radix-4: r = partner + s * own
    float win[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) win[n1] = window[64 * n1 + lane];
    const int PT = (T + 1) >> 1;                                      // frame pairs per item
    const int npairs = B * PT;
    const int stride = gridDim.x * FF_WAVES;
    for (int pair = blockIdx.x * FF_WAVES + wave; pair < ((FF_ABL & 32) ? 0 : npairs); pair += stride) {
        const int b = pair / PT, t0 = 2 * (pair - b * PT);
        const int L = lengths[b], nfr = L / HOP;
        const bool va = t0 < nfr, vb = t0 + 1 < nfr && t0 + 1 < T;
        float* oa = out + ((long)b * T + t0) * n_mels;
        float* ob = oa + n_mels;
        if (!va) {                                                   // zero padding AFTER the log (taco2_data.py:122-134); frame b is past the length too
            for (int m = lane; m < n_mels; m += 64) { oa[m] = 0.f; if (t0 + 1 < T) ob[m] = 0.f; }
            continue;
        }
        // (1) the samples of both frames: s[j] = audio[reflect(t0 * hop + 64 j + lane - pad)], j = 0 .. 19 (frame b = frame a moved by 4 j).
        // Every load is issued, from a clamped address (a predicated load per sample compiled into twenty branches)
        const SAMPLE* a = audio + (long)b * ld_audio;
        float sm[20];
        const int s0 = t0 * HOP - PAD;                               // first sample of frame a in the un-padded signal
        if (s0 >= 0 && s0 + 20 * 64 <= L) {                          // (wave-uniform) every sample of the pair lies inside the item: one base, constant offsets
            const SAMPLE* ab = a + s0 + lane;
#pragma unroll
            for (int j = 0; j < 20; ++j) sm[j] = (FF_ABL & 16) ? (float)j : (float)ab[64 * j] * pcm_scale;
        } else {
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                int s = s0 + 64 * j + lane;
                s = s < 0 ? -s : s;                                  // reflect without edge repeat (meldataset.py:69)
                s = s >= L ? 2 * (L - 1) - s : s;
                s = min(max(s, 0), L - 1);                           // (out of range only for frame b past the item's length: dropped below)
                sm[j] = (FF_ABL & 16) ? (float)s : (float)a[s] * pcm_scale;
            }
        }
        cf v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = cf{sm[n1] * win[n1], vb ? sm[n1 + 4] * win[n1] : 0.f};
#if !(FF_ABL & 2)
        dft16(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw1s[k - 1][lane]);
#endif
        // (2) transpose: row k1, column n2
#pragma unroll
        for (int k = 0; k < 16; ++k) zw[k * FF_PITCH + lane] = v[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (3) lane (k1, q): n2 = 4 r + q
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = zw[k1 * FF_PITCH + 4 * r + q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(FF_ABL & 2)
        dft16(v);
#pragma unroll
        for (int s = 1; s < 16; ++s) v[s] = cmul(v[s], tw2s[q][s]);
#endif
        // radix 4 across the quad: lanes hold x0 .. x3 (q = 0 .. 3); afterwards lane q holds X_u, u = bit-reversed q
#if !(FF_ABL & 4)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            cf c = v[s];
            const cf p = quad<0x4E>(c);                             // quad_perm [2, 3, 0, 1]
            c = c * s1 + p;                                          // q0: x0 + x2, q1: x1 + x3, q2: x0 - x2, q3: x1 - x3
            if (q == 3) c = mul_mi(c);
            const cf p2 = quad<0xB1>(c);                            // quad_perm [1, 0, 3, 2]
            v[s] = c * s2 + p2;                                      // q0: X0, q1: X2, q2: X1, q3: X3
        }
#endif
        const int u = ((q & 1) << 1) | (q >> 1);
        // (4) Z in natural order: f = k1 + 16 s + 256 u
        // (bin f sits at f + 4 (f / 256): the four lanes of a quad hold the same k1, and without the shift their stores meet in the same banks)
#pragma unroll
        for (int s = 0; s < 16; ++s) zw[k1 + 16 * s + 260 * u] = v[s];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // magnitudes of both frames, IN PLACE: bin f of this lane reads Z[f] and Z[N - f] -- no other lane reads either -- and leaves
        // (|X_a[f]|, |X_b[f]|) at zw[f] (unshifted: what sits there belongs to a bin at or below f, read in this or an earlier step)
#pragma unroll
        for (int i = 0; i < ((FF_ABL & 8) ? 2 : 9); ++i) {
            const int f = lane + 64 * i;
            if (f < NBINS) {
                const int fn = (N - f) & (N - 1);
                const cf zf = zw[f + 4 * (f >> 8)], zn = zw[fn + 4 * (fn >> 8)];
                const float ar = 0.5f * (zf.x + zn.x), ai = 0.5f * (zf.y - zn.y);
                const float br = 0.5f * (zf.y + zn.y), bi = -0.5f * (zf.x - zn.x);
                zw[f] = cf{__builtin_amdgcn_sqrtf(ar * ar + ai * ai + 1e-9f),     // meldataset.py:75 (v_sqrt_f32: 1 ulp)
                           __builtin_amdgcn_sqrtf(br * br + bi * bi + 1e-9f)};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the mel sums, four taps per step: one 16-byte read of weights, four 8-byte reads of magnitude pairs.  Steps past a filter's last
        // bin meet zero weights (cb is padded) and whatever finite value lies behind bin 512 in zw.
        auto span = [&](int off, int lo, int nsteps, float& ra, float& rb) {
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
            for (int st = 0; st < nsteps; ++st) {
                const float4 w = *(const float4*)(cb + off + 4 * st);
                const cf x0 = zw[lo + 4 * st], x1 = zw[lo + 4 * st + 1], x2 = zw[lo + 4 * st + 2], x3 = zw[lo + 4 * st + 3];
                a0 += w.x * x0.x; b0 += w.x * x0.y; a1 += w.y * x1.x; b1 += w.y * x1.y;
                a0 += w.z * x2.x; b0 += w.z * x2.y; a1 += w.w * x3.x; b1 += w.w * x3.y;
            }
            ra = a0 + a1; rb = b0 + b1;
        };
        // (a filterbank that does not fit cb: the plain loops over the global basis)
        auto span_global = [&](int m, int lo, int hi, float& ra, float& rb) {
            float a0 = 0.f, b0 = 0.f;
            const float* w = basis + (long)m * NBINS;
            for (int k = lo; k < hi; ++k) { const cf x = zw[k]; a0 += w[k] * x.x; b0 += w[k] * x.y; }
            ra = a0; rb = b0;
        };
        float sa0, sb0, sa1, sb1;
#if FF_ABL & 1
        sa0 = zw[lane].x; sb0 = zw[lane].y; sa1 = zw[lane + 64].x; sb1 = zw[lane + 64].y;
#else
        if (in_lds) {
            span(o0, lo0, ns0, sa0, sb0);
            span(o1, lo1, ns1, sa1, sb1);
        } else {
            sa0 = sb0 = sa1 = sb1 = 0.f;
            if (m0 < n_mels && m0 < FF_WIDE) span_global(m0, ranges[2 * m0], ranges[2 * m0 + 1], sa0, sb0);
            if (m1 < n_mels && q == 0) span_global(m1, ranges[2 * m1], ranges[2 * m1 + 1], sa1, sb1);
        }
#endif
        // the four quarters of a wide filter
        sa1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sa1), 0x4E, 0xf, 0xf, false));
        sb1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sb1), 0x4E, 0xf, 0xf, false));
        sa1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sa1), 0xB1, 0xf, 0xf, false));
        sb1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sb1), 0xB1, 0xf, 0xf, false));
        constexpr float LN2 = 0.69314718055994531f;
        if (m0 < n_mels && m0 < FF_WIDE) {
            oa[m0] = __builtin_amdgcn_logf(fmaxf(sa0, 1e-5f)) * LN2;              // meldataset.py:27-28, :78 (v_log_f32: log2, 1 ulp)
            if (t0 + 1 < T) ob[m0] = vb ? __builtin_amdgcn_logf(fmaxf(sb0, 1e-5f)) * LN2 : 0.f;
        }
        if (m1 < n_mels && q == 0) {
            oa[m1] = __builtin_amdgcn_logf(fmaxf(sa1, 1e-5f)) * LN2;
            if (t0 + 1 < T) ob[m1] = vb ? __builtin_amdgcn_logf(fmaxf(sb1, 1e-5f)) * LN2 : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the next pair's transpose overwrites zw
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

static int logmel_fft_launch(const void* audio, bool pcm16, float pcm_scale, int64_t ld_audio, const int32_t* lengths, const float* window, const float* basis,
                             const int32_t* ranges, float* out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream, const char* who) {
    if (!audio || !lengths || !window || !basis || !ranges || !out) return efts_fail(EFTS_EINVAL, "%s: null pointer", who);
    if (B <= 0 || T <= 0) return efts_fail(EFTS_ESHAPE, "%s: bad B / T", who);
    if (n_fft != 1024 || hop != 256 || n_mels <= 0 || n_mels > FF_WIDE + 16)
        return efts_fail(EFTS_ESHAPE, "%s: the fused kernel is built for n_fft 1024, hop 256, at most 80 mel bins (64 filters on one lane each + 16 on four lanes each; other configurations: efts_frame_pack_dit + efts_gemm + efts_logmel_dit)", who);
    const long pairs = (long)B * ((T + 1) / 2);
    long blocks = (pairs + FF_WAVES - 1) / FF_WAVES;
    const long cap = 3L * efts_num_cus();                     // persistent: three workgroups per CU (LDS, registers), each wave works through pairs / (12 CUs) frame pairs
    if (blocks > cap) blocks = cap;
    if (pcm16)
        hipLaunchKernelGGL(logmel_fft_kernel<short>, dim3((unsigned)blocks), dim3(64 * FF_WAVES), 0, (hipStream_t)stream, (const short*)audio, (long)ld_audio, lengths, window,
                           basis, ranges, out, B, T, n_mels, pcm_scale);
    else
        hipLaunchKernelGGL(logmel_fft_kernel<float>, dim3((unsigned)blocks), dim3(64 * FF_WAVES), 0, (hipStream_t)stream, (const float*)audio, (long)ld_audio, lengths, window,
                           basis, ranges, out, B, T, n_mels, 1.f);
    return efts_check_launch(who);
}

extern "C" int efts_logmel_fft(const float* audio, int64_t ld_audio, const int32_t* lengths, const float* window, const float* basis,
                               const int32_t* ranges, float* out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream) {
    return logmel_fft_launch(audio, false, 1.f, ld_audio, lengths, window, basis, ranges, out, B, T, n_fft, hop, n_mels, stream, "efts_logmel_fft");
}

extern "C" int efts_logmel_fft_pcm16(const int16_t* audio, int64_t ld_audio, float pcm_scale, const int32_t* lengths, const float* window, const float* basis,
                                     const int32_t* ranges, float* out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream) {
    return logmel_fft_launch(audio, true, pcm_scale, ld_audio, lengths, window, basis, ranges, out, B, T, n_fft, hop, n_mels, stream, "efts_logmel_fft_pcm16");
}
