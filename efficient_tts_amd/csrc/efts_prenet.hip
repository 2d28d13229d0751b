// efts_prenet.hip -- a Linear with few input features straight from the caller's fp32 frames into the row space:
//
//   y[b * Tp + t, :] = act( x[b, t, :cin] . W^T + bias ),  t < T        (cin <= 128, cin % 8 == 0, n % 128 == 0)
//
// i.e. `mel_prenet` of the reference (Linear(80, 512) + LeakyReLU, nntts/models/efficient_tts.py:76-80, applied at :161)
// WITHOUT the detour through an operand plane of the input (efts_pack_rows + an efts_gemm launch of two K steps per
// workgroup).  The launch is bound by its output: 4 B per frame and channel (hi + lo planes, or fp32) against 320 B of input
// per frame, so it is built like efts_expand (efts_align.hip): one workgroup = one item x one slice of 128 output channels,
// 8 waves; the slice's weight rows sit in LDS as MFMA B fragments (hi, or hi + lo); every wave converts its 32 frames to bf16
// fragments in registers -- 8 consecutive features per lane are 32 contiguous bytes of the caller's tensor -- runs its 5 (8)
// k-slices, and sweeps the block out as whole 128-byte lines (efts_rowsweep.h).  (Round 2's version swept 64-byte row
// segments from 4-wave workgroups: 62-78 us for the 105 MB of the mel prenet at 64 x 800 frames.)
// Results are bit-identical to efts_pack_rows + efts_gemm (same operand rounding, same k order; the zero-padded k-slices of the
// plane path add exact zeros).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_rowsweep.h"

namespace efts {

struct FlArgs {
    const float* x;        // [B][T][cin]
    const char* w;         // packed B plane [n][ldw] (efts_pack_weight, one tap)
    const float* bias;
    float* y_f32;          // [B * Tp][ldo] or null
    char* y;               // operand plane, row 0, or null
    char* y_lo;            // split-1 remainder plane or null
    long ldw, ldo, ldy;
    int B, T, Tp, cin, n;
    int act;
    float slope;
    int y_split;
};

constexpr int FL_NCB = 4;                                     // 32-column blocks per workgroup (128 output columns)

template <int SPLIT, int KS>                                  // KS: k-slices of 16 features held in registers (5: cin <= 80, 8: cin <= 128)
__global__ __launch_bounds__(512, 2) void frame_linear_kernel(FlArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WF_BYTES = KS * FL_NCB * SPLIT * 1024;     // weight fragments [KS][NCB][hi (, lo)][64 lanes][16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const int ncq = p.n / (32 * FL_NCB);
    const int ns = (p.cin + 15) >> 4;
    // a workgroup takes the (item, channel slice) units blockIdx.x, blockIdx.x + gridDim.x, ...: one each by default; the caller may
    // cap the grid (efts_frame_linear_args.max_workgroups) so that the launch leaves compute units to a concurrent one -- the
    // kernel is bound by HBM, which half the CUs already saturate
    for (int unit = blockIdx.x; unit < p.B * ncq; unit += gridDim.x) {
    const int b = unit / ncq, cq = unit - b * ncq;
    const int c0 = cq * 32 * FL_NCB;
    if (unit != (int)blockIdx.x) __syncthreads();            // every wave is done with the previous unit's weight fragments

    // ---- 1. the 128 weight rows of this slice -> LDS, in the order the lanes read them: lane l of (slice s, block cb) holds
    // output column c0 + 32 cb + (l & 31), features 16 s + 8 (l >> 5) .. + 7
    for (int idx = tid; idx < ns * FL_NCB * 64; idx += 512) {
        const int l = idx & 63, cb = (idx >> 6) % FL_NCB, s = idx / (64 * FL_NCB);
        const int k0 = 16 * s + 8 * (l >> 5);
        const char* src = p.w + (long)(c0 + cb * 32 + (l & 31)) * p.ldw + (SPLIT == 1 ? k0 * 2 : (k0 >> 5) * 128 + (k0 & 31) * 2);
        char* d = smem + ((s * FL_NCB + cb) * SPLIT) * 1024 + l * 16;
        *(u32x4*)d = *(const u32x4*)src;
        if constexpr (SPLIT == 2) *(u32x4*)(d + 1024) = *(const u32x4*)(src + 64);
    }
    float bv[FL_NCB];
#pragma unroll
    for (int cb = 0; cb < FL_NCB; ++cb) bv[cb] = p.bias ? p.bias[c0 + cb * 32 + lrow] : 0.f;
    __syncthreads();

    char* const st = smem + WF_BYTES + wave * 8192;
    const RowOut o{p.y_f32, p.y, p.y_lo, p.ldo, p.ldy, p.y_split};
    const int nrb = (p.T + 31) >> 5;
    for (int rb = wave; rb < nrb; rb += 8) {
        // ---- 2. this lane's frame: 8 consecutive features per k-slice, rounded to bf16 (hi) and to the bf16 remainder (lo)
        const int t = rb * 32 + lrow;
        const bool live = t < p.T;
        const float* xr = p.x + ((long)b * p.T + (live ? t : 0)) * p.cin;
        bf16x8 ah[KS], al[SPLIT == 2 ? KS : 1];
        (void)al;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = s * 16 + lhalf * 8;
            float f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = 0.f;
            if (live && s < ns && k0 + 8 <= p.cin) {
                const float4 q0 = *(const float4*)(xr + k0), q1 = *(const float4*)(xr + k0 + 4);
                f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w;
            }
            u32x4 h4, l4;
            split8(f, &h4, &l4);
            ah[s] = __builtin_bit_cast(bf16x8, h4);
            if constexpr (SPLIT == 2) al[s] = __builtin_bit_cast(bf16x8, l4);
        }
        // ---- 3. 32 frames x 128 columns per wave
        f32x16 acc[FL_NCB];
#pragma unroll
        for (int cb = 0; cb < FL_NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s < ns) {
                bf16x8 bh[FL_NCB], bl[SPLIT == 2 ? FL_NCB : 1];
                (void)bl;
#pragma unroll
                for (int cb = 0; cb < FL_NCB; ++cb) {
                    const char* src = smem + ((s * FL_NCB + cb) * SPLIT) * 1024 + lane * 16;
                    bh[cb] = *(const bf16x8*)src;
                    if constexpr (SPLIT == 2) bl[cb] = *(const bf16x8*)(src + 1024);
                }
                if constexpr (SPLIT == 2) {                 // (efts_gemm's order inside a k-slice: lo*hi, hi*lo, hi*hi per accumulator)
#pragma unroll
                    for (int cb = 0; cb < FL_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh[cb], acc[cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < FL_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl[cb], acc[cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < FL_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh[cb], acc[cb], 0, 0, 0);
            }
        }
        // ---- 4. bias + activation, then out as whole lines; frames t >= T are not stored (the gap rows stay zero)
#pragma unroll
        for (int hp = 0; hp < FL_NCB / 2; ++hp)
            sweep64<true>(acc[2 * hp], acc[2 * hp + 1], st, lane, o, (long)b * p.Tp + rb * 32, p.T - rb * 32, c0 + hp * 64, bv[2 * hp], bv[2 * hp + 1],
                          p.act, p.slope);
    }
    }   // units
}

}  // namespace efts

using namespace efts;

extern "C" int efts_frame_linear(const efts_frame_linear_args* a, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_frame_linear: null args");
    if (!a->x || !a->w || (!a->y && !a->y_f32)) return efts_fail(EFTS_EINVAL, "efts_frame_linear: null operand / no output");
    if (!(a->split == 1 || a->split == 2) || (a->y && !(a->y_split == 1 || a->y_split == 2))) return efts_fail(EFTS_EINVAL, "efts_frame_linear: split / y_split must be 1 or 2");
    if (a->B <= 0 || a->T <= 0 || a->Tp < a->T || a->cin <= 0 || a->cin > 128 || (a->cin & 7) || a->n <= 0 || a->n % (32 * FL_NCB))
        return efts_fail(EFTS_ESHAPE, "efts_frame_linear: cin must be a multiple of 8 up to 128, n a multiple of 128, T <= Tp");
    const int kch = a->split == 1 ? 64 : 32, nchunk = (a->cin + kch - 1) / kch;
    if (a->ldw < (int64_t)nchunk * 128 || (a->ldw & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)a->x & 15) || ((a->cin * 4) & 15))
        return efts_fail(EFTS_EALIGN, "efts_frame_linear: weight plane stride / alignment, frames must be 16-byte aligned rows");
    if ((a->y && ((a->ldy & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->y_lo & 15))) || (a->y_f32 && ((a->ldo & 3) || ((uintptr_t)a->y_f32 & 15))))
        return efts_fail(EFTS_EALIGN, "efts_frame_linear: output rows must be 16-byte aligned");
    if (a->y_split == 2 && a->y_lo) return efts_fail(EFTS_EINVAL, "efts_frame_linear: y_lo is for split-1 output planes");
    FlArgs k;
    k.x = a->x; k.w = (const char*)a->w; k.bias = a->bias; k.y_f32 = a->y_f32; k.y = (char*)a->y; k.y_lo = (char*)a->y_lo;
    k.ldw = a->ldw; k.ldo = a->ldo; k.ldy = a->ldy; k.B = a->B; k.T = a->T; k.Tp = a->Tp; k.cin = a->cin; k.n = a->n;
    k.act = a->act; k.slope = a->slope; k.y_split = a->y ? a->y_split : 0;
    int wgs = a->B * (a->n / (32 * FL_NCB));
    if (a->max_workgroups > 0 && a->max_workgroups < wgs) wgs = a->max_workgroups;
    const dim3 grid((unsigned)wgs);
#define EFTS_FL(S, N)                                                                                                                \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) { (void)hipFuncSetAttribute((const void*)frame_linear_kernel<S, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((frame_linear_kernel<S, N>), grid, dim3(512), (size_t)N * FL_NCB * S * 1024 + 8 * 8192, (hipStream_t)stream, k);  \
    } while (0)
    const bool small = a->cin <= 80;
    if (a->split == 1) { if (small) EFTS_FL(1, 5); else EFTS_FL(1, 8); }
    else { if (small) EFTS_FL(2, 5); else EFTS_FL(2, 8); }
#undef EFTS_FL
    return efts_check_launch("efts_frame_linear");
}
