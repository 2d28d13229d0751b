// efts_prenet.hip -- a Linear with few input features straight from the caller's fp32 frames into the row space:
//
//   y[b * Tp + t, :] = act( x[b, t, :cin] . W^T + bias ),  t < T        (cin <= 128, cin % 8 == 0, n % 256 == 0)
//
// i.e. `mel_prenet` of the reference (Linear(80, 512) + LeakyReLU, nntts/models/efficient_tts.py:76-80, applied at :161)
// WITHOUT the detour through an operand plane of the input: efts_pack_rows (20 us at 64 x 800 frames) + an efts_gemm launch
// whose 1 604 workgroups each run two K steps and are bound by their own start-up latency (45 us alone, 85-100 us beside the
// text-side stream).  Here the packed weight rows of a 256-column half (64 KiB bf16, 96 KiB bf16x3) sit in LDS once per
// workgroup, every wave converts its 32 frames to bf16 (hi, or hi + lo) fragments in registers -- 8 consecutive features per
// lane are 32 contiguous bytes of the caller's tensor -- and runs 5 k-slices x 8 column blocks of MFMAs; the epilogue is the
// wave-private staged sweep of efts_resconv.hip.  Results are bit-identical to efts_pack_rows + efts_gemm (same operand
// rounding, same k order; the zero-padded k-slices of the plane path add exact zeros).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_mma.h"

namespace efts {

struct FlArgs {
    const float* x;        // [B][T][cin]
    const char* w;         // packed B plane [n][ldw] (efts_pack_weight, one tap)
    const float* bias;
    float* y_f32;          // [B * Tp][ldo] or null
    char* y;               // operand plane, row 0, or null
    char* y_lo;            // split-1 remainder plane or null
    long ldw, ldo, ldy;
    int B, T, Tp, cin, n, tiles_per_item;
    int act;
    float slope;
    int y_split;
};

constexpr int FL_ROWS = 128;                                  // frames per workgroup (4 waves x 32)
constexpr int FL_COLS = 256;                                  // output columns per workgroup

template <int SPLIT, int NS>                                  // NS: k-slices of 16 features held in registers (5: cin <= 80, 8: cin <= 128)
__global__ __launch_bounds__(256) void frame_linear_kernel(FlArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KCH = SPLIT == 1 ? 64 : 32;                // features per 128-byte chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nchunk = (p.cin + KCH - 1) / KCH;
    const int wrow = nchunk * 128;                           // bytes of a weight row that are used
    const int ldl = wrow + 16;                               // LDS row stride: one 16-byte slot of padding -> a 16-lane read group hits 16 different bank groups
    const int nh = p.n / FL_COLS;
    const int half = blockIdx.x % nh;                        // the column half is fixed per workgroup: its weights are staged once
    const int n0 = half * FL_COLS;
    const int ntiles = p.B * p.tiles_per_item;

    // ---- 1. the 256 weight rows of this column half -> LDS
    {
        const int slots = wrow >> 4;
        for (int idx = tid; idx < FL_COLS * slots; idx += 256) {
            const int r = idx / slots, s = idx - r * slots;
            *(u32x4*)(smem + r * ldl + s * 16) = *(const u32x4*)(p.w + (long)(n0 + r) * p.ldw + s * 16);
        }
    }
    __syncthreads();
    const int ns = (p.cin + 15) >> 4;
    // persistent over the 128-frame tiles of this column half
    for (int tile = blockIdx.x / nh; tile < ntiles; tile += gridDim.x / nh) {
    const int b = tile / p.tiles_per_item, t0 = (tile - b * p.tiles_per_item) * FL_ROWS;
    // ---- 2. this lane's frame: 8 consecutive features per k-slice, rounded to bf16 (hi) and to the bf16 remainder (lo)
    const int t = t0 + wave * 32 + lrow;
    const bool live = t < p.T;
    const float* xr = p.x + ((long)b * p.T + (live ? t : 0)) * p.cin;
    bf16x8 ah[NS], al[SPLIT == 2 ? NS : 1];
    (void)al;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k0 = s * 16 + lhalf * 8;
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = 0.f;
        if (live && s < ns && k0 + 8 <= p.cin) {
            const float4 q0 = *(const float4*)(xr + k0), q1 = *(const float4*)(xr + k0 + 4);
            f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned short h = f32_to_bf16(f[u]);
            ah[s][u] = (short)h;
            if constexpr (SPLIT == 2) al[s][u] = (short)f32_to_bf16(f[u] - bf16_to_f32(h));
        }
    }

    // ---- 3. 32 frames x 256 columns per wave
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s < ns) {
            if constexpr (SPLIT == 1) {
                const int off = (s >> 2) * 128 + (((s & 3) * 2 + lhalf) << 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16x8 bf = *(const bf16x8*)(smem + (j * 32 + lrow) * ldl + off);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bf, acc[j], 0, 0, 0);
                }
            } else {
                const int off = (s >> 1) * 128 + (((s & 1) * 2 + lhalf) << 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16x8 bh = *(const bf16x8*)(smem + (j * 32 + lrow) * ldl + off);
                    const bf16x8 bl = *(const bf16x8*)(smem + (j * 32 + lrow) * ldl + off + 64);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc[j], 0, 0, 0);       // (efts_gemm's order inside a k-slice)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- 4. epilogue, wave-private: one 32 x 32 block at a time through 4 KiB of LDS behind the weights, swept out as 16-byte
    // row segments (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    char* st = smem + FL_COLS * ldl + wave * 4096;
    const int srow = lane >> 2, sc8 = (lane & 3) * 8;
    const long orow0 = (long)b * p.Tp + t0 + wave * 32;       // row-space row of this wave's frame 0
    const int rows_live = p.T - (t0 + wave * 32);             // frames of this wave that exist (may be <= 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float bv = p.bias ? p.bias[n0 + j * 32 + lrow] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            float vv = acc[j][r] + bv;
            if (p.act == EFTS_ACT_LEAKY) vv = vv > 0.f ? vv : vv * p.slope;
            else if (p.act == EFTS_ACT_RELU) vv = vv > 0.f ? vv : 0.f;
            *(float*)(st + rl * 128 + ((((lrow >> 2) ^ ((rl >> 1) & 1))) << 4) + (lrow & 3) * 4) = vv;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + srow;
            const int sw = (row >> 1) & 1;
            const float4 d0 = *(const float4*)(st + row * 128 + ((((lane & 3) * 2) ^ sw) << 4));
            const float4 d1 = *(const float4*)(st + row * 128 + ((((lane & 3) * 2 + 1) ^ sw) << 4));
            if (row >= rows_live) continue;
            const float y[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            const int col = n0 + j * 32 + sc8;
            const long orow = orow0 + row;
            if (p.y_f32) {
                *(float4*)(p.y_f32 + orow * p.ldo + col) = d0;
                *(float4*)(p.y_f32 + orow * p.ldo + col + 4) = d1;
            }
            if (p.y) {
                float rr[8];
                const u32x4 hi = {pack_bf16x2(y[0], y[1], &rr[0], &rr[1]), pack_bf16x2(y[2], y[3], &rr[2], &rr[3]),
                                  pack_bf16x2(y[4], y[5], &rr[4], &rr[5]), pack_bf16x2(y[6], y[7], &rr[6], &rr[7])};
                float e0, e1;
                const u32x4 lo = {pack_bf16x2(rr[0], rr[1], &e0, &e1), pack_bf16x2(rr[2], rr[3], &e0, &e1),
                                  pack_bf16x2(rr[4], rr[5], &e0, &e1), pack_bf16x2(rr[6], rr[7], &e0, &e1)};
                const long boff = p.y_split == 1 ? (long)col * 2 : (long)(col >> 5) * 128 + (col & 31) * 2;
                *(u32x4*)(p.y + orow * p.ldy + boff) = hi;
                if (p.y_split == 2) *(u32x4*)(p.y + orow * p.ldy + boff + 64) = lo;
                else if (p.y_lo) *(u32x4*)(p.y_lo + orow * p.ldy + boff) = lo;
            }
        }
    }
    }   // tile loop
}

}  // namespace efts

using namespace efts;

extern "C" int efts_frame_linear(const efts_frame_linear_args* a, void* stream) {
    if (!a) return efts_fail(EFTS_EINVAL, "efts_frame_linear: null args");
    if (!a->x || !a->w || (!a->y && !a->y_f32)) return efts_fail(EFTS_EINVAL, "efts_frame_linear: null operand / no output");
    if (!(a->split == 1 || a->split == 2) || (a->y && !(a->y_split == 1 || a->y_split == 2))) return efts_fail(EFTS_EINVAL, "efts_frame_linear: split / y_split must be 1 or 2");
    if (a->B <= 0 || a->T <= 0 || a->Tp < a->T || a->cin <= 0 || a->cin > 128 || (a->cin & 7) || a->n <= 0 || a->n % FL_COLS)
        return efts_fail(EFTS_ESHAPE, "efts_frame_linear: cin must be a multiple of 8 up to 128, n a multiple of 256, T <= Tp");
    const int kch = a->split == 1 ? 64 : 32, nchunk = (a->cin + kch - 1) / kch;
    if (a->ldw < (int64_t)nchunk * 128 || (a->ldw & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)a->x & 15) || ((a->cin * 4) & 15))
        return efts_fail(EFTS_EALIGN, "efts_frame_linear: weight plane stride / alignment, frames must be 16-byte aligned rows");
    if ((a->y && ((a->ldy & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)a->y_lo & 15))) || (a->y_f32 && ((a->ldo & 3) || ((uintptr_t)a->y_f32 & 15))))
        return efts_fail(EFTS_EALIGN, "efts_frame_linear: output rows must be 16-byte aligned");
    if (a->y_split == 2 && a->y_lo) return efts_fail(EFTS_EINVAL, "efts_frame_linear: y_lo is for split-1 output planes");
    FlArgs k;
    k.x = a->x; k.w = (const char*)a->w; k.bias = a->bias; k.y_f32 = a->y_f32; k.y = (char*)a->y; k.y_lo = (char*)a->y_lo;
    k.ldw = a->ldw; k.ldo = a->ldo; k.ldy = a->ldy; k.B = a->B; k.T = a->T; k.Tp = a->Tp; k.cin = a->cin; k.n = a->n;
    k.tiles_per_item = (a->T + FL_ROWS - 1) / FL_ROWS; k.act = a->act; k.slope = a->slope; k.y_split = a->y_split;
    const size_t lds = (size_t)FL_COLS * (nchunk * 128 + 16) + 4 * 4096;
    const int nh = a->n / FL_COLS, ntiles = a->B * k.tiles_per_item;
    int wgs = efts_num_cus() * (lds <= 80 * 1024 ? 2 : 1) / nh;                // persistent: as many workgroups as fit at once
    wgs = wgs < 1 ? 1 : (wgs > ntiles ? ntiles : wgs);
    const dim3 grid((unsigned)(wgs * nh));
#define EFTS_FL(S, N)                                                                                                                \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) { (void)hipFuncSetAttribute((const void*)frame_linear_kernel<S, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((frame_linear_kernel<S, N>), grid, dim3(256), lds, (hipStream_t)stream, k);                                \
    } while (0)
    const bool small = a->cin <= 80;
    if (a->split == 1) { if (small) EFTS_FL(1, 5); else EFTS_FL(1, 8); }
    else { if (small) EFTS_FL(2, 5); else EFTS_FL(2, 8); }
#undef EFTS_FL
    return efts_check_launch("efts_frame_linear");
}
