// efts_rowsweep.h -- the store side of the HBM-bound row producers (efts_expand, efts_frame_linear): a wave's 32-row x 64-column
// block of fp32 MFMA accumulators leaves through 8 KiB of wave-private LDS and is swept out 8 rows per pass with 8 lanes per
// row, so that every store instruction writes whole 128-byte lines of the row space -- the bf16 hi / lo operand planes the
// next contraction reads (either format of include/efts_abi.h) and / or the fp32 stream.
#pragma once
#include "efts_mma.h"

namespace efts {

struct RowOut {
    float* y_f32;      // [rows][ldo] fp32 or null
    char* y;           // operand plane (row 0) or null
    char* y_lo;        // y_split 1: remainder plane or null
    long ldo, ldy;     // elements / bytes
    int y_split;       // 1 | 2 (0 with y == null)
};

__device__ __forceinline__ void gstore_b128(void* p, u32x4 v) {
    *(u32x4*)p = v;
    asm volatile("s_nop 4" ::"v"(v));          // keeps the data registers untouched behind the wide store (efts_mma.h store_b128)
}

// 8 floats -> 8 bf16 (hi) + the bf16 of the remainders (lo)
__device__ __forceinline__ void split8(const float* f, u32x4* hi, u32x4* lo) {
    float r[8], d0, d1;
    *hi = u32x4{pack_bf16x2(f[0], f[1], &r[0], &r[1]), pack_bf16x2(f[2], f[3], &r[2], &r[3]),
                pack_bf16x2(f[4], f[5], &r[4], &r[5]), pack_bf16x2(f[6], f[7], &r[6], &r[7])};
    *lo = u32x4{pack_bf16x2(r[0], r[1], &d0, &d1), pack_bf16x2(r[2], r[3], &d0, &d1),
                pack_bf16x2(r[4], r[5], &d0, &d1), pack_bf16x2(r[6], r[7], &d0, &d1)};
}

// a0 / a1: the accumulators of two adjacent 32-column blocks (C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); st: this wave's 8 KiB ([32 rows][64 columns] fp32, 16-byte slots XORed with
// (row >> 1) & 1: conflict-free both ways).  Tile row r goes to row-space row orow0 + r, columns col0 .. col0 + 63
// (col0 % 64 == 0); rows >= rows_live are not stored.  EPI: value = act(acc + bias) with this lane's two biases b0, b1.
template <bool EPI>
__device__ __forceinline__ void sweep64(const f32x16& a0, const f32x16& a1, char* st, int lane, const RowOut& o, long orow0,
                                        int rows_live, int col0, float b0, float b1, int act, float slope) {
    const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhalf, col = jj * 32 + lrow;
            float v = jj ? a1[r] : a0[r];
            if (EPI) {
                v += jj ? b1 : b0;
                if (act == EFTS_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                else if (act == EFTS_ACT_RELU) v = v > 0.f ? v : 0.f;
            }
            *(float*)(st + rl * 256 + (((col >> 2) ^ ((rl >> 1) & 1)) << 4) + (col & 3) * 4) = v;
        }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + (lane >> 3), c8 = lane & 7, sw = (row >> 1) & 1;
        const long orow = orow0 + row;
        const bool keep = row < rows_live;
        if (o.y_f32 || (o.y && o.y_split == 1)) {
            const float4 d0 = *(const float4*)(st + row * 256 + (((2 * c8) ^ sw) << 4));
            const float4 d1 = *(const float4*)(st + row * 256 + (((2 * c8 + 1) ^ sw) << 4));
            const int col = col0 + c8 * 8;
            if (keep) {
                if (o.y_f32) {
                    gstore_b128(o.y_f32 + orow * o.ldo + col, __builtin_bit_cast(u32x4, d0));
                    gstore_b128(o.y_f32 + orow * o.ldo + col + 4, __builtin_bit_cast(u32x4, d1));
                }
                if (o.y && o.y_split == 1) {
                    const float f[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    u32x4 hi, lo;
                    split8(f, &hi, &lo);
                    gstore_b128(o.y + orow * o.ldy + (long)col * 2, hi);
                    if (o.y_lo) gstore_b128(o.y_lo + orow * o.ldy + (long)col * 2, lo);
                }
            }
        }
        if (o.y && o.y_split == 2) {
            // [32 hi | 32 lo] chunks: lanes 0-3 of a row write the hi slots, lanes 4-7 the lo slots of the same 32 columns
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int s0 = ch * 8 + 2 * (c8 & 3);
                const float4 d0 = *(const float4*)(st + row * 256 + ((s0 ^ sw) << 4));
                const float4 d1 = *(const float4*)(st + row * 256 + (((s0 + 1) ^ sw) << 4));
                const float f[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                u32x4 hi, lo;
                split8(f, &hi, &lo);
                if (keep) gstore_b128(o.y + orow * o.ldy + (long)((col0 >> 5) + ch) * 128 + c8 * 16, c8 < 4 ? hi : lo);
            }
        }
    }
}

}  // namespace efts
