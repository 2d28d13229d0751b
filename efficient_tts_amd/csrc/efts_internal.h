// efts_internal.h -- shared helpers of libefts_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/efts_abi.h"

// Records a thread-local error message and returns `code`.
int efts_fail(int code, const char* fmt, ...);
// hipGetLastError() after a launch -> EFTS_ELAUNCH with the HIP error string.
int efts_check_launch(const char* what);
// (internal: not an ABI symbol, hidden from the shared object's export table)
__attribute__((visibility("hidden"))) void efts_gemm_init(void);
// number of compute units of the current device (cached)
int efts_num_cus(void);

// geometry of a grouped (stream-K) wgrad launch, shared by efts_wgrad_tn_grouped and efts_wgrad_reduce_grouped (efts_wgrad.hip)
struct efts_wgrad_sk_geom { int steps_tile, tiles_item, ntn, nx, q, total, workgroups, maxseg; };
// the launch's geometry as efts_wgrad_tn_grouped leaves it behind the last slab of `part` (8 int32) and efts_wgrad_reduce_grouped expects to find it:
// a reduction called with other arguments than the launch that filled `part` would sum slabs that were never written -- it writes NaN instead
#define EFTS_WGRAD_STAMP_MAGIC 0x57475236
static inline void efts_wgrad_stamp(int* st, int count, int rows, int cout, int cin, int taps, int split, const efts_wgrad_sk_geom& gm) {
    st[0] = EFTS_WGRAD_STAMP_MAGIC; st[1] = count; st[2] = rows; st[3] = cout; st[4] = cin; st[5] = taps; st[6] = split; st[7] = gm.workgroups;
}
__attribute__((visibility("hidden"))) int efts_wgrad_sk_geometry(int count, int rows, int cout, int cin, int split, int workgroups, efts_wgrad_sk_geom* gm);

namespace efts {

// round-to-nearest-even fp32 -> bf16 bits (same rounding as torch's .to(bfloat16)), integer arithmetic: the reference form
__device__ __forceinline__ unsigned short f32_to_bf16_sw(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// gfx950's v_cvt_pk_bf16_f32: two values per instruction, round-to-nearest-even -- bit-identical to the integer form on every
// finite input (tests/test_align_gpu.py sweeps the bit patterns); the planes' producers spend most of their VALU time here
typedef __bf16 efts_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float efts_f32x2_t __attribute__((ext_vector_type(2)));
#ifndef EFTS_HW_BF16
#define EFTS_HW_BF16 1
#endif
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {      // bf16(a) | bf16(b) << 16
#if EFTS_HW_BF16
    return __builtin_bit_cast(unsigned, __builtin_convertvector((efts_f32x2_t){a, b}, efts_bf16x2_t));
#else
    return (unsigned)f32_to_bf16_sw(a) | ((unsigned)f32_to_bf16_sw(b) << 16);
#endif
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
#if EFTS_HW_BF16
    return (unsigned short)(cvt_pk_bf16(f, 0.f) & 0xffffu);
#else
    return f32_to_bf16_sw(f);
#endif
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// Byte offset of element k of an operand-plane row (see efts_abi.h "MFMA operand planes").
__device__ __forceinline__ long plane_off_hi(int k, int split) {
    return split == 1 ? (long)k * 2 : (long)(k >> 5) * 128 + (k & 31) * 2;
}

// Store 4 consecutive k's (k % 4 == 0) of one row into an operand plane.
__device__ __forceinline__ void plane_store4(char* row, int k, float v0, float v1, float v2, float v3, int split) {
    const unsigned p0 = cvt_pk_bf16(v0, v1), p1 = cvt_pk_bf16(v2, v3);
    char* d = row + plane_off_hi(k, split);
    *(uint2*)d = make_uint2(p0, p1);
    if (split == 2) {
        const unsigned q0 = cvt_pk_bf16(v0 - __uint_as_float(p0 << 16), v1 - __uint_as_float(p0 & 0xffff0000u));
        const unsigned q1 = cvt_pk_bf16(v2 - __uint_as_float(p1 << 16), v3 - __uint_as_float(p1 & 0xffff0000u));
        *(uint2*)(d + 64) = make_uint2(q0, q1);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// counter-based dropout mask: keep (and scale by 1/(1-p)) iff hash(seed, element index) >= p * 2^32.
// Stateless, so forward and backward regenerate the same mask from (seed, index).
__host__ __device__ __forceinline__ unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_scale(unsigned seed_h, unsigned idx, unsigned thresh, float inv_keep) {
    return hash_u32(idx ^ seed_h) >= thresh ? inv_keep : 0.f;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_incl(float v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

}  // namespace efts
