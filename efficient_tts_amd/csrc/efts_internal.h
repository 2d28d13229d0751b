// efts_internal.h -- shared helpers of libefts_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/efts_abi.h"

// Records a thread-local error message and returns `code`.
int efts_fail(int code, const char* fmt, ...);
// hipGetLastError() after a launch -> EFTS_ELAUNCH with the HIP error string.
int efts_check_launch(const char* what);
extern "C" void efts_gemm_init(void);
// number of compute units of the current device (cached)
int efts_num_cus(void);

namespace efts {

// round-to-nearest-even fp32 -> bf16 bits (same rounding as torch's .to(bfloat16))
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// Byte offset of element k of an operand-plane row (see efts_abi.h "MFMA operand planes").
__device__ __forceinline__ long plane_off_hi(int k, int split) {
    return split == 1 ? (long)k * 2 : (long)(k >> 5) * 128 + (k & 31) * 2;
}

// Store 4 consecutive k's (k % 4 == 0) of one row into an operand plane.
__device__ __forceinline__ void plane_store4(char* row, int k, float v0, float v1, float v2, float v3, int split) {
    const unsigned short h0 = f32_to_bf16(v0), h1 = f32_to_bf16(v1), h2 = f32_to_bf16(v2), h3 = f32_to_bf16(v3);
    uint2 hi = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
    char* d = row + plane_off_hi(k, split);
    *(uint2*)d = hi;
    if (split == 2) {
        const unsigned short l0 = f32_to_bf16(v0 - bf16_to_f32(h0)), l1 = f32_to_bf16(v1 - bf16_to_f32(h1));
        const unsigned short l2 = f32_to_bf16(v2 - bf16_to_f32(h2)), l3 = f32_to_bf16(v3 - bf16_to_f32(h3));
        *(uint2*)(d + 64) = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
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
