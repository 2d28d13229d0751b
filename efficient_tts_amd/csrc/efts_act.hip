// efts_act.hip -- the residual layer's non-linearity when it is NOT (Leaky)ReLU.
//
// The reference builds `getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)` into every ResConv1d and into the mel
// prenet (nntts/layers/efts_modules.py:32-35, nntts/models/efficient_tts.py:76-80).  Every shipped recipe says LeakyReLU, which (with ReLU =
// slope 0) lives in the epilogues of the contraction kernels.  Any other pointwise torch.nn activation takes this two-launch form: the
// contraction writes the pre-activation z = conv(x) + bias in fp32 (efts_gemm, no activation), and
//
//   efts_act_apply:  y = (x + Dropout(f(z))) * rowmask              -> fp32 stream and / or operand plane
//   efts_act_grad :  dZ = G * rowmask * Dropout'(.) * f'(z)          -> fp32 and / or operand plane, bias gradient (column sums)
//
// HBM-bound elementwise sweeps (12-18 B per element), not tuned beyond coalescing: no reference configuration reaches them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

// f(z) and f'(z) as torch.nn defines them (torch/nn/modules/activation.py); p0 / p1 are the module's scalar parameters
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
__device__ __forceinline__ float softplus1(float z) { return z > 20.f ? z : log1pf(expf(z)); }          // F.softplus(beta 1, threshold 20), as Mish uses it

__device__ __forceinline__ float act_f(int act, float z, float p0, float p1) {
    switch (act) {
        case EFTS_ACTFN_IDENTITY: return z;
        case EFTS_ACTFN_RELU: return fmaxf(z, 0.f);
        case EFTS_ACTFN_LEAKY_RELU: return z > 0.f ? z : z * p0;
        case EFTS_ACTFN_ELU: return z > 0.f ? z : p0 * expm1f(z);
        case EFTS_ACTFN_CELU: return fmaxf(z, 0.f) + fminf(0.f, p0 * expm1f(z / p0));
        case EFTS_ACTFN_SELU: return 1.0507009873554804934193349852946f * (z > 0.f ? z : 1.6732632423543772848170429916717f * expm1f(z));
        case EFTS_ACTFN_GELU: return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
        case EFTS_ACTFN_GELU_TANH: {
            const float u = 0.79788456080286535588f * (z + 0.044715f * z * z * z);
            return 0.5f * z * (1.f + tanhf(u));
        }
        case EFTS_ACTFN_SILU: return z * sigmoidf_(z);
        case EFTS_ACTFN_MISH: return z * tanhf(softplus1(z));
        case EFTS_ACTFN_TANH: return tanhf(z);
        case EFTS_ACTFN_SIGMOID: return sigmoidf_(z);
        case EFTS_ACTFN_SOFTPLUS: return z * p0 > p1 ? z : log1pf(expf(z * p0)) / p0;
        case EFTS_ACTFN_HARDTANH: return fminf(fmaxf(z, p0), p1);
        case EFTS_ACTFN_HARDSWISH: return z * fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case EFTS_ACTFN_HARDSIGMOID: return fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case EFTS_ACTFN_SOFTSIGN: return z / (1.f + fabsf(z));
        case EFTS_ACTFN_TANHSHRINK: return z - tanhf(z);
        case EFTS_ACTFN_LOGSIGMOID: return fminf(z, 0.f) - log1pf(expf(-fabsf(z)));
        default: return z;
    }
}

__device__ __forceinline__ float act_df(int act, float z, float p0, float p1) {
    switch (act) {
        case EFTS_ACTFN_IDENTITY: return 1.f;
        case EFTS_ACTFN_RELU: return z > 0.f ? 1.f : 0.f;
        case EFTS_ACTFN_LEAKY_RELU: return z > 0.f ? 1.f : p0;
        case EFTS_ACTFN_ELU: return z > 0.f ? 1.f : p0 * expf(z);
        case EFTS_ACTFN_CELU: return z > 0.f ? 1.f : expf(z / p0);
        case EFTS_ACTFN_SELU: return 1.0507009873554804934193349852946f * (z > 0.f ? 1.f : 1.6732632423543772848170429916717f * expf(z));
        case EFTS_ACTFN_GELU:
            return 0.5f * (1.f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
        case EFTS_ACTFN_GELU_TANH: {
            const float u = 0.79788456080286535588f * (z + 0.044715f * z * z * z), t = tanhf(u);
            return 0.5f * (1.f + t) + 0.5f * z * (1.f - t * t) * 0.79788456080286535588f * (1.f + 3.f * 0.044715f * z * z);
        }
        case EFTS_ACTFN_SILU: { const float s = sigmoidf_(z); return s * (1.f + z * (1.f - s)); }
        case EFTS_ACTFN_MISH: {
            const float t = tanhf(softplus1(z)), s = sigmoidf_(z);
            return t + z * (1.f - t * t) * s;
        }
        case EFTS_ACTFN_TANH: { const float t = tanhf(z); return 1.f - t * t; }
        case EFTS_ACTFN_SIGMOID: { const float s = sigmoidf_(z); return s * (1.f - s); }
        case EFTS_ACTFN_SOFTPLUS: return z * p0 > p1 ? 1.f : sigmoidf_(z * p0);
        case EFTS_ACTFN_HARDTANH: return (z > p0 && z < p1) ? 1.f : 0.f;
        case EFTS_ACTFN_HARDSWISH: return z < -3.f ? 0.f : (z > 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
        case EFTS_ACTFN_HARDSIGMOID: return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
        case EFTS_ACTFN_SOFTSIGN: { const float d = 1.f + fabsf(z); return 1.f / (d * d); }
        case EFTS_ACTFN_TANHSHRINK: { const float t = tanhf(z); return t * t; }
        case EFTS_ACTFN_LOGSIGMOID: return sigmoidf_(-z);
        default: return 1.f;
    }
}

// one thread = 4 consecutive channels of one row
__global__ __launch_bounds__(256) void act_apply_kernel(const float* __restrict__ z, const float* __restrict__ resid,
                                                        const float* __restrict__ rowmask, int act, float p0, float p1,
                                                        float* __restrict__ y, char* __restrict__ plane, long ldp, int split, long rows, int c,
                                                        unsigned drop_thresh, unsigned drop_seed_h, float drop_inv_keep) {
    const int q = c >> 2;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= rows * q) return;
    const long r = item / q;
    const int c4 = (int)(item - r * q) << 2;
    const long o = r * c + c4;
    const float4 zv = *(const float4*)(z + o);
    float v[4] = {act_f(act, zv.x, p0, p1), act_f(act, zv.y, p0, p1), act_f(act, zv.z, p0, p1), act_f(act, zv.w, p0, p1)};
    if (drop_thresh) {
        const unsigned e0 = (unsigned)o;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] *= drop_scale(drop_seed_h, e0 + u, drop_thresh, drop_inv_keep);
    }
    if (resid) {
        const float4 xv = *(const float4*)(resid + o);
        v[0] += xv.x; v[1] += xv.y; v[2] += xv.z; v[3] += xv.w;
    }
    const float rm = rowmask ? rowmask[r] : 1.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] *= rm;
    if (y) *(float4*)(y + o) = make_float4(v[0], v[1], v[2], v[3]);
    if (plane) plane_store4(plane + r * ldp, c4, v[0], v[1], v[2], v[3], split);
}

// block: 64 rows x 128 columns (blockIdx.y = column group), 32 column quads x 8 rows in flight; column sums through LDS, one atomic
// per column and block (the layout of act_bwd_kernel, csrc/efts_train.hip)
__global__ __launch_bounds__(256) void act_grad_kernel(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ rowmask,
                                                       int act, float p0, float p1, float* __restrict__ dz, char* __restrict__ plane, long ldp,
                                                       int split, float* __restrict__ dbias, int rows, int c, unsigned drop_thresh,
                                                       unsigned drop_seed_h, float drop_inv_keep) {
    __shared__ float colsum[8][128];
    const int q = threadIdx.x & 31, rsub = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 64, r1 = min(r0 + 64, rows);
    const int cbase = blockIdx.y * 128;
    const int c4 = cbase + (q << 2);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < c)
        for (int r = r0 + rsub; r < r1; r += 8) {
            const long o = (long)r * c + c4;
            float4 gv = *(const float4*)(g + o);
            const float4 zv = *(const float4*)(z + o);
            const float rm = rowmask ? rowmask[r] : 1.f;
            float m[4] = {act_df(act, zv.x, p0, p1), act_df(act, zv.y, p0, p1), act_df(act, zv.z, p0, p1), act_df(act, zv.w, p0, p1)};
            if (drop_thresh) {
                const unsigned e0 = (unsigned)o;
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] *= drop_scale(drop_seed_h, e0 + u, drop_thresh, drop_inv_keep);
            }
            gv.x *= m[0] * rm; gv.y *= m[1] * rm; gv.z *= m[2] * rm; gv.w *= m[3] * rm;
            if (dz) *(float4*)(dz + o) = gv;
            if (plane) plane_store4(plane + (long)r * ldp, c4, gv.x, gv.y, gv.z, gv.w, split);
            acc.x += gv.x; acc.y += gv.y; acc.z += gv.z; acc.w += gv.w;
        }
    if (dbias) {
        colsum[rsub][q * 4] = acc.x; colsum[rsub][q * 4 + 1] = acc.y; colsum[rsub][q * 4 + 2] = acc.z; colsum[rsub][q * 4 + 3] = acc.w;
        __syncthreads();
        if (threadIdx.x < 128 && cbase + threadIdx.x < c) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += colsum[k][threadIdx.x];
            atomicAdd(dbias + cbase + threadIdx.x, t);
        }
    }
}

}  // namespace efts

using namespace efts;

static int drop_params(float drop_p, uint32_t drop_seed, long n, unsigned* thresh, unsigned* seed_h, float* inv_keep, const char* who) {
    *thresh = 0; *seed_h = 0; *inv_keep = 1.f;
    if (drop_p > 0.f) {
        if (!(drop_p < 1.f) || n > 0xffffffffL) return efts_fail(EFTS_EINVAL, "%s: drop_p must be in [0, 1) and rows * c < 2^32", who);
        *thresh = (unsigned)((double)drop_p * 4294967296.0);
        *seed_h = hash_u32(drop_seed);
        *inv_keep = 1.f / (1.f - drop_p);
    }
    return 0;
}

extern "C" int efts_act_apply(const float* z, const float* resid, const float* rowmask, int32_t act, float p0, float p1, float* y_f32,
                              void* plane, int64_t ld_plane, int32_t split, int32_t rows, int32_t c, float drop_p, uint32_t drop_seed,
                              void* stream) {
    if (!z || (!y_f32 && !plane)) return efts_fail(EFTS_EINVAL, "efts_act_apply: null pointer");
    if (act < 0 || act >= EFTS_ACTFN_COUNT) return efts_fail(EFTS_EINVAL, "efts_act_apply: unknown activation %d", act);
    if (rows <= 0 || c <= 0 || c % 4 || (plane && !(split == 1 || split == 2))) return efts_fail(EFTS_ESHAPE, "efts_act_apply: c must be a positive multiple of 4");
    unsigned thresh, seed_h; float inv_keep;
    if (int rc = drop_params(drop_p, drop_seed, (long)rows * c, &thresh, &seed_h, &inv_keep, "efts_act_apply")) return rc;
    const long items = (long)rows * (c >> 2);
    hipLaunchKernelGGL(act_apply_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, resid, rowmask, act, p0, p1, y_f32,
                       (char*)plane, (long)ld_plane, split, (long)rows, c, thresh, seed_h, inv_keep);
    return efts_check_launch("efts_act_apply");
}

extern "C" int efts_act_grad(const float* g, const float* z, const float* rowmask, int32_t act, float p0, float p1, float* dz, void* plane,
                             int64_t ld_plane, int32_t split, float* dbias, int32_t rows, int32_t c, float drop_p, uint32_t drop_seed,
                             void* stream) {
    if (!g || !z || (!dz && !plane)) return efts_fail(EFTS_EINVAL, "efts_act_grad: null pointer");
    if (act < 0 || act >= EFTS_ACTFN_COUNT) return efts_fail(EFTS_EINVAL, "efts_act_grad: unknown activation %d", act);
    if (rows <= 0 || c <= 0 || c % 4 || (plane && !(split == 1 || split == 2))) return efts_fail(EFTS_ESHAPE, "efts_act_grad: c must be a positive multiple of 4");
    unsigned thresh, seed_h; float inv_keep;
    if (int rc = drop_params(drop_p, drop_seed, (long)rows * c, &thresh, &seed_h, &inv_keep, "efts_act_grad")) return rc;
    hipLaunchKernelGGL(act_grad_kernel, dim3((rows + 63) / 64, (c + 127) / 128), dim3(256), 0, (hipStream_t)stream, g, z, rowmask, act, p0, p1, dz,
                       (char*)plane, (long)ld_plane, split, dbias, rows, c, thresh, seed_h, inv_keep);
    return efts_check_launch("efts_act_grad");
}
