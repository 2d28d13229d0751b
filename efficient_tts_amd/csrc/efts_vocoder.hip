// efts_vocoder.hip -- the one elementwise step of the HiFi-GAN generator that is not an epilogue of
// the contraction kernel: the multi-receptive-field sum `xs = (r0 + r1 + r2) / num_kernels` followed by
// the LeakyReLU its consumer applies to its input (nntts/vocoders/hifigan_model.py:123-131).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "efts_internal.h"

namespace efts {

// rows x c fp32 (row stride ld elements, c % 4 == 0) -> optional fp32 mean and the operand plane of leaky(mean)
__global__ __launch_bounds__(256) void mean_act_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c3,
                                                       long ld, float scale, float slope, float* __restrict__ out, long ldo,
                                                       char* __restrict__ plane, long ldp, int split, int rows, int c) {
    const int q = c >> 2;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < (long)rows * q; it += (long)gridDim.x * 256) {
        const int r = (int)(it / q), c4 = (int)(it - (long)r * q) << 2;
        const long o = (long)r * ld + c4;
        float4 v = *(const float4*)(a + o);
        if (b) { const float4 t = *(const float4*)(b + o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (c3) { const float4 t = *(const float4*)(c3 + o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (out) *(float4*)(out + (long)r * ldo + c4) = v;
        if (plane) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
            plane_store4(plane + (long)r * ldp, c4, v.x, v.y, v.z, v.w, split);
        }
    }
}

}  // namespace efts

using namespace efts;

extern "C" int efts_mean_act_rows(const float* a, const float* b, const float* c3, int64_t ld, float scale, float slope, float* out,
                                  int64_t ldo, void* plane, int64_t ld_plane, int32_t split, int32_t rows, int32_t c, void* stream) {
    if (!a || (!out && !plane)) return efts_fail(EFTS_EINVAL, "efts_mean_act_rows: null pointer");
    if (rows <= 0 || c <= 0 || (c & 3) || (ld & 3) || (out && (ldo & 3))) return efts_fail(EFTS_ESHAPE, "efts_mean_act_rows: c, ld, ldo must be multiples of 4");
    if (plane && !(split == 1 || split == 2)) return efts_fail(EFTS_EINVAL, "efts_mean_act_rows: split must be 1 or 2");
    const long items = (long)rows * (c >> 2);
    const int blocks = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
    hipLaunchKernelGGL(mean_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, c3, (long)ld, scale, slope, out, (long)ldo,
                       (char*)plane, (long)ld_plane, split, rows, c);
    return efts_check_launch("efts_mean_act_rows");
}
