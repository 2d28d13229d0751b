// efts_resconv_bwd.hip -- efts_resconv5 in the training backward: the dgrad of residual layer l of a stack,
//
//   G'[row, :] = ( G[row, :] + sum_{tap<5} dZ_l[row + tap - 2, :] . W_l^T[tap] ) * rowmask[row]            (= d loss / d x_l)
//
// with the ACTIVATION BACKWARD OF LAYER l - 1 fused into its epilogue (RC_EPI_DGRAD_ACT of rc_tile, efts_resconv_tile.h):
//
//   dZ_{l-1}[row, c] = G'[row, c] * (sign_{l-1}[row, c] ? 1 : slope)        -> operand plane of the next dgrad / wgrad
//   bias_part[tile, c] = sum over the tile's rows of dZ_{l-1}[row, c]        -> summed by efts_wgrad_reduce_grouped
//
// (autograd of `x + LeakyReLU(conv(x))`, nntts/layers/efts_modules.py:48-51, under nntts/trainers/efficient_tts_trainer.py:146).
// The stand-alone efts_act_bwd launch read G' back (4 B per element) to write 2 B of plane; here the values are in registers.  The
// variant is a kernel of its own in its own translation unit: inlined beside the forward's epilogue variants it pushed the 8-wave
// kernel's register allocation into scratch (round 4), alone it has nothing to share registers with.
#include "efts_resconv_tile.h"

using namespace efts;

void efts_rc_launch_dgrad_act(int split, unsigned grid, void* stream, const efts::RcArgs& k) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)resconv5_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        (void)hipFuncSetAttribute((const void*)resconv5_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        attr = true;
    }
    if (split == 1) hipLaunchKernelGGL((resconv5_kernel<1, 1>), dim3(grid), dim3(512), RC_LDS, (hipStream_t)stream, k);
    else hipLaunchKernelGGL((resconv5_kernel<2, 1>), dim3(grid), dim3(512), RC_LDS, (hipStream_t)stream, k);
}
