// The lean convolution kernel on 8-wave workgroups (conv_fast_body.hpp, NW = 8): the 192-pixel x 128-channel weights-direct
// tile of the 128-channel levels (unet.py:12,15,66 forward and data gradients; the folded inference graph) with the tile's 24
// MFMA tiles dealt to eight waves -- 3 x 1 per wave, 48 accumulator registers, <= 128 VGPRs, so that the CU's two workgroups
// put FOUR waves on every SIMD.  Same tile, halo, grid, statistics rows and results as the 4-wave form (conv_fast.hip); only
// the wave -> MFMA-tile map and the occupancy differ.
// (DEBUG flavour only: an experiment measured in profiles/README.md "Round 5", not part of the production library)
#ifdef ABC_KERNEL_DEBUG
#include "conv_fast_body.hpp"

using namespace abc_cf;

// epi: 0 plain / 2 act_bwd in the epilogue (abc_conv_desc.actbwd_*); bf16 in, bf16 compute, bf16 out
int abc_conv_fast_launch8(const FastK& k, const abc_fast_geom& g, int epi, hipStream_t st) {
    if (epi == 2) return launch_st<bf16, bf16, bf16, 32, 128, 1, 6, false, 9, 2, 8>(k, g, st);
    return launch_st<bf16, bf16, bf16, 32, 128, 1, 6, false, 9, 0, 8>(k, g, st);
}
#endif
