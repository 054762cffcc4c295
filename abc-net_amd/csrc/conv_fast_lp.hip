// The lean convolution kernel with the lane = pixel epilogue (conv_fast_body.hpp, LP): the 192-pixel x 128-channel weights-direct
// tile of the 128-channel levels (unet.py:12,15,66 forward and data gradients; the folded inference graph) with the MFMA operands
// swapped, so that the tile leaves the accumulators in 16-byte stores with no LDS staging and no workgroup barrier at the tile's end.
// Same tile, halo, grid and results as the lane = channel form in conv_fast.hip (the statistics come in one partial row per wave row
// instead of one per tile: abc_conv_stat_blocks).
// (DEBUG flavour only: an experiment measured in profiles/README.md "Round 5", not part of the production library)
#ifdef ABC_KERNEL_DEBUG
#include "conv_fast_body.hpp"

using namespace abc_cf;

// epi: 0 plain / 2 act_bwd in the epilogue (abc_conv_desc.actbwd_*); bf16 in, bf16 compute, bf16 out
// g.var (experiments, abc_debug_conv_lp): bit 0 = 1 x 4 wave layout, bit 1 = s_setprio around the MFMA groups; g.lp = 0 with g.var != 0: the
// lane = channel epilogue with that variant
template <bool LP, int VAR>
static int launch_var(const FastK& k, const abc_fast_geom& g, int epi, hipStream_t st) {
    if (epi == 2) return launch_st<bf16, bf16, bf16, 32, 128, 1, 6, false, 9, 2, 4, LP, VAR>(k, g, st);
    return launch_st<bf16, bf16, bf16, 32, 128, 1, 6, false, 9, 0, 4, LP, VAR>(k, g, st);
}
int abc_conv_fast_launch_lp(const FastK& k, const abc_fast_geom& g, int epi, hipStream_t st) {
    if (g.lp) {
        switch (g.var) {
            case 1: return launch_var<true, 1>(k, g, epi, st);
            case 2: return launch_var<true, 2>(k, g, epi, st);
            case 3: return launch_var<true, 3>(k, g, epi, st);
            default: return launch_var<true, 0>(k, g, epi, st);
        }
    }
    switch (g.var) {
        case 1: return launch_var<false, 1>(k, g, epi, st);
        case 2: return launch_var<false, 2>(k, g, epi, st);
        default: return launch_var<false, 3>(k, g, epi, st);
    }
}
#endif
