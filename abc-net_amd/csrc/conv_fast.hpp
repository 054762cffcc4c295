// Internal interface between conv_igemm.hip (C-ABI entry points) and conv_fast.hip (lean kernel for plain NHWC inputs).
#pragma once
#include "../../include/abcnet_hip.h"

struct abc_fast_geom {
    int eligible;
    int CK, BN, MT, dy_min, dx_min, HH, HW, PS, RS, tg, ngroups, sA_bytes, a_bufs, sB_bytes, tap_off, coef_off, cstride, lds,
        tiles_x, tiles_y, nbn, ntiles, nwg, b_static, stg_off, red_off, wd, ystg_off, nw, lp, epi_off, var, m16;
};

int abc_conv_fast_geom(const abc_conv_desc* d, abc_fast_geom* g);
int abc_conv_fast_launch(const abc_conv_desc* d, const abc_fast_geom& g, abc_stream_t stream);
// n (2..4) convolutions of one lean-kernel geometry as ONE launch (the four phases of a ConvTranspose2d forward)
int abc_conv_fast_batch_ok(const abc_conv_desc* d, int n, abc_fast_geom* g0);
int abc_conv_fast_launch_batch(const abc_conv_desc* d, int n, abc_stream_t stream);

// plain-input 16 / 32-channel 3x3 convolution without statistics (conv_narrow.hip)
int abc_conv_narrow_ok(const abc_conv_desc* d);
int abc_conv_narrow_launch(const abc_conv_desc* d, abc_stream_t stream);
int abc_conv_narrow_stat_blocks(const abc_conv_desc* d);   // statistics rows it writes: one per workgroup

// one-channel first convolution (stem.hip)
int abc_conv_stem_ok(const abc_conv_desc* d, int* stat_blocks);
int abc_conv_stem_launch(const abc_conv_desc* d, abc_stream_t stream);

// heads' 1x1 convolution forward into NCHW f32 (heads.hip)
int abc_head_fwd_ok(const abc_conv_desc* d);
int abc_head_fwd_launch(const abc_conv_desc* d, abc_stream_t stream);
int abc_head_dgrad_ok(const abc_conv_desc* d);
int abc_head_dgrad_launch(const abc_conv_desc* d, abc_stream_t stream);

// 16-channel 3x3 weight gradient (wgrad_narrow.hip)
int abc_wgrad_narrow_ok(const abc_wgrad_desc* d);
int abc_wgrad_narrow_launch(const abc_wgrad_desc* d, abc_stream_t stream);
int abc_wgrad_n32r2_ok(const abc_wgrad_desc* d);
int abc_wgrad_n32r2_launch(const abc_wgrad_desc* d, abc_stream_t stream);
