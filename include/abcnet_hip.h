/* abcnet_hip.h -- C ABI of libabcnet_hip.so: the MI355X (gfx950) kernels behind
 * ABC-Net's U-Net hot path.
 *
 * The reference (zhang-xuan1314/ABC-Net) has NO native interface: every FLOP of
 * the path runs inside torch ops.  Each entry point
 * below therefore names the reference *torch call site* it replaces (file:line
 * under /root/reference).  The binding a maintainer adds is the ctypes stub in
 * INTEGRATION.md; abc-net_amd/_lib.py is that stub in full.
 *
 * Conventions
 *  - plain pointers and sizes only: device pointers are raw HBM addresses owned by
 *    the caller (torch tensors stay owned by torch); nothing is retained past a call;
 *  - every function enqueues on the given hipStream_t, never allocates, never
 *    synchronises (graph-capture safe), and returns 0 on success or a negative
 *    abc_status (abc_last_error() gives text); no exceptions cross the boundary;
 *  - activations are NHWC with an explicit pixel stride (ld*, in elements) and a
 *    channel offset, so concat buffers are written/read in place (unet.py:51-59);
 *  - dtype codes: 0 = float32, 1 = bfloat16.  Accumulation is always float32.
 *  - one host thread per GPU/process (mirrors mp.spawn, multi_gpu_train.py:36).
 */
#ifndef ABCNET_HIP_H
#define ABCNET_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* abc_stream_t; /* hipStream_t */

enum abc_status { ABC_OK = 0, ABC_EINVAL = -1, ABC_EUNSUPPORTED = -2, ABC_ELAUNCH = -3 };
enum abc_dtype { ABC_F32 = 0, ABC_BF16 = 1, ABC_FP8 = 2 /* OCP e4m3fn: the fp8 inference graph only (abc_conv_desc.out_scale) */ };
#define ABC_MAX_TAPS_C 49

/* A raw (pre-BatchNorm) activation tensor consumed through its BN affine +
 * activation, optional 2x2 max-pool and optional dropout, all applied ON LOAD:
 *   v = max(y, slope*y), y = scale[c]*x + shift[c]      (BN+ReLU: unet.py:13-17,
 *                                                         BN+LeakyReLU: unet.py:67-68)
 *   pool: max over the 2x2 window of v                   (nn.MaxPool2d(2): unet.py:30)
 *   dropout: v * keep/(1-p)                              (nn.Dropout(0.2): unet.py:69)
 * scale == NULL means identity.  Coefficient arrays are indexed by the absolute
 * channel inside x (so a concat buffer carries one array for both halves). */
typedef struct abc_act_src {
    const void* x;
    const float* scale;
    const float* shift;
    const float* slope;
    int32_t Hx, Wx, ldx; /* physical dims / pixel stride of x */
    int32_t pool;
    float drop_p;
    uint32_t drop_seed;
    int32_t planar;      /* 1: x is channel-planar f32 [B][ctot][Hx][Wx] (the reference's NCHW head maps and
                            their gradients, unet.py:119); ldx is ignored, no pool/dropout */
    int32_t ctot;
    const uint32_t* drop_salt; /* device scalar added to drop_seed (NULL = 0): a counter the caller bumps once per step
                                  (abc_counter_add_u32), so that a captured graph draws a fresh mask every replay, as
                                  nn.Dropout does every forward (unet.py:69) */
} abc_act_src;

/* One head's 1x1 convolution (unet.py:70) computed in the epilogue of the convolution that produces its 128 features
 * (abc_conv_desc.heads_epi; folded inference graph): w2 = the head's packed conv2 weights as abc_pack_conv_weights writes them
 * for the heads' 1x1 kernel (layout 0: bf16 [1][4 chunks][Cout_pad][32] / e4m3 [1][2 chunks][Cout_pad][64]), bias [Cout],
 * oscale [Cout] (e4m3 only: feature scale x weight-row scale), y = the head's NCHW f32 map [B][Cout][Hout][Wout]. */
typedef struct abc_heads_epi {
    const void* w2; const float* bias; const float* oscale; float* y;
    int32_t Cout, Cout_pad;
} abc_heads_epi;

/* Generic tap-list convolution as implicit GEMM on MFMA.  One descriptor covers
 *   nn.Conv2d 3x3/5x5/1x1 forward   (unet.py:12,15,66,70; unet2.py:56,59,66)
 *   its data gradient               (autograd of the same, train.py:140)
 *   nn.ConvTranspose2d(k3,s2) forward as 4 output-parity phases, written straight
 *   into the concat buffer with the crop of unet.py:51-57 folded in (unet.py:44)
 *   and its data gradient (stride-2 gather).
 * out[b, g*om+o0, cout_off+n] = bias[n] + sum_t sum_c  W[t][c][n] * in[b, g*stride + d_t, cin_off+c]
 * Weights are pre-packed by abc_pack_conv_weights as [tap][Cin/CK][Cout_pad][CK]. */
typedef struct abc_conv_desc {
    abc_act_src src;
    const void* w;
    const float* bias; /* [Cout] or NULL */
    void* y;
    float* stats;      /* NULL, or per-block partial (sum, sumsq) [nblk][2][Cout] of the f32 outputs */
    int32_t dtype_in, dtype_c, dtype_out;
    int32_t B, Hin, Win; /* logical input dims (after the optional pool) */
    int32_t cin_off, Cin; /* real channel count; padded to the K-chunk (16, or 32 for bf16 when Cin%32==0)
                             with zeros on load, so the 1-channel image (unet.py:83) goes through here too */
    int32_t Hg, Wg;       /* output grid iterated by the kernel */
    int32_t Hout, Wout, ldy, cout_off, Cout, Cout_pad;
    int32_t stride, om, oy0, ox0;
    int32_t ntaps;
    int8_t tap_dy[ABC_MAX_TAPS_C];
    int8_t tap_dx[ABC_MAX_TAPS_C];
    int32_t stats_rows;  /* 0/2: stats = [nblk][2][Cout] (sum, sumsq); 4: [nblk][4][Cout] adds (max, min) of the
                            f32 outputs per workgroup = per-image partials for CBAM's global pools (unet2.py:19-21) */
    int32_t accumulate;  /* 1: y += result (NHWC only): a second data-gradient summed into the same tensor
                            (residual branch of unet2.DoubleConv, unet2.py:72) */
    int32_t planar_out;  /* 1: y is channel-planar f32 [B][ctot_out][Hout][Wout] (NCHW logits written directly by
                            the heads' 1x1 conv: the list forward() returns needs no layout pass); ldy ignored */
    int32_t ctot_out;
    int32_t out_act;     /* 1: the epilogue stores max(v, out_slope * v) instead of v (v = result + bias): with BatchNorm folded into
                            the weights / bias (eval mode: abc_pack_desc.row_scale, abc_bn_eval_fold) the output is the ACTIVATED
                            tensor and its consumers load it with the identity transform; the statistics stay those of v */
    float out_slope;     /* 0 = ReLU, 0.01 = LeakyReLU */
    void* pool_y;        /* optional second output: nn.MaxPool2d(2) of the stored tensor (unet.py:30), NHWC [B][Hout/2][Wout/2][ld_pool],
                            same dtype as y -- saves the separate abc_pool_act pass of the folded inference graph.  Only where
                            abc_conv_variant() == 5 (the narrow-level kernel); abc_conv_fwd refuses it elsewhere */
    int32_t ld_pool;
    /* optional (narrow-level kernel, 16 input channels, 3x3): the input tensor is not read but COMPUTED on the fly as the
     * network's first convolution of the folded inference graph (unet.py:12-14, one input channel):
     *   in[p][c] = lrelu(sum_t stem_x[p + d_t] * stem_w[c][t] * stem_scale[c] + stem_bias[c]; stem_slope)
     * over the same 3x3 taps (zero padding), stem_x = the f32 image [B][Hin][Win]; src.x is ignored */
    const float* stem_x; const float* stem_w; const float* stem_scale; const float* stem_bias; float stem_slope;
    /* fp8 (e4m3) inference graph -- the 128-channel 3x3 convolutions of the BatchNorm-folded eval graph (img2smiles2.py:42-59;
     * SURVEY.md section 8f.4) on the block-scaled MFMA with unit scales (2 x the bf16 rate):
     *   dtype_c = ABC_FP8 (then dtype_in = ABC_FP8 too): x and the packed weights are e4m3, x = real value / s_in (per tensor),
     *   w = real (BatchNorm-folded) weight / s_w[n] (per output row); out_scale[n] = s_in * s_w[n] turns the f32 accumulator back:
     *   v = acc * out_scale[n] + bias[n], then the activation (out_act).
     *   dtype_out = ABC_FP8 (with bf16 or fp8 compute): the stored value is v * (*out_quant) rounded to e4m3 (saturating at 448),
     *   *out_quant = 1 / s_out, a DEVICE scalar (calibrated per tensor: abc_fp8_act_scale), read by the kernel.
     * Served by the weights-direct loop of the lean kernel only (3x3, stride 1, Cout a multiple of 128, Cin a multiple of 64). */
    const float* out_scale;
    const float* out_quant;
    int32_t out_quant_stride; /* 0: one scalar; 1: one value per 128-channel block of the output (out_quant[n / 128]): the eight heads'
                                 features side by side in one tensor, each head with its own scale */
    const abc_heads_epi* heads_epi; /* NULL, or a DEVICE array of Cout / 128 entries: every 128-channel block of this convolution's output is
                                 one head's finished feature slice (out_act set, BatchNorm folded) and the head's 1x1 convolution is computed
                                 in the tile's epilogue -- y is not written at all, the heads' maps are.  3x3, stride 1, bf16 or e4m3 compute,
                                 the weights-direct tile (abc_conv_variant == 1); abc_conv_fwd refuses it elsewhere */
    /* act_bwd in the epilogue (training's data gradients): NULL, or the raw convolution output y_raw [B, Hout, Wout, actbwd_ld] (bf16) of the
     * layer whose ACTIVATION OUTPUT this convolution differentiates (autograd of unet.py:12-17 under train.py:140).  The kernel then stores
     *   g = dA * (BatchNorm(y_raw) > 0 ? 1 : slope),   BatchNorm(y_raw) = actbwd_scale * y_raw + actbwd_shift,
     * instead of dA, and writes that layer's BatchNorm-backward partial sums to `stats` (stats_rows = 2; abc_conv_stat_blocks rows of
     * [2][Cout]: sum of g, sum of g * (y_raw - actbwd_mean) * actbwd_invstd) -- exactly what abc_act_bwd computes from dA in a pass of its
     * own (abc_act_bwd_desc: y_raw, scale, shift, slope, mean, invstd, partial), so abc_bn_finalize_bwd consumes them unchanged.
     * Served where abc_conv_actbwd_ok() says so (bf16, stride 1: whole tiles of the lean kernel, any shape of the 16-channel narrow-level
     * kernel); abc_conv_fwd refuses it elsewhere. */
    const void* actbwd_y;
    int32_t actbwd_ld, actbwd_coff;        /* y_raw's row length and first channel (elements) */
    const float *actbwd_scale, *actbwd_shift, *actbwd_slope, *actbwd_mean, *actbwd_invstd;   /* per output channel of THIS convolution */
    /* The peak-NMS outputs of img2smiles2.py:61-79 straight from the heads' 1x1 kernel (abc_conv_variant == 3 only; abc_conv_fwd and
     * abc_heads_batch refuse it elsewhere): head_aux = a second NCHW f32 output [B][Cout][Hout][Wout] holding
     *   head_aux_mode 1: |v|                                        (img2smiles2.py:73, the rho head)
     *   head_aux_mode 2: 1.0 where v >= both circular neighbours along the CHANNEL axis and v > -1, else 0.0
     *                    (img2smiles2.py:75-79, the omega head: Cout <= 64)
     * of the value v this launch stores to y -- the lane that computed v holds its channel neighbours (or gets them by one
     * cross-lane exchange), so the maps are not read back (abc_nms_peaks with n_omega = 0 then does the two 3x3 spatial masks
     * alone).  With head_aux set, y may be NULL: the raw map is then not stored.
     *   head_aux_mode 3: head_aux is a UINT8 map [B][Cout / 6][Hout][Wout] = arg max over g = 0..5 of channel g * (Cout / 6) + bin
     *                    (first maximum, as torch.argmax: img2smiles2.py:71,112 -- bond_types_pred.view(-1, 6, 60, H, W), .argmax(0) --
     *                    the only use the decoder makes of the 360-channel bond-type head); y must be NULL, Cout / 6 <= 64, no dropout.
     *                    abc_extract_desc.btype_idx consumes it. */
    float* head_aux;
    int32_t head_aux_mode;
} abc_conv_desc;

/* number of per-block stat partials abc_conv_fwd writes for this descriptor */
int abc_conv_stat_blocks(const abc_conv_desc* d);
/* n convolutions (an array of descriptors) issued as ONE launch where they share a tile geometry of the lean kernel -- the four
 * output-parity phases of nn.ConvTranspose2d(k3, s2) (unet.py:44: four 1 / 2 / 2 / 4-tap convolutions of the same input into
 * interleaved output pixels) -- and one after the other otherwise: the result is that of n abc_conv_fwd calls either way.
 * abc_conv_batch_ok tells which of the two it will be (1: one launch). */
int abc_conv_fwd_batch(const abc_conv_desc* d, int32_t n, abc_stream_t stream);

/* nn.ConvTranspose2d(Cin -> Cout, kernel 3, stride 2) + the crop of the first output row and column that unet.py:51-56 applies when the
 * skip tensor has 2n rows / columns (every level when the input size is a multiple of 32), forward, bias included: all four output-parity
 * phases in ONE pass over the input (convt_fused.hip) -- the same result as the four abc_conv_fwd phase calls (engine.convT_phase_taps).
 *   src          the input, NHWC bf16, with the producer's BatchNorm + activation applied on load (abc_act_src; no pool / dropout / planar)
 *   w            the NINE packed weight slices of the four phases one after the other -- phase (0,0): 1 tap, (0,1): 2, (1,0): 2, (1,1): 4,
 *                each packed by abc_pack_conv_weights mode 2 with layout 1 into [taps][Cin / 32][Cout_pad][32] -- i.e. phase p starts
 *                at slice {0, 1, 3, 5}[p] of one [9][Cin / 32][Cout_pad][32] buffer
 *   y            NHWC bf16 [B][Hout][Wout][ldy], written at channels [cout_off, cout_off + Cout); Hout = 2 Hin, Wout = 2 Win
 * abc_convt_fused_ok: 1 when the kernel serves the descriptor (bf16, both axes cropped, Cin % 32 == 0, Cout_pad % 64 == 0, Cout % 8 == 0). */
typedef struct abc_convt_desc {
    abc_act_src src;
    const void* w; const float* bias; void* y;
    int32_t dtype;                     /* ABC_BF16 */
    int32_t B, Hin, Win, cin_off, Cin;
    int32_t Hout, Wout, ldy, cout_off, Cout, Cout_pad;
} abc_convt_desc;
int abc_convt_fused_ok(const abc_convt_desc* d);
int abc_convt_fused_fwd(const abc_convt_desc* d, abc_stream_t stream);
int abc_conv_batch_ok(const abc_conv_desc* d, int32_t n);
/* 1 when abc_conv_fwd honours d->actbwd_* for this descriptor (fill everything first, `stats` included), else 0: the caller then clears
 * actbwd_y and runs abc_act_bwd as a pass of its own (the engine's fallback; same results up to the rounding of dA to bf16) */
int abc_conv_actbwd_ok(const abc_conv_desc* d);
/* which kernel abc_conv_fwd runs for this descriptor: 0 = general implicit GEMM (pool / dropout / planar / ragged
 * inputs), 1 = lean 4-wave kernel for plain NHWC inputs, 2 = one-channel first layer, 3 = heads' 1x1 into NCHW f32, 4 = its data gradient from NCHW f32.  Labels only. */
int abc_conv_variant(const abc_conv_desc* d);
int abc_conv_fwd(const abc_conv_desc* d, abc_stream_t stream);
/* tile the launcher picks for this descriptor: BN output channels x (2*mt x 16) pixels per workgroup, K-chunk ck */
int abc_conv_tile(const abc_conv_desc* d, int32_t* bn, int32_t* mt, int32_t* ck);

/* channels per K-chunk used by the packed weight layout for (compute dtype, Cin) */
int abc_conv_chunk(int dtype_c, int Cin);

/* Repack master f32 weights into the layout abc_conv_fwd / abc_wgrad consume.
 *   mode 0: Conv2d weight [Cout][Cin][kh][kw] -> forward packing   (taps = kh*kw)
 *   mode 1: Conv2d weight -> data-gradient packing (roles of Cin/Cout swapped, taps mirrored)
 *   mode 2: ConvTranspose2d weight [Cin][Cout][3][3] -> one forward parity phase (py,px)
 *   mode 3: ConvTranspose2d weight -> data-gradient packing (stride-2 gather, 9 taps)
 * Rows beyond the real channel counts are zero.  dst element type = dtype_c. */
typedef struct abc_pack_desc {
    const float* w; void* dst;
    int32_t mode, dtype_c, Cout, Cin, kh, kw, py, px;
    int32_t rows_pad;  /* padded row count of dst (Cout_pad of the consuming conv) */
    int32_t red_pad;   /* padded reduction-channel count of THIS weight (multiple of CK) */
    int32_t red_total; /* reduction-channel count of the whole dst (>= red_off + red_pad): several weights
                          may be packed side by side along the reduction axis (the 8 heads' conv1 data
                          gradient is ONE conv over the concatenated 8x128 channels) */
    int32_t red_off;   /* where this weight's reduction channels start in dst (multiple of CK) */
    int32_t ck;        /* K-chunk of dst: abc_conv_chunk(dtype_c, red_total) */
    int32_t rows_total; /* 0, or the row count of the whole dst (>= rows_off + rows_pad): several weights may be packed one
                          below the other along the ROW (output-channel) axis -- the 8 heads' conv1 (unet.py:66,116-118) run
                          as ONE 128 -> 8 x 128 convolution over the shared trunk activation */
    int32_t rows_off;  /* where this weight's rows start in dst */
    int32_t layout;    /* 0: rows of CK elements, [tap][chunk][row][CK].  1 (bf16, CK = 32 only; abc_conv_weight_layout() says which
                          convolution wants it): inside every block of 32 rows x 64 bytes the bytes are ordered
                          [kk = 16-byte half of a lane's 32 bytes][h = lane half][r = row][16 bytes], so that ONE fragment load of
                          the weights-direct conv loop (64 lanes x 16 bytes) reads 1 KB of consecutive bytes = 8 whole cache
                          lines instead of touching 16 */
    const float* row_scale; /* NULL, or one factor per output row (modes 0 and 2: per output channel): eval-mode BatchNorm folded
                               into the convolution in front of it, w'[n][..] = w[n][..] * gamma[n] / sqrt(running_var[n] + eps) */
} abc_pack_desc;
int abc_pack_conv_weights(const abc_pack_desc* d, abc_stream_t stream);
/* abc_pack_desc.layout the kernel that serves this convolution reads its weights in */
int abc_conv_weight_layout(const abc_conv_desc* d);
/* batched form: the caller builds a table of abc_pack_item_bytes()-sized entries with abc_pack_item_fill (host
 * memory, `first` = running element offset), copies it to the device once, and packs all weights of a step in ONE launch */
int abc_pack_item_bytes(void);
int64_t abc_pack_item_fill(void* item, const abc_pack_desc* d, int64_t first);
int abc_pack_batch(const void* items_dev, int32_t nitems, int64_t total, abc_stream_t stream);

/* fp8 inference graph, calibration and weight scales (all results stay on the device: no host sync, graph-safe).
 *   abc_absmax:            *out = max(*out, max |x[i]|) over n elements of dtype (f32 / bf16); zero *out first (abc_fill_f32)
 *   abc_fp8_act_scale:     s = max(*amax, 1e-12) * margin / 448, *s_out = s, *inv_s_out = 1 / s   (per-tensor activation scale)
 *   abc_fp8_weight_scales: per output row n of a Conv2d weight [rows][K] (K = Cin * kh * kw) whose BatchNorm fold factor is
 *                          fold[n] (NULL = 1):  s_w = max_k |w[n][k] * fold[n]| / 448 (1 when the row is zero),
 *                          qmul[n] = fold[n] / s_w  (abc_pack_desc.row_scale of the fp8 packing),
 *                          deq[n] = s_w * (*s_in)   (abc_conv_desc.out_scale) */
int abc_absmax(const void* x, int32_t dtype, int64_t n, float* out, abc_stream_t stream);
/* the same over the columns [c_off, c_off + C) of an [npix][ld] tensor (C, ld, c_off multiples of 8) */
int abc_absmax_cols(const void* x, int32_t dtype, int64_t npix, int32_t ld, int32_t c_off, int32_t C, float* out, abc_stream_t stream);
int abc_fp8_act_scale(const float* amax, float margin, float* s_out, float* inv_s_out, abc_stream_t stream);
int abc_fp8_weight_scales(const float* w, int32_t rows, int32_t K, const float* fold, const float* s_in, float* qmul, float* deq,
                          abc_stream_t stream);

/* BatchNorm2d, training mode (unet.py:13,16,67): reduce the conv's stat partials
 * in f64, write the on-load coefficients (scale, shift) for consumers, keep
 * (mean, invstd) for backward, update running stats with momentum 0.1 / unbiased
 * variance and bump num_batches_tracked. */
typedef struct abc_bn_fwd_desc {
    const float* partial; int32_t nblk; int32_t C; double count;
    int32_t rows; /* rows per workgroup in `partial`: 0/2 or 4 (see abc_conv_desc.stats_rows) */
    const float* gamma; const float* beta;
    float* scale; float* shift; float* mean; float* invstd;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    float eps, momentum;
} abc_bn_fwd_desc;
int abc_bn_finalize_fwd(const abc_bn_fwd_desc* d, abc_stream_t stream);
/* n <= 8 layers in one launch (the eight heads' BatchNorms, unet.py:67).  pstride > 0: the layers' stat partials are column
 * slices of ONE [nblk][rows][pstride] buffer (each descriptor's `partial` points at its first column: the heads' conv1
 * run as one convolution); 0: own buffers of width C */
int abc_bn_finalize_fwd_batch(const abc_bn_fwd_desc* descs, int32_t n, int32_t pstride, abc_stream_t stream);
/* eval mode: coefficients from running stats (model.eval(): img2smiles2.py:49) */
int abc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float* scale, float* shift, int32_t C, float eps, abc_stream_t stream);

/* eval mode with the BatchNorm folded into the convolution in front of it (img2smiles2.py:49 inference graph): the per-row
 * weight factor scale[c] = gamma / sqrt(running_var + eps) (-> abc_pack_desc.row_scale) and the folded bias
 * bias_out[c] = (conv_bias[c] - running_mean[c]) * scale[c] + beta[c]; conv_bias may be NULL */
int abc_bn_eval_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     const float* conv_bias, float* scale, float* bias_out, int32_t C, float eps, abc_stream_t stream);

/* Backward of [BN -> act -> (dropout) -> (maxpool)] in two passes (autograd of
 * unet.py:13-17,30,67-69):
 *   pass 1 (abc_act_bwd): G = (dA_same + unpool(dA_pooled)) * act'(y) * dropout;
 *           per-block partials of sum(G), sum(G*xhat)
 *   abc_bn_finalize_bwd: dgamma, dbeta (into the gradient arena) and k1=mean(G), k2=mean(G*xhat)
 *   pass 2 (abc_bn_apply_bwd): dY = gamma*invstd*(G - k1 - xhat*k2)   (in place on G) */
typedef struct abc_act_bwd_desc {
    const void* y_raw; int32_t ld_y;        /* raw conv output [B,H,W,*] */
    const void* dA_same; int32_t ld_same;   /* grad wrt activated tensor at full res, or NULL */
    const void* dA_pool; int32_t ld_pool;   /* grad wrt pooled activated tensor [B,H/2,W/2,*], or NULL */
    void* g; int32_t ld_g;                  /* out [B,H,W,C] */
    float* partial;                         /* [nblk][2][C] */
    const float* scale; const float* shift; const float* slope; const float* mean; const float* invstd;
    int32_t dtype, B, H, W, C, cy_off, csame_off, cpool_off;
    float drop_p; uint32_t drop_seed; int32_t drop_ld; /* dropout index = pixel*drop_ld + cy_off + c */
    const uint32_t* drop_salt;                         /* as abc_act_src.drop_salt: the SAME counter as the forward's */
} abc_act_bwd_desc;
int abc_act_bwd_blocks(const abc_act_bwd_desc* d);
int abc_act_bwd(const abc_act_bwd_desc* d, abc_stream_t stream);
typedef struct abc_bn_bwd_desc {
    const float* partial; int32_t nblk; int32_t C; double count;
    const float* gamma; const float* invstd;
    float* dgamma; float* dbeta; float* k1; float* k2; float* gscale; /* gscale = gamma*invstd */
    /* optional (all or none): the same correction as ONE per-channel affine of (g, y_raw),
     *   dY = ca*g + cb*y_raw + cc,  ca = gscale, cb = -gscale*k2*invstd, cc = gscale*(mean*invstd*k2 - k1),
     * for consumers that apply it on load (abc_wgrad_desc.p_dual) instead of a separate abc_bn_apply_bwd pass */
    const float* mean; float* ca; float* cb; float* cc;
    /* optional device scalar: the partial sums are of g / in_scale -- a producer that could not know a global factor of
     * its gradient yet (the fused heads kernel: the loss normalisers are batch sums); folded into the sums and into ca */
    const float* in_scale;
} abc_bn_bwd_desc;
int abc_bn_finalize_bwd(const abc_bn_bwd_desc* d, abc_stream_t stream);
/* n <= 8 layers in one launch; pstride > 0: their partial sums are column slices of one [nblk][2][pstride] buffer */
int abc_bn_finalize_bwd_batch(const abc_bn_bwd_desc* descs, int32_t n, int32_t pstride, abc_stream_t stream);
typedef struct abc_bn_apply_desc {
    void* g; int32_t ld_g; const void* y_raw; int32_t ld_y; int32_t cy_off;
    const float* mean; const float* invstd; const float* k1; const float* k2; const float* gscale;
    int32_t dtype, C; int64_t npix;
    void* out; int32_t ld_out;   /* NULL: dY replaces g in place; else dY goes to out[pixel * ld_out + c] and g is kept */
} abc_bn_apply_desc;
int abc_bn_apply_bwd(const abc_bn_apply_desc* d, abc_stream_t stream);

/* Weight gradient of a tap-list convolution (autograd of unet.py:12,15,44,66,70):
 *   dW[t][a][b] = sum_p P[p][a] * Q[stride*p + d_t][b]
 * regular conv: P = dY (a = cout), Q = activated input (b = cin), stride 1
 * transposed conv: P = activated input (a = cin), Q = dOut (b = cout), stride 2
 * Two stages: split-K partial slabs, then abc_wgrad_reduce sums the slabs and
 * writes the reference layout [a][b][taps] into the f32 gradient arena. */
typedef struct abc_wgrad_desc {
    abc_act_src p;  /* un-shifted operand */
    abc_act_src q;  /* shifted operand (halo) */
    float* partial; /* [nsplit][ntaps][Ca_pad][Cb_pad] */
    int32_t dtype_p, dtype_q, dtype_c;
    int32_t B, Hg, Wg;  /* grid of p */
    int32_t Hq, Wq;     /* logical dims of q */
    int32_t cp_off, Ca, cq_off, Cb;
    int32_t stride, ntaps, nsplit;
    int8_t tap_dy[ABC_MAX_TAPS_C];
    int8_t tap_dx[ABC_MAX_TAPS_C];
    /* BatchNorm-backward correction fused into the load of P (replaces abc_bn_apply_bwd for this layer):
     * p_dual = 1: P[p][a] = p.scale[a]*p.x[p][a] + p.slope[a]*p2[p][cp2_off + a] + p.shift[a]  (no activation), where p.x is
     * the act_bwd output g and p2 the layer's raw conv output; the corrected values are also stored to p_out
     * (NHWC, pixel stride ld_pout, same dtype as p) for the data-gradient conv that runs afterwards.
     * Only where abc_wgrad_fuses_apply() says so; otherwise abc_wgrad returns ABC_EUNSUPPORTED for p_dual. */
    const void* p2; int32_t ld_p2, cp2_off, p_dual; void* p_out; int32_t ld_pout;
    /* optional, heads' 1x1 kernel only (planar f32 P): per-split row sums of the transformed P, [nsplit][Ca_pad] --
     * the bias gradient of the conv (sum over pixels of dY) falls out of the operand the kernel holds anyway;
     * reduce with abc_wgrad_reduce(ntaps = 1, Cb = Cb_pad = 1).  abc_wgrad_rowsum_ok() says whether it is written. */
    float* rowsum_partial;
} abc_wgrad_desc;
int abc_wgrad_rowsum_ok(const abc_wgrad_desc* d);
int abc_wgrad_fuses_apply(const abc_wgrad_desc* d); /* 1: this descriptor (with p_dual set) is served by the fused path */
int abc_wgrad_pads(const abc_wgrad_desc* d, int32_t* ca_pad, int32_t* cb_pad);
int abc_wgrad_tile(const abc_wgrad_desc* d, int32_t* at, int32_t* bt); /* 32x32 tile pairs per workgroup: at x bt */
int abc_wgrad_blocks(const abc_wgrad_desc* d); /* workgroups per split: choose nsplit so that blocks*nsplit fills the GPU */
int abc_wgrad(const abc_wgrad_desc* d, abc_stream_t stream);
typedef struct abc_wgrad_reduce_desc {
    const float* partial; int32_t nsplit, ntaps, Ca, Cb, Ca_pad, Cb_pad;
    float* dw;    /* [Ca][Cb][ntaps] */
    int32_t accumulate; /* 1: add to dw instead of overwrite */
} abc_wgrad_reduce_desc;
/* the heads' 1x1 weight gradients of all heads in one launch (each descriptor with its own partial / rowsum slabs) */
int abc_wgrad_heads_batch(const abc_wgrad_desc* descs, int32_t n, abc_stream_t stream);
int abc_wgrad_reduce(const abc_wgrad_reduce_desc* d, abc_stream_t stream);
/* abc_wgrad_reduce(r) and abc_bn_finalize_bwd(f) of two DIFFERENT layers as one launch (both inputs complete, neither reads the other's
 * output): in the backward chain the finaliser is a dependent ~5 us launch and the slab reduction of the layer above is independent of it */
int abc_wgrad_reduce_bn_bwd(const abc_wgrad_reduce_desc* r, const abc_bn_bwd_desc* f, abc_stream_t stream);
/* up to 16 (small) reductions in one launch; results bit-identical to abc_wgrad_reduce item by item */
int abc_wgrad_reduce_batch(const abc_wgrad_reduce_desc* descs, int32_t n, abc_stream_t stream);

/* Per-channel column sums of an NHWC tensor (bias gradients of convs that do not
 * feed a BN: unet.py:44 up.bias, unet.py:70 conv2.bias), optional per-channel scale. */
int abc_colsum_blocks(int64_t npix);
int abc_colsum(const void* x, int32_t dtype, int64_t npix, int32_t ld, int32_t c_off, int32_t C,
               const float* chan_scale, float* work /* [abc_colsum_blocks(npix)][C] */, float* out, abc_stream_t stream);
/* The same pass with a second row of sums weighted by a one-channel f32 image w[npix]: out_sum[c] = sum x[p][c] (may be NULL),
 * out_w[c] = sum x[p][c] * w[p] -- bias and weight gradient of a 1x1 convolution over a one-channel input, i.e. autograd of
 * unet2.DoubleConv's res_conv in the first block (unet2.py:62,72,135: nn.Conv2d(1, 32, 1)) in ONE pass over d(out).
 * work: [abc_colsum_blocks(npix)][2][C] floats. */
int abc_colsum_w1(const void* x, int32_t dtype, int64_t npix, int32_t ld, int32_t c_off, int32_t C, const float* w,
                  float* work, float* out_sum, float* out_w, abc_stream_t stream);

/* Fused activation + loss + dlogits (train.py:95-137).  Logits and dlogits are the 8 NCHW f32
 * head maps of unet.forward (unet.py:119), targets the reference's NCHW tensors
 * (utils.py:254-300; rho/omega f64): one thread per pixel, every access coalesced.
 * Writes the UNNORMALISED per-term gradient d(numerator_i)/d(logit) and per-block
 * partials of the 8 numerators + 8 denominators; abc_loss_finalize reduces them,
 * forms the 8 terms, the uncertainty weighting with s (train.py:127-135), the
 * total, ds, and head_scale[i] = weight_i/denominator_i, replicated per channel into
 * chan_scale (head i at chan_off[i]) which the heads' backward applies on load. */
typedef struct abc_loss_desc {
    const float* logits[8]; float* dlogits[8];
    const float* t_atom; const float* t_types; const float* t_charges; const float* t_hs; const float* t_bond;
    const float* t_btypes; const double* t_rho; const double* t_omega;
    int32_t B, h, w;
    double* partial; /* [nblk][16] */
} abc_loss_desc;
int abc_loss_blocks(const abc_loss_desc* d);
int abc_loss_fwd_bwd(const abc_loss_desc* d, abc_stream_t stream);
typedef struct abc_loss_fin_desc {
    const double* partial; int32_t nblk;
    const float* s; float* ds;         /* [10] */
    double* out;                       /* [0]=total, [1+i]=weighted term of head i, [9+i]=raw term of head i
                                          (head order of unet.forward: atom_t, types, charges, hs, bond_t, btypes, rho, omega) */
    float* chan_scale; int32_t nchan;  /* [nchan] */
    int32_t chan_off[8]; int32_t head_c[8];
    float grad_scale;                  /* multiplies the scales (1/world for the DDP gradient mean) */
} abc_loss_fin_desc;
int abc_loss_finalize(const abc_loss_fin_desc* d, abc_stream_t stream);

/* per-channel sum over batch and pixels of a planar f32 tensor [B][C][HW], times chan_scale[c]:
 * bias gradient of the heads' 1x1 convs (unet.py:70) from the NCHW dlogits */
int abc_plane_sum_work(int32_t C); /* floats of workspace */
int abc_plane_sum(const float* x, int32_t B, int32_t C, int32_t HW, const float* chan_scale, float* work, float* out,
                  abc_stream_t stream);

/* torch.optim.Adam step over a flat f32 arena (train.py:55,141): L2 decay into the
 * gradient, bias correction from the device-side step counter. */
typedef struct abc_adam_desc {
    float* p; const float* g; float* m; float* v; int64_t n;
    int64_t* step; /* device scalar, incremented by the kernel */
    float lr, beta1, beta2, eps, weight_decay, grad_scale;
} abc_adam_desc;
int abc_adam_step(const abc_adam_desc* d, abc_stream_t stream);

/* The heads' 1x1 convolutions (unet.py:70, out_modules[i].conv2) of ALL heads in one launch: descs[0..n) are the same
 * descriptors abc_conv_fwd would take one by one (n <= 8, same batch and map size).  which = 0: forward into the NCHW
 * f32 logits; which = 1: data gradient from the NCHW f32 dlogits.  ABC_EUNSUPPORTED when a descriptor is not served by
 * the dedicated heads kernels (then call abc_conv_fwd per head). */
int abc_heads_batch(const abc_conv_desc* descs, int32_t n, int32_t which, abc_stream_t stream);

/* The heads' second half of a TRAINING step in one pass (bf16): out_modules[i].conv2 forward (unet.py:70, 116-118), the
 * activation + loss block with d(loss)/d(logits) (train.py:95-125, as abc_loss_fwd_bwd), conv2's data gradient, the
 * backward of Dropout and LeakyReLU (unet.py:67-69) and BatchNorm-backward partial sums, for the eight heads
 * [1,14,3,2,1,360,60,60] (train.py:47).  Replaces abc_heads_batch(which = 0) + abc_loss_fwd_bwd + abc_heads_batch(which
 * = 1) + abc_act_bwd; abc_heads_fused_wgrad replaces abc_wgrad_heads_batch.  Order of a step:
 *   abc_heads_fused_pack (weights changed) -> abc_heads_fused_fwd_bwd -> abc_loss_finalize(loss_partial, abc_heads_fused_loss_blocks())
 *   -> abc_heads_fused_wgrad, abc_bn_finalize_bwd(_batch) with in_scale = chan_scale + chan_off[i] and partial = bn_partial.
 * Everything the kernel writes is the gradient of each loss term's NUMERATOR (the normalisers are batch sums): g and the
 * BatchNorm sums lack the head's factor chan_scale[chan_off[i]], which the two consumers above apply. */
typedef struct abc_heads_fused_desc {
    const void* feat; int32_t ld;       /* raw conv1 outputs of all heads, NHWC bf16 [B*h*w][ld]; head i = channels [128 i, 128 i + 128) */
    const float* scale; const float* shift; const float* slope;   /* [ld]: BatchNorm + LeakyReLU applied on load */
    const float* mean; const float* invstd;                        /* [ld]: batch statistics (for xhat) */
    float drop_p; uint32_t drop_seed; const uint32_t* drop_salt;   /* Dropout after the activation (abc_act_src) */
    const float* w2[8]; const float* b2[8];   /* conv2.weight [C_i][128], conv2.bias [C_i] (reference layout, f32) */
    void* w2_pack;                      /* abc_heads_fused_pack_bytes() bytes, written by abc_heads_fused_pack */
    float* logits[8];                   /* out: NCHW f32 [B][C_i][h][w] (unet.py:119); an entry may be NULL when nothing
                                         * downstream reads that head's logits (they are then never stored: 0.3 GB less) */
    const float* t_atom; const float* t_types; const float* t_charges; const float* t_hs; const float* t_bond;
    const float* t_btypes; const double* t_rho; const double* t_omega;   /* targets, as abc_loss_desc */
    void* dl;                           /* out: d(numerator)/d(logits), bf16, abc_heads_fused_dl_elems() elements:
                                           per head [chunk][packed rows][128 pixels] (row order: abc_heads_fused_chan_of_row) */
    void* g;                            /* out: NHWC bf16 [B*h*w][ld]: gradient w.r.t. the BatchNorm outputs, without the head's factor */
    float* bn_partial;                  /* out: [abc_heads_fused_chunks()][2][ld]: sum g, sum g * xhat */
    double* loss_partial;               /* out: [abc_heads_fused_loss_blocks()][16], the layout abc_loss_finalize reduces */
    int32_t B, h, w;                    /* h * w a multiple of 128 */
    /* abc_heads_fused_wgrad (wgrad_work: both): */
    const float* chan_scale; int32_t chan_off[8];   /* abc_loss_finalize's factors, first channel of head i */
    float* dw2[8]; float* db2[8];       /* out: conv2.weight.grad [C_i][128], conv2.bias.grad [C_i] */
    float* wgrad_work;                  /* abc_heads_fused_wgrad_floats() floats; ALSO written by abc_heads_fused_fwd_bwd (the five
                                           small heads' weight-gradient partials), so set it for both calls */
    void* keep_mask;                    /* optional, 3 * B*h*w * 16 bytes: abc_heads_fused_fwd_bwd leaves the dropout keep bits of the three
                                           wide heads' features here and abc_heads_fused_wgrad reads them instead of hashing every
                                           element again (set it for both calls, or for neither) */
    /* optional (both or neither): abc_raster_desc.group_flags of the rasteriser that drew the target maps, and 512 zero bytes.  A wave whose
     * 32 pixels carry no target of a head reads that head's targets from the zero bytes instead of the maps (train.py:107-125 on
     * all-zero targets: identical arithmetic, no HBM traffic -- the maps hold ~62 non-zero 3x3 neighbourhoods per image) */
    const uint32_t* target_flags; const void* zero_bytes;
} abc_heads_fused_desc;
int64_t abc_heads_fused_pack_bytes(void);
int abc_heads_fused_chunks(const abc_heads_fused_desc* d);
int abc_heads_fused_loss_blocks(const abc_heads_fused_desc* d);
int64_t abc_heads_fused_dl_elems(const abc_heads_fused_desc* d);
int64_t abc_heads_fused_wgrad_floats(const abc_heads_fused_desc* d);
int abc_heads_fused_rows(int32_t head);                       /* packed rows of a head (multiple of 32) */
int abc_heads_fused_chan_of_row(int32_t head, int32_t row);   /* channel of a packed row, -1 = padding */
int abc_heads_fused_pack(const abc_heads_fused_desc* d, abc_stream_t stream);
int abc_heads_fused_fwd_bwd(const abc_heads_fused_desc* d, abc_stream_t stream);
int abc_heads_fused_wgrad(const abc_heads_fused_desc* d, abc_stream_t stream);

/* Inference NMS (img2smiles2.py:61-79) on the NCHW f32 head maps: atom/bond 3x3 peak
 * masks (logit > -1), |rho|, circular 3-tap omega peak mask; outputs NCHW f32 like the reference. */
typedef struct abc_nms_desc {
    const float* atom; const float* bond; const float* rho; const float* omega;
    int32_t B, h, w, n_omega;
    float* atom_mask; float* bond_mask; float* rho_abs; float* omega_mask;
} abc_nms_desc;
int abc_nms_peaks(const abc_nms_desc* d, abc_stream_t stream);

/* Target rasteriser (utils.py:83-228, MolecularImageDataset.__getitem__) from compact per-molecule records: zeroes the 8
 * target maps of the loss (same layouts / dtypes as abc_loss_desc) and rasterises atoms, then bonds, in record order
 * (the reference's slice assignments are order dependent).  atoms[b][i] = (x, y, type, charge, hs in {0,1} or -1);
 * bonds[b][i] = (x, y, type, omega bin, single: 1 = one direction (stereo types 4, 5), 0 = bins k and k + 30);
 * rho[b][i] float64.  x = row, y = column, inside the map.  The string parsing, vocabulary look-ups and atan of
 * utils.py:94-163 stay on the host (abcnet_amd/raster.py). */
typedef struct abc_raster_desc {
    float* t_atom; float* t_types; float* t_charges; float* t_hs; float* t_bond; float* t_btypes; double* t_rho; double* t_omega;
    int32_t B, h, w, max_atoms, max_bonds;
    const int32_t* atoms; const int32_t* n_atoms;   /* [B][max_atoms][5], [B] */
    const int32_t* bonds; const int32_t* n_bonds;   /* [B][max_bonds][5], [B] */
    const double* rho;                              /* [B][max_bonds] */
    /* Sparse use of the maps (optional; all NULL / 0 = the plain form: zero all eight maps, draw).
     *   group_flags  out, uint32 [B * h * w / 32]: bit i of word g set when head i's targets (atom 0, types 1, charges 2, hs 3, bond 4,
     *                bond types 5, rho 6, omega 7) may be non-zero somewhere in pixels [32 g, 32 g + 32) of the flattened batch -- the unit a
     *                wave of abc_heads_fused_fwd_bwd owns (abc_heads_fused_desc.target_flags); h * w a multiple of 32
     *   prev_*       device scratch of the records' shapes (atoms, bonds, rho, counts [2][B]): the records the maps currently hold.
     *   incremental  1: the maps hold exactly the drawing of prev_* (a previous call with the same buffers): ERASE those pixels instead of
     *                zeroing 23 MB per image; 0: zero everything first.  Either way prev_* <- the new records. */
    uint32_t* group_flags;
    int32_t* prev_atoms; int32_t* prev_bonds; double* prev_rho; int32_t* prev_counts;
    int32_t incremental;
} abc_raster_desc;
int abc_rasterize_targets(const abc_raster_desc* d, abc_stream_t stream);

/* Candidate extraction for the SMILES decoder (img2smiles2.py:113-191; replaces its per-pixel .cpu().item() loops):
 * from the NMS masks of abc_nms_peaks and the raw head maps (all NCHW f32) to compact ordered lists per image.
 *   atoms[b][i] = (x, y, type, charge, hs)      raster order, greedy suppression within squared distance < 4
 *   bonds[b][i] = (x, y, omega bin, type), bond_rho[b][i] = |rho|   raster order of bond peaks, bins ascending,
 *                                                 a bin kept unless its opposite direction wins (lines 141-157)
 *   counts[b]   = (atom peaks, atoms accepted, bond peaks, bond candidates) -- true totals; the lists hold at most
 *                 cap_atoms / cap_bonds entries (and at most 4096 bond peaks per image are expanded).
 * x = row, y = column as in the reference. */
typedef struct abc_extract_desc {
    const float* atom_mask; const float* bond_mask;               /* [B][1][h][w] */
    const float* types; const float* charges; const float* hs;    /* [B][14|3|2][h][w] logits */
    const float* btypes; const float* rho; const float* omega;    /* [B][360|60|60][h][w] raw maps */
    int32_t B, h, w, cap_atoms, cap_bonds;                        /* cap_atoms <= 2048 */
    int32_t* counts;      /* [B][4] */
    int32_t* atoms;       /* [B][cap_atoms][5] */
    int32_t* bonds;       /* [B][cap_bonds][4] */
    float* bond_rho;      /* [B][cap_bonds] */
    int32_t* work;        /* scratch, abc_extract_work_ints() int32 */
    uint64_t* work_masks; /* scratch, abc_extract_work_masks() uint64 */
    /* decode mode (InferenceRunner(decode=True)): the bond type of a (bin, pixel) from the heads kernel's arg-max map
     * (abc_conv_desc.head_aux_mode 3, uint8 [B][60][h][w]) instead of six raw planes -- btypes may then be NULL; rho may be the |rho|
     * map (the kernel takes the absolute value either way) */
    const uint8_t* btype_idx;
} abc_extract_desc;
int64_t abc_extract_work_ints(const abc_extract_desc* d);
int64_t abc_extract_work_masks(const abc_extract_desc* d);
int abc_extract_peaks(const abc_extract_desc* d, abc_stream_t stream);

/* The 17 training meters of train.py:145-215 (each an AverageMeter.update(num/den, den), meter.py:12-16), from the
 * NCHW f32 head maps and the targets of the loss; replaces 34 host round trips per step by one device-side table.
 * Meter order: atom_targets {precision, precision3, recall, recall3}, atom_types_acc, atom_charges_acc, atom_hs_acc,
 * bond_targets {precision, precision3, recall, recall3}, bond_types_acc, bond_rhos_mae, bond_omega {precision,
 * recall3, recall, precision3}.  last[2i], last[2i+1] = (num, den) of this batch; totals += the same. */
typedef struct abc_metrics_desc {
    const float* logits[8];
    const float* t_atom; const float* t_types; const float* t_charges; const float* t_hs; const float* t_bond;
    const float* t_btypes; const double* t_rho; const double* t_omega;
    int32_t B, h, w;
    uint8_t* peaks;    /* scratch [2][B][h][w] */
    double* partial;   /* scratch [abc_metrics_blocks][24] */
    double* totals;    /* [17][2] running (sum, count), accumulated in place */
    double* last;      /* [17][2] */
} abc_metrics_desc;
int abc_metrics_blocks(const abc_metrics_desc* d);
int abc_metrics_update(const abc_metrics_desc* d, abc_stream_t stream);

/* ---- unet2: CBAM attention + residual (unet2.py:6-74).  See csrc/cbam.hip for the pass structure. ---- */
typedef struct abc_cbam_channel_desc { /* ChannelAttentionModule (unet2.py:6-22), one MLP evaluation per image */
    const float* partial;  /* fwd: conv stats [B*tiles_per_img][4][C] (sum,sumsq,max,min of y2); bwd: [B*tiles_per_img][C] */
    int32_t tiles_per_img, B, C, mid; double HW;
    const float* scale; const float* shift;            /* BN2 affine of this block */
    const float* w1; const float* b1; const float* w2; const float* b2; /* shared_MLP.0 / .2 */
    float* ca; float* avgz; float* maxz; float* hid_avg; float* hid_max; /* [B][C], [B][C], [B][C], [B][mid] x2 */
    float* dw1; float* db1; float* dw2; float* db2; float* d_avgz; float* d_maxz; /* backward outputs */
    float* work;           /* backward scratch, B * (C + 2 mid) floats */
    /* forward outputs for the backward of AdaptiveMaxPool2d(1) (unet2.py:10,20): ext[n][c] = the extreme RAW value of y2 whose BN image is
     * max(z) (max of y2 for a positive BN scale, min for a negative one); first[n][c] is reset to INT32_MAX here and lowered by
     * abc_cbam_spatial_stats to the first pixel (row-major) holding that value -- the one torch routes the gradient to */
    float* ext; int32_t* first;
} abc_cbam_channel_desc;
int abc_cbam_channel_fwd(const abc_cbam_channel_desc* d, abc_stream_t stream);
int abc_cbam_channel_bwd(const abc_cbam_channel_desc* d, abc_stream_t stream);
/* abc_cbam_channel_bwd + the pending reduction of c7->dw_partial into c7->dw7 / c7->db7 (see abc_cbam_conv7_bwd_partial) */
struct abc_cbam_conv7_desc;
int abc_cbam_channel_bwd_c7(const abc_cbam_channel_desc* d, const struct abc_cbam_conv7_desc* c7, abc_stream_t stream);

typedef struct abc_cbam_pix_desc { /* per-pixel passes of SpatialAttentionModule / CBAM / residual (unet2.py:24-74) */
    const void* y; int32_t ld_y, cy_off;               /* raw second-conv output y2 */
    const float* scale; const float* shift; const float* mean; const float* invstd;
    const float* ca; const float* maxz; const float* d_avgz; const float* d_maxz;
    const float* sa; float* st; int32_t* amax; float* du; const float* dst;
    const void* res; int32_t ld_res, cres_off, res_pool; /* residual r (res_pool: 2x2 max of a 2x tensor) */
    void* out; int32_t ld_out, cout_off;               /* block output relu(sa*ca*z + r) */
    const void* d_same; int32_t ld_same, csame_off;    /* gradient sources wrt `out` */
    const void* d_pool; int32_t ld_pool, cpool_off;
    void* g; int32_t ld_g;                             /* dOut*[out>0] (the residual branch's gradient) */
    void* dz; int32_t ld_dz;                           /* d_o1, then d_z in place */
    float* partial;
    int32_t dtype, B, H, W, C;
    const float* ext; int32_t* first;  /* see abc_cbam_channel_desc: spatial_stats lowers first[n][c], bwd3 adds d_maxz at that pixel only */
} abc_cbam_pix_desc;
int abc_cbam_spatial_stats(const abc_cbam_pix_desc* d, abc_stream_t stream); /* y,ca -> st[B,H,W,2], amax */
int abc_cbam_apply_fwd(const abc_cbam_pix_desc* d, abc_stream_t stream);     /* -> out */
int abc_cbam_bwd1(const abc_cbam_pix_desc* d, abc_stream_t stream);          /* -> g, du */
int abc_cbam_bwd2_blocks(const abc_cbam_pix_desc* d);                        /* workgroups per image */
int abc_cbam_bwd2(const abc_cbam_pix_desc* d, abc_stream_t stream);          /* -> dz = d_o1, partial [B][blocks][C] */
int abc_cbam_bwd3_blocks(const abc_cbam_pix_desc* d);
int abc_cbam_bwd3(const abc_cbam_pix_desc* d, abc_stream_t stream);          /* dz -> d_z in place, BN partial [blocks][2][C] */

typedef struct abc_cbam_conv7_desc { /* SpatialAttentionModule.conv2d 7x7 (2->1) + sigmoid (unet2.py:27,34) */
    const float* st; const float* w7; const float* b7; float* sa;
    const float* du; float* dst; float* dw_partial; float* dw7; float* db7; /* backward */
    int32_t B, H, W;
} abc_cbam_conv7_desc;
int abc_cbam_conv7_fwd(const abc_cbam_conv7_desc* d, abc_stream_t stream);
int abc_cbam_conv7_blocks(const abc_cbam_conv7_desc* d); /* dw_partial = [blocks][99] */
int abc_cbam_conv7_bwd(const abc_cbam_conv7_desc* d, abc_stream_t stream);
/* the same without the reduction of dw_partial: abc_cbam_channel_bwd_c7 of the same block performs it inside its first launch
 * (the reduction is independent of the channel attention's backward and was a ~5 us launch of its own, 13 per step of unet2.py) */
int abc_cbam_conv7_bwd_partial(const abc_cbam_conv7_desc* d, abc_stream_t stream);

/* dst[.., cdst_off + c] += src[.., csrc_off + c] (identity residual gradient, unet2.py:62) */
int abc_add_into(void* dst, int32_t ld_dst, int32_t cdst_off, const void* src, int32_t ld_src, int32_t csrc_off, int32_t C,
                 int64_t npix, int32_t dtype, abc_stream_t stream);

/* nn.MaxPool2d(2) of an activated tensor (unet.py:30), materialised once: out[b][y][x][c] = max over the 2x2 window
 * of act(src[b][2y+dy][2x+dx][c_off + c]) (src->pool is ignored: src is read at full resolution).  The encoder's
 * first convolution of every level and its weight gradient then read a plain NHWC tensor on their prefetch paths
 * instead of pooling on load twice. */
int abc_pool_act(const abc_act_src* src, int32_t dtype_in, int32_t c_off, int32_t C, int32_t B, void* out, int32_t dtype_out,
                 int32_t ld_out, abc_stream_t stream);

/* layout conversion helpers (multi-channel NCHW input images -> NHWC) */
int abc_nhwc_to_nchw_f32(const float* src, int32_t ld, int32_t c_off, int32_t C, int32_t B, int32_t H, int32_t W,
                         float* dst, abc_stream_t stream);
int abc_nchw_to_nhwc_f32(const float* src, int32_t C, int32_t B, int32_t H, int32_t W, float* dst, int32_t ld,
                         int32_t c_off, abc_stream_t stream);
int abc_fill_f32(float* p, float v, int64_t n, abc_stream_t stream);
/* dst = srcs[0][0:counts[0]] ++ srcs[1][0:counts[1]] ++ ... (n <= 16 device arrays; `srcs` / `counts` are HOST arrays read at
 * call time): the eight heads' conv1 biases (unet.py:66) side by side for the one merged convolution */
int abc_concat_f32(const float* const* srcs, const int32_t* counts, int32_t n, float* dst, abc_stream_t stream);
/* *p += inc (one thread): the per-step dropout salt */
int abc_counter_add_u32(uint32_t* p, uint32_t inc, abc_stream_t stream);

/* sizeof(descriptor #which) in declaration order (abc_act_src = 0 ... abc_nms_desc = 12, abc_cbam_channel_desc = 13, abc_cbam_pix_desc = 14, abc_cbam_conv7_desc = 15, abc_metrics_desc = 16, abc_extract_desc = 17, abc_raster_desc = 18, abc_heads_fused_desc = 19, abc_heads_epi = 20, abc_convt_desc = 21):
 * lets a foreign-language binding check its mirror structs at load time */
int abc_sizeof(int which);
const char* abc_last_error(void);
int abc_version(void);
/* Leave `n` (rounded up to a multiple of 4; 0 <= n <= 128) of the 256 compute units out of the PERSISTENT convolution grids
 * (the launches sized to 2 or 3 workgroups per CU): process-wide, read when a launch sizes its grid -- set it before a plan is
 * built or captured.  For the data-parallel step (multi_gpu_train.py:44-53 / DDP's NCCL kernels beside backward): two 256-VGPR
 * workgroups per CU leave no registers for a communication kernel, which would wait for a persistent workgroup to drain. */
int abc_set_reserved_cus(int32_t n);
int abc_get_reserved_cus(void);

#ifdef __cplusplus
}
#endif
#endif
