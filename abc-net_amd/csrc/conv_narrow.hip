// 3x3 convolution of a PLAIN bf16 NHWC tensor with 16 or 32 input channels into 16 / 32 output channels: the narrow
// levels of the inference graph (BatchNorm folded, img2smiles2.py:42-49 over unet.py:12,15) and the data gradients of the
// narrow levels in training (unet.py:12,15 under autograd).  These layers are HBM-bound (10 KB in, 8 KB out per 256
// pixels) and the general lean kernel (conv_fast.hip) runs them at 2 TB/s: ~600 VALU instructions per wave and tile for
// staging with the transform on load, weight fragments through LDS, an LDS-transposed epilogue, two workgroup barriers.
// Nothing of that is needed when the input is a finished tensor:
//
//   * a WAVE owns an 8 x 16 pixel tile (four 32-pixel MFMA tiles) and walks a contiguous run of tiles: no workgroup
//     barriers at all, only its own LDS queue to wait for;
//   * the halo (10 x 18 pixels) goes global -> registers -> LDS untouched (6 / 12 16-byte loads per lane, issued one tile
//     AHEAD: they land under the current tile's MFMAs and stores);
//   * the weights live in REGISTERS for the whole kernel (9 taps x 1-2 K-steps = 36 / 72 VGPRs): the inner loop is one
//     conflict-free ds_read_b128 + one MFMA per (tile, tap, K-step), tap offsets as immediates;
//   * the MFMA runs "transposed" (A = weights, B = pixels), so a lane of the result holds ONE PIXEL and its registers
//     channels; with the weight rows permuted (bits 2 and 3 swapped) register k of lane half h is channel
//     (k & 7) + 8 h + 16 (k >> 3): bias, activation and 16-byte stores straight from the accumulators.
//
// ~250 instructions per wave and 128 pixels.  Served: stride 1, taps within +-1, unit output stride, no statistics.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "conv_fast.hpp"
#include <stdlib.h>

namespace {

struct NarrowK {
    const bf16* x; const bf16* w; const float* bias; bf16* y;
    const float *sc, *sh, *sl;                 // XF: previous layer's BatchNorm + activation applied on load (absolute channel index)
    // STEM: the input tensor is never materialised -- it is relu-like(conv3x3(one-channel f32 image) * scale + bias), computed into the halo
    const float *stem_x, *stem_w, *stem_scale, *stem_bias; float stem_slope;
    bf16* pool_y; int ld_pool;                 // optional: 2x2 max-pool of the stored tensor, [B][H/2][W/2][ld_pool]
    float* stats;                              // NST: [grid][2][Cout] sum, sum of squares of the f32 outputs (one row per workgroup)
    int B, H, W, ldx, cin_off, ldy, cout_off, Cout;
    int tiles_x, tiles_y, ntiles, tpw;         // tiles per wave (contiguous runs)
    int out_act; float out_slope;
    unsigned bytesX;
    // ACTB (abc_conv_desc.actbwd_*): this data gradient is d(activation output) of the producing 16-channel layer; the epilogue stores
    // d(BatchNorm output) and sums that layer's BatchNorm-backward statistics (bn_act.hip's act_bwd pass, not run)
    const bf16* ab_y; int ab_ld; unsigned bytesY; int cf_off;      // cf_off: LDS offset of the [4][16] coefficient table
    const float *ab_sc, *ab_sh, *ab_sl, *ab_mu, *ab_is;
    int8_t ty[25], tx[25];                     // tap offsets + R (0 .. 2 R)
};

// XF: transform on load; NST: 0 = no statistics, 1 / 2 = BatchNorm partial sums for Cout <= 16 / <= 32 (per-lane running sums
// over all tiles of the wave, reduced once at the end)
// R: tap radius (1: 3x3, nine taps; 2: 5x5, 25 taps -- unet2.py's 32-channel levels: weights in LDS, one workgroup per CU)
// STEM (inference, 16 channels): the halo is not loaded but COMPUTED from the one-channel image -- the network's first
// convolution + folded BatchNorm + ReLU (unet.py:12-14) fused in front of its second one: the 16-channel full-resolution
// tensor between them (0.5 GB written + read at 512 x 512, batch 64) never exists
template <int CK, bool XF, int NST, int R = 1, bool STEM = false, bool ACTB = false>
__global__ __launch_bounds__(256, R == 1 ? 2 : 1) void conv_narrow_kernel(const NarrowK a) {
    static_assert(!STEM || (CK == 16 && !XF && R == 1), "stem fusion: 16 channels, 3x3");
    static_assert(!ACTB || (CK == 16 && !XF && R == 1 && !STEM && NST == 1), "act_bwd epilogue: the plain 16-channel 3x3 data gradient");
    constexpr int NTAP = (2 * R + 1) * (2 * R + 1), HR = 8 + 2 * R, HC = 16 + 2 * R;
    constexpr int CKB = CK * 2;                // bytes of a pixel's channels
    constexpr int PS = CKB + 16;               // padded pixel stride: 16 consecutive pixels = 16 distinct 16-byte bank slots
    constexpr int RS = HC * PS;
    constexpr int SEGS = CKB / 16;             // 16-byte segments per pixel
    constexpr int NSEG = HR * HC * SEGS;       // of the halo
    constexpr int NL = (NSEG + 63) / 64;       // loads per lane
    constexpr int KS = CK / 16;                // K-steps per tap
    constexpr int LHB = CKB / 2;               // a lane half's bytes of a pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    char* halo = smem + wave * (HR * RS);

    // ---- weights: A fragments, row = output channel with bits 2 and 3 of the lane swapped (see above).  16 input channels:
    // 36 registers for the whole kernel; 32: 72 would not leave room for the prefetch -- they sit in LDS in fragment order
    // ([tap][K-step][lane] x 16 bytes: one conflict-free ds_read_b128 per four MFMAs)
    const int rw = (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1);
    constexpr bool WREG = CK == 16 && R == 1 && !STEM && !ACTB;   // (STEM keeps the FIRST convolution's weights in registers instead; ACTB needs the 36 registers for the y_raw prefetch and the coefficients)
    bf16x8 wf[WREG ? 9 : 1][WREG ? KS : 1];
    char* swt = smem + 4 * HR * RS + 128;
    if constexpr (WREG) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[t][ks] = *(const bf16x8*)(a.w + ((size_t)(t * 32 + rw) * CK + h * (LHB / 2) + 8 * ks));
    } else {
        for (int s2 = wave; s2 < NTAP * KS; s2 += 4)
            *(bf16x8*)(swt + (s2 * 64 + lane) * 16) = *(const bf16x8*)(a.w + ((size_t)((s2 / KS) * 32 + rw) * CK + h * (LHB / 2) + 8 * (s2 % KS)));
    }
    // bias -> LDS (read back per tile as the accumulators' initial value: register k <-> channel (k & 7) + 8 h + 16 (k >> 3))
    float* sbias = (float*)(smem + 4 * HR * RS);
    if (threadIdx.x < 32) sbias[threadIdx.x] = (a.bias != nullptr && (int)threadIdx.x < a.Cout) ? a.bias[threadIdx.x] : 0.f;
    if constexpr (ACTB) {
        if (threadIdx.x >= 64 && threadIdx.x < 128) {
            const int which = (threadIdx.x - 64) >> 4, n = threadIdx.x & 15;
            const float* src = which == 0 ? a.ab_sc : (which == 1 ? a.ab_sh : (which == 2 ? a.ab_sl : a.ab_mu));
            ((float*)(smem + a.cf_off))[which * 16 + n] = n < a.Cout ? src[n] : 0.f;
        }
    }
    // STEM: folded first-layer weights [tap][16 channels] + bias [16] (f32), and a 12 x 20 image patch per wave
    float* stw = (float*)(smem + 4 * HR * RS + 128 + (WREG ? 0 : NTAP * KS * 1024));
    float* simg = stw + 160 + wave * 240;
    if constexpr (STEM) {
        if (threadIdx.x < 144) { const int t = threadIdx.x / 16, ch = threadIdx.x % 16; stw[t * 16 + ch] = a.stem_w[ch * 9 + t] * a.stem_scale[ch]; }
        else if (threadIdx.x < 160) stw[threadIdx.x] = a.stem_bias[threadIdx.x - 144];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX);
    // STEM: the first convolution's folded weights of this lane's 8 channels (a lane always stages the same channel group) and
    // their bias, in registers: 80 VGPRs that turn the halo computation into 9 LDS reads + 72 FMAs per pixel segment
    float sw[STEM ? 10 : 1][8];
    if constexpr (STEM) {
        const int c0 = ((lane % SEGS) * 16) >> 1;
#pragma unroll
        for (int t = 0; t < 10; ++t) LoadVec<float, 8>::ld(stw + t * 16 + c0, sw[t]);
    }

    // segment i of this lane = 16-byte part sg of halo pixel p0 + (64 / SEGS) i (the same for every tile); row / column by a
    // multiply (pixel < 192), recomputed where needed: as stored arrays they cost 18-36 registers the kernel does not have
    const int sg16 = (lane % SEGS) * 16, p0 = lane / SEGS;
    // (a lane always stages the SAME 8 channels: their coefficients stay in registers)
    float csc[XF ? 8 : 1], csh[XF ? 8 : 1], csl[XF ? 8 : 1];
    if constexpr (XF) {
        const int c0 = a.cin_off + (sg16 >> 1);
        LoadVec<float, 8>::ld(a.sc + c0, csc); LoadVec<float, 8>::ld(a.sh + c0, csh); LoadVec<float, 8>::ld(a.sl + c0, csl);
    }
    float st1[NST ? 8 * NST : 1], st2[NST ? 8 * NST : 1];
#pragma unroll
    for (int k = 0; k < (NST ? 8 * NST : 1); ++k) { st1[k] = 0.f; st2[k] = 0.f; }
    // ACTB: a lane's accumulator registers 0..7 are channels 8 h .. 8 h + 7 of ITS pixel -- the producer's y_raw of that pixel is one
    // 16-byte load at the store's own address pattern (no transposition).  The layer's coefficients sit in LDS ([scale | shift | slope |
    // mean][16], written before the barrier above) and are read per tile in the epilogue: as registers through the MFMA loop they
    // cost 32 VGPRs the kernel does not have (61 spilled)
    __amdgpu_buffer_rsrc_t rsYR = abc_make_rsrc(a.x, 0u);
    if constexpr (ACTB) rsYR = abc_make_rsrc(a.ab_y, a.bytesY);
    auto seg_rc = [&](int i, int& hr, int& hc) {
        const int pix = p0 + (64 / SEGS) * i;
        hr = (pix * (R == 1 ? 3641 : 3277)) >> 16;          // pix / HC (18 or 20) for pix < 1024
        hc = pix - HC * hr;
    };
    const int wid = blockIdx.x * 4 + wave;
    const int t0 = wid * a.tpw, t1 = min(t0 + a.tpw, a.ntiles);
    const int wldx2 = a.W * a.ldx * 2, ldx2 = a.ldx * 2;
    u32x4 pre[STEM ? 1 : NL];
    float pimg[STEM ? 4 : 1];
    auto issue = [&](int tile) {
        if constexpr (STEM) {
            // the 12 x 20 patch of the image under the halo (one pixel wider on every side), 240 values over 64 lanes
            int id = tile;
            const int tx_i = id % a.tiles_x; id /= a.tiles_x;
            const int ty_i = id % a.tiles_y;
            const int b = id / a.tiles_y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = lane + 64 * j;
                const int pr = (v * 3277) >> 16, pc = v - 20 * pr;
                const int iy = ty_i * 8 - 2 + pr, ix = tx_i * 16 - 2 + pc;
                const bool ok = tile < t1 && v < 240 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                pimg[j] = ok ? a.stem_x[((size_t)b * a.H + iy) * a.W + ix] : 0.f;
            }
            return;
        }
        int id = tile;
        const int tx_i = id % a.tiles_x; id /= a.tiles_x;
        const int ty_i = id % a.tiles_y;
        const int b = id / a.tiles_y;
        const int iy0 = ty_i * 8 - R, ix0 = tx_i * 16 - R;
        const int tbase = (((b * a.H + iy0) * a.W + ix0) * a.ldx + a.cin_off) * 2 + sg16;   // (may point before the image: masked below)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int hr, hc;
            seg_rc(i, hr, hc);
            const int iy = iy0 + hr, ix = ix0 + hc;
            const bool ok = tile < t1 && hr < HR && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? (unsigned)(tbase + hr * wldx2 + hc * ldx2) : 0x80000000u, 0, 0);   // (out of range: zeros = the padding)
        }
    };
    if (t0 < t1) issue(t0);
    for (int tile = t0; tile < t1; ++tile) {
        if constexpr (STEM) {
            int id2 = tile;
            const int tx2 = id2 % a.tiles_x; id2 /= a.tiles_x;
            const int ty2 = id2 % a.tiles_y;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j < 240) simg[lane + 64 * j] = pimg[j];
            issue(tile + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // halo pixel (hr, hc), channels 8 sg .. 8 sg + 7 = first convolution at image position (iy, ix); zero outside the
            // image (the SECOND convolution's padding)
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                int hr, hc;
                seg_rc(i, hr, hc);
                if (hr < HR) {
                    const int iy = ty2 * 8 - 1 + hr, ix = tx2 * 16 - 1 + hc;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = sw[9][j];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float xv = simg[(hr + t / 3) * 20 + hc + t % 3];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaf(xv, sw[t][j], v[j]);
                    }
                    const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = in ? fmaxf(v[j], a.stem_slope * v[j]) : 0.f;
                    *(bf16x8*)(halo + hr * RS + hc * PS + sg16) = pack_frag<bf16>(v);
                }
            }
        }
        // ---- this tile's halo: registers -> LDS (the wave's previous reads are ahead of these writes in its LDS queue)
#pragma unroll
        for (int i = 0; i < (STEM ? 0 : NL); ++i) {
            int hr, hc;
            seg_rc(i, hr, hc);
            if constexpr (XF) {
                // BN + activation of the producer; the zero padding applies to the ACTIVATED tensor: out-of-image stays 0
                int id2 = tile;
                const int tx2 = id2 % a.tiles_x; id2 /= a.tiles_x;
                const int ty2 = id2 % a.tiles_y;
                const int iy = ty2 * 8 - R + hr, ix = tx2 * 16 - R + hc;
                const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(pre[i][j] << 16); v[2 * j + 1] = __uint_as_float(pre[i][j] & 0xFFFF0000u); }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = in ? abc_act(v[j], csc[j], csh[j], csl[j]) : 0.f;
                if (hr < HR) *(bf16x8*)(halo + hr * RS + hc * PS + sg16) = pack_frag<bf16>(v);
            } else {
                if (hr < HR) *(u32x4*)(halo + hr * RS + hc * PS + sg16) = pre[i];
            }
        }
        if constexpr (!STEM) issue(tile + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int id = tile;
        const int tx_i = id % a.tiles_x; id /= a.tiles_x;
        const int ty_i = id % a.tiles_y;
        const int b = id / a.tiles_y;
        u32x4 yq[ACTB ? 4 : 1];
        if constexpr (ACTB) {
            // (issued here: they land under the tile's 36 MFMAs)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gy = ty_i * 8 + 2 * i + (r >> 4), gx = tx_i * 16 + (r & 15);
                const bool ok = gy < a.H && gx < a.W && 8 * h < a.Cout;
                yq[i] = __builtin_amdgcn_raw_buffer_load_b128(rsYR, ok ? (unsigned)((((b * a.H + gy) * a.W + gx) * a.ab_ld + 8 * h) * 2) : 0x80000000u, 0, 0);
            }
        }
        f32x16 acc[4];
        {
            const f32x4 b0 = *(const f32x4*)(sbias + 8 * h), b1 = *(const f32x4*)(sbias + 8 * h + 4),
                        b2 = *(const f32x4*)(sbias + 16 + 8 * h), b3 = *(const f32x4*)(sbias + 16 + 8 * h + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { acc[i][k] = b0[k]; acc[i][4 + k] = b1[k]; acc[i][8 + k] = b2[k]; acc[i][12 + k] = b3[k]; }
        }
        const char* base = halo + (r >> 4) * RS + (r & 15) * PS + h * LHB;
        // K-steps (tap, 16-channel slice) software-pipelined by one: the four fragment reads of step s + 1 are issued in front
        // of the four MFMAs of step s (two named register sets; without the fences the scheduler hoists all 36-72 reads and spills)
        constexpr int NS = NTAP * KS;
        auto frag_off = [&](int s2) { const int t = s2 / KS, ks = s2 % KS; return a.ty[t] * RS + a.tx[t] * PS + 16 * ks; };
        bf16x8 pa[4], pb[4], wa, wb;
        auto wfrag = [&](int s2) -> bf16x8 {
            if constexpr (WREG) return wf[s2 / KS][s2 % KS];
            else return *(const bf16x8*)(swt + (s2 * 64 + lane) * 16);
        };
        {
            const int o = frag_off(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) pa[i] = *(const bf16x8*)(base + o + 2 * i * RS);
            wa = wfrag(0);
        }
#pragma unroll
        for (int s2 = 0; s2 < NS; s2 += 2) {
            if (s2 + 1 < NS) {
                const int o = frag_off(s2 + 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) pb[i] = *(const bf16x8*)(base + o + 2 * i * RS);
                wb = wfrag(s2 + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, pa[i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s2 + 1 < NS) {
                if (s2 + 2 < NS) {
                    const int o = frag_off(s2 + 2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) pa[i] = *(const bf16x8*)(base + o + 2 * i * RS);
                    wa = wfrag(s2 + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, pb[i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: lane = pixel r of each 32-pixel tile, registers = channels
        float asc[ACTB ? 8 : 1], ash[ACTB ? 8 : 1], asl[ACTB ? 8 : 1], amu[ACTB ? 8 : 1];
        if constexpr (ACTB) {
            const float* scf = (const float*)(smem + abc_launder(a.cf_off + 32 * h));     // (laundered: the reads are loop-invariant and would be hoisted)
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const f32x4 t0 = *(const f32x4*)(scf + 0 * 16 + 4 * q4), t1 = *(const f32x4*)(scf + 1 * 16 + 4 * q4),
                            t2 = *(const f32x4*)(scf + 2 * 16 + 4 * q4), t3 = *(const f32x4*)(scf + 3 * 16 + 4 * q4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { asc[4 * q4 + k] = t0[k]; ash[4 * q4 + k] = t1[k]; asl[4 * q4 + k] = t2[k]; amu[4 * q4 + k] = t3[k]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gy = ty_i * 8 + 2 * i + (r >> 4), gx = tx_i * 16 + (r & 15);
            const bool valid = gy < a.H && gx < a.W;
            bf16* dst = a.y + ((size_t)(b * a.H + gy) * a.W + gx) * a.ldy + a.cout_off + 8 * h;
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                if (16 * g8 < a.Cout) {        // (uniform: both lane halves run the shuffles below)
                    float vo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float v = acc[i][8 * g8 + q];
                        if constexpr (ACTB) {
                            // g = dA where BatchNorm(y_raw) > 0, slope * dA elsewhere; sums of g and g (y_raw - mean)  (channels 0..15 only)
                            const unsigned w2 = yq[i][q >> 1];
                            const float x = __uint_as_float((q & 1) ? (w2 & 0xFFFF0000u) : (w2 << 16));
                            const float gg = v * (fmaf(x, asc[q], ash[q]) > 0.f ? 1.f : asl[q]);
                            if (g8 == 0 && valid) { st1[q] += gg; st2[q] = fmaf(gg, x - amu[q], st2[q]); }
                            vo[q] = gg;
                        } else {
                            if constexpr (NST > 0) {
                                if (g8 < NST && valid) { st1[8 * (g8 < NST ? g8 : 0) + q] += v; st2[8 * (g8 < NST ? g8 : 0) + q] = fmaf(v, v, st2[8 * (g8 < NST ? g8 : 0) + q]); }
                            }
                            vo[q] = a.out_act ? fmaxf(v, a.out_slope * v) : v;
                        }
                    }
                    const bool chan = 16 * g8 + 8 * h < a.Cout;
                    if (valid && chan) *(bf16x8*)(dst + 16 * g8) = pack_frag<bf16>(vo);
                    if (a.pool_y != nullptr) {
                        // the 2x2 window of this 32-pixel tile (rows 2 i, 2 i + 1): lanes r, r ^ 1, r ^ 16, r ^ 17
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            float m = fmaxf(vo[q], __shfl_xor(vo[q], 1));
                            vo[q] = fmaxf(m, __shfl_xor(m, 16));
                        }
                        const int py = ty_i * 4 + i, px = tx_i * 8 + ((r & 15) >> 1);
                        if ((r & 17) == 0 && chan && 2 * py + 1 < a.H && 2 * px + 1 < a.W)
                            *(bf16x8*)(a.pool_y + ((size_t)(b * (a.H >> 1) + py) * (a.W >> 1) + px) * a.ld_pool + 16 * g8 + 8 * h) = pack_frag<bf16>(vo);
                    }
                }
            }
        }
    }
    if constexpr (NST > 0) {
        // per-lane sums -> per-channel sums of the wave (lanes of one half hold the same channels, one pixel column each),
        // then the four waves through LDS: ONE partial row per workgroup (abc_conv_stat_blocks = the grid)
        __syncthreads();                       // (every wave is done with its halo: the sums reuse wave 0's)
        float* red = (float*)smem;             // [4 waves][2][32]
#pragma unroll
        for (int k = 0; k < 8 * NST; ++k) {
            float v1 = st1[k], v2 = st2[k];
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) { v1 += __shfl_xor(v1, m); v2 += __shfl_xor(v2, m); }
            if (r == 0) {
                const int n = (k & 7) + 8 * h + 16 * (k >> 3);
                red[(wave * 2 + 0) * 32 + n] = v1;
                red[(wave * 2 + 1) * 32 + n] = v2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64 && (int)(threadIdx.x & 31) < a.Cout) {
            const int row = threadIdx.x >> 5, n = threadIdx.x & 31;
            float v = (red[(0 * 2 + row) * 32 + n] + red[(1 * 2 + row) * 32 + n]) + (red[(2 * 2 + row) * 32 + n] + red[(3 * 2 + row) * 32 + n]);
            if constexpr (ACTB) { if (row == 1) v *= a.ab_is[n]; }      // (the row act_bwd writes: sum of g (y_raw - mean) / std)
            a.stats[((size_t)blockIdx.x * 2 + row) * a.Cout + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// The 16 -> 16 channel training forward (unet.py:12,15 at full resolution: previous layer's BatchNorm + activation on load, raw
// output + BatchNorm partial sums out) on v_mfma_f32_16x16x32_bf16 (round 4).  The kernel above is bound by the bytes it keeps in
// flight: 256 registers (64 accumulators, 36 weight registers, coefficients, one tile of prefetch) = two waves per SIMD, 6 KB
// each -- 3.46 TB/s, 51 us for 151 MB.  With the smaller MFMA shape K = 32 is TWO taps of 16 input channels: nine taps = five
// MFMAs per 16-pixel row (20 weight registers), a row's result is ONE 4-register accumulator that is stored before the next row
// starts (lane = pixel, its registers = four consecutive channels: 8-byte stores, 512 contiguous bytes per row), so the kernel
// needs ~110 registers: four waves per SIMD, each with the next tile's halo in flight.  The pixel operand is read from the plain
// [pixel][channel] halo image (a lane's 8 k-values = 8 channels of one tap at its pixel: one aligned ds_read_b128).
struct N16K {
    const bf16* x; const bf16* w; const float* bias; bf16* y;
    const float *sc, *sh, *sl;
    float* stats;                               // [grid][2][16] or null
    int B, H, W, ldx, cin_off, ldy, cout_off, tiles_x, tiles_y, ntiles, tpw;
    unsigned bytesX;
    int out_act; float out_slope;               // XF = false (the folded inference graph): activation in the epilogue ...
    bf16* pool_y; int ld_pool;                  // ... and optionally the 2x2 max-pool of the stored tensor, [B][H/2][W/2][ld_pool]
};

template <bool XF>
__global__ __launch_bounds__(256, 4) void conv_n16_kernel(const N16K a) {
    // halo image per wave: [channel half][10 rows][18 columns] x 16 bytes, the second half 3072 bytes (a multiple of 256) behind the
    // first: a ds_read_b128 lane group takes its lanes from BOTH halves (K groups 0 and 1) -- 16 consecutive pixels of a row then sit
    // on 16 distinct 16-byte slots (interleaved [pixel][32 bytes + pad] put five of sixteen on a slot already taken)
    constexpr int PS = 16, RS = 18 * PS, HP = 3072, HB = 2 * HP;
    __shared__ __attribute__((aligned(16))) char smem[4 * HB];
    __shared__ __attribute__((aligned(16))) float scoef[3][16];
    __shared__ float sbias[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* halo = smem + wave * HB;
    if (XF && threadIdx.x < 48) { const int w = threadIdx.x >> 4, c = threadIdx.x & 15; scoef[w][c] = (w == 0 ? a.sc : (w == 1 ? a.sh : a.sl))[a.cin_off + c]; }
    if (threadIdx.x >= 64 && threadIdx.x < 80) sbias[threadIdx.x - 64] = a.bias != nullptr ? a.bias[threadIdx.x - 64] : 0.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX);
    // compute role: pixel column n of a row, K group kg: taps (2 p, 2 p + 1) of MFMA p -- kg < 2 the first, kg >= 2 the second -- and
    // input channels 8 (kg & 1) .. + 7; output channels 4 kg .. 4 kg + 3 of its pixel
    const int n = lane & 15, kg = lane >> 4;
    bf16x8 wfr[5];
    int boff[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int t = 2 * p + (kg >> 1);
        const int tc = t < 9 ? t : 8;
        // A fragment: row = output channel n (lane & 15), k = (tap, input channel): packed weights [tap][32 rows][16]
        const bf16x8 wv = *(const bf16x8*)(a.w + ((size_t)(tc * 32 + n) * 16 + 8 * (kg & 1)));
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        wfr[p] = t < 9 ? wv : z;
        boff[p] = (tc / 3) * RS + (tc % 3) * PS + HP * (kg & 1);      // (the padding tap reads tap 8's finite values against zero weights)
    }
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = sbias[4 * kg + i];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // staging role: halo segment s = lane + 64 i (s < 360): pixel s >> 1 (row / 18, column % 18), channels 8 (lane & 1) .. + 7
    const int half = lane & 1;
    u32x4 pre[6];
    const int wid = blockIdx.x * 4 + wave;
    const int t0 = wid * a.tpw, t1 = min(t0 + a.tpw, a.ntiles);
    auto issue = [&](int tile) {
        int id = tile;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int b = id / a.tiles_y;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = (lane >> 1) + 32 * i;
            const int hr = (q * 3641) >> 16, hc = q - 18 * hr;
            const int iy = ty * 8 - 1 + hr, ix = tx * 16 - 1 + hc;
            const bool ok = tile < t1 && q < 180 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)a.ldx + (unsigned)(a.cin_off + 8 * half)) * 2u : 0x80000000u, 0, 0);
        }
    };
    if (t0 < t1) issue(t0);
    for (int tile = t0; tile < t1; ++tile) {
        int id = tile;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int b = id / a.tiles_y;
        const int y0 = ty * 8, x0 = tx * 16;
        if constexpr (XF) {
            float csc[8], csh[8], csl[8];
            LoadVec<float, 8>::ld(&scoef[0][8 * half], csc); LoadVec<float, 8>::ld(&scoef[1][8 * half], csh); LoadVec<float, 8>::ld(&scoef[2][8 * half], csl);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int q = (lane >> 1) + 32 * i;
                const int hr = (q * 3641) >> 16, hc = q - 18 * hr;
                const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
                const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(pre[i][j] << 16); v[2 * j + 1] = __uint_as_float(pre[i][j] & 0xFFFF0000u); }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = in ? abc_act(v[j], csc[j], csh[j], csl[j]) : 0.f;     // (the zero padding applies to the ACTIVATED tensor)
                if (q < 180) *(bf16x8*)(halo + HP * half + hr * RS + hc * PS) = pack_frag<bf16>(v);
            }
        } else {
            // a finished tensor: registers -> LDS untouched (out-of-image loads returned zeros = the padding)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int q = (lane >> 1) + 32 * i;
                const int hr = (q * 3641) >> 16, hc = q - 18 * hr;
                if (q < 180) *(u32x4*)(halo + HP * half + hr * RS + hc * PS) = pre[i];
            }
        }
        issue(tile + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const char* base = halo + n * PS;
        bf16* dst = a.y + ((size_t)(b * a.H + y0) * a.W + x0 + n) * a.ldy + a.cout_off + 4 * kg;
        // two rows at a time: two independent chains of five MFMAs
#pragma unroll 1
        for (int r2 = 0; r2 < 8; r2 += 2) {
            f32x4 acc0 = (f32x4){bv[0], bv[1], bv[2], bv[3]}, acc1 = acc0;
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                const bf16x8 f0 = *(const bf16x8*)(base + r2 * RS + boff[p]);
                const bf16x8 f1 = *(const bf16x8*)(base + (r2 + 1) * RS + boff[p]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[p], f0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[p], f1, acc1, 0, 0, 0);
            }
            bf16x4 o0, o1;
            if (XF || a.stats != nullptr) {      // (the sums are of the f32 values before the activation)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s1[i] += acc0[i] + acc1[i];
                    s2[i] = fmaf(acc0[i], acc0[i], fmaf(acc1[i], acc1[i], s2[i]));
                }
            }
            {
                const float slope = (!XF && a.out_act) ? a.out_slope : 1.f;      // (max(v, 1 * v) = v)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (!XF) { acc0[i] = fmaxf(acc0[i], slope * acc0[i]); acc1[i] = fmaxf(acc1[i], slope * acc1[i]); }
                    o0[i] = (bf16)acc0[i]; o1[i] = (bf16)acc1[i];
                }
            }
            *(bf16x4*)(dst + (size_t)r2 * a.W * a.ldy) = o0;
            *(bf16x4*)(dst + (size_t)(r2 + 1) * a.W * a.ldy) = o1;
            if constexpr (!XF) {
                if (a.pool_y != nullptr) {
                    // the 2x2 windows of these two rows: the row pair in this lane's two accumulators, the column pair in lanes n, n ^ 1
                    // (of the values as STORED: rounded to bf16 first, as a pool over the stored tensor sees them)
                    bf16x4 pm;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float m = fmaxf((float)o0[i], (float)o1[i]);
                        pm[i] = (bf16)fmaxf(m, dpp_mov<0xB1>(m));       // quad_perm [1,0,3,2]
                    }
                    if ((n & 1) == 0)
                        *(bf16x4*)(a.pool_y + ((size_t)(b * (a.H >> 1) + ((y0 + r2) >> 1)) * (a.W >> 1) + ((x0 + n) >> 1)) * a.ld_pool + 4 * kg) = pm;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (a.stats != nullptr) {
        // sums over the 16 pixel columns of a DPP row, then the four waves through LDS: ONE partial row pair per workgroup
        __syncthreads();
        float* red = (float*)smem;         // [4 waves][2][16]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v1 = row_sum16(s1[i]), v2 = row_sum16(s2[i]);
            if (n == 0) { red[(wave * 2 + 0) * 16 + 4 * kg + i] = v1; red[(wave * 2 + 1) * 16 + 4 * kg + i] = v2; }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
            a.stats[((size_t)blockIdx.x * 2 + row) * 16 + c] = (red[(0 * 2 + row) * 16 + c] + red[(1 * 2 + row) * 16 + c]) + (red[(2 * 2 + row) * 16 + c] + red[(3 * 2 + row) * 16 + c]);
        }
    }
}

}  // namespace

// geometry shared by the eligibility test, abc_conv_stat_blocks and the launch
// ---------------------------------------------------------------------------
// unet2's 5x5 32 -> 32 convolutions at full resolution (unet2.py:52-58 and their data gradients) on v_mfma_f32_16x16x32_bf16 (round 4).
// K = 32 is ONE tap of 32 input channels: 25 taps x 2 output-channel tiles = 50 MFMAs per 16-pixel row (121 GFLOP per layer at
// b16, 384 x 384: 48 us of matrix time; the kernels above ran them in 128-139 us, one 4-wave workgroup per CU with 128 KB of LDS).
// Here the 25 x 32 x 32 weights sit in LDS once per workgroup (50 KB, rows swizzled like head_fwd_group_kernel's), a wave owns a
// 4 x 16 pixel tile (halo 8 x 20 pixels, planar by channel quarter: 10 KB), eight waves per workgroup: per kernel ROW of five
// taps a wave reads the ten weight fragments once and walks its four output rows (five pixel fragments each, shared by the two
// channel tiles) -- 150 LDS reads per 200 MFMAs.  A lane ends up with four consecutive output channels of one pixel per tile:
// 8-byte stores, 64 contiguous bytes per pixel.  XF: the producer's BatchNorm + activation on load; statistics when asked for.
struct N32K {
    const bf16* x; const bf16* w; const float* bias; bf16* y;
    const float *sc, *sh, *sl;
    float* stats;                               // [grid][2][32] or null
    int B, H, W, ldx, cin_off, ldy, cout_off, tiles_x, tiles_y, ntiles;
    unsigned bytesX, bytesW;
    int out_act; float out_slope;
    int mirror;                                 // the tap list runs (+2, +2) .. (-2, -2): a data gradient (weight block 24 - t at offset t)
    int wpi, rows4;                             // workgroups per image (a workgroup's tiles belong to ONE image: grid = B x wpi); rows4: the
                                                // statistics rows are [grid][4][32] = sum, sum of squares, max, min (of the values as stored) -- CBAM's
    // ACTB (abc_conv_desc.actbwd_*, plain input only): this data gradient is d(activation output) of the producing 32-channel layer; the
    // epilogue stores g = dA * (BatchNorm(y_raw) > 0 ? 1 : slope) and the statistics rows are that layer's BatchNorm-backward sums
    // (sum g, sum g (y_raw - mean) / std): bn_act.hip's act_bwd pass over the 384 x 384 tensor (three tensor passes) is not run
    const bf16* ab_y; int ab_ld; unsigned bytesY;
    const float *ab_sc, *ab_sh, *ab_sl, *ab_mu, *ab_is;
    bf16* pool_y; int ld_pool;                  // plain form, optional: the 2x2 max-pool of the stored tensor, [B][H/2][W/2][ld_pool] (unet.py:30: the
                                                // folded inference graph's 32 -> 32 level at 256 x 256)
    int accumulate;                             // y += result (abc_conv_desc.accumulate; plain form): the identity residual's gradient of unet2.DoubleConv
                                                // (unet2.py:72) summed into the tensor that already holds d(out) -- bn_act.hip's add_into pass is not run
    unsigned bytesO;
};
constexpr int N32_CF = 1024;                    // coefficient tables between the weights and the halo images: [sc | sh | sl][32], bias [32], ACTB's mean [32]

// R: tap radius (2: unet2.py's 5x5 layers, the kernel's first use; 1: unet.py's 3x3 32 -> 32 layer at 192 x 192, forward with the
// transform on load + statistics and data gradient with act_bwd in the epilogue, were conv_fast's BN = 32 tiles at 44 / 24 us)
template <bool XF, bool ACTB = false, int R = 2>
__global__ __launch_bounds__(512, 2) void conv_n32r2_kernel(const N32K a) {
    static_assert(!(XF && ACTB), "act_bwd in the epilogue: plain inputs only");
    constexpr int KW = 2 * R + 1, TAPS = KW * KW;
    constexpr int HR = 4 + 2 * R, HC = 16 + 2 * R;      // halo rows / columns of a 4 x 16 tile
    // a channel quarter's plane of the halo image: HR x HC pixels x 16 bytes, a multiple of 256 bytes (the lane groups of a
    // ds_read_b128 span two planes: their 16 slots stay distinct only if planes start on the same bank)
    constexpr int QS = (HR * HC * 16 + 255) / 256 * 256;
    constexpr int HB = 4 * QS;                  // per wave: 10 KB (R = 2), 7 KB (R = 1)
    constexpr int NSEG = (HR * HC * 4 + 63) / 64;       // halo segments per lane
    constexpr int DIVM = R == 2 ? 3277 : 3641;  // (q * DIVM) >> 16 = q / HC for q < HR * HC
    constexpr int WB = TAPS * 32 * 64;          // weights: [tap][32 rows][64 bytes]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    float* scoef = (float*)(smem + WB);         // [sc | sh | sl][32] (XF: of the input's channels; ACTB: of the producer's = output channels)
    float* sbias = scoef + 96;                  // [32]
    float* smu = scoef + 128;                   // [32] (ACTB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* halo = smem + WB + N32_CF + wave * HB;
    {
        const __amdgpu_buffer_rsrc_t rsW = abc_make_rsrc(a.w, a.bytesW);
        // packed weights [tap][32 rows][32 channels] -> LDS, 16 bytes at a time, slot s of row co at (s + 2 (co >> 2)) & 3: a
        // ds_read_b128 lane group holds rows c .. c + 3, c + 12 .. c + 15 of one K group and c + 4 .. c + 11 of the next -- with this
        // rotation its 16 lanes hit 16 distinct 16-byte slots (an XOR by (co >> 2) & 3 left 40 % of the LDS cycles bank conflicts)
        for (int i = threadIdx.x; i < WB / 16; i += 512) {
            const int row = i >> 2, sl = i & 3, co = row & 31;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, (unsigned)i * 16u, 0, 0);
            *(u32x4*)(wl + row * 64 + (((sl + 2 * (co >> 2)) & 3) * 16)) = v;
        }
        if (threadIdx.x < 96)
            scoef[threadIdx.x] = XF ? (threadIdx.x < 32 ? a.sc : (threadIdx.x < 64 ? a.sh : a.sl))[a.cin_off + (threadIdx.x & 31)]
                                    : (ACTB ? (threadIdx.x < 32 ? a.ab_sc : (threadIdx.x < 64 ? a.ab_sh : a.ab_sl))[threadIdx.x & 31] : 0.f);
        if (threadIdx.x >= 128 && threadIdx.x < 160) sbias[threadIdx.x - 128] = a.bias != nullptr ? a.bias[threadIdx.x - 128] : 0.f;
        if (ACTB && threadIdx.x >= 192 && threadIdx.x < 224) smu[threadIdx.x - 192] = a.ab_mu[threadIdx.x - 192];
    }
    __syncthreads();
    // (the second resource: ACTB's y_raw, or the output tensor itself where the result is added to it)
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX),
                                 rsYR = abc_make_rsrc(ACTB ? a.ab_y : (const bf16*)a.y, ACTB ? a.bytesY : (!XF && a.accumulate ? a.bytesO : 0u));
    // compute role: pixel column n, K group kg = input channels 8 kg .. + 7 of the tap; output channels 16 c2 + 4 kg .. + 3
    const int n = lane & 15, kg = lane >> 4;
    // this lane's weight-fragment address inside a tap's 2 KB block: row 16 c2 + n, slot kg (swizzled by the row)
    int woff[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) { const int co = 16 * c2 + n; woff[c2] = co * 64 + (((kg + 2 * (co >> 2)) & 3) * 16); }
    float bv[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[c2][i] = sbias[16 * c2 + 4 * kg + i];
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1[c2][i] = 0.f; s2[c2][i] = 0.f; }
    // staging role: halo segment s = lane + 64 i (i < 10): pixel s >> 2 (row / 20, column % 20), channel quarter s & 3 = lane & 3
    const int qt = lane & 3;
    u32x4 pre[NSEG];
    // this workgroup's image and its waves' tile run inside it
    const int b = blockIdx.x / a.wpi;
    const int tpi = a.tiles_x * a.tiles_y;
    const int wid = (blockIdx.x - b * a.wpi) * 8 + wave, nw = a.wpi * 8;
    auto issue = [&](int tile) {
        const bool live = tile < tpi;
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
#pragma unroll
        for (int i = 0; i < NSEG; ++i) {
            const int q = (lane >> 2) + 16 * i;
            const int hr = (q * DIVM) >> 16, hc = q - HC * hr;
            const int iy = ty * 4 - R + hr, ix = tx * 16 - R + hc;
            const bool ok = live && q < HR * HC && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)a.ldx + (unsigned)(a.cin_off + 8 * qt)) * 2u : 0x80000000u, 0, 0);
        }
    };
    float smx[2][4], smn[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int i = 0; i < 4; ++i) { smx[c2][i] = -3.0e38f; smn[c2][i] = 3.0e38f; }
    issue(wid);
    for (int tile = wid; tile < tpi; tile += nw) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * 4, x0 = tx * 16;
        if constexpr (XF) {
            float csc[8], csh[8], csl[8];
            LoadVec<float, 8>::ld(scoef + 8 * qt, csc); LoadVec<float, 8>::ld(scoef + 32 + 8 * qt, csh); LoadVec<float, 8>::ld(scoef + 64 + 8 * qt, csl);
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                const int q = (lane >> 2) + 16 * i;
                if (HR * HC % 16 != 0 && q >= HR * HC) continue;
                const int hr = (q * DIVM) >> 16, hc = q - HC * hr;
                const int iy = y0 - R + hr, ix = x0 - R + hc;
                const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(pre[i][j] << 16); v[2 * j + 1] = __uint_as_float(pre[i][j] & 0xFFFF0000u); }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = in ? abc_act(v[j], csc[j], csh[j], csl[j]) : 0.f;     // (the zero padding applies to the ACTIVATED tensor)
                *(bf16x8*)(halo + QS * qt + (hr * HC + hc) * 16) = pack_frag<bf16>(v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NSEG; ++i) {
                const int q = (lane >> 2) + 16 * i;
                if (HR * HC % 16 != 0 && q >= HR * HC) continue;
                *(u32x4*)(halo + QS * qt + q * 16) = pre[i];
            }
        }
        issue(tile + nw);
        // ACTB: the producer's raw output at this lane's 4 pixels x 8 channels (the accumulator layout), in flight under the MFMAs
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 yq[XF ? 1 : 4][2];
        if constexpr (!XF && !ACTB) {
            if (a.accumulate) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
                        yq[r][c2] = __builtin_amdgcn_raw_buffer_load_b64(rsYR, (unsigned)(((b * a.H + y0 + r) * a.W + x0 + n) * a.ldy + a.cout_off + 16 * c2 + 4 * kg) * 2u, 0, 0);
            }
        }
        if constexpr (ACTB) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
                    yq[r][c2] = __builtin_amdgcn_raw_buffer_load_b64(rsYR, (unsigned)(((b * a.H + y0 + r) * a.W + x0 + n) * a.ab_ld + 16 * c2 + 4 * kg) * 2u, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 acc[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) acc[r][c2] = (f32x4){bv[c2][0], bv[c2][1], bv[c2][2], bv[c2][3]};
        const char* hb = halo + QS * kg + n * 16;
        const char* wbase = wl + (a.mirror ? (TAPS - 1) * 2048 : 0);
        // software-pipelined by one tap: the four pixel fragments of tap (dy, dx + 1) -- and, at the end of a kernel row, the ten weight
        // fragments of the next row -- are issued in front of the eight MFMAs of tap (dy, dx) (two named fragment sets; the fences keep
        // the compiler from sinking the reads to their uses: 89 lgkmcnt waits with two MFMAs between them otherwise)
        bf16x8 wa[KW][2], fb0[4], fb1[4];
        auto load_w1 = [&](int dy, int dx) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) wa[dx][c2] = *(const bf16x8*)(wbase + (a.mirror ? -(dy * KW + dx) : dy * KW + dx) * 2048 + woff[c2]);
        };
        auto load_b = [&](bf16x8 (&f)[4], int dy, int dx) {
#pragma unroll
            for (int r = 0; r < 4; ++r) f[r] = *(const bf16x8*)(hb + ((r + dy) * HC + dx) * 16);
        };
        auto mma = [&](const bf16x8 (&f)[4], int dx) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[dx][0], f[r], acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[dx][1], f[r], acc[r][1], 0, 0, 0);
            }
        };
#pragma unroll
        for (int dx = 0; dx < KW; ++dx) load_w1(0, dx);
        load_b(fb0, 0, 0);
#pragma unroll
        for (int dy = 0; dy < KW; ++dy) {
#pragma unroll
            for (int dx = 0; dx < KW; ++dx) {
                const int t = dy * KW + dx;
                const int ndy = dx == KW - 1 ? dy + 1 : dy, ndx = dx == KW - 1 ? 0 : dx + 1;
                if (t < TAPS - 1) { if (t & 1) load_b(fb0, ndy, ndx); else load_b(fb1, ndy, ndx); }
                __builtin_amdgcn_sched_barrier(0);
                if (t & 1) mma(fb1, dx); else mma(fb0, dx);
                // this tap's two weight fragments are free: the next kernel row's take their place (needed KW taps from now)
                if (dy < KW - 1) load_w1(dy + 1, dx);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bf16* dst = a.y + ((size_t)(b * a.H + y0) * a.W + x0 + n) * a.ldy + a.cout_off + 4 * kg;
        const float slope = a.out_act ? a.out_slope : 1.f;      // (max(v, 1 * v) = v)
        if constexpr (ACTB) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                float csc[4], csh[4], csl[4], cmu[4];
                LoadVec<float, 4>::ld(scoef + 16 * c2 + 4 * kg, csc); LoadVec<float, 4>::ld(scoef + 32 + 16 * c2 + 4 * kg, csh);
                LoadVec<float, 4>::ld(scoef + 64 + 16 * c2 + 4 * kg, csl); LoadVec<float, 4>::ld(smu + 16 * c2 + 4 * kg, cmu);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // g = dA where BatchNorm(y_raw) > 0, slope * dA elsewhere (unet2.py:54,57 backward); sums of g and g (y_raw - mean)
                        const unsigned w = i < 2 ? yq[r][c2].x : yq[r][c2].y;
                        const float x = (i & 1) ? __uint_as_float(w & 0xFFFF0000u) : __uint_as_float(w << 16);
                        const float gg = acc[r][c2][i] * (fmaf(x, csc[i], csh[i]) > 0.f ? 1.f : csl[i]);
                        s1[c2][i] += gg; s2[c2][i] = fmaf(gg, x - cmu[i], s2[c2][i]);
                        o[i] = (bf16)gg;
                    }
                    *(bf16x4*)(dst + (size_t)r * a.W * a.ldy + 16 * c2) = o;
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[r][c2][i];
                    if constexpr (!XF) {
                        if (a.accumulate) {
                            const unsigned w = i < 2 ? yq[r][c2].x : yq[r][c2].y;
                            v += (i & 1) ? __uint_as_float(w & 0xFFFF0000u) : __uint_as_float(w << 16);
                        }
                    }
                    s1[c2][i] += v; s2[c2][i] = fmaf(v, v, s2[c2][i]);
                    o[i] = (bf16)fmaxf(v, slope * v);
                    // (max / min of the values AS STORED: CBAM's global max-pool and its backward see the tensor, unet2.py:10,20)
                    smx[c2][i] = fmaxf(smx[c2][i], (float)o[i]); smn[c2][i] = fminf(smn[c2][i], (float)o[i]);
                    if constexpr (!XF) acc[r][c2][i] = (float)o[i];      // (the stored value, for the pooled output below)
                }
                *(bf16x4*)(dst + (size_t)r * a.W * a.ldy + 16 * c2) = o;
            }
        if constexpr (!XF) {
            if (a.pool_y != nullptr) {
                // the 2x2 windows of the tile's two row pairs: the rows in this lane's accumulators, the column pair in lanes n, n ^ 1
                // (of the values as STORED: rounded to bf16 first, as a pool over the stored tensor sees them)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        bf16x4 pm;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float m = fmaxf(acc[2 * rp][c2][i], acc[2 * rp + 1][c2][i]);
                            pm[i] = (bf16)fmaxf(m, dpp_mov<0xB1>(m));       // quad_perm [1,0,3,2]
                        }
                        if ((n & 1) == 0)
                            *(bf16x4*)(a.pool_y + ((size_t)(b * (a.H >> 1) + ((y0 >> 1) + rp)) * (a.W >> 1) + ((x0 + n) >> 1)) * a.ld_pool + 16 * c2 + 4 * kg) = pm;
                    }
            }
        }
        }
    }
    if (a.stats != nullptr) {
        __syncthreads();
        float* red = (float*)(smem + WB + N32_CF);    // [8 waves][4][32] (the halo images are dead)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v1 = row_sum16(s1[c2][i]), v2 = row_sum16(s2[c2][i]), v3 = row_max16(smx[c2][i]), v4 = row_min16(smn[c2][i]);
                if (n == 0) {
                    const int c = 16 * c2 + 4 * kg + i;
                    red[(wave * 4 + 0) * 32 + c] = v1; red[(wave * 4 + 1) * 32 + c] = v2; red[(wave * 4 + 2) * 32 + c] = v3; red[(wave * 4 + 3) * 32 + c] = v4;
                }
            }
        __syncthreads();
        const int rows = a.rows4 ? 4 : 2;
        if (threadIdx.x < 32 * rows) {
            const int row = threadIdx.x >> 5, c = threadIdx.x & 31;
            float v = red[row * 32 + c];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const float u = red[(w * 4 + row) * 32 + c];
                v = row < 2 ? v + u : (row == 2 ? fmaxf(v, u) : fminf(v, u));
            }
            if constexpr (ACTB) { if (row == 1) v *= a.ab_is[c]; }      // (the row act_bwd writes: sum of g (y_raw - mean) / std)
            a.stats[((size_t)blockIdx.x * rows + row) * 32 + c] = v;
        }
    }
}

// 16 -> 16 channels over whole tiles with the full 3x3 square in row-major tap order go to conv_n16_kernel
static bool route_n16(const abc_conv_desc* d) {
    if (d->Cin != 16 || d->Cout != 16 || d->ntaps != 9) return false;
    if (d->stem_x != nullptr || d->actbwd_y != nullptr || d->stats_rows == 4) return false;
    // training forward: transform on load, raw output (+ statistics); folded inference graph: a finished input, activation in the
    // epilogue, optionally the pooled second output, no statistics
    // (must not depend on the statistics POINTER: abc_conv_stat_blocks is asked before that buffer exists)
    if (d->src.scale != nullptr && (d->pool_y != nullptr || d->out_act)) return false;
    if (d->pool_y != nullptr && (d->ld_pool % 4)) return false;
    if (d->Hin % 8 || d->Win % 16 || (d->ldy % 4) || (d->cout_off % 4)) return false;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) return false;
    return abc_knob("ABC_CONV_NON16") == nullptr;
}

// 32 -> 32 channels, the full 5x5 square in row-major tap order, whole 4 x 16 tiles go to conv_n32r2_kernel
static bool route_n32r2(const abc_conv_desc* d) {
    if (d->Cin != 32 || d->Cout != 32 || (d->ntaps != 25 && d->ntaps != 9)) return false;
    if (d->ntaps == 9 && (d->stats_rows == 4 || abc_knob("ABC_CONV_NON32R1"))) return false;
    if (d->stem_x != nullptr) return false;
    if (d->src.scale != nullptr && d->out_act) return false;
    // second output (the 2x2 max-pool of the stored tensor): the plain form
    if (d->pool_y != nullptr && (d->src.scale != nullptr || d->actbwd_y != nullptr || d->accumulate || (d->ld_pool % 4) || d->stats_rows == 4)) return false;
    // act_bwd in the epilogue: plain input, the two BatchNorm-backward rows
    if (d->actbwd_y != nullptr && (d->src.scale != nullptr || d->out_act || d->stats_rows == 4 || (d->actbwd_ld % 4) || (d->actbwd_coff % 4) ||
                                   (int64_t)d->B * d->Hin * d->Win * d->actbwd_ld * 2 >= (int64_t(1) << 31))) return false;
    if (d->Hin % 4 || d->Win % 16 || (d->ldy % 4) || (d->cout_off % 4) || d->B > 256) return false;
    bool fwd = true, mir = true;
    const int kw = d->ntaps == 25 ? 5 : 3, rr = kw / 2;
    for (int t = 0; t < d->ntaps; ++t) {
        fwd = fwd && d->tap_dy[t] == t / kw - rr && d->tap_dx[t] == t % kw - rr;
        mir = mir && d->tap_dy[t] == rr - t / kw && d->tap_dx[t] == rr - t % kw;
    }
    if (!fwd && !mir) return false;
    return abc_knob("ABC_CONV_NON32R2") == nullptr;
}
// workgroups per image (a workgroup's tiles belong to one image: the four-row statistics of unet2's CBAM are per image); grid = B x this
static int n32r2_wpi(const abc_conv_desc* d) {
    const int tpi = (d->Win / 16) * (d->Hin / 4);
    int w = abc_wg_slots(1) / d->B;
    if (w > abc_cdiv(tpi, 8)) w = abc_cdiv(tpi, 8);
    return w < 1 ? 1 : w;
}
static int n32r2_grid(const abc_conv_desc* d) { return d->B * n32r2_wpi(d); }

static void narrow_grid(const abc_conv_desc* d, int* nwg, int* tpw) {
    if (route_n32r2(d)) { *nwg = n32r2_grid(d); *tpw = 0; return; }
    const int ntiles = abc_cdiv(d->Win, 16) * abc_cdiv(d->Hin, 8) * d->B;
    int n = abc_wg_slots(route_n16(d) ? 4 : (d->ntaps == 9 ? 2 : 1));      // workgroups per CU: four (conv_n16), two (3x3), one (5x5: 128 KB of LDS)
    if (n * 4 > ntiles) n = abc_cdiv(ntiles, 4);
    *tpw = abc_cdiv(ntiles, n * 4);
    *nwg = abc_cdiv(ntiles, *tpw * 4);
}

// bf16 NHWC in and out, 16 / 32 input channels, <= 32 output channels, unit strides, the nine taps of a 3x3 (or, 32 input
// channels, the 25 of a 5x5 over a finished tensor); optionally the producer's BatchNorm + activation on load (3x3) and
// BatchNorm partial sums of the outputs (2 rows).  (32 input channels with BOTH the transform and the sums does not fit the
// registers: that one stays on conv_fast.)
int abc_conv_narrow_ok(const abc_conv_desc* d) {
    if (abc_knob("ABC_CONV_NONARROW")) return 0;
    if (d->dtype_in != ABC_BF16 || d->dtype_c != ABC_BF16 || d->dtype_out != ABC_BF16) return 0;
    if (d->src.pool || d->src.planar || d->src.drop_p > 0.f || d->planar_out) return 0;
    if (d->stride != 1 || d->om != 1 || d->oy0 || d->ox0) return 0;
    // y += result: the plain form of the 32 -> 32 kernel only
    if (d->accumulate && !(route_n32r2(d) && d->src.scale == nullptr && d->actbwd_y == nullptr && !d->out_act && d->stats_rows != 4 &&
                           (int64_t)d->B * d->Hin * d->Win * d->ldy * 2 < (int64_t(1) << 31))) return 0;
    if ((d->Cin != 16 && d->Cin != 32) || d->Cout_pad != 32 || d->Cout % 8 || (d->ntaps != 9 && d->ntaps != 25)) return 0;
    if (d->Hg != d->Hin || d->Wg != d->Win || d->Hout != d->Hg || d->Wout != d->Wg || d->src.Hx != d->Hin || d->src.Wx != d->Win) return 0;
    if ((d->src.ldx | d->cin_off | d->ldy | d->cout_off) % 8) return 0;
    // (must not depend on the statistics POINTER: abc_conv_stat_blocks is asked before that buffer exists)
    if (d->stats_rows == 4 && !route_n32r2(d)) return 0;                // unet2's CBAM rows: per image, per tile
    // act_bwd in the epilogue: the plain 16 -> <= 16 channel 3x3 data gradient, and the 5x5 32 -> 32 one of conv_n32r2_kernel
    if (d->actbwd_y != nullptr && route_n32r2(d)) {
        if (d->stats_rows != 2 || !d->actbwd_scale || !d->actbwd_shift || !d->actbwd_slope || !d->actbwd_mean || !d->actbwd_invstd) return 0;
    } else
    if (d->actbwd_y != nullptr && (d->Cin != 16 || d->ntaps != 9 || d->Cout > 16 || d->src.scale != nullptr || d->stem_x != nullptr || d->pool_y != nullptr ||
                                   d->out_act || d->stats_rows != 2 || d->actbwd_ld % 8 || d->actbwd_coff % 8 || !d->actbwd_scale || !d->actbwd_shift ||
                                   !d->actbwd_slope || !d->actbwd_mean || !d->actbwd_invstd ||
                                   (int64_t)d->B * d->Hin * d->Win * d->actbwd_ld * 2 >= (int64_t(1) << 31))) return 0;
    if (d->src.scale != nullptr && d->Cin == 32 && !route_n32r2(d)) return 0;             // transform + sums + 72 weight registers do not fit
    if (d->src.scale != nullptr && abc_knob("ABC_CONV_NONARROW_XF")) return 0;
    if (d->stem_x != nullptr && (d->Cin != 16 || d->ntaps != 9 || d->src.scale != nullptr || d->stats != nullptr || !d->stem_w || !d->stem_scale || !d->stem_bias)) return 0;
    const int R = d->ntaps == 9 ? 1 : 2;
    if (R == 2 && (d->Cin != 32 || (d->src.scale != nullptr && !route_n32r2(d)) || abc_knob("ABC_CONV_NONARROW5"))) return 0;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->tap_dy[t] < -R || d->tap_dy[t] > R || d->tap_dx[t] < -R || d->tap_dx[t] > R) return 0;
    return (int64_t)d->B * d->Hin * d->Win * d->src.ldx * 2 < (int64_t(1) << 31);
}

int abc_conv_narrow_stat_blocks(const abc_conv_desc* d) {
    int nwg, tpw;
    narrow_grid(d, &nwg, &tpw);
    return nwg;
}

template <int CK, bool XF, int NST, int R, bool STEM = false, bool ACTB = false>
static int narrow_launch_inst(const NarrowK& k, int nwg, int lds, hipStream_t st) {
    auto fn = conv_narrow_kernel<CK, XF, NST, R, STEM, ACTB>;
    if (lds > 64 * 1024) {
        static unsigned long long lds_ok = 0;
        if (int rc = abc_allow_lds((const void*)fn, lds, &lds_ok)) return rc;
    }
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), lds, st, k);
    return abc_check_launch("conv_narrow");
}

int abc_conv_narrow_launch(const abc_conv_desc* d, abc_stream_t stream) {
    if (route_n32r2(d)) {
        N32K q;
        q.x = (const bf16*)d->src.x; q.w = (const bf16*)d->w; q.bias = d->bias; q.y = (bf16*)d->y;
        q.sc = d->src.scale; q.sh = d->src.shift; q.sl = d->src.slope; q.stats = d->stats;
        q.B = d->B; q.H = d->Hin; q.W = d->Win; q.ldx = d->src.ldx; q.cin_off = d->cin_off; q.ldy = d->ldy; q.cout_off = d->cout_off;
        q.tiles_x = q.W / 16; q.tiles_y = q.H / 4; q.ntiles = q.tiles_x * q.tiles_y * q.B;
        q.bytesX = (unsigned)((int64_t)d->B * d->Hin * d->Win * d->src.ldx * 2);
        q.bytesW = (unsigned)d->ntaps * 32u * 64u;
        q.out_act = d->out_act; q.out_slope = d->out_slope;
        q.mirror = d->tap_dy[0] > 0 ? 1 : 0;
        q.wpi = n32r2_wpi(d); q.rows4 = d->stats_rows == 4 ? 1 : 0;
        q.ab_y = d->actbwd_y ? (const bf16*)d->actbwd_y + d->actbwd_coff : nullptr; q.ab_ld = d->actbwd_ld;
        q.bytesY = d->actbwd_y ? (unsigned)((int64_t)d->B * d->Hin * d->Win * d->actbwd_ld * 2) : 0u;
        q.ab_sc = d->actbwd_scale; q.ab_sh = d->actbwd_shift; q.ab_sl = d->actbwd_slope; q.ab_mu = d->actbwd_mean; q.ab_is = d->actbwd_invstd;
        q.pool_y = (bf16*)d->pool_y; q.ld_pool = d->ld_pool;
        q.accumulate = d->accumulate; q.bytesO = (unsigned)((int64_t)d->B * d->Hin * d->Win * d->ldy * 2);
        const int nwg = n32r2_grid(d);
        if (d->actbwd_y != nullptr && d->stats == nullptr) return abc_fail(ABC_EINVAL, "conv: actbwd_y needs stats (the BatchNorm-backward partial sums)");
        const int form = d->actbwd_y != nullptr ? 2 : (d->src.scale != nullptr ? 1 : 0);
        static unsigned long long okf[6] = {0, 0, 0, 0, 0, 0};
#define ABC_N32_LAUNCH(XF_, AB_, R_, SLOT)                                                                                        \
        do {                                                                                                                       \
            constexpr int qs_ = ((4 + 2 * R_) * (16 + 2 * R_) * 16 + 255) / 256 * 256;                                             \
            const int lds = (2 * R_ + 1) * (2 * R_ + 1) * 32 * 64 + N32_CF + 8 * 4 * qs_;                                          \
            if (int rc = abc_allow_lds((const void*)conv_n32r2_kernel<XF_, AB_, R_>, 160 * 1024, &okf[SLOT])) return rc;           \
            hipLaunchKernelGGL((conv_n32r2_kernel<XF_, AB_, R_>), dim3(nwg), dim3(512), lds, (hipStream_t)stream, q);              \
        } while (0)
        if (d->ntaps == 25) {
            if (form == 2) ABC_N32_LAUNCH(false, true, 2, 0); else if (form == 1) ABC_N32_LAUNCH(true, false, 2, 1); else ABC_N32_LAUNCH(false, false, 2, 2);
        } else {
            if (form == 2) ABC_N32_LAUNCH(false, true, 1, 3); else if (form == 1) ABC_N32_LAUNCH(true, false, 1, 4); else ABC_N32_LAUNCH(false, false, 1, 5);
        }
#undef ABC_N32_LAUNCH
        return abc_check_launch("conv_n32r2");
    }
    if (route_n16(d)) {
        N16K q;
        q.x = (const bf16*)d->src.x; q.w = (const bf16*)d->w; q.bias = d->bias; q.y = (bf16*)d->y;
        q.sc = d->src.scale; q.sh = d->src.shift; q.sl = d->src.slope; q.stats = d->stats;
        q.B = d->B; q.H = d->Hin; q.W = d->Win; q.ldx = d->src.ldx; q.cin_off = d->cin_off; q.ldy = d->ldy; q.cout_off = d->cout_off;
        q.tiles_x = q.W / 16; q.tiles_y = q.H / 8; q.ntiles = q.tiles_x * q.tiles_y * q.B;
        q.bytesX = (unsigned)((int64_t)d->B * d->Hin * d->Win * d->src.ldx * 2);
        q.out_act = d->out_act; q.out_slope = d->out_slope; q.pool_y = (bf16*)d->pool_y; q.ld_pool = d->ld_pool;
        int nwg;
        narrow_grid(d, &nwg, &q.tpw);
        if (d->src.scale != nullptr) hipLaunchKernelGGL(conv_n16_kernel<true>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, q);
        else hipLaunchKernelGGL(conv_n16_kernel<false>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, q);
        return abc_check_launch("conv_n16");
    }
    NarrowK k;
    k.x = (const bf16*)d->src.x; k.w = (const bf16*)d->w; k.bias = d->bias; k.y = (bf16*)d->y;
    k.sc = d->src.scale; k.sh = d->src.shift; k.sl = d->src.slope; k.stats = d->stats; k.pool_y = (bf16*)d->pool_y; k.ld_pool = d->ld_pool;
    k.stem_x = d->stem_x; k.stem_w = d->stem_w; k.stem_scale = d->stem_scale; k.stem_bias = d->stem_bias; k.stem_slope = d->stem_slope;
    k.B = d->B; k.H = d->Hin; k.W = d->Win; k.ldx = d->src.ldx; k.cin_off = d->cin_off; k.ldy = d->ldy; k.cout_off = d->cout_off; k.Cout = d->Cout;
    k.tiles_x = abc_cdiv(k.W, 16); k.tiles_y = abc_cdiv(k.H, 8); k.ntiles = k.tiles_x * k.tiles_y * k.B;
    k.out_act = d->out_act; k.out_slope = d->out_slope;
    k.bytesX = (unsigned)((int64_t)d->B * d->Hin * d->Win * d->src.ldx * 2);
    k.ab_y = d->actbwd_y ? (const bf16*)d->actbwd_y + d->actbwd_coff : nullptr; k.ab_ld = d->actbwd_ld;
    k.bytesY = d->actbwd_y ? (unsigned)((int64_t)d->B * d->Hin * d->Win * d->actbwd_ld * 2) : 0u;
    k.ab_sc = d->actbwd_scale; k.ab_sh = d->actbwd_shift; k.ab_sl = d->actbwd_slope; k.ab_mu = d->actbwd_mean; k.ab_is = d->actbwd_invstd;
    const int R = d->ntaps == 9 ? 1 : 2;
    for (int t = 0; t < d->ntaps; ++t) { k.ty[t] = (int8_t)(d->tap_dy[t] + R); k.tx[t] = (int8_t)(d->tap_dx[t] + R); }
    int nwg;
    narrow_grid(d, &nwg, &k.tpw);
    const int wlds = (d->Cin == 32 || R == 2 || d->stem_x != nullptr || d->actbwd_y != nullptr) ? d->ntaps * (d->Cin / 16) * 1024 : 0;
    const int lds = 4 * (8 + 2 * R) * (16 + 2 * R) * (d->Cin * 2 + 16) + 128 + wlds + (d->stem_x ? (160 + 4 * 240) * 4 : 0);
    k.cf_off = (lds + 15) & ~15;
    const int lds_all = d->actbwd_y ? k.cf_off + 256 : lds;
    hipStream_t st = (hipStream_t)stream;
    const bool xf = d->src.scale != nullptr;
    const int nst = d->stats != nullptr ? (d->Cout <= 16 ? 1 : 2) : 0;
    if (R == 2) {
        if (nst == 0) return narrow_launch_inst<32, false, 0, 2>(k, nwg, lds, st);
        if (nst == 1) return narrow_launch_inst<32, false, 1, 2>(k, nwg, lds, st);
        return narrow_launch_inst<32, false, 2, 2>(k, nwg, lds, st);
    }
    if (d->actbwd_y != nullptr) {
        if (d->stats == nullptr) return abc_fail(ABC_EINVAL, "conv: actbwd_y needs stats (the BatchNorm-backward partial sums)");
        return narrow_launch_inst<16, false, 1, 1, false, true>(k, nwg, lds_all, st);
    }
    if (d->stem_x != nullptr) return narrow_launch_inst<16, false, 0, 1, true>(k, nwg, lds, st);
    if (d->Cin == 16) {
        if (!xf && nst == 0) return narrow_launch_inst<16, false, 0, 1>(k, nwg, lds, st);
        if (!xf && nst == 1) return narrow_launch_inst<16, false, 1, 1>(k, nwg, lds, st);
        if (!xf && nst == 2) return narrow_launch_inst<16, false, 2, 1>(k, nwg, lds, st);
        if (nst == 0) return narrow_launch_inst<16, true, 0, 1>(k, nwg, lds, st);
        if (nst == 1) return narrow_launch_inst<16, true, 1, 1>(k, nwg, lds, st);
        return narrow_launch_inst<16, true, 2, 1>(k, nwg, lds, st);
    }
    if (nst == 0) return narrow_launch_inst<32, false, 0, 1>(k, nwg, lds, st);
    if (nst == 1) return narrow_launch_inst<32, false, 1, 1>(k, nwg, lds, st);
    return narrow_launch_inst<32, false, 2, 1>(k, nwg, lds, st);
}
