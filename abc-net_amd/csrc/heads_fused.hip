// The heads' second half of a TRAINING step as one pass (bf16 throughput mode): out_modules[i].conv2 (unet.py:70, 116-118)
// + the activation and loss block (train.py:95-125) + d(loss)/d(logits) + the data gradient of conv2 + the backward of
// Dropout / LeakyReLU (unet.py:67-69) + the BatchNorm-backward statistics, for all eight heads.
//
// The unfused chain makes five passes over 0.3 GB tensors at B = 16, 96 x 96 (1x1 forward, loss, 1x1 weight gradient,
// 1x1 data gradient, activation backward: 1.18 ms): logits written f32 and read back, d(logits) written f32 and read
// twice, the heads' features read four times.  Here a wave owns 32 pixels of one head and never lets the logits leave
// the CU before the loss has been applied:
//
//   features (raw conv1 output, BN + LeakyReLU + dropout on load) -> B fragments of its 32 pixels, kept in registers
//   per 32 output rows:  logits = W2 x features (8 MFMAs)  ->  + bias, stored NCHW f32 (the reference's interface)
//                        ->  loss terms and d(logits) in the accumulator layout (a lane = one pixel, its registers =
//                            output rows: the ROW ORDER of the packed weights is chosen so that every softmax group sits
//                            in one lane, see hf_chan_of_row)
//                        ->  d(logits) bf16: (a) through a wave-private LDS tile into the pixel-blocked buffer the weight
//                            gradient reads, (b) straight from the registers as the B fragments of the data gradient
//                            dA += W2^T x dL (the K order of the packed W2^T is the accumulator's register order)
//   dA -> wave-private LDS transpose -> g = dA * LeakyReLU' * dropout, 16-byte stores, with the sums BatchNorm's backward
//         needs (sum g, sum g * xhat) on the way.
//
// The loss normalisers are global sums, so everything here is the gradient of each term's NUMERATOR: abc_loss_finalize
// turns the partial sums into the per-head factors, which the weight gradient (abc_heads_fused_wgrad) applies on load and
// the BatchNorm finaliser folds into its coefficients (abc_bn_bwd_desc.in_scale).
//
// Work: grid = (128-pixel chunks, 2 head groups); group 0 = bond types + rho (they share the bond-type targets), group 1 =
// omega + the five small heads.  HBM-bound: features 0.30 GB + targets 0.37 GB read, logits 0.30 GB + d(logits) 0.23 GB
// + g 0.30 GB written.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "loss_math.hpp"
#include "heads_fused.hpp"
#include <stdlib.h>

namespace {

struct HFHead {
    const bf16* w2f;     // [4 chunks][Cpad][32]: A fragments of the forward GEMM (rows = packed output rows)
    const bf16* w2t;     // [Cpad / 16 K-steps][128][2][8]: A fragments of the data gradient (rows = feature channels)
    const float* biasp;  // [Cpad]
    float* logits;       // [B][C][HW]
    bf16* dlb;           // [chunk][Cpad][128 pixels]
    uint2* keep;         // [pixel][2 halves]: the dropout keep bits of the head's 128 features (heads 5-7, or null)
};

struct HFK {
    const bf16* y1;
    const float *sc, *sh, *sl, *mean, *invstd;
    bf16* g;
    int ld;
    float drop_p;
    uint32_t drop_seed, drop_thr;   // drop_thr: keep <=> hash24 >= drop_thr (abc_drop_threshold)
    const uint32_t* drop_salt;
    const float *t_atom, *t_types, *t_charges, *t_hs, *t_bond, *t_btypes;
    const double *t_rho, *t_omega;
    const uint32_t* tflags;   // abc_heads_fused_desc.target_flags (the rasteriser's 32-pixel group flags) or null
    const char* tzero;        // >= 512 zero bytes (with tflags)
    float* bnpart;      // [nchunk][2][ld]
    float* dwsmall;     // [nchunk][HF_SMALL_ROWS][128 + 1]: the small heads' conv2 weight / bias gradient partials (see run_head)
    double* losspart;   // [nchunk][16]
    int HW, nchunk, dbg;
    HFHead hd[HF_NH];
};

// Wave-private LDS (16 KB per wave):
//   [0, 4608)      d(logits) tile [32 rows][32 pixels] bf16 (rows of 80 B) during the row-tile loop; in the epilogue the
//                  transpose tile of half a slice, [32 pixels][64 channels] bf16 (rows of 144 B)
//   [2560, 5120)   the head's packed bias (<= 480 f32 from 2560; dead once the row-tile loop is over)
//   [5120, 7680)   the slice's per-channel coefficients [scale | shift | slope | mean | invstd][128] f32
//   [7680, 16384)  the five small heads: the activated features of the wave's pixels (conv2's weight gradient, see run_head)
//   [8704, 16384)  group 0: per-lane sum of the bond-type targets of each of its 30 omega bins (bond types -> rho);
//                  omega: the omega targets of the lane's 30 bins
constexpr int TROW = 32 * 2 + 16;          // row of the d(logits) tile
constexpr int HROW = 64 * 2 + 16;          // row of the epilogue's transpose tile
constexpr int WV_BIAS = 2560, WV_CF = 5120, WV_DN = 8704;
constexpr int WV_AIMG = 7680, AROW = 128 * 2 + 16;   // small heads: the wave's activated features [32 pixels][128 channels] bf16 (rows of 272 B)
constexpr int WV = 16384;
constexpr int LDS_BSUM = 4 * WV;           // the waves' BatchNorm sums: [4 waves][<= 2 slices][2][128] f32 = 8 KB
constexpr int LDS_LSUM = LDS_BSUM + 4 * 2 * 2 * 128 * 4;   // [4 waves][16] f64
constexpr int HF_LDS = LDS_LSUM + 4 * 16 * 8;

struct Ctx {
    int lane, r, h, wave, chunk, b, yx, nslice;
    uint32_t pix, pix0;
    char* ot;
    float* dnl;
    float* wsum;     // this wave's BatchNorm sums: [slice of the group][2][128]
    float dscale;
    uint32_t dseed;
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_f;
__device__ inline bf16x8 tr_read8f(const char* b0, const char* b1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_f*)b0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_f*)b1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The small heads (one row tile, <= 14 channels) also finish conv2's WEIGHT gradient in the fused pass: dW2[c][ci] = sum_p dL[c][p] a[p][ci]
// over the workgroup's 128 pixels is 8 MFMAs per wave (wave w = feature tile w) from the four waves' d(logits) tiles and
// their activated features in LDS -- one 32 x 128 partial per workgroup instead of a pass of the blocked weight-gradient
// kernel over the head's features (5 of its 11 units of work went to these 21 channels).  A ones-fragment gives the row
// sums of dL (the bias gradient) on the way.  (All four waves: two workgroup barriers.)
template <int HEAD>
__device__ inline void small_head_wgrad(const HFK& a, const Ctx& c, const bf16x8* fb) {
    constexpr int CH = hf_ch(HEAD);
    const int r = c.r, h = c.h, lane = c.lane;
    if (ABC_DBG(a.dbg) & 32) return;
    char* img = c.ot + WV_AIMG;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) *(bf16x8*)(img + r * AROW + (16 * kk + 8 * h) * 2) = fb[kk];
    __syncthreads();
    const int trow = 8 * h + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    f32x16 aw, ab;
#pragma unroll
    for (int k = 0; k < 16; ++k) { aw[k] = 0.f; ab[k] = 0.f; }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
        const char* tw = c.ot + (w2 - c.wave) * WV;          // wave w2's region
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 fa2 = *(const bf16x8*)(tw + r * TROW + (16 * s2 + 8 * h) * 2);
            const char* q0 = tw + WV_AIMG + (16 * s2 + trow) * AROW + (32 * c.wave + tcol) * 2;
            const bf16x8 fb2 = tr_read8f(q0, q0 + 4 * AROW);
            aw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2, fb2, aw, 0, 0, 0);
            // (every wave: an MFMA under a lane-dependent branch -- `wave == 0` is a VGPR compare -- came out wrong,
            //  the instruction ignores EXEC; only wave 0 stores the sums)
            ab = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2, ones, ab, 0, 0, 0);
        }
    }
    if (h == 0) {
        // register k of half 0 = packed row (k & 3) + 8 (k >> 2) = channel k (hf_chan_of_row)
        float* dst = a.dwsmall + ((size_t)c.chunk * HF_SMALL_ROWS + hf_small_row0(HEAD)) * 129;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            dst[k * 129 + 32 * c.wave + r] = aw[k];
            if (c.wave == 0 && r == 0) dst[k * 129 + 128] = ab[k];
        }
    }
    __syncthreads();   // the tiles are read by the other waves: the epilogue reuses their space
}

// A wave NONE of whose 32 pixels carries a target of a softmax head (atom types, charges, hydrogens, bond types; known from the
// rasteriser's group flags): every term of class_focal and every derivative is zero there (loss_math.hpp: t[k] == 0 adds nothing to
// the numerator or the denominator and leaves a[k] = 0, so dz[k] = q[k] * (0 - 0)) -- for finite logits exactly what the full path
// computes.  What is left of the head for this wave: its logits (when somebody reads them: hd.logits), zeros for d(logits), g and the
// BatchNorm sums.  The exp / log / focal arithmetic of the 15-tile bond-type head is the bulk of the fused pass's instruction issue,
// and most waves see no bond at all.
template <int HEAD>
__device__ inline void run_head_skip(const HFK& a, Ctx& c, double* lsum) {
    static_assert(HEAD == 1 || HEAD == 2 || HEAD == 3 || HEAD == 5, "softmax heads");
    constexpr int CH = hf_ch(HEAD), NT = hf_tiles(HEAD), CPAD = NT * 32;
    const HFHead& hd = a.hd[HEAD];
    const int slice = 128 * HEAD;
    const int r = c.r, h = c.h, lane = c.lane;
    const uint32_t lch = HEAD >= 5 ? 30u * h : 0u;
    const uint32_t loff = ((uint32_t)(c.b * CH) + lch) * (uint32_t)a.HW + (uint32_t)c.yx;
    auto at4w = [&](float* base, int chu) -> float* { return (float*)((char*)(base + (size_t)chu * a.HW) + 4u * loff); };
    float* cf = (float*)(c.ot + WV_CF);
    float* bl = (float*)(c.ot + WV_BIAS);
    const uint32_t e0 = c.pix * (uint32_t)a.ld + slice + 8 * h;
    const bool st_logits = hd.logits != nullptr;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    bf16x8 fb[8];
    uint32_t kbits[2] = {0u, 0u};
    if (st_logits || (HEAD >= 5 && hd.keep != nullptr)) {
        // ---- the forward half as in run_head: features -> BN + LeakyReLU + dropout -> B fragments (and the keep bits the weight gradient takes)
        u32x4 raw[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) raw[kk] = *(const u32x4*)(a.y1 + e0 + 16 * kk);
        {
            const int ch = slice + 2 * lane;
            const float2 v0 = *(const float2*)(a.sc + ch), v1 = *(const float2*)(a.sh + ch), v2 = *(const float2*)(a.sl + ch);
            *(float2*)(cf + 0 * 128 + 2 * lane) = v0; *(float2*)(cf + 1 * 128 + 2 * lane) = v1; *(float2*)(cf + 2 * 128 + 2 * lane) = v2;
#pragma unroll
            for (int i = 0; i < (CPAD + 63) / 64; ++i)
                if (lane + 64 * i < CPAD) bl[lane + 64 * i] = hd.biasp[lane + 64 * i];
        }
        lds_sync();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float v[8], sc[8], sh[8], sl[8];
            const int cc = 16 * kk + 8 * h;
            LoadVec<float, 8>::ld(cf + cc, sc); LoadVec<float, 8>::ld(cf + 128 + cc, sh); LoadVec<float, 8>::ld(cf + 256 + cc, sl);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(raw[kk][j] << 16); v[2 * j + 1] = __uint_as_float(raw[kk][j] & 0xFFFF0000u); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = fmaf(v[j], sc[j], sh[j]);
                const bool keep = a.drop_p > 0.f ? abc_drop_hash24(e0 + 16 * kk + j, c.dseed) >= a.drop_thr : true;
                kbits[kk >> 2] |= (keep ? 1u : 0u) << (8 * (kk & 3) + j);
                v[j] = keep ? fmaxf(y, sl[j] * y) * c.dscale : 0.f;
            }
            fb[kk] = pack_frag<bf16>(v);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) fb[kk] = __builtin_bit_cast(bf16x8, z4);
    }
    if (st_logits) {
        bf16x8 fa[8];
        auto load_fa = [&](int mt) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                fa[kk] = *(const bf16x8*)((const char*)(hd.w2f + (size_t)((kk >> 1) * CPAD + 32 * mt) * 32 + 16 * (kk & 1)) + (uint32_t)(r * 64 + h * 16));
        };
        load_fa(0);
#pragma unroll 1
        for (int mt = 0; mt < NT; ++mt) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *(const f32x4*)(bl + 32 * mt + 8 * q + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * q + j] = b4[j];
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[kk], acc, 0, 0, 0);
            // (the next tile's weights ahead of this tile's stores: loads and stores retire through one in-order counter)
            __builtin_amdgcn_sched_barrier(0);
            load_fa(mt + 1 < NT ? mt + 1 : mt);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HEAD == 5) {
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int k = 0; k < 6; ++k) *at4w(hd.logits, k * 60 + 2 * mt + gi) = acc[8 * gi + k];
            } else if (h == 0) {
#pragma unroll
                for (int k = 0; k < CH; ++k) *at4w(hd.logits, k) = acc[k];
            }
        }
    }
    // ---- d(logits) = 0: the blocked buffer of the weight gradient (the stores of run_head's tile loop), the wave's LDS tile (the
    // small heads' weight gradient reads it), and for the bond types the per-bin target sums rho takes
#pragma unroll 1
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = lane + 64 * j, row = q >> 2, part = q & 3;
            *(u32x4*)((char*)(hd.dlb + ((size_t)c.chunk * CPAD + 32 * mt) * 128) + (uint32_t)((row * 128 + 32 * c.wave + part * 8) * 2)) = z4;
        }
    if constexpr (HEAD == 5) {
#pragma unroll
        for (int j = 0; j < 30; ++j) c.dnl[j * 64 + lane] = 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = lane + 64 * j, row = q >> 2, part = q & 3;
            *(u32x4*)(c.ot + row * TROW + part * 16) = z4;
        }
        lds_sync();
        small_head_wgrad<HEAD>(a, c, fb);
    }
    // ---- g = 0 and its BatchNorm sums (run_head's epilogue: dA = W2^T x 0)
    {
        const int sg = lane & 7, pg = lane >> 3;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                *(u32x4*)(a.g + (size_t)(c.pix0 + it * 8 + pg) * a.ld + slice + 64 * half + sg * 8) = z4;
        float* ws = c.wsum + c.nslice * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) ws[lane + 64 * i] = 0.f;
    }
    if constexpr (HEAD >= 5) {
        if (hd.keep != nullptr) hd.keep[2u * c.pix + h] = make_uint2(kbits[0], kbits[1]);
    }
    c.nslice += 1;
    if (lane == 0) {
        lsum[HEAD] = 0.0;
        lsum[8 + HEAD] = 0.0;
        if (HEAD == 5) lsum[8 + 6] = 0.0;
    }
}

// One head for the wave's 32 pixels; its loss numerator / denominator (summed over the wave) go to lsum[HEAD] / lsum[8 + HEAD].
//
// The row-tile loop is software-pipelined by hand: on this hardware loads and stores retire through ONE in-order counter, so
// a wait for a load also waits for every store issued before it.  Each iteration therefore issues everything the NEXT one
// needs (its weight fragments, its targets, the data-gradient fragments of this tile) BEFORE its own stores, and the
// data-gradient MFMAs of a tile run at the top of the following iteration.
template <int HEAD>
__device__ inline void run_head(const HFK& a, Ctx& c, double* lsum) {
    double num = 0.0, den = 0.0;
    constexpr int CH = hf_ch(HEAD), NT = hf_tiles(HEAD), CPAD = NT * 32;
    const HFHead& hd = a.hd[HEAD];
    const int slice = 128 * HEAD;
    const int r = c.r, h = c.h, lane = c.lane;
    // NCHW planes: address = base + [uniform: plane chu of the head] + [this lane: image, the lane half's 30 bins, pixel].  The
    // uniform part stays in scalar registers (one VGPR offset serves every plane access of the head; per-plane VGPR pointers,
    // strength-reduced over the tile loop, cost 50 spilled registers)
    const uint32_t lch = HEAD >= 5 ? 30u * h : 0u;
    const uint32_t loff = ((uint32_t)(c.b * CH) + lch) * (uint32_t)a.HW + (uint32_t)c.yx;   // elements (32-bit: checked on the host)
    // TARGET reads: where the rasteriser's group flags say that none of the wave's 32 pixels carries a target of this head (the maps hold
    // ~62 non-zero 3x3 neighbourhoods per image), every target address is redirected into 512 zero bytes -- the same loads, the same
    // arithmetic on the same zeros, no HBM traffic (0.37 GB of target planes per step otherwise).  tz is wave-uniform: one scalar select of
    // the plane stride and base, one vector select of the lane's offset.
    const uint32_t tword = a.tflags ? (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tflags[c.pix0 >> 5]) : 0xFFFFFFFFu;
    const bool tz = !((tword >> (HEAD == 6 ? 5 : HEAD)) & 1u);      // (rho reads the bond-type targets' bins: their flag)
    if constexpr (HEAD == 1 || HEAD == 2 || HEAD == 3 || HEAD == 5) {
        // the softmax heads contribute nothing where no pixel of the wave has a target: run_head_skip (ABC_HF_DBG bit 6: the A/B)
        if (tz && !(ABC_DBG(a.dbg) & 64)) { run_head_skip<HEAD>(a, c, lsum); return; }
    }
    const size_t thw = tz ? (size_t)0 : (size_t)a.HW;
    const uint32_t tloff = tz ? (uint32_t)lane : loff;
    auto tbase4 = [&](const float* base) -> const float* { return tz ? (const float*)a.tzero : base; };
    auto tbase8 = [&](const double* base) -> const double* { return tz ? (const double*)a.tzero : base; };
    auto at4 = [&](const float* base, int chu) -> const float* { return (const float*)((const char*)(tbase4(base) + (size_t)chu * thw) + 4u * tloff); };
    auto at4w = [&](float* base, int chu) -> float* { return (float*)((char*)(base + (size_t)chu * a.HW) + 4u * loff); };
    auto at8 = [&](const double* base, int chu) -> const double* { return (const double*)((const char*)(tbase8(base) + (size_t)chu * thw) + 8u * tloff); };
    float* cf = (float*)(c.ot + WV_CF);
    float* bl = (float*)(c.ot + WV_BIAS);
    const uint32_t e0 = c.pix * (uint32_t)a.ld + slice + 8 * h;   // this lane's first feature element

    // ---- issue: features, coefficients + bias (-> LDS), the first tile's weights and targets
    u32x4 raw[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) raw[kk] = *(const u32x4*)(a.y1 + e0 + 16 * kk);
    {
        const int ch = slice + 2 * lane;
        const float2 v0 = *(const float2*)(a.sc + ch), v1 = *(const float2*)(a.sh + ch), v2 = *(const float2*)(a.sl + ch),
                     v3 = *(const float2*)(a.mean + ch), v4 = *(const float2*)(a.invstd + ch);
        *(float2*)(cf + 0 * 128 + 2 * lane) = v0; *(float2*)(cf + 1 * 128 + 2 * lane) = v1; *(float2*)(cf + 2 * 128 + 2 * lane) = v2;
        *(float2*)(cf + 3 * 128 + 2 * lane) = v3; *(float2*)(cf + 4 * 128 + 2 * lane) = v4;
#pragma unroll
        for (int i = 0; i < (CPAD + 63) / 64; ++i)
            if (lane + 64 * i < CPAD) bl[lane + 64 * i] = hd.biasp[lane + 64 * i];
    }
    bf16x8 fa[8];
    auto load_fa = [&](int mt) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            fa[kk] = *(const bf16x8*)((const char*)(hd.w2f + (size_t)((kk >> 1) * CPAD + 32 * mt) * 32 + 16 * (kk & 1)) + (uint32_t)(r * 64 + h * 16));
    };
    float tn[HEAD == 5 ? 12 : 1];
    auto load_t5 = [&](int mt) {
        if constexpr (HEAD == 5) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int k = 0; k < 6; ++k) tn[6 * gi + k] = *at4(a.t_btypes, k * 60 + 2 * mt + gi);
        }
    };
    // (only the 15-tile head is worth the registers the prefetch holds across the loss code; the others load at the point of use)
    constexpr bool PIPE = HEAD == 5;
    if constexpr (PIPE) { load_fa(0); load_t5(0); }

    // omega: the per-pixel weight is the sum of the pixel's 60 omega targets (train.py:124): this lane's 30 + the other half's
    double wpix = 0.0;
    if constexpr (HEAD == 7) {
        double wp = 0.0;
#pragma unroll 6
        for (int j = 0; j < 30; ++j) {
            const double t = *at8(a.t_omega, j);
            wp += t;
            c.dnl[j * 64 + lane] = (float)t;
        }
        wpix = wp + __shfl_xor(wp, 32);
        den = (h == 0) ? wpix : 0.0;
    }

    // ---- features as B fragments: lane (pixel r, half h) holds channels 16 kk + 8 h .. + 8, BN + LeakyReLU + dropout applied
    lds_sync();
    // (bit 8 (kk & 3) + j of word kk >> 2 = element j of fragment kk: kept by the dropout / BatchNorm output positive -- the
    //  epilogue's LeakyReLU' and dropout mask without a second hash)
    bf16x8 fb[8];
    uint32_t kbits[2] = {0u, 0u}, pbits[2] = {0u, 0u};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        float v[8], sc[8], sh[8], sl[8];
        const int cc = 16 * kk + 8 * h;
        LoadVec<float, 8>::ld(cf + cc, sc); LoadVec<float, 8>::ld(cf + 128 + cc, sh); LoadVec<float, 8>::ld(cf + 256 + cc, sl);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(raw[kk][j] << 16); v[2 * j + 1] = __uint_as_float(raw[kk][j] & 0xFFFF0000u); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = fmaf(v[j], sc[j], sh[j]);
            const bool keep = a.drop_p > 0.f ? abc_drop_hash24(e0 + 16 * kk + j, c.dseed) >= a.drop_thr : true;
            const int bit = 8 * (kk & 3) + j;
            kbits[kk >> 2] |= (keep ? 1u : 0u) << bit;
            pbits[kk >> 2] |= (y > 0.f ? 1u : 0u) << bit;
            v[j] = keep ? fmaxf(y, sl[j] * y) * c.dscale : 0.f;
        }
        fb[kk] = pack_frag<bf16>(v);
    }
    // (pin the masks HERE: left alone the compiler sinks the sign-bit computation behind the row-tile loop, keeps the 64 BatchNorm
    //  outputs in scratch across it and brings them back one by one, each behind a full `s_waitcnt vmcnt(0)`)
    asm volatile("" : "+v"(kbits[0]), "+v"(kbits[1]), "+v"(pbits[0]), "+v"(pbits[1]));

    const bool st_logits = hd.logits != nullptr;   // (uniform: a kernel argument)
    f32x16 accD[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int k = 0; k < 16; ++k) accD[mi][k] = 0.f;
    bf16x8 wt[8], bqp[2];
    auto dgrad_mfma = [&]() {
        if (ABC_DBG(a.dbg) & 8) return;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) accD[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[4 * u + mi], bqp[u], accD[mi], 0, 0, 0);
    };

#pragma unroll 1
    for (int mt = 0; mt < NT; ++mt) {
        // ---- data gradient of the previous tile: dA[ci][p] += W2^T x dL, K-step u = registers 8 u .. 8 u + 7 of both halves
        if (PIPE && mt > 0) dgrad_mfma();
        if constexpr (!PIPE) load_fa(mt);
        // ---- logits of 32 packed rows x 32 pixels (the accumulators start from the bias)
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *(const f32x4*)(bl + 32 * mt + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * q + j] = b4[j];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[kk], acc, 0, 0, 0);
        float v[16], dlv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = acc[k]; dlv[k] = 0.f; }
        // ---- loss + d(logits): register k of lane half h = packed row 32 mt + (k & 3) + 8 (k >> 2) + 4 h (hf_chan_of_row)
        if constexpr (HEAD == 5) {
            // registers 8 gi .. 8 gi + 5 = the six bond types of omega bin 30 h + 2 mt + gi (train.py:101 view)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                float z[6], t[6], dz[6], dn = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) { z[k] = v[8 * gi + k]; t[k] = tn[6 * gi + k]; }
                if (ABC_DBG(a.dbg) & 1) { for (int k = 0; k < 6; ++k) dz[k] = z[k] * t[k]; dn = t[0]; } else
                num += (double)class_focal<6, true>(z, t, nullptr, dz, &dn);
                den += (double)dn;
                c.dnl[(2 * mt + gi) * 64 + lane] = dn;
#pragma unroll
                for (int k = 0; k < 6; ++k) dlv[8 * gi + k] = dz[k];
            }
            // ---- loads of the next iteration (weights, targets) and this tile's data-gradient fragments: after the loss
            // arithmetic (its registers are free again), still AHEAD of this tile's stores
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler would hoist these loads over the loss arithmetic: 80 spilled registers)
            {
                const int mtn = mt + 1 < NT ? mt + 1 : mt;
                load_fa(mtn);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        wt[4 * u + mi] = *(const bf16x8*)((const char*)(hd.w2t + (size_t)((2 * mt + u) * 128 + 32 * mi) * 16) + (uint32_t)(r * 32 + h * 16));
                load_t5(mtn);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st_logits && !(ABC_DBG(a.dbg) & 2)) {
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int k = 0; k < 6; ++k) *at4w(hd.logits, k * 60 + 2 * mt + gi) = v[8 * gi + k];
            }
        } else if constexpr (HEAD == 6) {
            // rho: |abs(pred) - rho| * sum_types(t)   (train.py:105,121), f64 like the reference
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = 16 * mt + k;
                if (j < 30) {
                    const float zr = v[k];
                    const double tr = *at8(a.t_rho, j);
                    if (st_logits) *at4w(hd.logits, j) = zr;
                    const float dn = c.dnl[j * 64 + lane];
                    const double diff = (double)fabsf(zr) - tr;
                    num += fabs(diff) * (double)dn;
                    const float sg = (diff > 0.0) ? 1.f : ((diff < 0.0) ? -1.f : 0.f);
                    const float sz = (zr > 0.f) ? 1.f : ((zr < 0.f) ? -1.f : 0.f);
                    dlv[k] = sg * sz * dn;
                }
                if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the 16 bins from being interleaved: registers)
            }
        } else if constexpr (HEAD == 7) {
            // omega: focal per bin weighted by the pixel's weight (train.py:124-125)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = 16 * mt + k;
                if (j < 30) {
                    if (st_logits) *at4w(hd.logits, j) = v[k];
                    float dz;
                    num += (double)center_focal<true>(v[k], c.dnl[j * 64 + lane], (float)wpix, &dz);
                    dlv[k] = dz;
                }
                if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else if (h == 0) {
            if constexpr (HEAD == 0 || HEAD == 4) {
                const float t = *at4(HEAD == 0 ? a.t_atom : a.t_bond, 0);
                if (st_logits) *at4w(hd.logits, 0) = v[0];
                float dz;
                num += (double)center_focal<true>(v[0], t, 1.f, &dz);
                den += (t == 1.f) ? 1.0 : 0.0;
                dlv[0] = dz;
            } else {
                const float* tg = HEAD == 1 ? a.t_types : (HEAD == 2 ? a.t_charges : a.t_hs);
                float z[CH], t[CH], dz[CH], dn = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) t[k] = *at4(tg, k);
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    z[k] = v[k];
                    if (st_logits) *at4w(hd.logits, k) = z[k];
                }
                num += (double)class_focal<CH, true>(z, t, HEAD == 1 ? c_type_w : nullptr, dz, &dn);
                den += (double)dn;
#pragma unroll
                for (int k = 0; k < CH; ++k) dlv[k] = dz[k];
            }
        }

        // ---- d(logits) as bf16: [row][pixel] tile -> the blocked buffer of the weight gradient
        bqp[0] = pack_frag<bf16>(dlv);
        bqp[1] = pack_frag<bf16>(dlv + 8);
        if (!(ABC_DBG(a.dbg) & 4)) {
            char* tl = c.ot;
#pragma unroll
            for (int k = 0; k < 16; ++k) *(bf16*)(tl + ((k & 3) + 8 * (k >> 2) + 4 * h) * TROW + r * 2) = bqp[k >> 3][k & 7];
            lds_sync();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = lane + 64 * j, row = q >> 2, part = q & 3;
                const u32x4 t = *(const u32x4*)(tl + row * TROW + part * 16);
                *(u32x4*)((char*)(hd.dlb + ((size_t)c.chunk * CPAD + 32 * mt) * 128) + (uint32_t)((row * 128 + 32 * c.wave + part * 8) * 2)) = t;
            }
            lds_sync();
        }
        if constexpr (!PIPE) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    wt[4 * u + mi] = *(const bf16x8*)((const char*)(hd.w2t + (size_t)((2 * mt + u) * 128 + 32 * mi) * 16) + (uint32_t)(r * 32 + h * 16));
            if (mt + 1 < NT) dgrad_mfma();
        }
    }
    // the raw features again (the epilogue's LeakyReLU' / dropout mask / xhat), issued ahead of the last tile's MFMAs
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) raw[kk] = *(const u32x4*)(a.y1 + e0 + 16 * kk);
    dgrad_mfma();

    // ---- the small heads also finish conv2's WEIGHT gradient here (small_head_wgrad)
    if constexpr (HEAD < 5) small_head_wgrad<HEAD>(a, c, fb);

    // ---- dA -> g.  The packed W2^T puts feature channel 32 mi + 16 (k >> 3) + 8 h + (k & 7) in register k of accumulator mi:
    // element (kk = 2 mi + (k >> 3), j = k & 7) of this lane's own feature fragments.  g and g * xhat are formed here, then go
    // through the wave's LDS tile (half a slice at a time) so that a lane owns 8 consecutive channels of a pixel: 16-byte stores
    // of g, and per-channel sums over the wave's pixels for BatchNorm's backward.
    if (!(ABC_DBG(a.dbg) & 16)) {
        char* ot = c.ot;
        const int sg = lane & 7, pg = lane >> 3;
        float* ws = c.wsum + c.nslice * 256;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float gv[2][16], gx[2][16];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const int mi = 2 * half + m2;
#pragma unroll
                for (int k8 = 0; k8 < 2; ++k8) {
                    const int kk = 2 * mi + k8, cc = 16 * kk + 8 * h;
                    float sl[8], mu[8], is[8];
                    LoadVec<float, 8>::ld(cf + 256 + cc, sl); LoadVec<float, 8>::ld(cf + 384 + cc, mu); LoadVec<float, 8>::ld(cf + 512 + cc, is);
                    const uint32_t kb = kbits[kk >> 2] >> (8 * (kk & 3)), pb = pbits[kk >> 2] >> (8 * (kk & 3));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t w = raw[kk][j >> 1];
                        const float x = __uint_as_float((j & 1) ? (w & 0xFFFF0000u) : (w << 16));
                        const float m1 = ((pb >> j) & 1u) ? c.dscale : sl[j] * c.dscale;
                        const float gg = ((kb >> j) & 1u) ? accD[mi][8 * k8 + j] * m1 : 0.f;
                        gv[m2][8 * k8 + j] = gg;
                        gx[m2][8 * k8 + j] = gg * ((x - mu[j]) * is[j]);
                    }
                }
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (bf16)(pass ? gx[m2][4 * q + j] : gv[m2][4 * q + j]);
                        *(bf16x4*)(ot + r * HROW + (32 * m2 + 16 * (q >> 1) + 8 * h + 4 * (q & 1)) * 2) = o;
                    }
                lds_sync();
                float acc8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] = 0.f;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int px = it * 8 + pg;
                    const bf16x8 tv = *(const bf16x8*)(ot + px * HROW + sg * 16);
                    if (pass == 0) *(bf16x8*)(a.g + (size_t)(c.pix0 + px) * a.ld + slice + 64 * half + sg * 8) = tv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc8[j] += (float)tv[j];
                }
                lds_sync();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc8[j] += __shfl_xor(acc8[j], 8); acc8[j] += __shfl_xor(acc8[j], 16); acc8[j] += __shfl_xor(acc8[j], 32);
                }
                if (lane < 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ws[pass * 128 + 64 * half + sg * 8 + j] = acc8[j];
                }
            }
        }
    }
    // the blocked conv2 weight gradient (wgrad.hip) re-reads these features: it takes the keep bits from here instead of
    // hashing 128 elements per pixel again (byte kk of this lane half = channels 16 kk + 8 h .. + 7).  (Stored HERE, at the
    // end of the head: placed where the bits are made it cost the 15-tile head 166 spilled registers.)
    if constexpr (HEAD >= 5) {
        if (hd.keep != nullptr) hd.keep[2u * c.pix + h] = make_uint2(kbits[0], kbits[1]);
    }
    c.nslice += 1;
    {
        const double s1 = wave_sum_d(num), s2 = wave_sum_d(den);
        if (lane == 0) {
            lsum[HEAD] = s1;
            if (HEAD != 6) lsum[8 + HEAD] = s2;
            if (HEAD == 5) lsum[8 + 6] = s2;     // rho is normalised by the same sum of bond-type targets (train.py:121)
        }
    }
}

__global__ __launch_bounds__(256, 2) void heads_fused_kernel(const HFK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx c;
    c.lane = threadIdx.x & 63; c.wave = threadIdx.x >> 6;
    c.r = c.lane & 31; c.h = c.lane >> 5;
    c.chunk = blockIdx.x;
    c.pix0 = (uint32_t)c.chunk * 128u + 32u * c.wave;
    c.pix = c.pix0 + c.r;
    c.b = (int)(c.pix0 / (uint32_t)a.HW);
    c.yx = (int)(c.pix - (uint32_t)c.b * (uint32_t)a.HW);
    c.ot = smem + c.wave * WV;
    c.dnl = (float*)(smem + c.wave * WV + WV_DN);
    const int group = blockIdx.y;
    if (ABC_DBG(a.dbg) & (256 << group)) return;    // (debug build: time the work types one by one)
    c.wsum = (float*)(smem + LDS_BSUM) + c.wave * 512;
    c.nslice = 0;
    c.dscale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    c.dseed = a.drop_seed + ((a.drop_p > 0.f && a.drop_salt) ? *a.drop_salt : 0u);
    double* ls = (double*)(smem + LDS_LSUM);
    double* lsum = ls + c.wave * 16;
    if (c.lane < 16) lsum[c.lane] = 0.0;
    lds_sync();
    // (one workgroup = one of 7 work types: every path on its own keeps its registers; run back to back in one wave the
    //  compiler's cross-head scheduling spilled 50 registers)
    switch (group) {
        case 0: run_head<5>(a, c, lsum); run_head<6>(a, c, lsum); break;
        case 1: run_head<7>(a, c, lsum); break;
        case 2: run_head<0>(a, c, lsum); break;
        case 3: run_head<1>(a, c, lsum); break;
        case 4: run_head<2>(a, c, lsum); break;
        case 5: run_head<3>(a, c, lsum); break;
        default: run_head<4>(a, c, lsum); break;
    }
    __syncthreads();
    // one row of 16 sums per 128-pixel chunk; a work type writes the columns of ITS heads (together they cover all 16)
    if (threadIdx.x < 16) {
        const int i = threadIdx.x & 7;   // head of this column (numerator i, denominator 8 + i)
        const bool mine = group == 0 ? (i == 5 || i == 6) : (group == 1 ? i == 7 : i == group - 2);
        if (mine)
            a.losspart[(size_t)c.chunk * 16 + threadIdx.x] = (ls[threadIdx.x] + ls[16 + threadIdx.x]) + (ls[32 + threadIdx.x] + ls[48 + threadIdx.x]);
    }
    // BatchNorm sums of the workgroup's 128 pixels: the four waves' rows, slice by slice in the order the heads ran
    {
        const int nsl = group == 0 ? 2 : 1;
        const float* base = (const float*)(smem + LDS_BSUM);
        for (int i = threadIdx.x; i < nsl * 256; i += 256) {
            const int sl_i = i >> 8, row = (i >> 7) & 1, ch = i & 127;
            const float s = (base[i] + base[512 + i]) + (base[1024 + i] + base[1536 + i]);
            const int head = group == 0 ? 5 + sl_i : (group == 1 ? 7 : group - 2);
            a.bnpart[((size_t)c.chunk * 2 + row) * a.ld + 128 * head + ch] = s;
        }
    }
}

// conv2 weights / biases of all heads into the two fragment layouts above (every step: the weights change)
struct HFPackK {
    const float* w2[HF_NH];
    const float* b2[HF_NH];
    bf16* pack;
};

__global__ __launch_bounds__(256) void heads_fused_pack_kernel(const HFPackK a) {
    const int head = blockIdx.y;
    const int cpad = hf_tiles(head) * 32;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cpad * 128) return;
    char* base = (char*)a.pack + hf_pack_off(head);
    bf16* w2f = (bf16*)base;
    bf16* w2t = (bf16*)(base + (size_t)cpad * 256);
    float* biasp = (float*)(base + (size_t)cpad * 512);
    const float* w = a.w2[head];
    {   // forward: [chunk][row m][32]
        const int within = idx & 31, m = (idx >> 5) % cpad, chunk = idx / (32 * cpad);
        const int ch = hf_chan_of_row(head, m);
        w2f[idx] = (bf16)(ch >= 0 ? w[ch * 128 + chunk * 32 + within] : 0.f);
    }
    {   // data gradient: [K-step s][feature ci][half h][8]: slot j of half h = accumulator register 8 (s & 1) + j
        const int j = idx & 7, h = (idx >> 3) & 1, mrow = (idx >> 4) & 127, s = idx >> 11;
        const int ci = (mrow & ~12) | ((mrow & 4) << 1) | ((mrow & 8) >> 1);   // accumulator row -> feature channel (see the epilogue)
        const int k = 8 * (s & 1) + j;
        const int m = 32 * (s >> 1) + (k & 3) + 8 * (k >> 2) + 4 * h;
        const int ch = hf_chan_of_row(head, m);
        w2t[idx] = (bf16)(ch >= 0 ? w[ch * 128 + ci] : 0.f);
    }
    if (idx < cpad) {
        const int ch = hf_chan_of_row(head, idx);
        biasp[idx] = ch >= 0 ? a.b2[head][ch] : 0.f;
    }
}

static int hf_check(const abc_heads_fused_desc* d) {
    if (d->B < 1 || d->h < 1 || d->w < 1 || (d->h * d->w) % 128) return abc_fail(ABC_EINVAL, "heads_fused: the map must hold whole 128-pixel chunks");
    if (d->ld < 128 * HF_NH || d->ld % 8) return abc_fail(ABC_EINVAL, "heads_fused: feature stride");
    if ((int64_t)d->B * d->h * d->w * d->ld >= (int64_t(1) << 32) || (int64_t)d->B * d->h * d->w * 360 >= (int64_t(1) << 31))
        return abc_fail(ABC_EUNSUPPORTED, "heads_fused: 32-bit element offsets");
    return ABC_OK;
}

}  // namespace

extern "C" int64_t abc_heads_fused_pack_bytes(void) { return hf_pack_off(HF_NH); }
extern "C" int abc_heads_fused_chunks(const abc_heads_fused_desc* d) { return d->B * d->h * d->w / 128; }
extern "C" int abc_heads_fused_loss_blocks(const abc_heads_fused_desc* d) { return abc_heads_fused_chunks(d); }
extern "C" int64_t abc_heads_fused_dl_elems(const abc_heads_fused_desc* d) { return (int64_t)abc_heads_fused_chunks(d) * hf_rows_total() * 128; }
extern "C" int abc_heads_fused_rows(int32_t head) { return head >= 0 && head < HF_NH ? hf_tiles(head) * 32 : -1; }
extern "C" int abc_heads_fused_chan_of_row(int32_t head, int32_t row) {
    return (head >= 0 && head < HF_NH && row >= 0 && row < hf_tiles(head) * 32) ? hf_chan_of_row(head, row) : -1;
}

extern "C" int abc_heads_fused_pack(const abc_heads_fused_desc* d, abc_stream_t stream) {
    HFPackK k;
    for (int i = 0; i < HF_NH; ++i) { k.w2[i] = d->w2[i]; k.b2[i] = d->b2[i]; }
    k.pack = (bf16*)d->w2_pack;
    hipLaunchKernelGGL(heads_fused_pack_kernel, dim3(abc_cdiv(480 * 128, 256), HF_NH), dim3(256), 0, (hipStream_t)stream, k);
    return abc_check_launch("heads_fused_pack");
}

extern "C" int abc_heads_fused_fwd_bwd(const abc_heads_fused_desc* d, abc_stream_t stream) {
    if (int rc = hf_check(d)) return rc;
    HFK k;
    k.y1 = (const bf16*)d->feat; k.sc = d->scale; k.sh = d->shift; k.sl = d->slope; k.mean = d->mean; k.invstd = d->invstd;
    k.g = (bf16*)d->g; k.ld = d->ld;
    k.drop_p = d->drop_p; k.drop_seed = d->drop_seed; k.drop_salt = d->drop_salt; k.drop_thr = abc_drop_threshold(d->drop_p);
    k.t_atom = d->t_atom; k.t_types = d->t_types; k.t_charges = d->t_charges; k.t_hs = d->t_hs; k.t_bond = d->t_bond;
    k.t_btypes = d->t_btypes; k.t_rho = d->t_rho; k.t_omega = d->t_omega;
    if ((d->target_flags != nullptr) != (d->zero_bytes != nullptr)) return abc_fail(ABC_EINVAL, "heads_fused: target_flags and zero_bytes go together");
    k.tflags = d->target_flags; k.tzero = (const char*)d->zero_bytes;
    k.bnpart = d->bn_partial; k.losspart = d->loss_partial; k.dwsmall = d->wgrad_work;
    k.HW = d->h * d->w; k.nchunk = abc_heads_fused_chunks(d);
    { const char* e = abc_knob("ABC_HF_DBG"); k.dbg = e ? atoi(e) : 0; }   // (debug build only: phase ablations)
    size_t row0 = 0;
    for (int i = 0; i < HF_NH; ++i) {
        const int cpad = hf_tiles(i) * 32;
        char* base = (char*)d->w2_pack + hf_pack_off(i);
        k.hd[i].w2f = (const bf16*)base;
        k.hd[i].w2t = (const bf16*)(base + (size_t)cpad * 256);
        k.hd[i].biasp = (const float*)(base + (size_t)cpad * 512);
        k.hd[i].logits = d->logits[i];
        k.hd[i].dlb = (bf16*)d->dl + row0 * (size_t)k.nchunk * 128;
        k.hd[i].keep = (i >= 5 && d->keep_mask != nullptr && d->drop_p > 0.f) ? (uint2*)d->keep_mask + (size_t)(i - 5) * 2 * d->B * d->h * d->w : nullptr;
        row0 += cpad;
    }
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)heads_fused_kernel, HF_LDS, &lds_ok)) return rc;
    hipLaunchKernelGGL(heads_fused_kernel, dim3(k.nchunk, HF_GROUPS), dim3(256), HF_LDS, (hipStream_t)stream, k);
    return abc_check_launch("heads_fused_fwd_bwd");
}
