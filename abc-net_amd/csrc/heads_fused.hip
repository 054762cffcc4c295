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
};

struct HFK {
    const bf16* y1;
    const float *sc, *sh, *sl, *mean, *invstd;
    bf16* g;
    int ld;
    float drop_p;
    uint32_t drop_seed;
    const uint32_t* drop_salt;
    const float *t_atom, *t_types, *t_charges, *t_hs, *t_bond, *t_btypes;
    const double *t_rho, *t_omega;
    float* bnpart;      // [nchunk][2][ld]
    double* losspart;   // [2 nchunk][16]
    int HW, nchunk, dbg;
    HFHead hd[HF_NH];
};

constexpr int OROW = 128 * 2 + 16;         // row of the g transpose tile: 128 feature channels bf16 + pad
constexpr int TROW = 32 * 2 + 16;          // row of the d(logits) tile: 32 pixels bf16 + pad
constexpr int WV_OT = 32 * OROW;           // 8704 (the d(logits) tile aliases it: 32 x 80)
constexpr int WV_DN = 30 * 64 * 4;         // per-lane sum of the bond-type targets of each of its 30 omega bins
constexpr int WV = WV_OT + WV_DN;
constexpr int LDS_BSUM = 4 * WV;           // [2 buffers][4 waves][2][128] f32
constexpr int LDS_LSUM = LDS_BSUM + 2 * 4 * 2 * 128 * 4;   // [4 waves][16] f64
constexpr int HF_LDS = LDS_LSUM + 4 * 16 * 8;

struct Ctx {
    int lane, r, h, wave, chunk, b, yx, parity;
    uint32_t pix, pix0;
    char* ot;
    float* dnl;
    float* bsum;
    float dscale;
    uint32_t dseed;
};

__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One head for the wave's 32 pixels.  num / den: this lane's share of the head's loss numerator / denominator.
template <int HEAD>
__device__ inline void run_head(const HFK& a, Ctx& c, double& num, double& den) {
    constexpr int CH = hf_ch(HEAD), NT = hf_tiles(HEAD), CPAD = NT * 32;
    const HFHead& hd = a.hd[HEAD];
    const int slice = 128 * HEAD;
    const int r = c.r, h = c.h, lane = c.lane;
    const size_t HW = (size_t)a.HW;
    // plane (b, ch) of a C-channel NCHW map at this lane's pixel
    auto pl = [&](int C, int ch) -> size_t { return ((size_t)c.b * C + ch) * HW + c.yx; };

    // ---- features of the wave's pixels as B fragments: lane (pixel r, half h) holds channels 16 kk + 8 h .. + 8
    bf16x8 fb[8];
    {
        const uint32_t e0 = c.pix * (uint32_t)a.ld + slice + 8 * h;
        u32x4 raw[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) raw[kk] = *(const u32x4*)(a.y1 + e0 + 16 * kk);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float v[8], sc[8], sh[8], sl[8];
            const int cc = slice + 16 * kk + 8 * h;
            LoadVec<float, 8>::ld(a.sc + cc, sc); LoadVec<float, 8>::ld(a.sh + cc, sh); LoadVec<float, 8>::ld(a.sl + cc, sl);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(raw[kk][j] << 16); v[2 * j + 1] = __uint_as_float(raw[kk][j] & 0xFFFF0000u); }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = abc_act(v[j], sc[j], sh[j], sl[j]);
            if (a.drop_p > 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = abc_drop_keep(e0 + 16 * kk + j, c.dseed, a.drop_p) ? v[j] * c.dscale : 0.f;
            }
            fb[kk] = pack_frag<bf16>(v);
        }
    }

    // omega: the per-pixel weight is the sum of the pixel's 60 omega targets (train.py:124): this lane's 30 + the other half's
    // (the targets are parked in the wave's LDS slots the bond-type group uses for its sums: this group has no such sums)
    double wpix = 0.0;
    if constexpr (HEAD == 7) {
        double wp = 0.0;
#pragma unroll 6
        for (int j = 0; j < 30; ++j) {
            const double t = a.t_omega[pl(60, 30 * h + j)];
            wp += t;
            c.dnl[j * 64 + lane] = (float)t;
        }
        wpix = wp + __shfl_xor(wp, 32);
        den = (h == 0) ? wpix : 0.0;
    }

    f32x16 accD[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int k = 0; k < 16; ++k) accD[mi][k] = 0.f;

#pragma unroll 1
    for (int mt = 0; mt < NT; ++mt) {
        // ---- logits of 32 packed rows x 32 pixels
        bf16x8 fa[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            fa[kk] = *(const bf16x8*)(hd.w2f + ((size_t)((kk >> 1) * CPAD + 32 * mt + r) * 32 + 16 * (kk & 1) + 8 * h));
        f32x4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = *(const f32x4*)(hd.biasp + 32 * mt + 8 * q + 4 * h);
        f32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[kk], acc, 0, 0, 0);
        float v[16], dlv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = acc[k] + b4[k >> 2][k & 3]; dlv[k] = 0.f; }

        // ---- loss + d(logits): register k of lane half h = packed row 32 mt + (k & 3) + 8 (k >> 2) + 4 h (hf_chan_of_row)
        if constexpr (HEAD == 5) {
            // registers 8 gi .. 8 gi + 5 = the six bond types of omega bin 30 h + 2 mt + gi (train.py:101 view)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int o = 30 * h + 2 * mt + gi;
                float z[6], t[6], dz[6], dn = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    z[k] = v[8 * gi + k];
                    const size_t at = pl(360, k * 60 + o);
                    t[k] = a.t_btypes[at];
                    if (!(ABC_DBG(a.dbg) & 2)) hd.logits[at] = z[k];
                }
                if (ABC_DBG(a.dbg) & 1) { for (int k = 0; k < 6; ++k) dz[k] = z[k] * t[k]; dn = t[0]; } else
                num += (double)class_focal<6>(z, t, nullptr, dz, &dn);
                den += (double)dn;
                c.dnl[(2 * mt + gi) * 64 + lane] = dn;
#pragma unroll
                for (int k = 0; k < 6; ++k) dlv[8 * gi + k] = dz[k];
            }
        } else if constexpr (HEAD == 6) {
            // rho: |abs(pred) - rho| * sum_types(t)   (train.py:105,121), f64 like the reference
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = 16 * mt + k;
                if (j < 30) {
                    const size_t at = pl(60, 30 * h + j);
                    const float zr = v[k];
                    hd.logits[at] = zr;
                    const double tr = a.t_rho[at];
                    const float dn = c.dnl[j * 64 + lane];
                    const double diff = (double)fabsf(zr) - tr;
                    num += fabs(diff) * (double)dn;
                    const float sg = (diff > 0.0) ? 1.f : ((diff < 0.0) ? -1.f : 0.f);
                    const float sz = (zr > 0.f) ? 1.f : ((zr < 0.f) ? -1.f : 0.f);
                    dlv[k] = sg * sz * dn;
                }
            }
        } else if constexpr (HEAD == 7) {
            // omega: focal per bin weighted by the pixel's weight (train.py:124-125)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = 16 * mt + k;
                if (j < 30) {
                    hd.logits[pl(60, 30 * h + j)] = v[k];
                    float dz;
                    num += (double)center_focal(v[k], c.dnl[j * 64 + lane], (float)wpix, &dz);
                    dlv[k] = dz;
                }
            }
        } else if (h == 0) {
            if constexpr (HEAD == 0 || HEAD == 4) {
                const size_t at = pl(1, 0);
                const float t = (HEAD == 0 ? a.t_atom : a.t_bond)[at];
                hd.logits[at] = v[0];
                float dz;
                num += (double)center_focal(v[0], t, 1.f, &dz);
                den += (t == 1.f) ? 1.0 : 0.0;
                dlv[0] = dz;
            } else {
                const float* tg = HEAD == 1 ? a.t_types : (HEAD == 2 ? a.t_charges : a.t_hs);
                float z[CH], t[CH], dz[CH], dn = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const size_t at = pl(CH, k);
                    z[k] = v[k];
                    t[k] = tg[at];
                    hd.logits[at] = z[k];
                }
                num += (double)class_focal<CH>(z, t, HEAD == 1 ? c_type_w : nullptr, dz, &dn);
                den += (double)dn;
#pragma unroll
                for (int k = 0; k < CH; ++k) dlv[k] = dz[k];
            }
        }

        // ---- d(logits) as bf16: [row][pixel] tile -> the blocked buffer of the weight gradient
        bf16x8 bq[2];
        bq[0] = pack_frag<bf16>(dlv);
        bq[1] = pack_frag<bf16>(dlv + 8);
        if (!(ABC_DBG(a.dbg) & 4)) {
            char* tl = c.ot;
#pragma unroll
            for (int k = 0; k < 16; ++k) *(bf16*)(tl + ((k & 3) + 8 * (k >> 2) + 4 * h) * TROW + r * 2) = bq[k >> 3][k & 7];
            lds_sync();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = lane + 64 * j, row = q >> 2, part = q & 3;
                const u32x4 t = *(const u32x4*)(tl + row * TROW + part * 16);
                *(u32x4*)(hd.dlb + ((size_t)c.chunk * CPAD + 32 * mt + row) * 128 + 32 * c.wave + part * 8) = t;
            }
            lds_sync();
        }
        // ---- data gradient: dA[ci][p] += sum over the tile's 32 rows; K-step u = registers 8 u .. 8 u + 7 of both halves
        if (!(ABC_DBG(a.dbg) & 8))
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const bf16x8 af = *(const bf16x8*)(hd.w2t + ((size_t)((2 * mt + u) * 128 + 32 * mi + r) * 2 + h) * 8);
                accD[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[u], accD[mi], 0, 0, 0);
            }
    }

    // ---- dA -> g: transpose through the wave's LDS tile ([32 pixels][128 channels]) so that a lane owns 8 consecutive
    // channels of a pixel (16-byte loads of the raw feature, 16-byte stores of g); LeakyReLU' and the dropout mask from
    // the raw feature; per-channel sums for BatchNorm's backward
    if (!(ABC_DBG(a.dbg) & 16)) {
        char* ot = c.ot;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)accD[mi][4 * q + j];
                *(bf16x4*)(ot + r * OROW + (32 * mi + 8 * q + 4 * h) * 2) = o;
            }
        lds_sync();
        const int sg = lane & 15, pg = lane >> 4;
        const int cc = slice + sg * 8;
        float sc[8], sh[8], sl[8], mu[8], is[8], a1[8], a2[8];
        LoadVec<float, 8>::ld(a.sc + cc, sc); LoadVec<float, 8>::ld(a.sh + cc, sh); LoadVec<float, 8>::ld(a.sl + cc, sl);
        LoadVec<float, 8>::ld(a.mean + cc, mu); LoadVec<float, 8>::ld(a.invstd + cc, is);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = it * 4 + pg;
            const bf16x8 dav = *(const bf16x8*)(ot + px * OROW + sg * 16);
            const uint32_t e = (c.pix0 + px) * (uint32_t)a.ld + cc;
            const bf16x8 raw = *(const bf16x8*)(a.y1 + e);
            float out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = (float)raw[j];
                const float y = fmaf(x, sc[j], sh[j]);
                float gg = (float)dav[j] * (y > 0.f ? 1.f : sl[j]);
                if (a.drop_p > 0.f) gg = abc_drop_keep(e + j, c.dseed, a.drop_p) ? gg * c.dscale : 0.f;
                out[j] = gg;
                a1[j] += gg;
                a2[j] += gg * ((x - mu[j]) * is[j]);
            }
            *(bf16x8*)(a.g + e) = pack_frag<bf16>(out);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a1[j] += __shfl_xor(a1[j], 16); a1[j] += __shfl_xor(a1[j], 32);
            a2[j] += __shfl_xor(a2[j], 16); a2[j] += __shfl_xor(a2[j], 32);
        }
        float* bs = c.bsum + c.parity * (4 * 2 * 128);
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bs[(c.wave * 2 + 0) * 128 + sg * 8 + j] = a1[j];
                bs[(c.wave * 2 + 1) * 128 + sg * 8 + j] = a2[j];
            }
        }
        __syncthreads();   // (every wave of the workgroup runs the same list of heads)
        {
            const int row = threadIdx.x >> 7, ch = threadIdx.x & 127;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += bs[(w * 2 + row) * 128 + ch];
            a.bnpart[((size_t)c.chunk * 2 + row) * a.ld + slice + ch] = s;
        }
        c.parity ^= 1;
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
    c.dnl = (float*)(smem + c.wave * WV + WV_OT);
    c.bsum = (float*)(smem + LDS_BSUM);
    c.parity = 0;
    c.dscale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    c.dseed = a.drop_seed + ((a.drop_p > 0.f && a.drop_salt) ? *a.drop_salt : 0u);
    double num[8], den[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { num[i] = 0.0; den[i] = 0.0; }
    const int group = blockIdx.y;
    if (group == 0) {
        run_head<5>(a, c, num[5], den[5]);
        den[6] = den[5];                     // rho is normalised by the same sum of bond-type targets (train.py:121)
        double unused = 0.0;
        run_head<6>(a, c, num[6], unused);
    } else {
        run_head<7>(a, c, num[7], den[7]);
        run_head<0>(a, c, num[0], den[0]);
        run_head<1>(a, c, num[1], den[1]);
        run_head<2>(a, c, num[2], den[2]);
        run_head<3>(a, c, num[3], den[3]);
        run_head<4>(a, c, num[4], den[4]);
    }
    double* ls = (double*)(smem + LDS_LSUM);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double s1 = wave_sum_d(num[i]), s2 = wave_sum_d(den[i]);
        if (c.lane == 0) { ls[c.wave * 16 + i] = s1; ls[c.wave * 16 + 8 + i] = s2; }
    }
    __syncthreads();
    if (threadIdx.x < 16)
        a.losspart[((size_t)group * a.nchunk + c.chunk) * 16 + threadIdx.x] =
            (ls[threadIdx.x] + ls[16 + threadIdx.x]) + (ls[32 + threadIdx.x] + ls[48 + threadIdx.x]);
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
        const int j = idx & 7, h = (idx >> 3) & 1, ci = (idx >> 4) & 127, s = idx >> 11;
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
    if ((int64_t)d->B * d->h * d->w * d->ld >= (int64_t(1) << 32)) return abc_fail(ABC_EUNSUPPORTED, "heads_fused: 32-bit element offsets");
    return ABC_OK;
}

}  // namespace

extern "C" int64_t abc_heads_fused_pack_bytes(void) { return hf_pack_off(HF_NH); }
extern "C" int abc_heads_fused_chunks(const abc_heads_fused_desc* d) { return d->B * d->h * d->w / 128; }
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
    k.drop_p = d->drop_p; k.drop_seed = d->drop_seed; k.drop_salt = d->drop_salt;
    k.t_atom = d->t_atom; k.t_types = d->t_types; k.t_charges = d->t_charges; k.t_hs = d->t_hs; k.t_bond = d->t_bond;
    k.t_btypes = d->t_btypes; k.t_rho = d->t_rho; k.t_omega = d->t_omega;
    k.bnpart = d->bn_partial; k.losspart = d->loss_partial;
    k.HW = d->h * d->w; k.nchunk = abc_heads_fused_chunks(d);
    { const char* e = getenv("ABC_HF_DBG"); k.dbg = e ? atoi(e) : 0; }
    size_t row0 = 0;
    for (int i = 0; i < HF_NH; ++i) {
        const int cpad = hf_tiles(i) * 32;
        char* base = (char*)d->w2_pack + hf_pack_off(i);
        k.hd[i].w2f = (const bf16*)base;
        k.hd[i].w2t = (const bf16*)(base + (size_t)cpad * 256);
        k.hd[i].biasp = (const float*)(base + (size_t)cpad * 512);
        k.hd[i].logits = d->logits[i];
        k.hd[i].dlb = (bf16*)d->dl + row0 * (size_t)k.nchunk * 128;
        row0 += cpad;
    }
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)heads_fused_kernel, HF_LDS, &lds_ok)) return rc;
    hipLaunchKernelGGL(heads_fused_kernel, dim3(k.nchunk, 2), dim3(256), HF_LDS, (hipStream_t)stream, k);
    return abc_check_launch("heads_fused_fwd_bwd");
}
