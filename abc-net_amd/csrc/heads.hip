// Forward of the heads' 1x1 convolutions (unet.py:70, out_modules[i].conv2): logits[co][p] = b[co] + sum_ci W[co][ci] a[p][ci]
// with a = dropout(LeakyReLU(BN(h))) applied on load and the output in the reference's NCHW f32 layout.
//
// GEMM view: M = output channels (<= 360), N = pixels, K = 128.  Both operands already have the MFMA fragment layout in
// memory: a pixel's 128 channels are contiguous (NHWC) = the B operand's 8 consecutive k per lane, the packed weights
// give the A operand, and the accumulator layout (lane = pixel column, registers = output channels) stores straight into
// channel-planar rows, 128 contiguous bytes per row.  So: no LDS, no barriers; every wave owns 64 pixels (two 32-pixel
// tiles whose activated fragments stay in registers) and walks the m-tiles.  HBM-bound on the logits it writes.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "conv_fast.hpp"
#include <stdlib.h>

namespace {

struct HeadFwdK {
    const void* x;
    const float *sc, *sh, *sl;
    const void* w;        // packed [1 tap][4 chunks][Cout_pad][32] bf16
    const float* bias;
    float* y;             // [B][ctot][HW]
    int HW, ldx, cin_off, Cout, Cout_pad, ctot, cout_off, npairs;
    float drop_p;
    uint32_t drop_seed;
    const uint32_t* drop_salt;
    unsigned bytesX, bytesW;
    int f8;                // e4m3 features and weights (the fp8 inference graph): oscale[co] = s_x * s_w[co]
    const float* oscale;
    float* aux;            // abc_conv_desc.head_aux: [B][Cout][HW]
    int aux_mode;          // 1: |v|; 2: circular 3-tap local maximum along the channel axis and v > -1; 3: arg max over six channel groups (uint8)
};

// e4m3 form (fp8 inference graph: finished features, no transform on load): a pixel's 128 channels are 128 bytes = two K = 64
// steps of the block-scaled MFMA (unit scales); packed weights [1 tap][2 chunks][Cout_pad][64]; logits = acc * oscale[co] + b[co]
__device__ inline void head_fwd_f8_body(const HeadFwdK& a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.npairs) return;
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX), rsW = abc_make_rsrc(a.w, a.bytesW);
    i32x8 fb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t pix = (uint32_t)pair * 64u + t * 32 + r;
        const uint32_t e0 = pix * (uint32_t)a.ldx + (uint32_t)a.cin_off + 32 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            fb[t][s] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsX, e0 + 64 * s, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsX, e0 + 64 * s + 16, 0, 0));
    }
    const int b = (pair * 64) / a.HW, pp = pair * 64 - b * a.HW;
    const int mtiles = a.Cout_pad / 32;
    for (int mt = 0; mt < mtiles; ++mt) {
        i32x8 fa[2];
        const int co = mt * 32 + r;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned off = (unsigned)((s * a.Cout_pad + co) * 64 + 32 * h);
            fa[s] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsW, off + 16, 0, 0));
        }
        float bvv[16], osc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
            const bool ok = oc < a.Cout;
            bvv[k] = (a.bias && ok) ? a.bias[oc] : 0.f;
            osc[k] = ok ? a.oscale[oc] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) mma32B_f8(acc, fa[s], fb[t][s]);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
                if (oc < a.Cout) {
                    const float v = fmaf(acc[k], osc[k], bvv[k]);
                    if (a.y) a.y[((size_t)(b * a.ctot + a.cout_off + oc)) * a.HW + pp + t * 32 + r] = v;
                    if (a.aux_mode == 1) a.aux[((size_t)(b * a.Cout + oc)) * a.HW + pp + t * 32 + r] = fabsf(v);
                }
            }
        }
    }
}

__device__ inline void head_fwd_body(const HeadFwdK& a) {
    if (a.f8) { head_fwd_f8_body(a); return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pair = blockIdx.x * 4 + wave;   // 64 consecutive pixels (never straddling an image: HW % 64 == 0)
    if (pair >= a.npairs) return;
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX), rsW = abc_make_rsrc(a.w, a.bytesW);
    const bool tr = a.sc != nullptr;
    const float dscale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const uint32_t dseed = a.drop_seed + ((a.drop_p > 0.f && a.drop_salt) ? *a.drop_salt : 0u);

    // ---- B fragments of both pixel tiles: lane (pixel r, half h) takes channels 16 kk + 8 h .. + 8 of its pixel
    bf16x8 fb[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t pix = (uint32_t)pair * 64u + t * 32 + r;
        const uint32_t e0 = pix * (uint32_t)a.ldx + (uint32_t)a.cin_off + 8 * h;
        u32x4 raw[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) raw[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (e0 + 16 * kk) * 2u, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(raw[kk][j] << 16); v[2 * j + 1] = __uint_as_float(raw[kk][j] & 0xFFFF0000u); }
            const int c = a.cin_off + 16 * kk + 8 * h;
            if (tr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = abc_act(v[j], a.sc[c + j], a.sh[c + j], a.sl[c + j]);
            }
            if (a.drop_p > 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = abc_drop_keep(e0 + 16 * kk + j, dseed, a.drop_p) ? v[j] * dscale : 0.f;
            }
            fb[t][kk] = pack_frag<bf16>(v);
        }
    }
    const int b = (pair * 64) / a.HW, pp = pair * 64 - b * a.HW;
    const int mtiles = a.Cout_pad / 32;
    for (int mt = 0; mt < mtiles; ++mt) {
        // A fragments of this m-tile: row co = 32 mt + r, channels 16 kk + 8 h .. + 8 (chunk kk / 2 of the packed layout)
        bf16x8 fa[8];
        const int co = mt * 32 + r;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const unsigned off = (unsigned)((((kk >> 1) * a.Cout_pad + co) * 32 + 16 * (kk & 1) + 8 * h) * 2);
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
            fa[kk] = *(const bf16x8*)&t;
        }
        float bvv[16];  // bias of this lane's 16 output rows of the m-tile (loaded once, both pixel tiles use it)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
            bvv[k] = (a.bias && oc < a.Cout) ? a.bias[oc] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[t][kk], acc, 0, 0, 0);
            // accumulator: column = pixel r of tile t, register k = output channel 32 mt + (k & 3) + 8 (k >> 2) + 4 h
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
                if (oc < a.Cout) {
                    const float v = acc[k] + bvv[k];
                    if (a.y) a.y[((size_t)(b * a.ctot + a.cout_off + oc)) * a.HW + pp + t * 32 + r] = v;
                    if (a.aux_mode == 1) a.aux[((size_t)(b * a.Cout + oc)) * a.HW + pp + t * 32 + r] = fabsf(v);
                }
            }
        }
    }
}


__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadFwdK a) { head_fwd_body(a); }

// ---- the omega head with its NMS mask (abc_conv_desc.head_aux_mode == 2; img2smiles2.py:75-79): a kernel of its own -- the mask
// needs all 60 channels of a pixel at once, 2 x 16 accumulator values per lane and as many from the partner lane, which the plain
// form's register budget (both pixel tiles' fragments resident) does not have.  Here a wave walks its two 32-pixel tiles one
// after the other.  The lane (pixel r, half h) holds the channels 32 mt + (k & 3) + 8 (k >> 2) + 4 h of its pixel; the partner lane
// r + 32 the other half of every group of eight (one cross-lane exchange).  mask = (max(prev, cur, next) == cur && cur > -1) over
// the circular channel axis -- on the very f32 values that are stored as the raw map (when y is given).
template <bool F8>
__global__ __launch_bounds__(256, 2) void head_fwd_omega_kernel(const HeadFwdK a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.npairs) return;
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX), rsW = abc_make_rsrc(a.w, a.bytesW);
    const int b = (pair * 64) / a.HW, pp = pair * 64 - b * a.HW;
    const bool tr = a.sc != nullptr;
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        const uint32_t pix = (uint32_t)pair * 64u + t * 32 + r;
        float v[2][16];
        if constexpr (F8) {
            const uint32_t e0 = pix * (uint32_t)a.ldx + (uint32_t)a.cin_off + 32 * h;
            i32x8 fb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                fb[s] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsX, e0 + 64 * s, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsX, e0 + 64 * s + 16, 0, 0));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                i32x8 fa[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned off = (unsigned)((s * a.Cout_pad + mt * 32 + r) * 64 + 32 * h);
                    fa[s] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsW, off + 16, 0, 0));
                }
                f32x16 acc;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) mma32B_f8(acc, fa[s], fb[s]);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
                    const bool ok = oc < a.Cout;
                    v[mt][k] = fmaf(acc[k], ok ? a.oscale[oc] : 0.f, (a.bias && ok) ? a.bias[oc] : 0.f);
                }
            }
        } else {
            const uint32_t e0 = pix * (uint32_t)a.ldx + (uint32_t)a.cin_off + 8 * h;
            bf16x8 fb[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsX, (e0 + 16 * kk) * 2u, 0, 0);
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(raw[j] << 16); f[2 * j + 1] = __uint_as_float(raw[j] & 0xFFFF0000u); }
                if (tr) {
                    const int c = a.cin_off + 16 * kk + 8 * h;
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = abc_act(f[j], a.sc[c + j], a.sh[c + j], a.sl[c + j]);
                }
                fb[kk] = pack_frag<bf16>(f);      // (no dropout: an evaluation-time output)
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const unsigned off = (unsigned)((((kk >> 1) * a.Cout_pad + mt * 32 + r) * 32 + 16 * (kk & 1) + 8 * h) * 2);
                    const u32x4 tw = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&tw, fb[kk], acc, 0, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
                    v[mt][k] = acc[k] + ((a.bias && oc < a.Cout) ? a.bias[oc] : 0.f);
                }
            }
        }
        float o[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) o[m][k] = __shfl_xor(v[m][k], 32);
        // value of channel cc as a lane of half hh sees it (compile-time cc, hh: one fixed register of v or o)
        auto at = [&](int cc, int hh) -> float {
            const int mm = cc >> 5, rem = cc & 31, hc = (rem >> 2) & 1, kk = (rem & 3) + 4 * (rem >> 3);
            return hc == hh ? v[mm][kk] : o[mm][kk];
        };
        const size_t p0 = (size_t)pp + t * 32 + r;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c0 = 32 * m + (k & 3) + 8 * (k >> 2), c1 = c0 + 4;      // this register's channel for h = 0 / h = 1
                float l0 = 0.f, r0 = 0.f, l1 = 0.f, r1 = 0.f;                      // (60 channels: the wrap-around pairs 59 with 0)
                if (c0 < 60) { l0 = at(c0 == 0 ? 59 : c0 - 1, 0); r0 = at(c0 == 59 ? 0 : c0 + 1, 0); }
                if (c1 < 60) { l1 = at(c1 - 1, 1); r1 = at(c1 == 59 ? 0 : c1 + 1, 1); }
                const float cur = v[m][k];
                const float l = h ? l1 : l0, rr = h ? r1 : r0;
                const int c = h ? c1 : c0;
                if (c < 60) {
                    if (a.y) a.y[((size_t)(b * a.ctot + a.cout_off + c)) * a.HW + p0] = cur;
                    a.aux[((size_t)(b * 60 + c)) * a.HW + p0] = (cur >= l && cur >= rr && cur > -1.f) ? 1.f : 0.f;
                }
            }
        }
    }
}

// ---- a head whose maps are consumed as an arg-max over GROUPS of channel planes (abc_conv_desc.head_aux_mode == 3; the bond-type head:
// img2smiles2.py:71 views its 360 channels as 6 types x 60 omega bins and :112 takes argmax over the 6): the raw maps are not stored at
// all (1.5 GB of f32 per batch of 64 at 512 x 512 that only ever feed six-way comparisons), one byte per (bin, pixel) is.  A wave owns
// 64 pixels as in the plain form and walks m-tiles made of the SAME 32 bins of each of the six groups one after the other (row
// g * nb + 32 bb + r of the packed weights; a row past the bins of the last block reads zeros), so the six candidates of a
// (bin, pixel) arrive in the same accumulator register of the same lane: a running (max, arg) pair per register, first maximum wins
// as in torch.argmax.  The values compared are the f32 values the plain form would have stored.
// Persistent 8-wave workgroups with the head's packed weights in LDS (98 KB bf16 / 49 KB e4m3: every wave of the plain form
// re-fetches them from the L2 per 64 pixels, and with two waves per SIMD each of the twelve m-tile steps then waits out an L2 round
// trip: that form ran this head at 1.1 TB/s, 319 us per batch of 64).  The 64-byte rows are stored with their four 16-byte slots
// XOR-ed by (row >> 2) & 3: the 16 lanes of a ds_read_b128 group (rows c .. c+3, c+12 .. c+15, c+20 .. c+27) then hit 16 banks.
template <bool F8>
__global__ __launch_bounds__(512, 1) void head_fwd_group_kernel(const HeadFwdK a) {
    extern __shared__ __attribute__((aligned(16))) char wl[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsX = abc_make_rsrc(a.x, a.bytesX), rsW = abc_make_rsrc(a.w, a.bytesW);
    const int nb = a.Cout / 6, nbb = (nb + 31) >> 5;
    const bool tr = a.sc != nullptr;
    constexpr int ROWB = 64;                      // bytes of a packed weight row of one chunk: 64 e4m3 / 32 bf16
    constexpr int SLOTS = ROWB / 16;
    // bias / dequantisation factor per (group, bin), bins padded to whole blocks of 32: [6][64] floats each behind the weights
    float* sbias = (float*)(wl + a.bytesW);
    float* sosc = sbias + 6 * 64;
    {
        // packed weights [chunk][Cout_pad][ROWB] -> LDS, 16 bytes at a time, slot s of row co at s ^ ((co >> 2) & 3)
        const int nseg = (int)(a.bytesW / 16);
        for (int i = threadIdx.x; i < nseg; i += 512) {
            const int row = i / SLOTS, sl = i - row * SLOTS;
            const int co = row % a.Cout_pad;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, (unsigned)i * 16u, 0, 0);
            *(u32x4*)(wl + row * ROWB + ((sl ^ ((co >> 2) & 3)) * 16)) = v;
        }
        for (int i = threadIdx.x; i < 6 * 64; i += 512) {
            const int g = i >> 6, bin = i & 63;
            const bool ok = bin < nb;
            sbias[i] = (a.bias && ok) ? a.bias[g * nb + bin] : 0.f;
            sosc[i] = (F8 && ok) ? a.oscale[g * nb + bin] : 0.f;
        }
    }
    __syncthreads();
    // the arg-max map through a buffer descriptor: a lane's part of the address is its pixel, the (image, bin) plane a scalar offset
    const __amdgpu_buffer_rsrc_t rsO = abc_make_rsrc(a.aux, (unsigned)((size_t)(a.npairs * 64 / a.HW) * nb * a.HW));
    for (int pair = blockIdx.x * 8 + wave; pair < a.npairs; pair += gridDim.x * 8) {
        const int b = (pair * 64) / a.HW, pp = pair * 64 - b * a.HW;
        bf16x8 fb[F8 ? 1 : 2][F8 ? 1 : 8];
        i32x8 fq[F8 ? 2 : 1][F8 ? 2 : 1];
        if constexpr (F8) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t pix = (uint32_t)pair * 64u + t * 32 + r;
                const uint32_t e0 = pix * (uint32_t)a.ldx + (uint32_t)a.cin_off + 32 * h;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    fq[t][s] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsX, e0 + 64 * s, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsX, e0 + 64 * s + 16, 0, 0));
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t pix = (uint32_t)pair * 64u + t * 32 + r;
                const uint32_t e0 = pix * (uint32_t)a.ldx + (uint32_t)a.cin_off + 8 * h;
                u32x4 raw[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) raw[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (e0 + 16 * kk) * 2u, 0, 0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (tr) {
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(raw[kk][j] << 16); f[2 * j + 1] = __uint_as_float(raw[kk][j] & 0xFFFF0000u); }
                        const int c = a.cin_off + 16 * kk + 8 * h;
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = abc_act(f[j], a.sc[c + j], a.sh[c + j], a.sl[c + j]);
                        fb[t][kk] = pack_frag<bf16>(f);      // (no dropout: an evaluation-time output)
                    } else {
                        fb[t][kk] = *(const bf16x8*)&raw[kk];
                    }
                }
            }
        }
        for (int bb = 0; bb < nbb; ++bb) {
            float best[2][16];
            int arg[2][16];
            const bool row_ok = bb * 32 + r < nb;
            // one m-tile: the 32 bins of block bb of group g, both pixel tiles -> acc (the f32 values the plain form would store)
            auto tile = [&](int g, f32x16* acc) {
                const int co = row_ok ? g * nb + bb * 32 + r : a.Cout_pad - 1;   // this lane's weight row (a zero padding row past the bins)
                const int sw = (co >> 2) & 3;
                float bvv[16], osc[16];
                const float* bp = sbias + g * 64 + bb * 32 + 4 * h;
                const float* op = sosc + g * 64 + bb * 32 + 4 * h;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 b4 = *(const f32x4*)(bp + 8 * k4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bvv[4 * k4 + j] = b4[j];
                    if constexpr (F8) {
                        const f32x4 o4 = *(const f32x4*)(op + 8 * k4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) osc[4 * k4 + j] = o4[j];
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[t][k] = F8 ? 0.f : bvv[k];
                if constexpr (F8) {
                    i32x8 fa[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const char* rowp = wl + (s2 * a.Cout_pad + co) * 64;      // bytes 32 h .. 32 h + 32 of the row: slots 2 h, 2 h + 1
                        fa[s2] = abc_join32B(*(const u32x4*)(rowp + (((2 * h) ^ sw) * 16)), *(const u32x4*)(rowp + (((2 * h + 1) ^ sw) * 16)));
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) mma32B_f8(acc[t], fa[s2], fq[t][s2]);
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[t][k] = fmaf(acc[t][k], osc[k], bvv[k]);
                    }
                } else {
                    bf16x8 fa[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)      // packed byte offset ((kk >> 1) Cout_pad + co) 64 + 32 (kk & 1) + 16 h: slot 2 (kk & 1) + h
                        fa[kk] = *(const bf16x8*)(wl + ((kk >> 1) * a.Cout_pad + co) * 64 + (((2 * (kk & 1) + h) ^ sw) * 16));
                    // (the bias as the accumulators' initial value: acc + bias in the plain form is the same f32 sum order up to the
                    //  position of the bias term -- NOT bit-identical; keep the plain form's order: add it afterwards)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[t][kk], acc[t], 0, 0, 0);
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[t][k] += bvv[k];
                    }
                }
            };
            {
                f32x16 acc[2];
                tile(0, acc);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 16; ++k) { best[t][k] = acc[t][k]; arg[t][k] = 0; }
            }
#pragma unroll 1
            for (int g = 1; g < 6; ++g) {
                f32x16 acc[2];
                tile(g, acc);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const bool gt = acc[t][k] > best[t][k];
                        best[t][k] = gt ? acc[t][k] : best[t][k];
                        arg[t][k] = gt ? g : arg[t][k];
                    }
            }
            // Stores.  Registers k = 4 q .. 4 q + 3 of a lane are four consecutive bins of ONE pixel; a dword of the map is four
            // consecutive PIXELS (the lanes of a quad) of one bin: a 4 x 4 byte transpose inside the quad (two butterfly steps of one
            // DPP move + one v_perm each), after which lane j of the quad holds bin 4 q' + j for the quad's four pixels and every lane
            // stores one dword per register group.  (Byte stores are not merged on their way to the L2; a broadcast per byte cost
            // 7 VALU instructions per value.)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    unsigned w = (unsigned)arg[t][4 * q4] | ((unsigned)arg[t][4 * q4 + 1] << 8) | ((unsigned)arg[t][4 * q4 + 2] << 16) | ((unsigned)arg[t][4 * q4 + 3] << 24);
#if defined(__HIP_DEVICE_COMPILE__)
                    {   // step 1: exchange with lane ^ 1 -- even lanes keep bytes 0, 2 and take the partner's 0, 2 into 1, 3; odd lanes the mirror
                        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
                        w = (lane & 1) ? __builtin_amdgcn_perm(w, o, 0x07030501u) : __builtin_amdgcn_perm(o, w, 0x06020400u);
                    }
                    {   // step 2: exchange with lane ^ 2 on 16-bit halves
                        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
                        w = (lane & 2) ? __builtin_amdgcn_perm(w, o, 0x07060302u) : __builtin_amdgcn_perm(o, w, 0x05040100u);
                    }
#endif
                    // this lane now holds bin 32 bb + 8 q4 + 4 h + (lane & 3) for pixels pp + 32 t + (r & ~3) .. + 3
                    const int bin = bb * 32 + 8 * q4 + 4 * h + (lane & 3);
                    const unsigned voff = bin < nb ? (unsigned)(bin * a.HW + pp + t * 32 + (r & ~3)) : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b32(w, rsO, voff, (unsigned)(b * nb) * (unsigned)a.HW, 0);
                }
        }
    }
}

// (dynamic LDS of the group kernel: the head's packed weights)
static int group_launch(const HeadFwdK& k, hipStream_t st) {
    static unsigned long long ok8 = 0, ok16 = 0;
    const int lds = (int)k.bytesW + 2 * 6 * 64 * 4;
    if (lds > 150 * 1024) return abc_fail(ABC_EUNSUPPORTED, "head_fwd (arg-max form): the head's packed weights exceed the LDS");
    const int want = abc_cdiv(k.npairs, 8);
    const int grid = want < 256 ? want : 256;
    if (k.f8) {
        if (int rc = abc_allow_lds((const void*)head_fwd_group_kernel<true>, 160 * 1024, &ok8)) return rc;
        hipLaunchKernelGGL(head_fwd_group_kernel<true>, dim3(grid), dim3(512), lds, st, k);
    } else {
        if (int rc = abc_allow_lds((const void*)head_fwd_group_kernel<false>, 160 * 1024, &ok16)) return rc;
        hipLaunchKernelGGL(head_fwd_group_kernel<false>, dim3(grid), dim3(512), lds, st, k);
    }
    return 0;
}

// All heads of the network in ONE launch (blockIdx.y = head): eight back-to-back launches each drained the chip before the
// next began, and the five small heads (1 .. 14 channels) are far too short to fill it on their own.
constexpr int MAX_HEADS = 8;
struct HeadFwdBatch { HeadFwdK k[MAX_HEADS]; };
__global__ __launch_bounds__(256) void head_fwd_batch_kernel(const HeadFwdBatch bt) { head_fwd_body(bt.k[blockIdx.y]); }

// ---------------------------------------------------------------------------
// Data gradient of the heads' 1x1 convolutions: dH[p][ci] = sum_co act(dL[co][p]) * W[co][ci] with dL the NCHW f32
// logit gradients (act = the per-channel loss scale) and dH a 128-channel slice of an NHWC tensor.
// GEMM view (transposed so that dL's memory layout is usable): D[ci][p] = W^T[ci][co] x dL[co][p]; A = W^T from the
// packed data-gradient weights, B = dL: rows of pixels per output channel = k-major, which is what
// ds_read_b64_tr_b16 turns into per-lane k-fragments -> each wave stages 16 co x 64 pixels (f32 -> bf16, 128-byte
// line reads) in a private LDS tile, no workgroup barriers.  HBM-bound on dL (read once) and the dH slice written.
struct HeadDgK {
    const float* dl;
    const float *sc, *sh, *sl;
    const void* w;       // packed [1][nchunks][128][CK] bf16 (rows = ci, reduction = co)
    bf16* y;             // NHWC
    int HW, hc, CK, ksteps, ldy, cout_off, npairs;
    unsigned bytesDL, bytesW;
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_h;
__device__ inline bf16x8 tr_read8h(const char* b0, const char* b1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)b0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)b1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ inline void head_dgrad_body(const HeadDgK& a) {
    constexpr int ROWB = 64 * 2 + 16;          // LDS row: 64 pixels bf16 + pad
    __shared__ __attribute__((aligned(16))) char smem[4][2][16 * ROWB];
    __shared__ __attribute__((aligned(16))) char stile[4][32 * (128 * 2 + 16)];  // per-wave output transpose tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.npairs) return;
    const __amdgpu_buffer_rsrc_t rsD = abc_make_rsrc(a.dl, a.bytesDL), rsW = abc_make_rsrc(a.w, a.bytesW);
    const int b = (pair * 64) / a.HW, pp = pair * 64 - b * a.HW;
    // staging role of this lane: row co = lane >> 2 of the 16-row step, pixels 16 * (lane & 3) .. + 16
    const int srow = lane >> 2, spx = (lane & 3) * 16;
    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][m][k] = 0.f;

    u32x4 raw[4];
    auto issue = [&](int ks) {
        const int co = ks * 16 + srow;
        const unsigned off = co < a.hc ? (unsigned)((((unsigned)(b * a.hc + co)) * (unsigned)a.HW + (unsigned)(pp + spx)) * 4u) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsD, off + 16u * i, 0, 0);
    };
    auto commit = [&](int ks, char* buf) {
        const int co = ks * 16 + srow;
        const bool ok = co < a.hc;
        const float sc = (ok && a.sc) ? a.sc[co] : 1.f, sh = (ok && a.sh) ? a.sh[co] : 0.f, sl = (ok && a.sl) ? a.sl[co] : 1.f;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = __uint_as_float(raw[i][j]);
                v[4 * i + j] = (ok && a.sc) ? abc_act(x, sc, sh, sl) : x;
            }
        *(bf16x8*)(buf + srow * ROWB + spx * 2) = pack_frag<bf16>(v);
        *(bf16x8*)(buf + srow * ROWB + spx * 2 + 16) = pack_frag<bf16>(v + 8);
    };
    issue(0);
    commit(0, smem[wave][0]);
    // transposing read: lane supplies row 8 h + ((lane & 15) >> 2) (+4), columns 16 ((lane >> 4) & 1) + 4 (lane & 3) (+32 for tile 1)
    const int trow = 8 * h + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    for (int ks = 0; ks < a.ksteps; ++ks) {
        const bool has_next = ks + 1 < a.ksteps;
        if (has_next) issue(ks + 1);
        const char* buf = smem[wave][ks & 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's own LDS writes of the tile are complete
        // A fragments: rows ci = 32 m + r, reduction co = 16 ks + 8 h .. + 8
        bf16x8 fa[4];
        const int chunk = (16 * ks) / a.CK, within = (16 * ks) % a.CK + 8 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsW, (unsigned)(((chunk * 128 + 32 * m + r) * a.CK + within) * 2), 0, 0);
            fa[m] = *(const bf16x8*)&t;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char* q0 = buf + trow * ROWB + (tcol + 32 * t) * 2;
            const bf16x8 fb = tr_read8h(q0, q0 + 4 * ROWB);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb, acc[t][m], 0, 0, 0);
        }
        if (has_next) commit(ks + 1, smem[wave][(ks + 1) & 1]);
    }
    // D[ci][p]: column = pixel r of tile t, registers 4 q .. 4 q + 3 = ci 32 m + 8 q + 4 h .. + 4.  Transposed through
    // the wave's LDS tile ([32 pixels][128 ci], reusing the staging buffers' space) into 16-byte stores: a pixel's 256
    // bytes leave as one contiguous run instead of 32 scattered 8-byte pieces.
    constexpr int OROW = 128 * 2 + 16;
    char* ot = stile[wave];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)acc[t][m][4 * q + j];
                *(bf16x4*)(ot + r * OROW + (32 * m + 8 * q + 4 * h) * 2) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = it * 4 + (lane >> 4), sg = lane & 15;
            const f32x4 v = *(const f32x4*)(ot + px * OROW + sg * 16);
            *(f32x4*)(a.y + ((size_t)(pair * 64 + 32 * t + px)) * a.ldy + a.cout_off + sg * 8) = v;
        }
    }
}

__global__ __launch_bounds__(256) void head_dgrad_kernel(const HeadDgK a) { head_dgrad_body(a); }
struct HeadDgBatch { HeadDgK k[MAX_HEADS]; };
__global__ __launch_bounds__(256) void head_dgrad_batch_kernel(const HeadDgBatch bt) { head_dgrad_body(bt.k[blockIdx.y]); }

static void fill_fwd(HeadFwdK& k, const abc_conv_desc* d) {
    k.x = d->src.x; k.sc = d->src.scale; k.sh = d->src.shift; k.sl = d->src.slope; k.w = d->w; k.bias = d->bias; k.y = (float*)d->y;
    k.HW = d->Hg * d->Wg; k.ldx = d->src.ldx; k.cin_off = d->cin_off; k.Cout = d->Cout; k.Cout_pad = d->Cout_pad;
    k.ctot = d->ctot_out; k.cout_off = d->cout_off; k.npairs = d->B * k.HW / 64;
    k.drop_p = d->src.drop_p; k.drop_seed = d->src.drop_seed; k.drop_salt = d->src.drop_salt;
    k.f8 = d->dtype_c == ABC_FP8 ? 1 : 0; k.oscale = d->out_scale;
    k.aux = d->head_aux; k.aux_mode = d->head_aux != nullptr ? d->head_aux_mode : 0;
    k.bytesX = (unsigned)((int64_t)d->B * k.HW * d->src.ldx * (k.f8 ? 1 : 2));
    k.bytesW = (unsigned)((int64_t)128 * d->Cout_pad * (k.f8 ? 1 : 2));
}

static void fill_dg(HeadDgK& k, const abc_conv_desc* d) {
    k.dl = (const float*)d->src.x; k.sc = d->src.scale; k.sh = d->src.shift; k.sl = d->src.slope; k.w = d->w; k.y = (bf16*)d->y;
    k.HW = d->Hg * d->Wg; k.hc = d->Cin; k.CK = abc_conv_chunk(d->dtype_c, d->Cin);
    k.ksteps = abc_roundup(d->Cin, k.CK) / 16; k.ldy = d->ldy; k.cout_off = d->cout_off; k.npairs = d->B * k.HW / 64;
    k.bytesDL = (unsigned)((int64_t)d->B * d->Cin * k.HW * 4);
    k.bytesW = (unsigned)((int64_t)abc_roundup(d->Cin, k.CK) * 128 * 2);
}

}  // namespace

int abc_head_fwd_ok(const abc_conv_desc* d) {
    if (abc_knob("ABC_CONV_NOHEAD")) return 0;
    if (!d->planar_out || d->ntaps != 1 || d->tap_dy[0] != 0 || d->tap_dx[0] != 0 || d->stride != 1 || d->om != 1 || d->oy0 || d->ox0) return 0;
    const bool f8 = d->dtype_in == ABC_FP8 && d->dtype_c == ABC_FP8;
    // (e4m3: finished features only -- no transform, no dropout -- with the dequantisation factors in out_scale)
    if (f8 && (d->out_scale == nullptr || d->src.scale != nullptr || d->src.drop_p > 0.f || (d->src.ldx % 16) || (d->cin_off % 16))) return 0;
    if (d->Cin != 128 || !(f8 || (d->dtype_in == ABC_BF16 && d->dtype_c == ABC_BF16)) || d->dtype_out != ABC_F32) return 0;
    if (d->src.pool || d->src.planar || d->accumulate || d->stats != nullptr || d->out_act) return 0;
    if (d->Hg != d->Hin || d->Wg != d->Win || d->Hout != d->Hg || d->Wout != d->Wg || (d->Hg * d->Wg) % 64) return 0;
    if ((d->src.ldx % 8) || (d->cin_off % 8) || d->Cout_pad % 32) return 0;
    if (d->head_aux != nullptr && !(d->head_aux_mode == 1 || (d->head_aux_mode == 2 && d->Cout == 60 && d->Cout_pad == 64 && d->src.drop_p <= 0.f) ||
                                    (d->head_aux_mode == 3 && d->Cout % 6 == 0 && d->Cout / 6 <= 64 && d->y == nullptr && d->src.drop_p <= 0.f))) return 0;
    if (d->y == nullptr && d->head_aux == nullptr) return 0;
    // (all offsets in the kernel are unsigned 32-bit bytes: a batch-64 512x512 feature buffer of 8 x 128 channels is 2^31)
    const int64_t bx = (int64_t)d->B * d->Hg * d->Wg * d->src.ldx * (f8 ? 1 : 2);
    return bx < (int64_t(1) << 32) - 4096;
}

int abc_head_fwd_launch(const abc_conv_desc* d, abc_stream_t stream) {
    HeadFwdK k;
    fill_fwd(k, d);
    if (k.aux_mode == 2 && k.f8) hipLaunchKernelGGL(head_fwd_omega_kernel<true>, dim3(abc_cdiv(k.npairs, 4)), dim3(256), 0, (hipStream_t)stream, k);
    else if (k.aux_mode == 2) hipLaunchKernelGGL(head_fwd_omega_kernel<false>, dim3(abc_cdiv(k.npairs, 4)), dim3(256), 0, (hipStream_t)stream, k);
    else if (k.aux_mode == 3) { if (int rc = group_launch(k, (hipStream_t)stream)) return rc; }
    else hipLaunchKernelGGL(head_fwd_kernel, dim3(abc_cdiv(k.npairs, 4)), dim3(256), 0, (hipStream_t)stream, k);
    return abc_check_launch("head_fwd");
}

int abc_head_dgrad_ok(const abc_conv_desc* d) {
    if (abc_knob("ABC_CONV_NOHEAD")) return 0;
    if (!d->src.planar || d->dtype_in != ABC_F32 || d->dtype_c != ABC_BF16 || d->dtype_out != ABC_BF16 || d->planar_out) return 0;
    if (d->ntaps != 1 || d->tap_dy[0] != 0 || d->tap_dx[0] != 0 || d->stride != 1 || d->om != 1 || d->oy0 || d->ox0) return 0;
    if (d->Cout != 128 || d->Cout_pad != 128 || d->cin_off != 0 || d->src.ctot != d->Cin || d->bias != nullptr || d->stats != nullptr || d->accumulate || d->out_act) return 0;
    if (d->Hg != d->Hin || d->Wg != d->Win || d->Hout != d->Hg || d->Wout != d->Wg || (d->Hg * d->Wg) % 64) return 0;
    if ((d->ldy % 8) || (d->cout_off % 8)) return 0;
    return (int64_t)d->B * d->Cin * d->Hg * d->Wg * 4 < (int64_t(1) << 31);
}

int abc_head_dgrad_launch(const abc_conv_desc* d, abc_stream_t stream) {
    HeadDgK k;
    fill_dg(k, d);
    hipLaunchKernelGGL(head_dgrad_kernel, dim3(abc_cdiv(k.npairs, 4)), dim3(256), 0, (hipStream_t)stream, k);
    return abc_check_launch("head_dgrad");
}

// All heads' 1x1 convolutions in one launch.  which = 0: forward (every descriptor must satisfy abc_head_fwd_ok),
// 1: data gradient (abc_head_dgrad_ok); all with the same batch and map size.
extern "C" int abc_heads_batch(const abc_conv_desc* descs, int32_t n, int32_t which, abc_stream_t stream) {
    if (n < 1 || n > MAX_HEADS) return abc_fail(ABC_EINVAL, "heads_batch: 1..8 heads");
    for (int i = 0; i < n; ++i) {
        const abc_conv_desc* d = descs + i;
        if (!(which ? abc_head_dgrad_ok(d) : abc_head_fwd_ok(d))) return abc_fail(ABC_EUNSUPPORTED, "heads_batch: a descriptor does not fit the heads' 1x1 kernels");
        if (d->B != descs->B || d->Hg != descs->Hg || d->Wg != descs->Wg) return abc_fail(ABC_EINVAL, "heads_batch: all heads must share batch and map size");
    }
    const int npairs = descs->B * descs->Hg * descs->Wg / 64;
    if (which == 0) {
        // (a head with the channel-axis NMS mask runs on its own kernel, beside the batch of the others)
        HeadFwdBatch bt;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            HeadFwdK k;
            fill_fwd(k, descs + i);
            if (k.aux_mode == 2 && k.f8) hipLaunchKernelGGL(head_fwd_omega_kernel<true>, dim3(abc_cdiv(npairs, 4)), dim3(256), 0, (hipStream_t)stream, k);
            else if (k.aux_mode == 2) hipLaunchKernelGGL(head_fwd_omega_kernel<false>, dim3(abc_cdiv(npairs, 4)), dim3(256), 0, (hipStream_t)stream, k);
            else if (k.aux_mode == 3) { if (int rc = group_launch(k, (hipStream_t)stream)) return rc; }
            else bt.k[m++] = k;
        }
        if (m > 0) {
            for (int i = m; i < MAX_HEADS; ++i) bt.k[i] = bt.k[0];
            hipLaunchKernelGGL(head_fwd_batch_kernel, dim3(abc_cdiv(npairs, 4), m), dim3(256), 0, (hipStream_t)stream, bt);
        }
    } else {
        HeadDgBatch bt;
        for (int i = 0; i < n; ++i) fill_dg(bt.k[i], descs + i);
        for (int i = n; i < MAX_HEADS; ++i) bt.k[i] = bt.k[0];
        hipLaunchKernelGGL(head_dgrad_batch_kernel, dim3(abc_cdiv(npairs, 4), n), dim3(256), 0, (hipStream_t)stream, bt);
    }
    return abc_check_launch("heads_batch");
}
