// Weight gradient of a tap-list convolution on the gfx950 matrix cores.
//
//   dW[t][a][b] = sum over pixels p of  P[p][a] * Q[stride*p + d_t][b]
//
// GEMM view per tap: M = a-channels, N = b-channels, K = pixels (split-K over 8x16
// spatial patches across workgroups; every workgroup writes its own f32 slab and
// abc_wgrad_reduce sums the slabs in a fixed order -> bitwise reproducible, no atomics).
// Both operands have the reduction index (pixel) as the SLOW memory axis (NHWC), so
// the [pixel][channel] LDS images are read column-wise: in bf16 mode with the
// hardware-transposing ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group),
// in exact-f32 mode with plain ds_read_b32 (v_mfma_f32_32x32x2_f32 wants one value per lane).
// One K-step = one 16-pixel row segment of the patch; the un-shifted operand's
// fragment is read once per K-step and reused by all taps.
//
// Reference ops covered: autograd of nn.Conv2d / nn.ConvTranspose2d weights
// (unet.py:12,15,44,66,70 under loss.backward(), train.py:140).
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

constexpr int MAXT = 9;  // taps per workgroup (accumulator budget: 9 x 16 VGPRs)

struct WgK {
    ActSrc p, q;
    float* partial;
    int B, Hg, Wg, Hq, Wq;
    int cp_off, Ca, cq_off, Cb, Ca_pad, Cb_pad;
    int ntaps, tgw, nsplit, npatch, tiles_x, tiles_y;
    int dy_min, dx_min, HH, HW, PSW, sQ_off, nta, ntb;
    int8_t ty[ABC_MAX_TAPS], tx[ABC_MAX_TAPS];
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ inline bf16x8 tr_read8(const char* base0, const char* base1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)base0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)base1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename PT, typename QT, typename CT, int CW, int STRIDE>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgK a) {
    constexpr bool WIDE = (CW == 64);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;
    char* sQ = smem + a.sQ_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int id = blockIdx.x;
    const int bt = id % a.ntb; id /= a.ntb;
    const int at = id % a.nta; id /= a.nta;
    const int split = id;
    const int t0 = blockIdx.y * a.tgw;
    const int tcnt = min(a.tgw, a.ntaps - t0);

    const int ai = WIDE ? (wave >> 1) : 0, bi = WIDE ? (wave & 1) : 0;
    const int row_lo = WIDE ? 0 : 2 * wave, row_hi = WIDE ? 8 : 2 * wave + 2;
    const int PSW = a.PSW;

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    int tapoff[MAXT];  // LDS byte offset of each tap inside the Q halo
#pragma unroll
    for (int t = 0; t < MAXT; ++t) tapoff[t] = (t < tcnt) ? (a.ty[t0 + t] * a.HW + a.tx[t0 + t]) * PSW : 0;

    // per-lane channel byte offsets inside a pixel
    int pch, qch;
    if constexpr (sizeof(CT) == 2) {
        const int sub = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        pch = (ai * 32 + sub) * 2;
        qch = (bi * 32 + sub) * 2;
    } else {
        pch = (ai * 32 + r) * 4;
        qch = (bi * 32 + r) * 4;
    }

    for (int patch = split; patch < a.npatch; patch += a.nsplit) {
        int pid = patch;
        const int tx_i = pid % a.tiles_x; pid /= a.tiles_x;
        const int ty_i = pid % a.tiles_y; pid /= a.tiles_y;
        const int b = pid;
        const int gy0 = ty_i * 8, gx0 = tx_i * 16;
        __syncthreads();
        stage_halo<PT, CT, CW>(sP, 16 * PSW, PSW, 8, 16, b, gy0, gx0, a.Hg, a.Wg, a.p, a.cp_off + at * CW, tid, 256,
                               a.Ca - at * CW);
        stage_halo<QT, CT, CW>(sQ, a.HW * PSW, PSW, a.HH, a.HW, b, gy0 * STRIDE + a.dy_min, gx0 * STRIDE + a.dx_min,
                               a.Hq, a.Wq, a.q, a.cq_off + bt * CW, tid, 256, a.Cb - bt * CW);
        __syncthreads();
        for (int row = row_lo; row < row_hi; ++row) {
            if constexpr (sizeof(CT) == 2) {
                // lane supplies the address of pixel k = 8h + 4q + ((lane&15)>>2), 4 channels
                const int kq = 8 * h + ((lane & 15) >> 2);
                const char* pa = sP + (row * 16 + kq) * PSW + pch;
                const bf16x8 fa = tr_read8(pa, pa + 4 * PSW);
                const char* qb = sQ + ((row * STRIDE) * a.HW + kq * STRIDE) * PSW + qch;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (t < tcnt) {
                        const bf16x8 fb = tr_read8(qb + tapoff[t], qb + tapoff[t] + 4 * STRIDE * PSW);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll 2
                for (int s = 0; s < 8; ++s) {
                    const int k = 2 * s + h;
                    const float fa = *(const float*)(sP + (row * 16 + k) * PSW + pch);
                    const char* qb = sQ + ((row * STRIDE) * a.HW + k * STRIDE) * PSW + qch;
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) {
                        if (t < tcnt) {
                            const float fb = *(const float*)(qb + tapoff[t]);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    if constexpr (!WIDE) {
        // the 4 waves hold partial sums over different patch rows: fold into wave 0
        float* red = (float*)smem;  // [3][16][64]
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < tcnt) {
                __syncthreads();
                if (wave > 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) red[((wave - 1) * 16 + k) * 64 + lane] = acc[t][k];
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int w = 0; w < 3; ++w)
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[t][k] += red[(w * 16 + k) * 64 + lane];
                }
            }
        }
        if (wave != 0) return;
    }

#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < tcnt) {
            float* out = a.partial + ((size_t)(split * a.ntaps + t0 + t) * a.Ca_pad + at * CW + ai * 32) * a.Cb_pad
                         + bt * CW + bi * 32 + r;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int arow = (k & 3) + 8 * (k >> 2) + 4 * h;
                out[(size_t)arow * a.Cb_pad] = acc[t][k];
            }
        }
    }
}

// Sum the split-K slabs.  One thread per (tap, a, b) output element; consecutive threads walk
// b, so every slab read is a coalesced 4-byte stream and the whole reduction is one pass at
// HBM/L2 speed (slab order fixed -> bitwise reproducible).  Output = reference layout [a][b][tap].
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const abc_wgrad_reduce_desc d) {
    const int64_t n = (int64_t)d.ntaps * d.Ca * d.Cb;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int bi = (int)(idx % d.Cb);
    const int ai = (int)((idx / d.Cb) % d.Ca);
    const int t = (int)(idx / ((int64_t)d.Cb * d.Ca));
    const size_t slab = (size_t)d.Ca_pad * d.Cb_pad;
    const float* p = d.partial + (size_t)t * slab + (size_t)ai * d.Cb_pad + bi;
    const size_t step = (size_t)d.ntaps * slab;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= d.nsplit; k += 4) {
        s0 += p[(size_t)k * step]; s1 += p[(size_t)(k + 1) * step]; s2 += p[(size_t)(k + 2) * step]; s3 += p[(size_t)(k + 3) * step];
    }
    for (; k < d.nsplit; ++k) s0 += p[(size_t)k * step];
    const float s = (s0 + s1) + (s2 + s3);
    float* o = d.dw + ((size_t)ai * d.Cb + bi) * d.ntaps + t;
    *o = d.accumulate ? (*o + s) : s;
}

struct WGeom { int CW, dy_min, dx_min, HH, HW, PSW, sP_bytes, lds, tgw, ngroups, nta, ntb, npatch, tiles_x, tiles_y; };

static int wgeom(const abc_wgrad_desc* d, WGeom* g) {
    if (d->ntaps < 1 || d->ntaps > ABC_MAX_TAPS) return abc_fail(ABC_EINVAL, "wgrad: ntaps");
    if (d->stride != 1 && d->stride != 2) return abc_fail(ABC_EUNSUPPORTED, "wgrad: stride");
    const int csz = d->dtype_c == ABC_BF16 ? 2 : 4;
    const int ca32 = abc_roundup(d->Ca, 32), cb32 = abc_roundup(d->Cb, 32);
    g->CW = (d->stride == 1 && ca32 % 64 == 0 && cb32 % 64 == 0) ? 64 : 32;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    g->dy_min = dymin; g->dx_min = dxmin;
    g->HH = 7 * d->stride + (dymax - dymin) + 1;
    g->HW = 15 * d->stride + (dxmax - dxmin) + 1;
    // bf16: 4 consecutive pixels' 64-byte column blocks must fall on distinct bank quarters
    if (csz == 2) g->PSW = (g->CW == 32) ? 64 : 192;
    else g->PSW = g->CW * 4;
    g->sP_bytes = abc_roundup(128 * g->PSW, 256);
    g->lds = g->sP_bytes + abc_roundup(g->HH * g->HW * g->PSW, 256);
    if (g->lds < 3 * 16 * 64 * 4) g->lds = 3 * 16 * 64 * 4;
    if (g->lds > 160 * 1024) return abc_fail(ABC_EUNSUPPORTED, "wgrad: LDS tile too large");
    g->ngroups = abc_cdiv(d->ntaps, MAXT);
    g->tgw = abc_cdiv(d->ntaps, g->ngroups);
    g->nta = abc_cdiv(d->Ca, g->CW); g->ntb = abc_cdiv(d->Cb, g->CW);
    g->tiles_x = abc_cdiv(d->Wg, 16); g->tiles_y = abc_cdiv(d->Hg, 8);
    g->npatch = g->tiles_x * g->tiles_y * d->B;
    return ABC_OK;
}

template <typename PT, typename QT, typename CT, int CW, int STRIDE>
static int wlaunch(const WgK& k, const WGeom& g, int nsplit, hipStream_t st) {
    auto fn = wgrad_kernel<PT, QT, CT, CW, STRIDE>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3(g.nta * g.ntb * nsplit, g.ngroups), dim3(256), g.lds, st, k);
    return abc_check_launch("wgrad");
}

template <typename PT, typename QT, typename CT>
static int wdispatch(const WgK& k, const WGeom& g, int stride, int nsplit, hipStream_t st) {
    if (g.CW == 64) return wlaunch<PT, QT, CT, 64, 1>(k, g, nsplit, st);
    if (stride == 1) return wlaunch<PT, QT, CT, 32, 1>(k, g, nsplit, st);
    return wlaunch<PT, QT, CT, 32, 2>(k, g, nsplit, st);
}

}  // namespace

extern "C" int abc_wgrad_pads(const abc_wgrad_desc* d, int32_t* ca_pad, int32_t* cb_pad) {
    WGeom g;
    int rc = wgeom(d, &g);
    if (rc) return rc;
    *ca_pad = g.nta * g.CW;
    *cb_pad = g.ntb * g.CW;
    return ABC_OK;
}

extern "C" int abc_wgrad(const abc_wgrad_desc* d, abc_stream_t stream) {
    WGeom g;
    int rc = wgeom(d, &g);
    if (rc) return rc;
    if (d->nsplit < 1) return abc_fail(ABC_EINVAL, "wgrad: nsplit");
    WgK k;
    auto cp = [](ActSrc& o, const abc_act_src& i) {
        o.x = i.x; o.scale = i.scale; o.shift = i.shift; o.slope = i.slope; o.Hx = i.Hx; o.Wx = i.Wx; o.ldx = i.ldx;
        o.pool = i.pool; o.drop_p = i.drop_p; o.drop_seed = i.drop_seed; o.planar = i.planar; o.ctot = i.ctot;
    };
    cp(k.p, d->p); cp(k.q, d->q);
    const int php = d->p.pool ? d->p.Hx / 2 : d->p.Hx, pwp = d->p.pool ? d->p.Wx / 2 : d->p.Wx;
    const int qhp = d->q.pool ? d->q.Hx / 2 : d->q.Hx, qwp = d->q.pool ? d->q.Wx / 2 : d->q.Wx;
    if (php != d->Hg || pwp != d->Wg || qhp != d->Hq || qwp != d->Wq) return abc_fail(ABC_EINVAL, "wgrad: dims mismatch");
    k.partial = d->partial; k.B = d->B; k.Hg = d->Hg; k.Wg = d->Wg; k.Hq = d->Hq; k.Wq = d->Wq;
    k.cp_off = d->cp_off; k.Ca = d->Ca; k.cq_off = d->cq_off; k.Cb = d->Cb;
    k.Ca_pad = g.nta * g.CW; k.Cb_pad = g.ntb * g.CW;
    k.ntaps = d->ntaps; k.tgw = g.tgw; k.nsplit = d->nsplit; k.npatch = g.npatch; k.tiles_x = g.tiles_x; k.tiles_y = g.tiles_y;
    k.dy_min = g.dy_min; k.dx_min = g.dx_min; k.HH = g.HH; k.HW = g.HW; k.PSW = g.PSW; k.sQ_off = g.sP_bytes;
    k.nta = g.nta; k.ntb = g.ntb;
    for (int t = 0; t < d->ntaps; ++t) { k.ty[t] = (int8_t)(d->tap_dy[t] - g.dy_min); k.tx[t] = (int8_t)(d->tap_dx[t] - g.dx_min); }
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype_c == ABC_F32) {
        if (d->dtype_p != ABC_F32 || d->dtype_q != ABC_F32) return abc_fail(ABC_EUNSUPPORTED, "wgrad: f32 compute needs f32 operands");
        return wdispatch<float, float, float>(k, g, d->stride, d->nsplit, st);
    }
    if (d->dtype_p == ABC_BF16 && d->dtype_q == ABC_BF16) return wdispatch<bf16, bf16, bf16>(k, g, d->stride, d->nsplit, st);
    if (d->dtype_p == ABC_F32 && d->dtype_q == ABC_BF16) return wdispatch<float, bf16, bf16>(k, g, d->stride, d->nsplit, st);
    if (d->dtype_p == ABC_BF16 && d->dtype_q == ABC_F32) return wdispatch<bf16, float, bf16>(k, g, d->stride, d->nsplit, st);
    return abc_fail(ABC_EUNSUPPORTED, "wgrad: dtype combination");
}

extern "C" int abc_wgrad_reduce(const abc_wgrad_reduce_desc* d, abc_stream_t stream) {
    const int64_t n = (int64_t)d->ntaps * d->Ca * d->Cb;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("wgrad_reduce");
}
