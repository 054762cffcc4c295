// Generic tap-list convolution as implicit GEMM on the gfx950 matrix cores.
//
// GEMM view: M = output pixels (8x16 spatial patch of one image per workgroup),
// N = output channels (BN per workgroup), K = taps x input channels, walked as
// [Cin chunk of 64 B][tap].  Per chunk the activated input halo tile is staged ONCE
// in LDS (previous layer's BN + activation (+pool, +dropout) applied on the way in)
// and reused by every tap; the weight slices of TG taps are staged beside it.
// A wave computes TM x TN tiles of 32x32 with v_mfma_f32_32x32x16_bf16 (bf16 mode) or
// v_mfma_f32_32x32x2_f32 (exact-f32 parity mode): lane-half h of the wave owns bytes
// [32h, 32h+32) of each pixel's/row's 64-byte chunk for BOTH operands, so one LDS image
// serves both dtypes and fragment reads are plain ds_read_b128.
//
// LDS images: pixel stride PS = chunk bytes + 16, A row stride a multiple of 256 B:
// every ds_read_b128 lane group then covers 16 distinct 16-byte slots (conflict-free).
//
// Reference ops covered: see include/abcnet_hip.h (abc_conv_desc).
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

struct ConvK {
    ActSrc src;
    const void* w;
    const float* bias;
    void* y;
    float* stats;
    int B, Hin, Win, cin_off, Cin, nchunks;
    int Hg, Wg, Hout, Wout, ldy, cout_off, Cout, Cout_pad;
    int om, oy0, ox0;
    int ntaps, tg, dy_min, dx_min, HH, HW, RS;
    int tiles_x, tiles_y, nblocks_n, sB_off, tap_off, planar_out, ctot_out;
    int8_t ty[ABC_MAX_TAPS], tx[ABC_MAX_TAPS];
};

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvK a) {
    constexpr int CKB = CK * (int)sizeof(CT);
    constexpr int PS = CKB + 16;
    constexpr int LHB = CKB / 2;
    constexpr int NR = LHB / 16;
    constexpr int NV = Frag<CT>::NV;
    constexpr int SEGS = CKB / 16;
    constexpr int NT = BN / 32;
    constexpr int WN = (NT >= 2) ? 2 : 1;
    constexpr int WM = 4 / WN;
    constexpr int TM = 4 / WM;
    constexpr int TN = NT / WN;
    typedef typename Frag<CT>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + a.sB_off;
    int* sTap = (int*)(smem + a.tap_off);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- block -> (n-block, patch, image); n-block fastest so neighbours share the halo in L2
    const int nwg = gridDim.x;
    int id = abc_xcd_remap(blockIdx.x, nwg);
    const int nb = id % a.nblocks_n; id /= a.nblocks_n;
    const int mblock = id;
    const int tx_i = id % a.tiles_x; id /= a.tiles_x;
    const int ty_i = id % a.tiles_y; id /= a.tiles_y;
    const int b = id;
    const int gy0 = ty_i * 8, gx0 = tx_i * 16;
    const int n0 = nb * BN;

    if (tid < a.ntaps) sTap[tid] = a.ty[tid] * a.RS + a.tx[tid] * PS;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    int aBase[TM], bBase[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int prow = 2 * (wm * TM + i) + (r >> 4), pcol = r & 15;
        aBase[i] = prow * STRIDE * a.RS + pcol * STRIDE * PS + h * LHB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) bBase[j] = ((wn * TN + j) * 32 + r) * PS + h * LHB;

    const CT* wp = (const CT*)a.w;
    const int iy0 = gy0 * STRIDE + a.dy_min, ix0 = gx0 * STRIDE + a.dx_min;

    for (int c = 0; c < a.nchunks; ++c) {
        __syncthreads();  // everyone done with the previous chunk's A and B
        stage_halo<InT, CT, CK>(sA, a.RS, PS, a.HH, a.HW, b, iy0, ix0, a.Hin, a.Win, a.src, a.cin_off + c * CK, tid, 256,
                                a.Cin - c * CK);
        for (int t0 = 0; t0 < a.ntaps; t0 += a.tg) {
            if (t0 > 0) __syncthreads();  // B of the previous tap group consumed
            const int tcnt = min(a.tg, a.ntaps - t0);
            // ---- stage B: tcnt x BN rows of CKB bytes (contiguous per tap in the packed layout)
            for (int s = tid; s < tcnt * BN * SEGS; s += 256) {
                const int tl = s / (BN * SEGS);
                const int rem = s - tl * (BN * SEGS);
                const int row = rem / SEGS, part = rem - row * SEGS;
                const CT* g = wp + ((size_t)((t0 + tl) * a.nchunks + c) * a.Cout_pad + n0 + row) * CK + part * NV;
                *(frag_t*)(sB + (tl * BN + row) * PS + part * 16) = *(const frag_t*)g;
            }
            __syncthreads();
            // ---- MFMA over the staged taps
            for (int tl = 0; tl < tcnt; ++tl) {
                const int aoff = sTap[t0 + tl];
                const int boff = tl * BN * PS;
                frag_t fa[TM][NR], fb[TN][NR];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < NR; ++q) fa[i][q] = *(const frag_t*)(sA + aBase[i] + aoff + q * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < NR; ++q) fb[j][q] = *(const frag_t*)(sB + bBase[j] + boff + q * 16);
#pragma unroll
                for (int q = 0; q < NR; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) mma16B(acc[i][j], fa[i][q], fb[j][q]);
            }
        }
    }

    // ---- epilogue: bias, statistics of the f32 values, store
    OutT* yo = (OutT*)a.y;
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        const int n = n0 + (wn * TN + j) * 32 + r;
        const bool nvalid = n < a.Cout;
        const float bv = (a.bias != nullptr && nvalid) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int rit = (k & 3) + 8 * (k >> 2) + 4 * h;
                const int gy = gy0 + 2 * (wm * TM + i) + (rit >> 4), gx = gx0 + (rit & 15);
                if (nvalid && gy < a.Hg && gx < a.Wg) {
                    const float v = acc[i][j][k] + bv;
                    s1[j] += v; s2[j] += v * v;
                    if constexpr (sizeof(OutT) == 4) {
                        if (a.planar_out) {
                            // NCHW: registers k..k+3 of a lane are 4 consecutive x of one plane -> one 16-byte store
                            const size_t o = ((size_t)(b * a.ctot_out + a.cout_off + n) * a.Hout + gy) * a.Wout + gx;
                            if ((k & 3) == 0 && gx + 3 < a.Wg && (a.Wout & 3) == 0) {
                                f32x4 t;
                                t[0] = v; t[1] = acc[i][j][k + 1] + bv; t[2] = acc[i][j][k + 2] + bv; t[3] = acc[i][j][k + 3] + bv;
                                *(f32x4*)((float*)a.y + o) = t;
                            } else if (!(gx - (k & 3) + 3 < a.Wg && (a.Wout & 3) == 0)) {
                                ((float*)a.y)[o] = v;
                            }
                            continue;
                        }
                    }
                    const size_t o = ((size_t)(b * a.Hout + gy * a.om + a.oy0) * a.Wout + gx * a.om + a.ox0) * a.ldy
                                     + a.cout_off + n;
                    yo[o] = (OutT)v;
                }
            }
        }
    }
    if (a.stats != nullptr) {
        __syncthreads();  // LDS reuse
        float* red = (float*)smem;  // [WM][2][BN]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v1 = s1[j] + __shfl_xor(s1[j], 32);
            float v2 = s2[j] + __shfl_xor(s2[j], 32);
            if (h == 0) {
                const int nl = (wn * TN + j) * 32 + r;
                red[(wm * 2 + 0) * BN + nl] = v1;
                red[(wm * 2 + 1) * BN + nl] = v2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cout) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { v1 += red[(w * 2 + 0) * BN + tid]; v2 += red[(w * 2 + 1) * BN + tid]; }
            a.stats[((size_t)mblock * 2 + 0) * a.Cout + n0 + tid] = v1;
            a.stats[((size_t)mblock * 2 + 1) * a.Cout + n0 + tid] = v2;
        }
    }
}

struct Geom {
    int CK, BN, dy_min, dx_min, HH, HW, PS, RS, tg, sA_bytes, sB_bytes, tap_off, lds, tiles_x, tiles_y, nbn, grid;
};

static int conv_geom(const abc_conv_desc* d, Geom* g) {
    if (d->ntaps < 1 || d->ntaps > ABC_MAX_TAPS) return abc_fail(ABC_EINVAL, "conv: ntaps out of range");
    if (d->stride != 1 && d->stride != 2) return abc_fail(ABC_EUNSUPPORTED, "conv: stride must be 1 or 2");
    const int csz = d->dtype_c == ABC_BF16 ? 2 : 4;
    g->CK = abc_conv_chunk(d->dtype_c, d->Cin);
    if (g->CK <= 0) return abc_fail(ABC_EINVAL, "conv: Cin must be positive");
    if (d->Cout_pad % 32 || d->Cout_pad < d->Cout) return abc_fail(ABC_EINVAL, "conv: Cout_pad must be a multiple of 32 >= Cout");
    g->BN = (d->Cout_pad % 128 == 0) ? 128 : (d->Cout_pad % 64 == 0 ? 64 : 32);
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    g->dy_min = dymin; g->dx_min = dxmin;
    g->HH = 7 * d->stride + (dymax - dymin) + 1;
    g->HW = 15 * d->stride + (dxmax - dxmin) + 1;
    const int CKB = g->CK * csz;
    g->PS = CKB + 16;
    g->RS = abc_roundup(g->HW * g->PS, 256);
    g->sA_bytes = abc_roundup(g->HH * g->RS, 256);
    // taps per weight stage: keep the B image <= ~40 KB so that 2-3 workgroups fit a CU
    int tg = 40960 / (g->BN * g->PS);
    if (tg < 1) tg = 1;
    if (tg > d->ntaps) tg = d->ntaps;
    // balance the groups
    const int ngroups = abc_cdiv(d->ntaps, tg);
    tg = abc_cdiv(d->ntaps, ngroups);
    g->tg = tg;
    g->sB_bytes = abc_roundup(tg * g->BN * g->PS, 256);
    g->tap_off = g->sA_bytes + g->sB_bytes;
    g->lds = g->tap_off + 256;
    if (g->lds < 4 * 2 * 128 * 4 + 256) g->lds = 4 * 2 * 128 * 4 + 256;
    if (g->lds > 160 * 1024) return abc_fail(ABC_EUNSUPPORTED, "conv: LDS tile too large");
    g->tiles_x = abc_cdiv(d->Wg, 16);
    g->tiles_y = abc_cdiv(d->Hg, 8);
    g->nbn = d->Cout_pad / g->BN;
    g->grid = g->nbn * g->tiles_x * g->tiles_y * d->B;
    return ABC_OK;
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE>
static int launch_inst(const ConvK& k, const Geom& g, hipStream_t st) {
    auto fn = conv_igemm_kernel<InT, CT, OutT, CK, BN, STRIDE>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3(g.grid), dim3(256), g.lds, st, k);
    return abc_check_launch("conv_igemm");
}

template <typename InT, typename CT, typename OutT, int CK>
static int launch_bn(const ConvK& k, const Geom& g, int stride, hipStream_t st) {
#define ABC_L(BN_)                                                                        \
    (stride == 1 ? launch_inst<InT, CT, OutT, CK, BN_, 1>(k, g, st) : launch_inst<InT, CT, OutT, CK, BN_, 2>(k, g, st))
    switch (g.BN) {
        case 128: return ABC_L(128);
        case 64: return ABC_L(64);
        default: return ABC_L(32);
    }
#undef ABC_L
}

}  // namespace

extern "C" int abc_conv_chunk(int dtype_c, int Cin) {
    if (Cin <= 0) return -1;
    const int cp = abc_roundup(Cin, 16);
    if (dtype_c == ABC_BF16) return (cp % 32 == 0) ? 32 : 16;
    return 16;
}

extern "C" int abc_conv_stat_blocks(const abc_conv_desc* d) {
    return abc_cdiv(d->Wg, 16) * abc_cdiv(d->Hg, 8) * d->B;
}

extern "C" int abc_conv_fwd(const abc_conv_desc* d, abc_stream_t stream) {
    Geom g;
    int rc = conv_geom(d, &g);
    if (rc) return rc;
    if (d->src.pool && (d->src.Hx / 2 != d->Hin || d->src.Wx / 2 != d->Win))
        return abc_fail(ABC_EINVAL, "conv: pooled dims mismatch");
    if (!d->src.pool && (d->src.Hx != d->Hin || d->src.Wx != d->Win)) return abc_fail(ABC_EINVAL, "conv: dims mismatch");
    if (!d->src.planar && d->Cin % 16 == 0 && ((d->src.ldx * (d->dtype_in == ABC_BF16 ? 2 : 4)) % 16 || (d->cin_off % 8)))
        return abc_fail(ABC_EINVAL, "conv: input stride/offset must keep 16-byte alignment");
    ConvK k;
    k.src.x = d->src.x; k.src.scale = d->src.scale; k.src.shift = d->src.shift; k.src.slope = d->src.slope;
    k.src.Hx = d->src.Hx; k.src.Wx = d->src.Wx; k.src.ldx = d->src.ldx; k.src.pool = d->src.pool;
    k.src.drop_p = d->src.drop_p; k.src.drop_seed = d->src.drop_seed;
    k.src.planar = d->src.planar; k.src.ctot = d->src.ctot;
    k.planar_out = d->planar_out; k.ctot_out = d->ctot_out;
    if (d->src.planar && (d->dtype_in != ABC_F32 || d->src.pool || d->src.drop_p > 0.f))
        return abc_fail(ABC_EUNSUPPORTED, "conv: planar input must be f32 without pool/dropout");
    if (d->planar_out && (d->dtype_out != ABC_F32 || d->om != 1 || d->oy0 || d->ox0))
        return abc_fail(ABC_EUNSUPPORTED, "conv: planar output must be f32, unit output stride");
    k.w = d->w; k.bias = d->bias; k.y = d->y; k.stats = d->stats;
    k.B = d->B; k.Hin = d->Hin; k.Win = d->Win; k.cin_off = d->cin_off; k.Cin = d->Cin; k.nchunks = abc_cdiv(d->Cin, g.CK);
    k.Hg = d->Hg; k.Wg = d->Wg; k.Hout = d->Hout; k.Wout = d->Wout; k.ldy = d->ldy; k.cout_off = d->cout_off;
    k.Cout = d->Cout; k.Cout_pad = d->Cout_pad; k.om = d->om; k.oy0 = d->oy0; k.ox0 = d->ox0;
    k.ntaps = d->ntaps; k.tg = g.tg; k.dy_min = g.dy_min; k.dx_min = g.dx_min; k.HH = g.HH; k.HW = g.HW; k.RS = g.RS;
    k.tiles_x = g.tiles_x; k.tiles_y = g.tiles_y; k.nblocks_n = g.nbn; k.sB_off = g.sA_bytes; k.tap_off = g.tap_off;
    for (int t = 0; t < d->ntaps; ++t) {
        k.ty[t] = (int8_t)(d->tap_dy[t] - g.dy_min);
        k.tx[t] = (int8_t)(d->tap_dx[t] - g.dx_min);
    }
    // the kernel's grid must address only in-range output pixels
    if ((d->Hg - 1) * d->om + d->oy0 >= d->Hout || (d->Wg - 1) * d->om + d->ox0 >= d->Wout)
        return abc_fail(ABC_EINVAL, "conv: output grid exceeds output tensor");
    hipStream_t st = (hipStream_t)stream;
    const int di = d->dtype_in, dc = d->dtype_c, dout = d->dtype_out;
    if (dc == ABC_F32) {
        if (di != ABC_F32 || dout != ABC_F32) return abc_fail(ABC_EUNSUPPORTED, "conv: f32 compute needs f32 in/out");
        return launch_bn<float, float, float, 16>(k, g, d->stride, st);
    }
    if (di == ABC_BF16 && dout == ABC_BF16)
        return g.CK == 32 ? launch_bn<bf16, bf16, bf16, 32>(k, g, d->stride, st) : launch_bn<bf16, bf16, bf16, 16>(k, g, d->stride, st);
    if (di == ABC_BF16 && dout == ABC_F32)
        return g.CK == 32 ? launch_bn<bf16, bf16, float, 32>(k, g, d->stride, st) : launch_bn<bf16, bf16, float, 16>(k, g, d->stride, st);
    if (di == ABC_F32 && dout == ABC_BF16)
        return g.CK == 32 ? launch_bn<float, bf16, bf16, 32>(k, g, d->stride, st) : launch_bn<float, bf16, bf16, 16>(k, g, d->stride, st);
    return abc_fail(ABC_EUNSUPPORTED, "conv: dtype combination");
}
