// First convolution of the network (unet.py:12 with in_channels = 1): a ONE-channel f32 image into <= 64 output
// channels.  K = 9 is no matrix shape: plain FMAs with the tap weights in registers, bound by writing the output once
// (the implicit-GEMM kernels pad K to 16 and spend an MFMA stage on it).  A thread owns 8 output channels of one
// pixel-column slot; the image rows a row of output needs sit in LDS.  BatchNorm statistics (sum, sum of squares of
// the f32 values) are reduced per workgroup like in the GEMM kernels' epilogue.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "conv_fast.hpp"
#include <stdlib.h>

namespace {

struct StemK {
    const float* x;
    const void* w;       // packed [tap][Cout_pad][16] in the compute type (column 0 = the only input channel)
    const float* bias;
    void* y;
    float* stats;
    int B, H, W, Cout, Cout_pad, ldy, cout_off, ntaps, dy_min, dy_max, rows_per_wg, out_act;
    float out_slope;
    int8_t ty[25], tx[25];
};

// NT = tap capacity (9: the 3x3 stem of unet.py; 25: the 5x5 stem of unet2.py:135), CPT = output channels per thread
// (the tap weights live in registers: NT * CPT of them)
template <typename CT, typename OutT, int NT, int CPT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemK a) {
    constexpr int SROWS = 8;                       // output rows per workgroup (STEM_ROWS below)
    constexpr int NR = SROWS + (NT > 9 ? 4 : 2);    // image rows they need
    __shared__ float sx[NR][512 + 8];
    __shared__ float red[4][2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncg = a.Cout / CPT;
    const int cg = tid % ncg, slot = tid / ncg;
    const int nslot = 256 / ncg;
    const int nrows = a.B * a.H;
    // (H is a multiple of rows_per_wg: a workgroup's rows belong to one image)
    const int r0 = blockIdx.x * a.rows_per_wg, r1 = min(r0 + a.rows_per_wg, nrows);
    const int nxr = a.dy_max - a.dy_min + 1;
    float wv[NT][CPT], bv[CPT], s1[CPT], s2[CPT];
    const CT* wp = (const CT*)a.w;
    // the tap weights reach the registers through LDS: one cooperative pass over the <= 25 x 64 values per workgroup
    // (every thread fetching its own 72 .. 100 values from global memory was a 72-deep latency chain per workgroup)
    __shared__ float sw[NT * 64];
    for (int i = tid; i < a.ntaps * a.Cout; i += 256) {
        const int t = i / a.Cout, c = i - t * a.Cout;
        sw[t * 64 + c] = (float)wp[((size_t)t * a.Cout_pad + c) * 16];
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) { bv[j] = a.bias ? a.bias[cg * CPT + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    OutT* yo = (OutT*)a.y;
    // all image rows of the workgroup's output rows in ONE staging pass (a barrier pair per output row left the kernel
    // latency-bound at 1.1 TB/s of stores)
    const int b = r0 / a.H, y0 = r0 - b * a.H;
    const int nload = (r1 - r0) + nxr - 1;
    for (int i = tid; i < nload * (a.W + 8); i += 256) {
        const int rr = i / (a.W + 8), xx = i - rr * (a.W + 8) - 4;
        const int yy = y0 + a.dy_min + rr;
        sx[rr][xx + 4] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? a.x[((size_t)b * a.H + yy) * a.W + xx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CPT; ++j) wv[t][j] = (t < a.ntaps) ? sw[t * 64 + cg * CPT + j] : 0.f;
    for (int row = r0; row < r1; ++row) {
        const int rl = row - r0;
        for (int x0 = slot; x0 < a.W; x0 += nslot) {
            float v[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) v[j] = bv[j];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < a.ntaps) {
                    const float xv = sx[rl + a.ty[t]][x0 + 4 + a.tx[t]];
#pragma unroll
                    for (int j = 0; j < CPT; ++j) v[j] = fmaf(wv[t][j], xv, v[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
            if (a.out_act) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) v[j] = fmaxf(v[j], a.out_slope * v[j]);
            }
            OutT* dst = yo + ((size_t)row * a.W + x0) * a.ldy + a.cout_off + cg * CPT;
            if constexpr (CPT == 4) {
                if constexpr (sizeof(OutT) == 2) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
                    *(bf16x4*)dst = o;
                } else {
                    f32x4 lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lo[j] = v[j];
                    *(f32x4*)dst = lo;
                }
            } else if constexpr (sizeof(OutT) == 2) {
                *(bf16x8*)dst = pack_frag<bf16>(v);
            } else {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = v[j]; hi[j] = v[4 + j]; }
                *(f32x4*)dst = lo; *(f32x4*)(dst + 4) = hi;
            }
        }
    }
    if (a.stats != nullptr) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            float u = s1[j], q = s2[j];
            for (int m = ncg; m < 64; m <<= 1) { u += __shfl_xor(u, m); q += __shfl_xor(q, m); }
            s1[j] = u; s2[j] = q;
        }
        __syncthreads();
        if (lane < ncg) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) { red[wave][lane * CPT + j] = s1[j]; red[wave][64 + lane * CPT + j] = s2[j]; }
        }
        __syncthreads();
        if (tid < a.Cout) {
            a.stats[((size_t)blockIdx.x * 2 + 0) * a.Cout + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            a.stats[((size_t)blockIdx.x * 2 + 1) * a.Cout + tid] = red[0][64 + tid] + red[1][64 + tid] + red[2][64 + tid] + red[3][64 + tid];
        }
    }
}

// Four pixels of a row per thread (round 4).  The form above reads one LDS value per CPT FMAs and issues every FMA on its own:
// the 5x5 stem of unet2.py ran at 1.2 TB/s of output (138 us, b16 at 384 x 384; 800 FMAs per pixel).  Here a thread owns CPT
// channels of FOUR neighbouring pixels: a kernel row's taps read one 8-pixel window (three LDS reads: 8 + 16 + 8 bytes, kept as
// register pairs), and every FMA is half of a v_pk_fma_f32 over a channel pair with the pixel value broadcast by op_sel
// (common.hpp) -- same products, same order per output value as the scalar form: bit-identical convolution outputs.
// KW x KW taps in row-major order (checked on the host), W a multiple of 4.
template <typename CT, typename OutT, int KW, int CPT>
__global__ __launch_bounds__(256) void stem_conv4_kernel(const StemK a) {
    constexpr int NT = KW * KW, R = KW / 2, SROWS = 8, NR = SROWS + 2 * R, NP = CPT / 2;
    __shared__ __attribute__((aligned(16))) float sx[NR][512 + 8];
    __shared__ float red[4][2 * 64];
    __shared__ __attribute__((aligned(8))) float sw[NT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncg = a.Cout / CPT;
    const int cg = tid % ncg, slot = tid / ncg;
    const int nslot = 256 / ncg;
    const int nrows = a.B * a.H;
    const int r0 = blockIdx.x * a.rows_per_wg, r1 = min(r0 + a.rows_per_wg, nrows);
    const CT* wp = (const CT*)a.w;
    for (int i = tid; i < NT * a.Cout; i += 256) {
        const int t = i / a.Cout, c = i - t * a.Cout;
        sw[t * 64 + c] = (float)wp[((size_t)t * a.Cout_pad + c) * 16];
    }
    const int b = r0 / a.H, y0 = r0 - b * a.H;
    const int nload = (r1 - r0) + 2 * R;
    // image rows as 16-byte loads, four in flight per thread (element by element the staging was a chain of dependent round trips:
    // 40 % of the workgroup's life); the four halo columns either side are zeros
    const int NQ = a.W >> 2, nq = nload * NQ;
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        f32x4 tq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + tid + 256 * u;
            const int rr = i / NQ, q = i - rr * NQ, yy = y0 - R + rr;
            tq[u] = (i < nq && yy >= 0 && yy < a.H) ? *(const f32x4*)(a.x + ((size_t)b * a.H + yy) * a.W + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + tid + 256 * u;
            const int rr = i / NQ, q = i - rr * NQ;
            if (i < nq) *(f32x4*)&sx[rr][4 + 4 * q] = tq[u];
        }
    }
    for (int i = tid; i < nload * 8; i += 256) sx[i >> 3][(i & 7) < 4 ? (i & 7) : a.W + (i & 7)] = 0.f;
    __syncthreads();
    f32pair wv[NT][NP], bv[NP], s1[NP], s2[NP];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < NP; ++j) wv[t][j] = *(const f32pair*)&sw[t * 64 + cg * CPT + 2 * j];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        bv[j] = a.bias ? (f32pair){a.bias[cg * CPT + 2 * j], a.bias[cg * CPT + 2 * j + 1]} : (f32pair){0.f, 0.f};
        s1[j] = (f32pair){0.f, 0.f}; s2[j] = (f32pair){0.f, 0.f};
    }
    OutT* yo = (OutT*)a.y;
    const int nitems = (r1 - r0) * NQ;
    for (int it = slot; it < nitems; it += nslot) {
        const int rl = it / NQ, x0 = (it - rl * NQ) << 2;
        f32pair v[4][NP];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < NP; ++j) v[p][j] = bv[j];
#pragma unroll
        for (int dy = 0; dy < KW; ++dy) {
            // pixels x0 - 2 .. x0 + 5 of image row (output row + dy - R)
            const float* rp = &sx[rl + dy][x0 + 2];
            const f32x4 mid = *(const f32x4*)(rp + 2);
            const f32pair win[4] = {*(const f32pair*)rp, (f32pair){mid[0], mid[1]}, (f32pair){mid[2], mid[3]}, *(const f32pair*)(rp + 6)};
#pragma unroll
            for (int dx = 0; dx < KW; ++dx)
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        constexpr int dummy = 0; (void)dummy;
                        const int e = p + dx + 2 - R;     // (compile-time after unrolling)
                        if (e & 1) pk_fma_hi(v[p][j], win[e >> 1], wv[dy * KW + dx][j]); else pk_fma_lo(v[p][j], win[e >> 1], wv[dy * KW + dx][j]);
                    }
        }
        const size_t row = (size_t)(r0 + rl);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int j = 0; j < NP; ++j) { s1[j] += v[p][j]; s2[j] = __builtin_elementwise_fma(v[p][j], v[p][j], s2[j]); }
            float o[CPT];
#pragma unroll
            for (int j = 0; j < NP; ++j) { o[2 * j] = v[p][j][0]; o[2 * j + 1] = v[p][j][1]; }
            if (a.out_act) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) o[j] = fmaxf(o[j], a.out_slope * o[j]);
            }
            OutT* dst = yo + (row * a.W + x0 + p) * a.ldy + a.cout_off + cg * CPT;
            if constexpr (CPT == 4) {
                if constexpr (sizeof(OutT) == 2) {
                    bf16x4 q;
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[j] = (bf16)o[j];
                    *(bf16x4*)dst = q;
                } else {
                    f32x4 lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lo[j] = o[j];
                    *(f32x4*)dst = lo;
                }
            } else if constexpr (sizeof(OutT) == 2) {
                *(bf16x8*)dst = pack_frag<bf16>(o);
            } else {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = o[j]; hi[j] = o[4 + j]; }
                *(f32x4*)dst = lo; *(f32x4*)(dst + 4) = hi;
            }
        }
    }
    if (a.stats != nullptr) {
        float t1[CPT], t2[CPT];
#pragma unroll
        for (int j = 0; j < NP; ++j) { t1[2 * j] = s1[j][0]; t1[2 * j + 1] = s1[j][1]; t2[2 * j] = s2[j][0]; t2[2 * j + 1] = s2[j][1]; }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            float u = t1[j], q = t2[j];
            for (int m = ncg; m < 64; m <<= 1) { u += __shfl_xor(u, m); q += __shfl_xor(q, m); }
            t1[j] = u; t2[j] = q;
        }
        __syncthreads();
        if (lane < ncg) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) { red[wave][lane * CPT + j] = t1[j]; red[wave][64 + lane * CPT + j] = t2[j]; }
        }
        __syncthreads();
        if (tid < a.Cout) {
            a.stats[((size_t)blockIdx.x * 2 + 0) * a.Cout + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            a.stats[((size_t)blockIdx.x * 2 + 1) * a.Cout + tid] = red[0][64 + tid] + red[1][64 + tid] + red[2][64 + tid] + red[3][64 + tid];
        }
    }
}

constexpr int STEM_ROWS = 8;

}  // namespace

// 1 when the one-channel kernel takes this descriptor; *stat_blocks = statistics partials it writes
int abc_conv_stem_ok(const abc_conv_desc* d, int* stat_blocks) {
    if (abc_knob("ABC_CONV_NOSTEM")) return 0;
    if (d->Cin != 1 || d->cin_off != 0 || d->src.ldx != 1 || d->dtype_in != ABC_F32 || d->src.scale || d->src.pool || d->src.planar ||
        d->src.drop_p > 0.f)
        return 0;
    if (d->planar_out || d->accumulate || d->stats_rows == 4 || d->stride != 1 || d->om != 1 || d->oy0 || d->ox0 || d->ntaps > 25) return 0;
    if (d->Cout % 8 || d->Cout > 64 || (d->Cout & (d->Cout - 1)) || d->Wg > 512 || d->Hg != d->Hin || d->Wg != d->Win ||
        d->Hout != d->Hg || d->Wout != d->Wg)
        return 0;
    const int esz = d->dtype_out == ABC_BF16 ? 2 : 4;
    if ((d->ldy * esz) % 16 || (d->cout_off * esz) % 16) return 0;
    if (d->dtype_c == ABC_F32 && d->dtype_out != ABC_F32) return 0;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    if (dymax - dymin > (d->ntaps > 9 ? 4 : 2) || dxmin < -4 || dxmax > 4 || d->Hg % STEM_ROWS) return 0;
    if (stat_blocks) *stat_blocks = abc_cdiv(d->B * d->Hg, STEM_ROWS);
    return 1;
}

int abc_conv_stem_launch(const abc_conv_desc* d, abc_stream_t stream) {
    StemK k;
    k.x = (const float*)d->src.x; k.w = d->w; k.bias = d->bias; k.y = d->y; k.stats = d->stats;
    k.B = d->B; k.H = d->Hg; k.W = d->Wg; k.Cout = d->Cout; k.Cout_pad = d->Cout_pad; k.ldy = d->ldy; k.cout_off = d->cout_off;
    k.ntaps = d->ntaps; k.rows_per_wg = STEM_ROWS; k.out_act = d->out_act; k.out_slope = d->out_slope;
    int dymin = 127, dymax = -127;
    for (int t = 0; t < d->ntaps; ++t) { dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax; }
    k.dy_min = dymin; k.dy_max = dymax;
    for (int t = 0; t < d->ntaps; ++t) { k.ty[t] = (int8_t)(d->tap_dy[t] - dymin); k.tx[t] = (int8_t)d->tap_dx[t]; }
    const int nwg = abc_cdiv(d->B * d->Hg, STEM_ROWS);
    hipStream_t st = (hipStream_t)stream;
    // the four-pixel form: a full 3 x 3 / 5 x 5 square in row-major tap order, whole pixel quads
    const int kw = d->ntaps == 9 ? 3 : (d->ntaps == 25 ? 5 : 0);
    bool square = kw != 0 && (d->Wg % 4) == 0;
    for (int t = 0; square && t < d->ntaps; ++t) square = d->tap_dy[t] == t / kw - kw / 2 && d->tap_dx[t] == t % kw - kw / 2;
    if (square && !abc_knob("ABC_STEM_SCALAR")) {
        if (kw == 5) {
            if (d->dtype_c == ABC_F32) hipLaunchKernelGGL((stem_conv4_kernel<float, float, 5, 4>), dim3(nwg), dim3(256), 0, st, k);
            else if (d->dtype_out == ABC_BF16) hipLaunchKernelGGL((stem_conv4_kernel<bf16, bf16, 5, 4>), dim3(nwg), dim3(256), 0, st, k);
            else hipLaunchKernelGGL((stem_conv4_kernel<bf16, float, 5, 4>), dim3(nwg), dim3(256), 0, st, k);
        } else if (d->dtype_c == ABC_F32) hipLaunchKernelGGL((stem_conv4_kernel<float, float, 3, 8>), dim3(nwg), dim3(256), 0, st, k);
        else if (d->dtype_out == ABC_BF16) hipLaunchKernelGGL((stem_conv4_kernel<bf16, bf16, 3, 8>), dim3(nwg), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((stem_conv4_kernel<bf16, float, 3, 8>), dim3(nwg), dim3(256), 0, st, k);
        return abc_check_launch("stem_conv4");
    }
    if (d->ntaps > 9) {
        if (d->dtype_c == ABC_F32) hipLaunchKernelGGL((stem_conv_kernel<float, float, 25, 4>), dim3(nwg), dim3(256), 0, st, k);
        else if (d->dtype_out == ABC_BF16) hipLaunchKernelGGL((stem_conv_kernel<bf16, bf16, 25, 4>), dim3(nwg), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((stem_conv_kernel<bf16, float, 25, 4>), dim3(nwg), dim3(256), 0, st, k);
    } else if (d->dtype_c == ABC_F32) hipLaunchKernelGGL((stem_conv_kernel<float, float, 9, 8>), dim3(nwg), dim3(256), 0, st, k);
    else if (d->dtype_out == ABC_BF16) hipLaunchKernelGGL((stem_conv_kernel<bf16, bf16, 9, 8>), dim3(nwg), dim3(256), 0, st, k);
    else hipLaunchKernelGGL((stem_conv_kernel<bf16, float, 9, 8>), dim3(nwg), dim3(256), 0, st, k);
    return abc_check_launch("stem_conv");
}
