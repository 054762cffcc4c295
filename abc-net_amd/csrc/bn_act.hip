// BatchNorm statistics finalisation, eval coefficients, and the backward of
// [BN -> activation -> dropout -> 2x2 max-pool] as two HBM-bound element-wise passes
// with deterministic two-stage per-channel reductions (per-block f32 partials, f64
// finalisation; no atomics).
//
// Reference: nn.BatchNorm2d / ReLU / LeakyReLU / Dropout / MaxPool2d and their autograd
// (unet.py:13-17, 30, 67-69 under train.py:94,140).
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "reduce_bn.hpp"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------ helpers
// pstride = floats between consecutive partial rows (the layer's own C, or the width of a shared partial buffer)
__device__ inline void bn_finalize_fwd_body(const abc_bn_fwd_desc& d, int pstride) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    if (c >= d.C) return;
    const int rows = d.rows == 4 ? 4 : 2;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < d.nblk; k += 256) {
        s1 += (double)d.partial[((size_t)k * rows + 0) * pstride + c];
        s2 += (double)d.partial[((size_t)k * rows + 1) * pstride + c];
    }
    s1 = block_sum_f64(s1, sm);
    s2 = block_sum_f64(s2, sm);
    if (threadIdx.x == 0) {
        const double mean = s1 / d.count;
        double var = s2 / d.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)d.eps));
        const float sc = d.gamma[c] * invstd;
        d.scale[c] = sc;
        d.shift[c] = d.beta[c] - (float)mean * sc;
        d.mean[c] = (float)mean;
        d.invstd[c] = invstd;
        if (d.running_mean != nullptr) {
            const double unb = d.count > 1.0 ? var * d.count / (d.count - 1.0) : var;
            d.running_mean[c] = (1.f - d.momentum) * d.running_mean[c] + d.momentum * (float)mean;
            d.running_var[c] = (1.f - d.momentum) * d.running_var[c] + d.momentum * (float)unb;
            if (c == 0 && d.num_batches_tracked != nullptr) *d.num_batches_tracked += 1;
        }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const abc_bn_fwd_desc d) { bn_finalize_fwd_body(d, d.C); }
// several BatchNorms (the eight heads') in one launch: blockIdx.y = layer
constexpr int MAX_BNB = 8;
struct BnFwdBatch { abc_bn_fwd_desc d[MAX_BNB]; int pstride; };
__global__ __launch_bounds__(256) void bn_finalize_fwd_batch_kernel(const BnFwdBatch bt) {
    bn_finalize_fwd_body(bt.d[blockIdx.y], bt.pstride > 0 ? bt.pstride : bt.d[blockIdx.y].C);
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float* scale,
                               float* shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

__global__ void bn_eval_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, const float* cb, float* scale,
                                    float* bias_out, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    bias_out[c] = ((cb ? cb[c] : 0.f) - rm[c]) * sc + beta[c];
}

__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const abc_bn_bwd_desc d) { bn_finalize_bwd_body(d, d.C, blockIdx.x); }
struct BnBwdBatch { abc_bn_bwd_desc d[MAX_BNB]; int pstride; };
__global__ __launch_bounds__(256) void bn_finalize_bwd_batch_kernel(const BnBwdBatch bt) {
    bn_finalize_bwd_body(bt.d[blockIdx.y], bt.pstride > 0 ? bt.pstride : bt.d[blockIdx.y].C, blockIdx.x);
}

// ------------------------------------------------------------------ pass 1
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16> { static constexpr int N = 8; };

template <typename T, int N> __device__ inline void ldv(const T* p, float* v) { LoadVec<T, N>::ld(p, v); }
template <int N> __device__ inline void stv(float* p, const float* v) {
    f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    *(f32x4*)p = t;
}
template <int N> __device__ inline void stv(bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16)v[j];
    *(bf16x8*)p = t;
}

// The common case -- one full-resolution gradient source, no dropout: two items per thread in flight, the x-hat
// product folded into one FMA per element (sum of g*(x - mean), scaled by invstd once per block).
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_plain_kernel(const abc_act_bwd_desc d) {
    constexpr int N = VecOf<T>::N;
    __shared__ float red[256][2 * N + 1];
    const int ncv = d.C / N;
    const unsigned nitems = (unsigned)((int64_t)d.B * d.H * d.W * ncv);
    const unsigned stride = gridDim.x * 256u;
    const int tid = threadIdx.x;
    const int cv = (int)((blockIdx.x * 256 + tid) % ncv);
    const int c = cv * N;
    const int lg = 31 - __builtin_clz((unsigned)ncv);
    float sc[N], sh[N], sl[N], mu[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] = d.scale[c + j]; sh[j] = d.shift[c + j]; sl[j] = d.slope[c + j]; mu[j] = d.mean[c + j]; }
    float a1[N], a2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    const T* yr = (const T*)d.y_raw + d.cy_off + c;
    const T* ds = (const T*)d.dA_same + d.csame_off + c;
    T* g = (T*)d.g + c;
    auto one = [&](const float* x, const float* da, size_t p) {
        float out[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float y = fmaf(x[j], sc[j], sh[j]);
            const float gg = y > 0.f ? da[j] : sl[j] * da[j];
            out[j] = gg;
            a1[j] += gg;
            a2[j] = fmaf(gg, x[j] - mu[j], a2[j]);
        }
        stv<N>(g + p * d.ld_g, out);
    };
    unsigned it = blockIdx.x * 256u + tid;
    for (; it + stride < nitems; it += 2 * stride) {
        const size_t p0 = it >> lg, p1 = (it + stride) >> lg;
        float x0[N], d0[N], x1[N], d1[N];
        ldv<T, N>(yr + p0 * d.ld_y, x0); ldv<T, N>(ds + p0 * d.ld_same, d0);
        ldv<T, N>(yr + p1 * d.ld_y, x1); ldv<T, N>(ds + p1 * d.ld_same, d1);
        one(x0, d0, p0);
        one(x1, d1, p1);
    }
    if (it < nitems) {
        const size_t p0 = it >> lg;
        float x0[N], d0[N];
        ldv<T, N>(yr + p0 * d.ld_y, x0); ldv<T, N>(ds + p0 * d.ld_same, d0);
        one(x0, d0, p0);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) { red[tid][j] = a1[j]; red[tid][N + j] = a2[j]; }
    __syncthreads();
    for (int cc = tid; cc < d.C; cc += 256) {
        const int v = cc / N, j = cc % N;
        float s1 = 0.f, s2 = 0.f;
        for (int t = v; t < 256; t += ncv) { s1 += red[t][j]; s2 += red[t][N + j]; }
        d.partial[((size_t)blockIdx.x * 2 + 0) * d.C + cc] = s1;
        d.partial[((size_t)blockIdx.x * 2 + 1) * d.C + cc] = s2 * d.invstd[cc];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const abc_act_bwd_desc d) {
    constexpr int N = VecOf<T>::N;
    __shared__ float red[256][2 * N + 1];
    const int ncv = d.C / N;
    const bool pooled = d.dA_pool != nullptr;
    const int Hw = pooled ? d.H / 2 : d.H, Ww = pooled ? d.W / 2 : d.W;  // work grid (windows or pixels)
    const int64_t nitems = (int64_t)d.B * Hw * Ww * ncv;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int tid = threadIdx.x;
    const int cv = (int)((blockIdx.x * 256 + tid) % ncv);  // constant per thread: stride % ncv == 0
    const int c = cv * N;
    float sc[N], sh[N], sl[N], mu[N], is[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        sc[j] = d.scale[c + j]; sh[j] = d.shift[c + j]; sl[j] = d.slope[c + j]; mu[j] = d.mean[c + j]; is[j] = d.invstd[c + j];
    }
    float a1[N], a2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    const T* yr = (const T*)d.y_raw;
    const T* ds = (const T*)d.dA_same;
    const T* dp = (const T*)d.dA_pool;
    T* g = (T*)d.g;
    const float dscale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
    const uint32_t dseed = d.drop_seed + ((d.drop_p > 0.f && d.drop_salt) ? *d.drop_salt : 0u);

    // 32-bit index arithmetic (checked on the host: fewer than 2^31 items); the vector count per pixel is a power of
    // two for every layer of the network -> shift instead of a division
    const bool pow2 = (ncv & (ncv - 1)) == 0;
    const int lg = 31 - __builtin_clz((unsigned)ncv);
    const unsigned n32 = (unsigned)nitems, st32 = (unsigned)stride;
    for (unsigned it = blockIdx.x * 256u + tid; it < n32; it += st32) {
        unsigned pix = pow2 ? (it >> lg) : (it / (unsigned)ncv);
        if (!pooled) {
            const size_t p = pix;  // dense pixel index: (b, y, x) are not needed
            float x[N], da[N], out[N];
            ldv<T, N>(yr + p * d.ld_y + d.cy_off + c, x);
            ldv<T, N>(ds + p * d.ld_same + d.csame_off + c, da);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float y = fmaf(x[j], sc[j], sh[j]);
                float gg = da[j] * (y > 0.f ? 1.f : sl[j]);
                if (d.drop_p > 0.f)
                    gg = abc_drop_keep((uint32_t)(p * d.drop_ld + d.cy_off + c + j), dseed, d.drop_p) ? gg * dscale : 0.f;
                out[j] = gg;
                a1[j] += gg;
                a2[j] += gg * ((x[j] - mu[j]) * is[j]);
            }
            stv<N>(g + p * d.ld_g + c, out);
        } else {
            const int wx = (int)(pix % (unsigned)Ww); pix /= (unsigned)Ww;
            const int wy = (int)(pix % (unsigned)Hw);
            const int b = (int)(pix / (unsigned)Hw);
            float x[4][N], dpool[N];
            size_t pq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pq[q] = ((size_t)b * d.H + 2 * wy + (q >> 1)) * d.W + 2 * wx + (q & 1);
                ldv<T, N>(yr + pq[q] * d.ld_y + d.cy_off + c, x[q]);
            }
            ldv<T, N>(dp + (((size_t)b * Hw + wy) * Ww + wx) * d.ld_pool + d.cpool_off + c, dpool);
            int arg[N];
            float yv[4][N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float best = 0.f;
                arg[j] = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float y = fmaf(x[q][j], sc[j], sh[j]);
                    yv[q][j] = y;
                    const float av = fmaxf(y, sl[j] * y);
                    if (q == 0 || av > best) { best = av; arg[j] = q; }  // first maximum wins (torch max_pool2d)
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float da[N], out[N];
                if (ds != nullptr) ldv<T, N>(ds + pq[q] * d.ld_same + d.csame_off + c, da);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float up = (arg[j] == q) ? dpool[j] : 0.f;
                    if (ds != nullptr) up += da[j];
                    const float gg = up * (yv[q][j] > 0.f ? 1.f : sl[j]);
                    out[j] = gg;
                    a1[j] += gg;
                    a2[j] += gg * ((x[q][j] - mu[j]) * is[j]);
                }
                stv<N>(g + pq[q] * d.ld_g + c, out);
            }
        }
    }
    if (pooled && ((d.H | d.W) & 1)) {
        // odd height / width (input sizes that are not multiples of 32): MaxPool2d(2) floors, so the last row / column
        // belongs to no window -- its gradient is the full-resolution source alone (or zero)
        const int wodd = d.W & 1, hodd = d.H & 1;
        const int ncol = wodd ? d.H : 0;                 // pixels (y, W - 1)
        const int nrow = hodd ? 2 * Ww : 0;              // pixels (H - 1, x < 2 Ww)
        const unsigned nl = (unsigned)(ncol + nrow);
        const unsigned n2 = (unsigned)d.B * nl * (unsigned)ncv;
        for (unsigned it = blockIdx.x * 256u + tid; it < n2; it += st32) {
            unsigned l = pow2 ? (it >> lg) : (it / (unsigned)ncv);
            const int b = (int)(l / nl);
            l -= (unsigned)b * nl;
            const int y = (int)l < ncol ? (int)l : d.H - 1, x = (int)l < ncol ? d.W - 1 : (int)l - ncol;
            const size_t p = ((size_t)b * d.H + y) * d.W + x;
            float xv[N], da[N], out[N];
            ldv<T, N>(yr + p * d.ld_y + d.cy_off + c, xv);
            if (ds != nullptr) ldv<T, N>(ds + p * d.ld_same + d.csame_off + c, da);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float yv = fmaf(xv[j], sc[j], sh[j]);
                const float gg = (ds != nullptr ? da[j] : 0.f) * (yv > 0.f ? 1.f : sl[j]);
                out[j] = gg;
                a1[j] += gg;
                a2[j] += gg * ((xv[j] - mu[j]) * is[j]);
            }
            stv<N>(g + p * d.ld_g + c, out);
        }
    }
    // block reduction per channel: threads with equal (tid % ncv) share a channel vector
#pragma unroll
    for (int j = 0; j < N; ++j) { red[tid][j] = a1[j]; red[tid][N + j] = a2[j]; }
    __syncthreads();
    for (int cc = tid; cc < d.C; cc += 256) {
        const int v = cc / N, j = cc % N;
        float s1 = 0.f, s2 = 0.f;
        for (int t = v; t < 256; t += ncv) { s1 += red[t][j]; s2 += red[t][N + j]; }
        d.partial[((size_t)blockIdx.x * 2 + 0) * d.C + cc] = s1;
        d.partial[((size_t)blockIdx.x * 2 + 1) * d.C + cc] = s2;
    }
}

// ------------------------------------------------------------------ pass 2
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const abc_bn_apply_desc d) {
    constexpr int N = VecOf<T>::N;
    const int ncv = d.C / N;
    const int64_t nitems = d.npix * ncv;
    const int64_t stride = (int64_t)gridDim.x * 256;
    T* g = (T*)d.g;
    T* o = d.out ? (T*)d.out : g;
    const int ld_o = d.out ? d.ld_out : d.ld_g;
    const T* yr = (const T*)d.y_raw;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < nitems; it += stride) {
        const int64_t p = it / ncv;
        const int c = (int)(it % ncv) * N;
        float gv[N], x[N], out[N];
        ldv<T, N>(g + p * d.ld_g + c, gv);
        ldv<T, N>(yr + p * d.ld_y + d.cy_off + c, x);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float xh = (x[j] - d.mean[c + j]) * d.invstd[c + j];
            out[j] = d.gscale[c + j] * (gv[j] - d.k1[c + j] - xh * d.k2[c + j]);
        }
        stv<N>(o + p * ld_o + c, out);
    }
}

// ------------------------------------------------------------------ column sums
// 256 threads = PL pixel lanes x CW channel lanes (CW = power of two <= 64 covering C or a slice of it);
// a thread keeps up to 8 channel accumulators (C <= 8*CW = 512), block-level LDS reduce over the pixel lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int64_t npix, int ld, int c_off, int C, const float* cs,
                                                      float* work, int CW) {
    __shared__ float red[256];
    const int PL = 256 / CW;
    const int cl = threadIdx.x % CW, pl = threadIdx.x / CW;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < npix; p += (int64_t)gridDim.x * PL) {
        const T* row = x + p * ld + c_off;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cl + j * CW;
            if (c < C) acc[j] += (float)row[c];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cl + j * CW;
        __syncthreads();
        red[threadIdx.x] = acc[j];
        __syncthreads();
        if (pl == 0 && c < C) {
            float s = 0.f;
            for (int q = 0; q < PL; ++q) s += red[q * CW + cl];
            work[(size_t)blockIdx.x * C + c] = s * (cs ? cs[c_off + c] : 1.f);
        }
    }
}
// Vector form (C a multiple of the 16-byte vector, aligned rows): a thread owns one group of N channels and walks
// pixels with 16-byte loads (the scalar form above read 2 bytes per load: 1 TB/s on the 151 MB tensors of unet2's first
// blocks); per-workgroup partials in the same [blocks][C] layout.
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* x, int64_t npix, int ld, int c_off, int C, const float* cs, float* work) {
    constexpr int N = VecOf<T>::N;
    __shared__ float red[256][N + 1];
    const int ncv = C / N;                 // divides 256 (checked on the host)
    const int cv = threadIdx.x % ncv, pl = threadIdx.x / ncv, PL = 256 / ncv;
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    const T* base = x + c_off + cv * N;
    int64_t p = (int64_t)blockIdx.x * PL + pl;
    const int64_t step = (int64_t)gridDim.x * PL;
    for (; p + step < npix; p += 2 * step) {   // two loads in flight
        float a[N], b[N];
        ldv<T, N>(base + p * ld, a);
        ldv<T, N>(base + (p + step) * ld, b);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] += a[j] + b[j];
    }
    if (p < npix) {
        float a[N];
        ldv<T, N>(base + p * ld, a);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] += a[j];
    }
#pragma unroll
    for (int j = 0; j < N; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int v = c / N, j = c % N;
        float s = 0.f;
        for (int q = 0; q < PL; ++q) s += red[q * ncv + v][j];
        work[(size_t)blockIdx.x * C + c] = s * (cs ? cs[c_off + c] : 1.f);
    }
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* work, int nblk, int C, float* out) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) s += (double)work[(size_t)k * C + c];
    s = block_sum_f64(s, sm);
    if (threadIdx.x == 0) out[c] = (float)s;
}
// column sums of x and of x * w[pixel] in one pass (w: one f32 value per pixel): bias and weight gradient of a 1x1 convolution over a
// ONE-channel input (unet2.py:62 res_conv of the first block, unet2.py:135): work = [blocks][2][C]
template <typename T>
__global__ __launch_bounds__(256) void colsum_w1_kernel(const T* x, int64_t npix, int ld, int c_off, int C, const float* w, float* work) {
    constexpr int N = VecOf<T>::N;
    __shared__ float red[256][2 * N + 1];
    const int ncv = C / N;                 // divides 256 (checked on the host)
    const int cv = threadIdx.x % ncv, pl = threadIdx.x / ncv, PL = 256 / ncv;
    float acc[N], accw[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { acc[j] = 0.f; accw[j] = 0.f; }
    const T* base = x + c_off + cv * N;
    int64_t p = (int64_t)blockIdx.x * PL + pl;
    const int64_t step = (int64_t)gridDim.x * PL;
    for (; p + 3 * step < npix; p += 4 * step) {   // four loads in flight
        float a[4][N], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ldv<T, N>(base + (p + u * step) * ld, a[u]); wv[u] = w[p + u * step]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < N; ++j) { acc[j] += a[u][j]; accw[j] = fmaf(a[u][j], wv[u], accw[j]); }
    }
    for (; p < npix; p += step) {
        float a[N];
        ldv<T, N>(base + p * ld, a);
        const float wv = w[p];
#pragma unroll
        for (int j = 0; j < N; ++j) { acc[j] += a[j]; accw[j] = fmaf(a[j], wv, accw[j]); }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) { red[threadIdx.x][j] = acc[j]; red[threadIdx.x][N + j] = accw[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int row = c / C, cc = c - row * C;
        const int v = cc / N, j = cc % N;
        float s = 0.f;
        for (int q = 0; q < PL; ++q) s += red[q * ncv + v][row * N + j];
        work[((size_t)blockIdx.x * 2 + row) * C + cc] = s;
    }
}
__global__ __launch_bounds__(256) void colsum_w1_reduce_kernel(const float* work, int nblk, int C, float* out_sum, float* out_w) {
    __shared__ double sm[4];
    const int c = blockIdx.x, row = blockIdx.y;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) s += (double)work[((size_t)k * 2 + row) * C + c];
    s = block_sum_f64(s, sm);
    if (threadIdx.x == 0) {
        if (row == 0) { if (out_sum != nullptr) out_sum[c] = (float)s; }
        else out_w[c] = (float)s;
    }
}

__global__ void fill_kernel(float* p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// NHWC(f32, strided) <-> NCHW(f32) through a 32x32 LDS transpose tile
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* src, int ld, int c_off, int C, int HW, float* dst) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        tile[k][tx] = (p < HW && c < C) ? src[((size_t)b * HW + p) * ld + c_off + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (p < HW && c < C) dst[((size_t)b * C + c) * HW + p] = tile[tx][k];
    }
}
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, int C, int HW, float* dst, int ld, int c_off) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (p < HW && c < C) ? src[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        if (p < HW && c < C) dst[((size_t)b * HW + p) * ld + c_off + c] = tile[tx][k];
    }
}

static int ew_blocks(int64_t nitems) {
    int64_t b = (nitems + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int abc_bn_finalize_fwd(const abc_bn_fwd_desc* d, abc_stream_t stream) {
    if (d->C < 1 || d->nblk < 1) return abc_fail(ABC_EINVAL, "bn_finalize_fwd: empty");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(d->C), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("bn_finalize_fwd");
}

// n <= 8 BatchNorm finalisations in one launch (the heads').  bwd: pstride > 0 = the layers' partial sums are column
// slices of ONE [nblk][2][pstride] buffer (each descriptor's `partial` points at its first column), 0 = own buffers.
extern "C" int abc_bn_finalize_fwd_batch(const abc_bn_fwd_desc* descs, int32_t n, int32_t pstride, abc_stream_t stream) {
    if (n < 1 || n > MAX_BNB) return abc_fail(ABC_EINVAL, "bn_finalize_fwd_batch: 1..8 layers");
    BnFwdBatch bt;
    int cmax = 0;
    for (int i = 0; i < n; ++i) { bt.d[i] = descs[i]; cmax = descs[i].C > cmax ? descs[i].C : cmax; if (descs[i].C < 1 || descs[i].nblk < 1) return abc_fail(ABC_EINVAL, "bn_finalize_fwd_batch: empty"); }
    for (int i = n; i < MAX_BNB; ++i) bt.d[i] = descs[0];
    bt.pstride = pstride;
    hipLaunchKernelGGL(bn_finalize_fwd_batch_kernel, dim3(cmax, n), dim3(256), 0, (hipStream_t)stream, bt);
    return abc_check_launch("bn_finalize_fwd_batch");
}

extern "C" int abc_bn_finalize_bwd_batch(const abc_bn_bwd_desc* descs, int32_t n, int32_t pstride, abc_stream_t stream) {
    if (n < 1 || n > MAX_BNB) return abc_fail(ABC_EINVAL, "bn_finalize_bwd_batch: 1..8 layers");
    BnBwdBatch bt;
    int cmax = 0;
    for (int i = 0; i < n; ++i) { bt.d[i] = descs[i]; cmax = descs[i].C > cmax ? descs[i].C : cmax; if (descs[i].C < 1 || descs[i].nblk < 1) return abc_fail(ABC_EINVAL, "bn_finalize_bwd_batch: empty"); }
    for (int i = n; i < MAX_BNB; ++i) bt.d[i] = descs[0];
    bt.pstride = pstride;
    hipLaunchKernelGGL(bn_finalize_bwd_batch_kernel, dim3(cmax, n), dim3(256), 0, (hipStream_t)stream, bt);
    return abc_check_launch("bn_finalize_bwd_batch");
}

extern "C" int abc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float* scale, float* shift, int32_t C, float eps, abc_stream_t stream) {
    hipLaunchKernelGGL(bn_eval_kernel, dim3(abc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, scale, shift, C, eps);
    return abc_check_launch("bn_eval_coeffs");
}

extern "C" int abc_bn_eval_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                const float* conv_bias, float* scale, float* bias_out, int32_t C, float eps, abc_stream_t stream) {
    hipLaunchKernelGGL(bn_eval_fold_kernel, dim3(abc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, conv_bias, scale, bias_out, C, eps);
    return abc_check_launch("bn_eval_fold");
}

extern "C" int abc_bn_finalize_bwd(const abc_bn_bwd_desc* d, abc_stream_t stream) {
    if (d->C < 1 || d->nblk < 1) return abc_fail(ABC_EINVAL, "bn_finalize_bwd: empty");
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(d->C), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("bn_finalize_bwd");
}

static int act_bwd_check(const abc_act_bwd_desc* d) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    if (d->C % N) return abc_fail(ABC_EINVAL, "act_bwd: C must be a multiple of the vector width");
    const int ncv = d->C / N;
    if (ncv > 256 || (256 % ncv)) return abc_fail(ABC_EUNSUPPORTED, "act_bwd: C/vec must divide 256");
    if (d->dA_pool && (((d->H | d->W) & 1) && d->drop_p > 0.f)) return abc_fail(ABC_EUNSUPPORTED, "act_bwd: odd pooled dims with dropout");
    if (!d->dA_pool && !d->dA_same) return abc_fail(ABC_EINVAL, "act_bwd: no gradient source");
    if ((d->ld_y | d->ld_g | d->cy_off) % N) return abc_fail(ABC_EINVAL, "act_bwd: alignment");
    if ((int64_t)d->B * d->H * d->W * ncv >= (int64_t(1) << 31)) return abc_fail(ABC_EUNSUPPORTED, "act_bwd: tensor too large for 32-bit indexing");
    return ABC_OK;
}

extern "C" int abc_act_bwd_blocks(const abc_act_bwd_desc* d) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    const bool pooled = d->dA_pool != nullptr;
    const int64_t n = (int64_t)d->B * (pooled ? d->H / 2 : d->H) * (pooled ? d->W / 2 : d->W) * (d->C / N);
    return ew_blocks(n);
}

extern "C" int abc_act_bwd(const abc_act_bwd_desc* d, abc_stream_t stream) {
    int rc = act_bwd_check(d);
    if (rc) return rc;
    const int nb = abc_act_bwd_blocks(d);
    if (d->dA_pool == nullptr && d->drop_p <= 0.f) {
        if (d->dtype == ABC_BF16) hipLaunchKernelGGL(act_bwd_plain_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
        else hipLaunchKernelGGL(act_bwd_plain_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
        return abc_check_launch("act_bwd");
    }
    if (d->dtype == ABC_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("act_bwd");
}

extern "C" int abc_bn_apply_bwd(const abc_bn_apply_desc* d, abc_stream_t stream) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    if (d->C % N || (d->ld_g | d->ld_y | d->cy_off) % N || (d->out && d->ld_out % N)) return abc_fail(ABC_EINVAL, "bn_apply: alignment");
    const int nb = ew_blocks(d->npix * (d->C / N));
    if (d->dtype == ABC_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("bn_apply_bwd");
}

extern "C" int abc_colsum_blocks(int64_t npix) {
    int64_t b = (npix + 63) / 64;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int abc_colsum(const void* x, int32_t dtype, int64_t npix, int32_t ld, int32_t c_off, int32_t C,
                          const float* chan_scale, float* work, float* out, abc_stream_t stream) {
    const int nb = abc_colsum_blocks(npix);
    hipStream_t st = (hipStream_t)stream;
    if (C > 512) return abc_fail(ABC_EUNSUPPORTED, "colsum: C > 512");
    {
        const int N = dtype == ABC_BF16 ? 8 : 4;
        const int ncv = C / N;
        if (C % N == 0 && ncv >= 1 && ncv <= 256 && 256 % ncv == 0 && ld % N == 0 && c_off % N == 0 && !abc_knob("ABC_COLSUM_SCALAR")) {
            if (dtype == ABC_BF16)
                hipLaunchKernelGGL(colsum_vec_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)x, npix, ld, c_off, C, chan_scale, work);
            else
                hipLaunchKernelGGL(colsum_vec_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, npix, ld, c_off, C, chan_scale, work);
            hipLaunchKernelGGL(colsum_reduce_kernel, dim3(C), dim3(256), 0, st, (const float*)work, nb, C, out);
            return abc_check_launch("colsum");
        }
    }
    int CW = 1;
    while (CW < C && CW < 64) CW <<= 1;
    if (dtype == ABC_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)x, npix, ld, c_off, C, chan_scale, work, CW);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, npix, ld, c_off, C, chan_scale, work, CW);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(C), dim3(256), 0, st, (const float*)work, nb, C, out);
    return abc_check_launch("colsum");
}

extern "C" int abc_colsum_w1(const void* x, int32_t dtype, int64_t npix, int32_t ld, int32_t c_off, int32_t C, const float* w,
                             float* work, float* out_sum, float* out_w, abc_stream_t stream) {
    const int nb = abc_colsum_blocks(npix);
    const int N = dtype == ABC_BF16 ? 8 : 4;
    const int ncv = C / N;
    if (x == nullptr || w == nullptr || work == nullptr || out_w == nullptr) return abc_fail(ABC_EINVAL, "colsum_w1: x, w, work and out_w are required");
    if (C % N || ncv < 1 || ncv > 256 || 256 % ncv || ld % N || c_off % N)
        return abc_fail(ABC_EUNSUPPORTED, "colsum_w1: C / vector width must divide 256, ld and c_off be multiples of the vector width");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ABC_BF16) hipLaunchKernelGGL(colsum_w1_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)x, npix, ld, c_off, C, w, work);
    else hipLaunchKernelGGL(colsum_w1_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, npix, ld, c_off, C, w, work);
    hipLaunchKernelGGL(colsum_w1_reduce_kernel, dim3(C, 2), dim3(256), 0, st, (const float*)work, nb, C, out_sum, out_w);
    return abc_check_launch("colsum_w1");
}

// 2x2 max-pool of the activated tensor, 8 channels (16 / 32 bytes) per thread
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void pool_act_kernel(const InT* x, const float* sc, const float* sh, const float* sl, int Hx, int Wx,
                                                        int ldx, int c_off, int C, int64_t nseg, OutT* out, int ld_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nseg) return;
    const int segs = C / 8;
    const int sg = (int)(i % segs);
    int64_t pix = i / segs;
    const int Wo = Wx / 2, Ho = Hx / 2;
    const int xo = (int)(pix % Wo); pix /= Wo;
    const int yo = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int c = c_off + sg * 8;
    float a[8], s_[8], t_[8];
    const bool tr = sc != nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = tr ? sc[c + j] : 1.f; s_[j] = tr ? sh[c + j] : 0.f; t_[j] = tr ? sl[c + j] : 1.f; }
    float m[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[8];
        LoadVec<InT, 8>::ld(x + ((size_t)(b * Hx + 2 * yo + (q >> 1)) * Wx + 2 * xo + (q & 1)) * ldx + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = tr ? abc_act(v[j], a[j], s_[j], t_[j]) : v[j];
            m[j] = (q == 0) ? y : fmaxf(m[j], y);
        }
    }
    OutT* dst = out + ((size_t)(b * Ho + yo) * Wo + xo) * ld_out + sg * 8;
    if constexpr (sizeof(OutT) == 2) {
        *(bf16x8*)dst = pack_frag<bf16>(m);
    } else {
        f32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo[j] = m[j]; hi[j] = m[4 + j]; }
        *(f32x4*)dst = lo; *(f32x4*)(dst + 4) = hi;
    }
}

extern "C" int abc_pool_act(const abc_act_src* src, int32_t dtype_in, int32_t c_off, int32_t C, int32_t B, void* out, int32_t dtype_out,
                            int32_t ld_out, abc_stream_t stream) {
    // (odd Hx / Wx: MaxPool2d(2) floors -- the last row / column is dropped, unet.py:30 on sizes that are not multiples of 32)
    if (C % 8 || c_off % 8 || src->ldx % 8 || ld_out % 8 || src->Hx < 2 || src->Wx < 2) return abc_fail(ABC_EINVAL, "pool_act: alignment");
    if (src->planar || src->drop_p > 0.f) return abc_fail(ABC_EUNSUPPORTED, "pool_act: planar / dropout source");
    const int64_t nseg = (int64_t)B * (src->Hx / 2) * (src->Wx / 2) * (C / 8);
    const dim3 grid((unsigned)((nseg + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype_in == ABC_BF16 && dtype_out == ABC_BF16)
        hipLaunchKernelGGL((pool_act_kernel<bf16, bf16>), grid, dim3(256), 0, st, (const bf16*)src->x, src->scale, src->shift, src->slope, src->Hx,
                           src->Wx, src->ldx, c_off, C, nseg, (bf16*)out, ld_out);
    else if (dtype_in == ABC_F32 && dtype_out == ABC_F32)
        hipLaunchKernelGGL((pool_act_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src->x, src->scale, src->shift, src->slope,
                           src->Hx, src->Wx, src->ldx, c_off, C, nseg, (float*)out, ld_out);
    else if (dtype_in == ABC_F32 && dtype_out == ABC_BF16)
        hipLaunchKernelGGL((pool_act_kernel<float, bf16>), grid, dim3(256), 0, st, (const float*)src->x, src->scale, src->shift, src->slope,
                           src->Hx, src->Wx, src->ldx, c_off, C, nseg, (bf16*)out, ld_out);
    else return abc_fail(ABC_EUNSUPPORTED, "pool_act: dtype combination");
    return abc_check_launch("pool_act");
}

extern "C" int abc_fill_f32(float* p, float v, int64_t n, abc_stream_t stream) {
    if (n <= 0) return ABC_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
    return abc_check_launch("fill");
}

extern "C" int abc_nhwc_to_nchw_f32(const float* src, int32_t ld, int32_t c_off, int32_t C, int32_t B, int32_t H, int32_t W,
                                    float* dst, abc_stream_t stream) {
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(abc_cdiv(HW, 32), abc_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, src,
                       ld, c_off, C, HW, dst);
    return abc_check_launch("nhwc_to_nchw");
}

extern "C" int abc_nchw_to_nhwc_f32(const float* src, int32_t C, int32_t B, int32_t H, int32_t W, float* dst, int32_t ld,
                                    int32_t c_off, abc_stream_t stream) {
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(abc_cdiv(HW, 32), abc_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, src,
                       C, HW, dst, ld, c_off);
    return abc_check_launch("nchw_to_nhwc");
}
