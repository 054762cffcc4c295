// CBAM attention + residual of the deeper variant (reference: src/unet2.py:6-74), forward and backward.
//
//   z  = BN2(y2)                          (y2 = raw output of the block's second conv, affine on load)
//   ca = sigmoid(MLP(avgpool(z)) + MLP(maxpool(z)))        per (image, channel)     unet2.py:19-22
//   o1 = ca * z
//   sa = sigmoid(conv7x7([mean_c(o1), max_c(o1)]))         per pixel                unet2.py:30-35
//   out = relu(sa * o1 + r),  r = x or conv1x1(x)                                   unet2.py:69-74
//
// avgpool/maxpool of z come from the conv epilogue's per-workgroup (sum, max, min) partials of y2 (BN is a per-channel
// affine: max(z) = scale*max(y)+shift for scale >= 0, scale*min(y)+shift otherwise), so the global pools cost no
// extra pass over the tensor.  Everything else is HBM-bound element-wise work, one pass each:
//   forward : spatial_stats (y2 -> mean/max over channels), conv7 (-> sa), apply (-> out)
//   backward: bwd1 (g = dOut*[out>0], du), conv7_bwd (-> d[mean,max], dW7), bwd2 (-> d_o1, d_ca partials),
//             channel_bwd (MLP), bwd3 (-> d_z, BN partials), then the generic BN/conv backward.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

template <typename T> struct V8;
template <> struct V8<float> { static constexpr int N = 4; };
template <> struct V8<bf16> { static constexpr int N = 8; };

template <typename T, int N> __device__ inline void ld8(const T* p, float* v) { LoadVec<T, N>::ld(p, v); }
__device__ inline void st8(float* p, const float* v) { f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; *(f32x4*)p = t; }
__device__ inline void st8(bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16)v[j];
    *(bf16x8*)p = t;
}
__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------ channel attention (forward)
// one workgroup per image
__global__ __launch_bounds__(256) void cbam_channel_fwd_kernel(const abc_cbam_channel_desc d) {
    extern __shared__ float sm[];
    float* av = sm;              // [C] avg(z)
    float* mx = sm + d.C;        // [C] max(z)
    float* ha = sm + 2 * d.C;    // [mid]
    float* hm = ha + d.mid;      // [mid]
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < d.C; c += 256) {
        double s = 0.0;
        float vmax = -3.0e38f, vmin = 3.0e38f;
        for (int k = 0; k < d.tiles_per_img; ++k) {
            const float* p = d.partial + ((size_t)(n * d.tiles_per_img + k) * 4) * d.C + c;
            s += (double)p[0];
            vmax = fmaxf(vmax, p[2 * d.C]);
            vmin = fminf(vmin, p[3 * d.C]);
        }
        const float sc = d.scale[c], sh = d.shift[c];
        const float a = sc * (float)(s / d.HW) + sh;
        const float m = (sc >= 0.f ? sc * vmax : sc * vmin) + sh;
        av[c] = a; mx[c] = m;
        d.avgz[(size_t)n * d.C + c] = a;
        d.maxz[(size_t)n * d.C + c] = m;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d.mid; j += 256) {
        float sa = d.b1[j], sb = d.b1[j];
        for (int c = 0; c < d.C; ++c) { const float w = d.w1[(size_t)j * d.C + c]; sa += w * av[c]; sb += w * mx[c]; }
        sa = fmaxf(sa, 0.f); sb = fmaxf(sb, 0.f);
        ha[j] = sa; hm[j] = sb;
        d.hid_avg[(size_t)n * d.mid + j] = sa;
        d.hid_max[(size_t)n * d.mid + j] = sb;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.C; c += 256) {
        float t = 2.f * d.b2[c];
        for (int j = 0; j < d.mid; ++j) t += d.w2[(size_t)c * d.mid + j] * (ha[j] + hm[j]);
        d.ca[(size_t)n * d.C + c] = sigmoidf_(t);
    }
}

// ------------------------------------------------------------------ spatial statistics (forward)
// thread group of C/N lanes per pixel; mean and max over channels of o1 = ca*(scale*y+shift), + argmax channel
template <typename T>
__global__ __launch_bounds__(256) void cbam_spatial_stats_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    const int ncv = d.C / N;
    const int cpp = ncv < 64 ? ncv : 64;  // lanes per pixel (power of two <= 64); a lane walks vectors sub, sub+cpp, ...
    const int64_t npix = (int64_t)d.B * d.H * d.W;
    const int ppb = 256 / cpp;  // pixels per workgroup pass
    const int sub = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const T* y = (const T*)d.y;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < npix; p += (int64_t)gridDim.x * ppb) {
        const int n = (int)(p / ((int64_t)d.H * d.W));
        float s = 0.f, m = -3.0e38f;
        int am = 0;
        for (int vi = sub; vi < ncv; vi += cpp) {
            float v[N];
            ld8<T, N>(y + p * d.ld_y + d.cy_off + vi * N, v);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int c = vi * N + j;
                const float o1 = d.ca[(size_t)n * d.C + c] * fmaf(v[j], d.scale[c], d.shift[c]);
                s += o1;
                if (o1 > m) { m = o1; am = c; }
            }
        }
        for (int o = 1; o < cpp; o <<= 1) {
            s += __shfl_xor(s, o);
            const float m2 = __shfl_xor(m, o);
            const int a2 = __shfl_xor(am, o);
            if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }  // first maximum (torch.max over dim)
        }
        if (sub == 0) {
            d.st[p * 2 + 0] = s / d.C;
            d.st[p * 2 + 1] = m;
            d.amax[p] = am;
        }
    }
}

// ------------------------------------------------------------------ 7x7 conv (2 -> 1) + sigmoid, and its backward
__global__ __launch_bounds__(256) void cbam_conv7_fwd_kernel(const abc_cbam_conv7_desc d) {
    __shared__ float tile[22][22][2];
    __shared__ float w[98];
    const int b = blockIdx.z, y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (threadIdx.x < 98) w[threadIdx.x] = d.w7[threadIdx.x];  // [ch][ky][kx]
    for (int i = threadIdx.x; i < 22 * 22; i += 256) {
        const int hy = i / 22, hx = i % 22;
        const int yy = y0 + hy - 3, xx = x0 + hx - 3;
        float a = 0.f, m = 0.f;
        if (yy >= 0 && yy < d.H && xx >= 0 && xx < d.W) {
            const float* p = d.st + (((size_t)b * d.H + yy) * d.W + xx) * 2;
            a = p[0]; m = p[1];
        }
        tile[hy][hx][0] = a; tile[hy][hx][1] = m;
    }
    __syncthreads();
    const int yy = y0 + ty, xx = x0 + tx;
    if (yy < d.H && xx < d.W) {
        float s = d.b7[0];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
                s += w[ky * 7 + kx] * tile[ty + ky][tx + kx][0] + w[49 + ky * 7 + kx] * tile[ty + ky][tx + kx][1];
        d.sa[((size_t)b * d.H + yy) * d.W + xx] = sigmoidf_(s);
    }
}

// d_st[pix][ch] = sum_taps du[pix - off] * w[ch][tap]; weight/bias gradient partials per workgroup
__global__ __launch_bounds__(256) void cbam_conv7_bwd_kernel(const abc_cbam_conv7_desc d) {
    __shared__ float tdu[22][22];
    __shared__ float tst[22][22][2];
    __shared__ float w[98];
    __shared__ float red[4][99];
    const int b = blockIdx.z, y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (threadIdx.x < 98) w[threadIdx.x] = d.w7[threadIdx.x];
    for (int i = threadIdx.x; i < 22 * 22; i += 256) {
        const int hy = i / 22, hx = i % 22;
        const int yy = y0 + hy - 3, xx = x0 + hx - 3;
        const bool in = yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
        const size_t o = ((size_t)b * d.H + yy) * d.W + xx;
        tdu[hy][hx] = in ? d.du[o] : 0.f;
        tst[hy][hx][0] = in ? d.st[o * 2] : 0.f;
        tst[hy][hx][1] = in ? d.st[o * 2 + 1] : 0.f;
    }
    __syncthreads();
    const int yy = y0 + ty, xx = x0 + tx;
    const bool valid = yy < d.H && xx < d.W;
    // data gradient: correlation with the flipped kernel
    if (valid) {
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float u = tdu[ty + 6 - ky][tx + 6 - kx];  // du at (y + 3 - ky, x + 3 - kx)
                g0 += u * w[ky * 7 + kx];
                g1 += u * w[49 + ky * 7 + kx];
            }
        d.dst[(((size_t)b * d.H + yy) * d.W + xx) * 2 + 0] = g0;
        d.dst[(((size_t)b * d.H + yy) * d.W + xx) * 2 + 1] = g1;
    }
    // weight gradient: dW[ch][ky][kx] = sum_pix st[pix + (ky-3, kx-3)][ch] * du[pix]
    const float u = valid ? tdu[ty + 3][tx + 3] : 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < 99; ++t) {
        float v;
        if (t < 98) {
            const int ch = t / 49, ky = (t % 49) / 7, kx = t % 7;
            v = u * tst[ty + ky][tx + kx][ch];
        } else v = u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 99) {
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        d.dw_partial[(size_t)blk * 99 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// one workgroup per output (98 weights + bias): 256 lanes stride the per-workgroup partials, fixed tree -> reproducible
// (one THREAD per output walking 9216 partials serially cost 6 ms per step)
__global__ __launch_bounds__(256) void cbam_conv7_reduce_kernel(const float* partial, int nblk, float* dw7, float* db7) {
    __shared__ double red[4];
    const int t = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256) s += (double)partial[(size_t)k * 99 + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        if (t < 98) dw7[t] = (float)tot; else db7[0] = (float)tot;
    }
}

// ------------------------------------------------------------------ apply (forward): out = relu(sa*ca*z + r)
template <typename T>
__global__ __launch_bounds__(256) void cbam_apply_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    const int ncv = d.C / N;
    const int64_t nitems = (int64_t)d.B * d.H * d.W * ncv;
    const T* y = (const T*)d.y;
    const T* rs = (const T*)d.res;
    T* out = (T*)d.out;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < nitems; it += (int64_t)gridDim.x * 256) {
        const int64_t p = it / ncv;
        const int c = (int)(it % ncv) * N;
        const int x = (int)(p % d.W);
        const int yy = (int)((p / d.W) % d.H);
        const int n = (int)(p / ((int64_t)d.H * d.W));
        float v[N], r[N], o[N];
        ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
        if (!d.res_pool) {
            ld8<T, N>(rs + p * d.ld_res + d.cres_off + c, r);
        } else {  // residual = 2x2 max-pool of a tensor at twice the resolution (unet2.Down: MaxPool2d then DoubleConv)
            float t[N];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t pp = ((size_t)n * 2 * d.H + 2 * yy + (q >> 1)) * (2 * d.W) + 2 * x + (q & 1);
                ld8<T, N>(rs + pp * d.ld_res + d.cres_off + c, t);
#pragma unroll
                for (int j = 0; j < N; ++j) r[j] = (q == 0) ? t[j] : fmaxf(r[j], t[j]);
            }
        }
        const float sa = d.sa[p];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float z = fmaf(v[j], d.scale[c + j], d.shift[c + j]);
            o[j] = fmaxf(sa * d.ca[(size_t)n * d.C + c + j] * z + r[j], 0.f);
        }
        st8(out + p * d.ld_out + d.cout_off + c, o);
    }
}

// ------------------------------------------------------------------ backward pass 1
// g = (dOut_same + unpool(dOut_pool)) * [out > 0];   du = (sum_c g*o1) * sa*(1-sa)
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd1_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    const int ncv = d.C / N;
    const int cpp = ncv < 64 ? ncv : 64;
    const int64_t npix = (int64_t)d.B * d.H * d.W;
    const int ppb = 256 / cpp;
    const int sub = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const T* y = (const T*)d.y;
    const T* out = (const T*)d.out;
    const T* ds = (const T*)d.d_same;
    const T* dp = (const T*)d.d_pool;
    T* g = (T*)d.g;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < npix; p += (int64_t)gridDim.x * ppb) {
        const int x = (int)(p % d.W);
        const int yy = (int)((p / d.W) % d.H);
        const int n = (int)(p / ((int64_t)d.H * d.W));
        float dsa = 0.f;
        for (int vi = sub; vi < ncv; vi += cpp) {
            const int c = vi * N;
            float ov[N], gv[N], v[N];
            ld8<T, N>(out + p * d.ld_out + d.cout_off + c, ov);
#pragma unroll
            for (int j = 0; j < N; ++j) gv[j] = 0.f;
            if (ds != nullptr) {
                float t[N];
                ld8<T, N>(ds + p * d.ld_same + d.csame_off + c, t);
#pragma unroll
                for (int j = 0; j < N; ++j) gv[j] += t[j];
            }
            if (dp != nullptr) {
                // this pixel receives the pooled gradient iff it is the FIRST maximum of its 2x2 window of `out`
                float t[N], w[N];
                const int wy = yy >> 1, wx = x >> 1, me = (yy & 1) * 2 + (x & 1);
                ld8<T, N>(dp + (((size_t)n * (d.H / 2) + wy) * (d.W / 2) + wx) * d.ld_pool + d.cpool_off + c, t);
                int arg[N];
                float best[N];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t pp = ((size_t)n * d.H + 2 * wy + (q >> 1)) * d.W + 2 * wx + (q & 1);
                    ld8<T, N>(out + pp * d.ld_out + d.cout_off + c, w);
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (q == 0 || w[j] > best[j]) { best[j] = w[j]; arg[j] = q; }
                }
#pragma unroll
                for (int j = 0; j < N; ++j) gv[j] += (arg[j] == me) ? t[j] : 0.f;
            }
            ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                gv[j] = (ov[j] > 0.f) ? gv[j] : 0.f;
                const float o1 = d.ca[(size_t)n * d.C + c + j] * fmaf(v[j], d.scale[c + j], d.shift[c + j]);
                dsa += gv[j] * o1;
            }
            st8(g + p * d.ld_g + c, gv);
        }
        for (int o = 1; o < cpp; o <<= 1) dsa += __shfl_xor(dsa, o);
        if (sub == 0) {
            const float sa = d.sa[p];
            d.du[p] = dsa * sa * (1.f - sa);
        }
    }
}

// ------------------------------------------------------------------ backward pass 2
// d_o1 = g*sa + d_mean/C + [c == argmax] * d_max ;  per-image partial of sum_pix d_o1 * z  (for d_ca)
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd2_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    __shared__ float red[256][N + 1];
    const int ncv = d.C / N;
    const int hw = d.H * d.W;
    const int n = blockIdx.y;  // image
    const int64_t nitems = (int64_t)hw * ncv;
    const int tid = threadIdx.x;
    const int cv = (int)((blockIdx.x * 256 + tid) % ncv);
    const int c = cv * N;
    const T* y = (const T*)d.y;
    const T* g = (const T*)d.g;
    T* dz = (T*)d.dz;
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int64_t it = (int64_t)blockIdx.x * 256 + tid; it < nitems; it += (int64_t)gridDim.x * 256) {
        const int64_t p = (int64_t)n * hw + it / ncv;
        float gv[N], v[N], o[N];
        ld8<T, N>(g + p * d.ld_g + c, gv);
        ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
        const float sa = d.sa[p], dm = d.dst[p * 2] / d.C, dx = d.dst[p * 2 + 1];
        const int am = d.amax[p];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float z = fmaf(v[j], d.scale[c + j], d.shift[c + j]);
            const float t = gv[j] * sa + dm + ((c + j) == am ? dx : 0.f);
            o[j] = t;
            acc[j] += t * z;
        }
        st8(dz + p * d.ld_dz + c, o);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) red[tid][j] = acc[j];
    __syncthreads();
    for (int cc = tid; cc < d.C; cc += 256) {
        const int v = cc / N, j = cc % N;
        float s = 0.f;
        for (int t = v; t < 256; t += ncv) s += red[t][j];
        d.partial[((size_t)n * gridDim.x + blockIdx.x) * d.C + cc] = s;
    }
}

// ------------------------------------------------------------------ channel attention backward
// Every workgroup first rebuilds the small per-image intermediates in LDS (dt[B][C], d_hidden[B][mid]: a few hundred
// thousand MACs), then the outputs -- weight / bias gradients summed over the images in REGISTERS in image order, and the
// per-image pool gradients -- are partitioned over the grid.  (The first version was one workgroup walking the images
// serially with global read-modify-write accumulation: 400 us per call.)
__global__ __launch_bounds__(256) void cbam_channel_bwd_kernel(const abc_cbam_channel_desc d) {
    extern __shared__ float sm[];
    float* dt = sm;                      // [B][C]
    float* dha = sm + d.B * d.C;         // [B][mid]
    float* dhm = dha + d.B * d.mid;      // [B][mid]
    const int tid = threadIdx.x;
    const int C_ = d.C, mid = d.mid, B = d.B, T = d.tiles_per_img;
    for (int idx = tid; idx < B * C_; idx += 256) {
        const int n = idx / C_, c = idx - n * C_;
        double s = 0.0;
        for (int k = 0; k < T; ++k) s += (double)d.partial[((size_t)n * T + k) * C_ + c];
        const float ca = d.ca[idx];
        dt[idx] = (float)s * ca * (1.f - ca);
    }
    __syncthreads();
    for (int idx = tid; idx < B * mid; idx += 256) {
        const int n = idx / mid, j = idx - n * mid;
        float s = 0.f;
        for (int c = 0; c < C_; ++c) s += d.w2[(size_t)c * mid + j] * dt[n * C_ + c];
        dha[idx] = d.hid_avg[idx] > 0.f ? s : 0.f;
        dhm[idx] = d.hid_max[idx] > 0.f ? s : 0.f;
    }
    __syncthreads();
    const int gsz = gridDim.x * 256, gt = blockIdx.x * 256 + tid;
    for (int i = gt; i < C_ * mid; i += gsz) {
        const int c = i / mid, j = i - c * mid;
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += dt[n * C_ + c] * (d.hid_avg[n * mid + j] + d.hid_max[n * mid + j]);
        d.dw2[i] = s;
    }
    for (int i = gt; i < mid * C_; i += gsz) {
        const int j = i / C_, c = i - j * C_;
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += dha[n * mid + j] * d.avgz[(size_t)n * C_ + c] + dhm[n * mid + j] * d.maxz[(size_t)n * C_ + c];
        d.dw1[i] = s;
    }
    for (int c = gt; c < C_; c += gsz) {
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += 2.f * dt[n * C_ + c];
        d.db2[c] = s;
    }
    for (int j = gt; j < mid; j += gsz) {
        float s = 0.f;
        for (int n = 0; n < B; ++n) s += dha[n * mid + j] + dhm[n * mid + j];
        d.db1[j] = s;
    }
    for (int idx = gt; idx < B * C_; idx += gsz) {
        const int n = idx / C_, c = idx - n * C_;
        float a = 0.f, m = 0.f;
        for (int j = 0; j < mid; ++j) { const float w = d.w1[(size_t)j * C_ + c]; a += w * dha[n * mid + j]; m += w * dhm[n * mid + j]; }
        d.d_avgz[idx] = a;
        d.d_maxz[idx] = m;
    }
}

// ------------------------------------------------------------------ backward pass 3
// d_z = d_o1*ca + d_avgz/HW + [z == maxz]*d_maxz ; BN partials (sum d_z, sum d_z*xhat) per workgroup; in place
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd3_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    __shared__ float red[256][2 * N + 1];
    const int ncv = d.C / N;
    const int64_t hw = (int64_t)d.H * d.W;
    const int64_t nitems = (int64_t)d.B * hw * ncv;
    const int tid = threadIdx.x;
    const int cv = (int)((blockIdx.x * 256 + tid) % ncv);
    const int c = cv * N;
    const T* y = (const T*)d.y;
    T* dz = (T*)d.dz;
    float a1[N], a2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    for (int64_t it = (int64_t)blockIdx.x * 256 + tid; it < nitems; it += (int64_t)gridDim.x * 256) {
        const int64_t p = it / ncv;
        const int n = (int)(p / hw);
        float t[N], v[N], o[N];
        ld8<T, N>(dz + p * d.ld_dz + c, t);
        ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const size_t nc = (size_t)n * d.C + c + j;
            const float z = fmaf(v[j], d.scale[c + j], d.shift[c + j]);
            float gz = t[j] * d.ca[nc] + d.d_avgz[nc] / (float)hw;
            if (z == d.maxz[nc]) gz += d.d_maxz[nc];
            o[j] = gz;
            a1[j] += gz;
            a2[j] += gz * ((v[j] - d.mean[c + j]) * d.invstd[c + j]);
        }
        st8(dz + p * d.ld_dz + c, o);
    }
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

// dst[.., coff + c] += src[.., c]  (identity residual: d_x += g)
template <typename T>
__global__ __launch_bounds__(256) void add_into_kernel(T* dst, int ld_dst, int cdst_off, const T* src, int ld_src, int csrc_off, int C,
                                                       int64_t npix) {
    constexpr int N = V8<T>::N;
    const int ncv = C / N;
    const int64_t nitems = npix * ncv;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < nitems; it += (int64_t)gridDim.x * 256) {
        const int64_t p = it / ncv;
        const int c = (int)(it % ncv) * N;
        float a[N], b[N];
        ld8<T, N>(dst + p * ld_dst + cdst_off + c, a);
        ld8<T, N>(src + p * ld_src + csrc_off + c, b);
#pragma unroll
        for (int j = 0; j < N; ++j) a[j] += b[j];
        st8(dst + p * ld_dst + cdst_off + c, a);
    }
}

static int pix_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

static int check_pix(const abc_cbam_pix_desc* d) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    if (d->C % N) return abc_fail(ABC_EINVAL, "cbam: C must be a multiple of the vector width");
    const int ncv = d->C / N;
    if (ncv > 256 || 256 % ncv) return abc_fail(ABC_EUNSUPPORTED, "cbam: C/vec must divide 256");
    return ABC_OK;
}

}  // namespace

extern "C" int abc_cbam_channel_fwd(const abc_cbam_channel_desc* d, abc_stream_t stream) {
    const size_t sh = (size_t)(2 * d->C + 2 * d->mid) * sizeof(float);
    hipLaunchKernelGGL(cbam_channel_fwd_kernel, dim3(d->B), dim3(256), sh, (hipStream_t)stream, *d);
    return abc_check_launch("cbam_channel_fwd");
}

extern "C" int abc_cbam_channel_bwd(const abc_cbam_channel_desc* d, abc_stream_t stream) {
    const size_t sh = (size_t)d->B * (d->C + 2 * d->mid) * sizeof(float);
    if (sh > 150 * 1024) return abc_fail(ABC_EUNSUPPORTED, "cbam_channel_bwd: B * (C + 2 mid) floats exceed the LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)cbam_channel_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const int nb = abc_cdiv(d->C * d->mid, 256 * 8) < 1 ? 1 : (abc_cdiv(d->C * d->mid, 256 * 8) > 32 ? 32 : abc_cdiv(d->C * d->mid, 256 * 8));
    hipLaunchKernelGGL(cbam_channel_bwd_kernel, dim3(nb), dim3(256), sh, (hipStream_t)stream, *d);
    return abc_check_launch("cbam_channel_bwd");
}

extern "C" int abc_cbam_bwd2_blocks(const abc_cbam_pix_desc* d) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    return pix_blocks((int64_t)d->H * d->W * (d->C / N)) > 64 ? 64 : pix_blocks((int64_t)d->H * d->W * (d->C / N));
}
extern "C" int abc_cbam_bwd3_blocks(const abc_cbam_pix_desc* d) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    return pix_blocks((int64_t)d->B * d->H * d->W * (d->C / N));
}

#define ABC_PIX_LAUNCH(KERNEL, GRID)                                                                                 \
    do {                                                                                                             \
        int rc = check_pix(d);                                                                                       \
        if (rc) return rc;                                                                                           \
        if (d->dtype == ABC_BF16) hipLaunchKernelGGL(KERNEL<bf16>, GRID, dim3(256), 0, (hipStream_t)stream, *d);     \
        else hipLaunchKernelGGL(KERNEL<float>, GRID, dim3(256), 0, (hipStream_t)stream, *d);                         \
    } while (0)

extern "C" int abc_cbam_spatial_stats(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    const int ppb = 256 / ((d->C / N) < 64 ? (d->C / N) : 64);
    ABC_PIX_LAUNCH(cbam_spatial_stats_kernel, dim3(pix_blocks(((int64_t)d->B * d->H * d->W + ppb - 1) / ppb * 256)));
    return abc_check_launch("cbam_spatial_stats");
}

extern "C" int abc_cbam_apply_fwd(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    ABC_PIX_LAUNCH(cbam_apply_kernel, dim3(pix_blocks((int64_t)d->B * d->H * d->W * (d->C / N))));
    return abc_check_launch("cbam_apply_fwd");
}

extern "C" int abc_cbam_bwd1(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    if (d->d_pool && ((d->H | d->W) & 1)) return abc_fail(ABC_EUNSUPPORTED, "cbam: pooled dims must be even");
    const int ppb = 256 / ((d->C / N) < 64 ? (d->C / N) : 64);
    ABC_PIX_LAUNCH(cbam_bwd1_kernel, dim3(pix_blocks(((int64_t)d->B * d->H * d->W + ppb - 1) / ppb * 256)));
    return abc_check_launch("cbam_bwd1");
}

extern "C" int abc_cbam_bwd2(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    ABC_PIX_LAUNCH(cbam_bwd2_kernel, dim3(abc_cbam_bwd2_blocks(d), d->B));
    return abc_check_launch("cbam_bwd2");
}

extern "C" int abc_cbam_bwd3(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    ABC_PIX_LAUNCH(cbam_bwd3_kernel, dim3(abc_cbam_bwd3_blocks(d)));
    return abc_check_launch("cbam_bwd3");
}

extern "C" int abc_cbam_conv7_fwd(const abc_cbam_conv7_desc* d, abc_stream_t stream) {
    hipLaunchKernelGGL(cbam_conv7_fwd_kernel, dim3(abc_cdiv(d->W, 16), abc_cdiv(d->H, 16), d->B), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("cbam_conv7_fwd");
}

extern "C" int abc_cbam_conv7_blocks(const abc_cbam_conv7_desc* d) { return abc_cdiv(d->W, 16) * abc_cdiv(d->H, 16) * d->B; }

extern "C" int abc_cbam_conv7_bwd(const abc_cbam_conv7_desc* d, abc_stream_t stream) {
    hipLaunchKernelGGL(cbam_conv7_bwd_kernel, dim3(abc_cdiv(d->W, 16), abc_cdiv(d->H, 16), d->B), dim3(256), 0, (hipStream_t)stream, *d);
    hipLaunchKernelGGL(cbam_conv7_reduce_kernel, dim3(99), dim3(256), 0, (hipStream_t)stream, (const float*)d->dw_partial,
                       abc_cbam_conv7_blocks(d), d->dw7, d->db7);
    return abc_check_launch("cbam_conv7_bwd");
}

extern "C" int abc_add_into(void* dst, int32_t ld_dst, int32_t cdst_off, const void* src, int32_t ld_src, int32_t csrc_off, int32_t C,
                            int64_t npix, int32_t dtype, abc_stream_t stream) {
    const int N = dtype == ABC_BF16 ? 8 : 4;
    if (C % N || (ld_dst | cdst_off | ld_src | csrc_off) % N) return abc_fail(ABC_EINVAL, "add_into: alignment");
    const int nb = pix_blocks(npix * (C / N));
    if (dtype == ABC_BF16)
        hipLaunchKernelGGL(add_into_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (bf16*)dst, ld_dst, cdst_off, (const bf16*)src,
                           ld_src, csrc_off, C, npix);
    else
        hipLaunchKernelGGL(add_into_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (float*)dst, ld_dst, cdst_off,
                           (const float*)src, ld_src, csrc_off, C, npix);
    return abc_check_launch("add_into");
}
