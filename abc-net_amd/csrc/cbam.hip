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
// ONE launch, one workgroup per image (round 4: the pool and the MLP were two dependent ~6 us launches per block, 26 per step):
// pass 1: global avg / max pool of z per (image, channel) from the conv epilogue's per-tile partials, 64 channels at a time:
// 4 tile groups x 64 channels, so the partial rows are read as 256-byte lines by 4 waves at once
// (a thread per channel walking 576 tiles serially: 51 us per call);
// pass 2: the shared MLP on both pooled vectors + sigmoid.
__global__ __launch_bounds__(1024) void cbam_channel_fwd_kernel(const abc_cbam_channel_desc d) {
    extern __shared__ float sm[];
    float* av = sm;              // [C] avg(z)
    float* mx = sm + d.C;        // [C] max(z)
    float* ha = sm + 2 * d.C;    // [mid]
    float* hm = ha + d.mid;      // [mid]
    constexpr int NG = 16;       // tile groups (waves) per 64 channel lanes
    __shared__ double ssum[NG][64];
    __shared__ float smax[NG][64], smin[NG][64];
    const int n = blockIdx.x, cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int cb = 0; cb < d.C; cb += 64) {
        const int c = cb + cl;
        double s = 0.0;
        float vmax = -3.0e38f, vmin = 3.0e38f;
        if (c < d.C) {
            // (sixteen groups x four independent chains: at the 384 x 384 levels an image has 576 tiles, and one chain of dependent
            //  loads and f64 adds took 51-59 us for 3.6 MB; four groups of 256 threads still 23 us)
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
            float mx4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, mn4[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
            int k = grp;
            for (; k + 3 * NG < d.tiles_per_img; k += 4 * NG) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* p = d.partial + ((size_t)(n * d.tiles_per_img + k + NG * u) * 4) * d.C + c;
                    s4[u] += (double)p[0];
                    mx4[u] = fmaxf(mx4[u], p[2 * d.C]);
                    mn4[u] = fminf(mn4[u], p[3 * d.C]);
                }
            }
            for (; k < d.tiles_per_img; k += NG) {
                const float* p = d.partial + ((size_t)(n * d.tiles_per_img + k) * 4) * d.C + c;
                s4[0] += (double)p[0];
                mx4[0] = fmaxf(mx4[0], p[2 * d.C]);
                mn4[0] = fminf(mn4[0], p[3 * d.C]);
            }
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            vmax = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
            vmin = fminf(fminf(mn4[0], mn4[1]), fminf(mn4[2], mn4[3]));
        }
        ssum[grp][cl] = s; smax[grp][cl] = vmax; smin[grp][cl] = vmin;
        __syncthreads();
        if (grp == 0 && c < d.C) {
            s = ssum[0][cl]; vmax = smax[0][cl]; vmin = smin[0][cl];
#pragma unroll
            for (int g = 1; g < NG; ++g) { s += ssum[g][cl]; vmax = fmaxf(vmax, smax[g][cl]); vmin = fminf(vmin, smin[g][cl]); }
            const float sc = d.scale[c], sh = d.shift[c];
            const float a_ = sc * (float)(s / d.HW) + sh, m_ = (sc >= 0.f ? sc * vmax : sc * vmin) + sh;
            d.avgz[(size_t)n * d.C + c] = a_;
            d.maxz[(size_t)n * d.C + c] = m_;
            d.ext[(size_t)n * d.C + c] = sc >= 0.f ? vmax : vmin;
            d.first[(size_t)n * d.C + c] = 0x7FFFFFFF;
            av[c] = a_; mx[c] = m_;
        }
        __syncthreads();
    }
    // hidden unit j: 256 / mid lanes share the dot products (mid <= 32 is a power of two)
    {
        // (the first 256 threads, as when the workgroup had no more)
        const int per = 256 / d.mid;                  // lanes per hidden unit (>= 8)
        const int j = (threadIdx.x & 255) / per, sub = (threadIdx.x & 255) % per;
        float sa = 0.f, sb = 0.f;
        if (threadIdx.x < 256)
            for (int c = sub; c < d.C; c += per) { const float w = d.w1[(size_t)j * d.C + c]; sa += w * av[c]; sb += w * mx[c]; }
        for (int o = 1; o < per && o < 64; o <<= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        if (sub == 0 && threadIdx.x < 256) {
            sa = fmaxf(sa + d.b1[j], 0.f); sb = fmaxf(sb + d.b1[j], 0.f);
            ha[j] = sa; hm[j] = sb;
            d.hid_avg[(size_t)n * d.mid + j] = sa;
            d.hid_max[(size_t)n * d.mid + j] = sb;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.C; c += 1024) {
        float t = 2.f * d.b2[c];
        for (int j = 0; j < d.mid; ++j) t += d.w2[(size_t)c * d.mid + j] * (ha[j] + hm[j]);
        d.ca[(size_t)n * d.C + c] = sigmoidf_(t);
    }
}

// ------------------------------------------------------------------ spatial statistics (forward)
// The per-pixel passes below share one decomposition: grid = (workgroups per image, B), so the image index is uniform
// per workgroup, and a thread keeps ONE fixed group of N channels (C / N is a power of two that divides 256), so every
// per-channel and per-(image, channel) coefficient is loaded once into registers and the loop body is the tensor
// traffic alone.  (The first versions re-read ca[n][c], scale[c], ... per element: 8-10 scalar loads per 16-byte vector
// load, 4-5x off the HBM time.)
//
// thread group of C/N lanes per pixel (NVL vectors per lane when C/N > 64); mean and max over channels of
// o1 = ca*(scale*y+shift), + argmax channel
template <typename T, int NVL>
__global__ __launch_bounds__(256) void cbam_spatial_stats_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    const int ncv = d.C / N;
    const int cpp = ncv / NVL;            // lanes per pixel (power of two <= 64)
    const int hw = d.H * d.W;
    const int n = blockIdx.y;
    const int ppb = 256 / cpp;            // pixels per workgroup pass
    const int sub = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const T* y = (const T*)d.y;
    float sc[NVL][N], sh[NVL][N], ca[NVL][N], ex[NVL][N];
#pragma unroll
    for (int u = 0; u < NVL; ++u)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int c = (sub + u * cpp) * N + j;
            sc[u][j] = d.scale[c]; sh[u][j] = d.shift[c]; ca[u][j] = d.ca[(size_t)n * d.C + c]; ex[u][j] = d.ext[(size_t)n * d.C + c];
        }
    for (int q = blockIdx.x * ppb + pl; q < hw; q += gridDim.x * ppb) {
        const int64_t p = (int64_t)n * hw + q;
        float s = 0.f, m = -3.0e38f;
        int am = 0;
#pragma unroll
        for (int u = 0; u < NVL; ++u) {
            const int vi = sub + u * cpp;
            float v[N];
            ld8<T, N>(y + p * d.ld_y + d.cy_off + vi * N, v);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float o1 = ca[u][j] * fmaf(v[j], sc[u][j], sh[u][j]);
                s += o1;
                if (o1 > m) { m = o1; am = vi * N + j; }
                // the global max-pool's arg-max: FIRST pixel holding the extreme value (an integer min: order-independent)
                if (v[j] == ex[u][j]) atomicMin(d.first + (size_t)n * d.C + vi * N + j, q);
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
    __shared__ __attribute__((aligned(8))) float tile[22][22][2];
    const int b = blockIdx.z, y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const float* __restrict__ w = d.w7;  // [ch][ky][kx]: workgroup-uniform -> scalar registers
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
            {
                const float2 t2 = *(const float2*)&tile[ty + ky][tx + kx][0];
                s += w[ky * 7 + kx] * t2.x + w[49 + ky * 7 + kx] * t2.y;
            }
        d.sa[((size_t)b * d.H + yy) * d.W + xx] = sigmoidf_(s);
    }
}

// d_st[pix][ch] = sum_taps du[pix - off] * w[ch][tap]; weight/bias gradient partials per workgroup.
// Persistent workgroups: a thread keeps its 99 weight-gradient sums in registers over ALL the tiles its workgroup walks and the
// cross-lane reduction runs once per workgroup, not once per tile.
//
// PX pixels of one row per thread (tile = 16 rows x 16 PX columns).  PX = 1 was the round-1..3 kernel: per pixel 49 reads of du,
// 49 8-byte reads of st and 98 broadcast reads of the weights for 196 FMAs -- LDS-issue-bound, 118 us per call at 384 x 384 (b16)
// against 12 us of FMAs.  With PX = 4 the 7 taps of a kernel row of four neighbouring pixels share one 10-pixel window: per
// kernel row 3 reads of du (2 x 16 bytes + 8), 5 reads of st (16 bytes) and 4 reads of the weight row (kept [tap][channel], so
// that (d_mean, d_max) and the (dW_mean, dW_max) pair of a tap are one packed FMA each) serve 56 packed FMAs, whose broadcast
// operand (du, du) is a half of a register pair picked by op_sel (the compiler builds every such pair with two moves).
// The next tile's halo is in flight (registers) under the FMAs of this one.
typedef f32pair c7x2;

template <int PX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void cbam_conv7_bwd_kernel(const abc_cbam_conv7_desc d) {
    typedef c7x2 f32x2;
    constexpr int TW = 16 * PX, HWD = TW + 6;                         // tile / halo width
    constexpr int NI = (22 * HWD + 255) / 256;                        // halo elements per thread
    constexpr bool SWZ = PX == 4;
    // row strides: 16-byte aligned windows; PX = 4: a multiple of 256 bytes, so that the lanes of the two tile rows that meet in one
    // ds_read_b128 lane group fall on distinct 16-byte slots.  A lane's st window is 80 bytes at a lane stride of 32: slots
    // 16 .. 31 of a row are stored with their lowest bit flipped, which puts lanes tx and tx + 8 on different banks.
    constexpr int SU = PX == 4 ? 128 : 24, SS = PX == 4 ? 96 : 24;   // floats per du row, float2 per st row
    __shared__ __attribute__((aligned(16))) float tdu[22 * SU];
    __shared__ __attribute__((aligned(16))) f32x2 tst[22 * SS];
    __shared__ __attribute__((aligned(16))) f32x2 w2[7 * 8];         // [ky][kx (7, padded to 8)] = (w[0][ky][kx], w[1][ky][kx])
    __shared__ float red[16][100];
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + 15) / 16;
    const int ntiles = tiles_x * tiles_y * d.B;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    if (threadIdx.x < 56) {
        const int ky = threadIdx.x >> 3, kx = threadIdx.x & 7;
        w2[threadIdx.x] = kx < 7 ? (f32x2){d.w7[ky * 7 + kx], d.w7[49 + ky * 7 + kx]} : (f32x2){0.f, 0.f};
    }
    float pu[NI];
    f32x2 ps[NI];
    auto issue = [&](int tile) {
        const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = (int)threadIdx.x + 256 * k;
            const int hy = i / HWD, hx = i - hy * HWD;
            const int yy = by * 16 + hy - 3, xx = bx * TW + hx - 3;
            const bool in = i < 22 * HWD && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
            const size_t o = ((size_t)b * d.H + yy) * d.W + xx;
            pu[k] = in ? d.du[o] : 0.f;
            ps[k] = in ? *(const f32x2*)(d.st + o * 2) : (f32x2){0.f, 0.f};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = (int)threadIdx.x + 256 * k;
            const int hy = i / HWD, hx = i - hy * HWD;
            if (i < 22 * HWD) {
                tdu[hy * SU + hx] = pu[k];
                const int slot = hx >> 1;
                tst[hy * SS + (SWZ ? ((slot ^ ((slot >> 4) & 1)) << 1) + (hx & 1) : hx)] = ps[k];
            }
        }
    };
    int soff[SWZ ? 5 : 1];    // this lane's five 16-byte st slots of a window row (f32x2 index)
#pragma unroll
    for (int c = 0; c < (SWZ ? 5 : 1); ++c) { const int slot = 2 * tx + c; soff[c] = (slot ^ ((slot >> 4) & 1)) << 1; }
    f32x2 acc[49];
    float accb = 0.f;
#pragma unroll
    for (int t = 0; t < 49; ++t) acc[t] = (f32x2){0.f, 0.f};
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    while (tile < ntiles) {
        const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        __syncthreads();   // previous tile's readers are done (and w2[] is visible on the first pass)
        commit();
        __syncthreads();
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) issue(next);
        const int yy = by * 16 + ty, xx = bx * TW + tx * PX;
        f32x2 g[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) g[p] = (f32x2){0.f, 0.f};
        if constexpr (PX == 4) {
            // du at this thread's pixels (zero outside the map: the halo fill), as two register pairs
            const float* ucp = &tdu[(ty + 3) * SU + tx * 4 + 3];
            const f32x2 uc01 = (f32x2){ucp[0], ucp[1]}, uc23 = (f32x2){ucp[2], ucp[3]};
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                // window row j of the halo: du for the data gradient's kernel row 6 - j, st for the weight gradient's kernel row j
                f32x2 dw_[5], st_w[10], wr[8];
                {
                    const f32x4 a0 = *(const f32x4*)&tdu[(ty + j) * SU + tx * 4], a1 = *(const f32x4*)&tdu[(ty + j) * SU + tx * 4 + 4];
                    dw_[0] = (f32x2){a0[0], a0[1]}; dw_[1] = (f32x2){a0[2], a0[3]}; dw_[2] = (f32x2){a1[0], a1[1]}; dw_[3] = (f32x2){a1[2], a1[3]};
                    dw_[4] = *(const f32x2*)&tdu[(ty + j) * SU + tx * 4 + 8];
                }
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const f32x4 s4 = *(const f32x4*)&tst[(ty + j) * SS + soff[c]];
                    st_w[2 * c] = (f32x2){s4[0], s4[1]}; st_w[2 * c + 1] = (f32x2){s4[2], s4[3]};
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 q = *(const f32x4*)&w2[(6 - j) * 8 + 2 * c];
                    wr[2 * c] = (f32x2){q[0], q[1]}; wr[2 * c + 1] = (f32x2){q[2], q[3]};
                }
                // data gradient (correlation with the flipped kernel): g[p] += du[y + 3 - ky][x + p + 3 - kx] * w[:, ky, kx], ky = 6 - j
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int c = p + 6 - kx;
                        if (c & 1) pk_fma_hi(g[p], dw_[c >> 1], wr[kx]); else pk_fma_lo(g[p], dw_[c >> 1], wr[kx]);
                    }
                // weight gradient: dW[:, ky, kx] += sum_p du[pix p] * st[pix p + (ky - 3, kx - 3)], ky = j
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    pk_fma_lo(acc[j * 7 + kx], uc01, st_w[kx]);
                    pk_fma_hi(acc[j * 7 + kx], uc01, st_w[kx + 1]);
                    pk_fma_lo(acc[j * 7 + kx], uc23, st_w[kx + 2]);
                    pk_fma_hi(acc[j * 7 + kx], uc23, st_w[kx + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);   // one window row at a time (hoisted, the seven rows' reads took 240 more registers)
            }
            accb += (uc01[0] + uc01[1]) + (uc23[0] + uc23[1]);
        } else {
            float uc[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) uc[p] = tdu[(ty + 3) * SU + tx * PX + p + 3];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                float du_w[PX + 6];
                f32x2 st_w[PX + 6], wr[8];
#pragma unroll
                for (int c = 0; c < PX + 6; ++c) { du_w[c] = tdu[(ty + j) * SU + tx * PX + c]; st_w[c] = tst[(ty + j) * SS + tx * PX + c]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 q = *(const f32x4*)&w2[(6 - j) * 8 + 2 * c];
                    wr[2 * c] = (f32x2){q[0], q[1]}; wr[2 * c + 1] = (f32x2){q[2], q[3]};
                }
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
                        const float u = du_w[p + 6 - kx];
                        g[p] = __builtin_elementwise_fma((f32x2){u, u}, wr[kx], g[p]);
                    }
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[j * 7 + kx] = __builtin_elementwise_fma((f32x2){uc[p], uc[p]}, st_w[p + kx], acc[j * 7 + kx]);
                // (the compiler sinks the FMAs of all seven rows behind the last row's reads otherwise: an empty asm that "modifies" the
                //  row's sums pins them here -- no instruction)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int p = 0; p < PX; ++p) asm volatile("" : "+v"(g[p]));
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) asm volatile("" : "+v"(acc[j * 7 + kx]));
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < PX; ++p) accb += uc[p];
        }
#pragma unroll
        for (int p = 0; p < PX; ++p)
            if (yy < d.H && xx + p < d.W) *(f32x2*)(d.dst + (((size_t)b * d.H + yy) * d.W + xx + p) * 2) = g[p];
        tile = next;
    }
    // 99 sums over the workgroup: the 16 lanes of a row by DPP rotations, then the 16 rows through LDS in row order
    // (594 ds_bpermute round trips, six deep per sum, were ~20 us of the 25 us a small map's call took)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 99; ++t) {
        const float v = row_sum16(t < 49 ? acc[t][0] : (t < 98 ? acc[t - 49][1] : accb));
        if ((lane & 15) == 0) red[wave * 4 + (lane >> 4)][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 99) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
        d.dw_partial[(size_t)blockIdx.x * 99 + threadIdx.x] = s;
    }
}

// one workgroup per output (98 weights + bias): 256 lanes stride the per-workgroup partials, fixed tree -> reproducible
// (one THREAD per output walking 9216 partials serially cost 6 ms per step)
__device__ __forceinline__ void conv7_reduce_body(const float* partial, int nblk, float* dw7, float* db7, int t);
__global__ __launch_bounds__(256) void cbam_conv7_reduce_kernel(const float* partial, int nblk, float* dw7, float* db7) {
    conv7_reduce_body(partial, nblk, dw7, db7, (int)blockIdx.x);
}

// ------------------------------------------------------------------ apply (forward): out = relu(sa*ca*z + r)
template <typename T>
__global__ __launch_bounds__(256) void cbam_apply_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    const int ncv = d.C / N;
    const int hw = d.H * d.W;
    const int n = blockIdx.y;
    const int gt = blockIdx.x * 256 + threadIdx.x;
    const int c = (gt % ncv) * N;
    const int ppass = gridDim.x * (256 / ncv);
    const T* y = (const T*)d.y;
    const T* rs = (const T*)d.res;
    T* out = (T*)d.out;
    float sc[N], sh[N], ca[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] = d.scale[c + j]; sh[j] = d.shift[c + j]; ca[j] = d.ca[(size_t)n * d.C + c + j]; }
    for (int q = gt / ncv; q < hw; q += ppass) {
        const int64_t p = (int64_t)n * hw + q;
        float v[N], r[N], o[N];
        ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
        if (!d.res_pool) {
            ld8<T, N>(rs + p * d.ld_res + d.cres_off + c, r);
        } else {  // residual = 2x2 max-pool of a tensor at twice the resolution (unet2.Down: MaxPool2d then DoubleConv)
            const int yy = q / d.W, x = q - yy * d.W;
            float t[N];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t pp = ((size_t)n * 2 * d.H + 2 * yy + (k >> 1)) * (2 * d.W) + 2 * x + (k & 1);
                ld8<T, N>(rs + pp * d.ld_res + d.cres_off + c, t);
#pragma unroll
                for (int j = 0; j < N; ++j) r[j] = (k == 0) ? t[j] : fmaxf(r[j], t[j]);
            }
        }
        const float sa = d.sa[p];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float z = fmaf(v[j], sc[j], sh[j]);
            o[j] = fmaxf(sa * ca[j] * z + r[j], 0.f);
        }
        st8(out + p * d.ld_out + d.cout_off + c, o);
    }
}

// ------------------------------------------------------------------ backward pass 1
// g = (dOut_same + unpool(dOut_pool)) * [out > 0];   du = (sum_c g*o1) * sa*(1-sa)
template <typename T, int NVL>
__global__ __launch_bounds__(256) void cbam_bwd1_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    const int ncv = d.C / N;
    const int cpp = ncv / NVL;
    const int hw = d.H * d.W;
    const int n = blockIdx.y;
    const int ppb = 256 / cpp;
    const int sub = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const T* y = (const T*)d.y;
    const T* out = (const T*)d.out;
    const T* ds = (const T*)d.d_same;
    const T* dp = (const T*)d.d_pool;
    T* g = (T*)d.g;
    float sc[NVL][N], sh[NVL][N], ca[NVL][N];
#pragma unroll
    for (int u = 0; u < NVL; ++u)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int c = (sub + u * cpp) * N + j;
            sc[u][j] = d.scale[c]; sh[u][j] = d.shift[c]; ca[u][j] = d.ca[(size_t)n * d.C + c];
        }
    for (int q = blockIdx.x * ppb + pl; q < hw; q += gridDim.x * ppb) {
        const int64_t p = (int64_t)n * hw + q;
        const int yy = q / d.W, x = q - yy * d.W;
        float dsa = 0.f;
#pragma unroll
        for (int u = 0; u < NVL; ++u) {
            const int c = (sub + u * cpp) * N;
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
            // (odd H / W: nn.MaxPool2d(2) floors -- the last row / column lies in no window and receives nothing, unet2.py:83)
            if (dp != nullptr && (yy >> 1) < d.H / 2 && (x >> 1) < d.W / 2) {
                // this pixel receives the pooled gradient iff it is the FIRST maximum of its 2x2 window of `out`
                float t[N], w[N];
                const int wy = yy >> 1, wx = x >> 1, me = (yy & 1) * 2 + (x & 1);
                ld8<T, N>(dp + (((size_t)n * (d.H / 2) + wy) * (d.W / 2) + wx) * d.ld_pool + d.cpool_off + c, t);
                int arg[N];
                float best[N];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const size_t pp = ((size_t)n * d.H + 2 * wy + (k >> 1)) * d.W + 2 * wx + (k & 1);
                    ld8<T, N>(out + pp * d.ld_out + d.cout_off + c, w);
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        if (k == 0 || w[j] > best[j]) { best[j] = w[j]; arg[j] = k; }
                }
#pragma unroll
                for (int j = 0; j < N; ++j) gv[j] += (arg[j] == me) ? t[j] : 0.f;
            }
            ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                gv[j] = (ov[j] > 0.f) ? gv[j] : 0.f;
                const float o1 = ca[u][j] * fmaf(v[j], sc[u][j], sh[u][j]);
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
    const int tid = threadIdx.x;
    const int gt = blockIdx.x * 256 + tid;
    const int c = (gt % ncv) * N;
    const int ppass = gridDim.x * (256 / ncv);
    const T* y = (const T*)d.y;
    const T* g = (const T*)d.g;
    T* dz = (T*)d.dz;
    float acc[N], sc[N], sh[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { acc[j] = 0.f; sc[j] = d.scale[c + j]; sh[j] = d.shift[c + j]; }
    const float invC = 1.f / d.C;
    for (int q = gt / ncv; q < hw; q += ppass) {
        const int64_t p = (int64_t)n * hw + q;
        float gv[N], v[N], o[N];
        ld8<T, N>(g + p * d.ld_g + c, gv);
        ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
        const float sa = d.sa[p], dm = d.dst[p * 2] * invC, dx = d.dst[p * 2 + 1];
        const int am = d.amax[p];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float z = fmaf(v[j], sc[j], sh[j]);
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
// Three small launches (the first version was one workgroup walking the images serially: 400 us per call; the second
// had EVERY workgroup rebuild the per-image intermediates from the pass-2 partials: 158 us):
//   pre1: dt[n][c] = (sum of the pass-2 partials) * ca (1 - ca), one thread per (image, channel)      -> work[0 .. B C)
//   pre2: d_hidden[n][j] for both pooled branches, one workgroup per image                              -> work[B C ..)
//   main: weight / bias gradients summed over the images in REGISTERS in image order, and the per-image pool
//         gradients, partitioned over the grid; the intermediates come from `work` into LDS.
// (8 lanes per (image, channel), each summing every 8th partial: one thread walking up to 128 partials was a 32-43 us chain
//  of dependent f64 adds for a few KB of data; the combination order is fixed -- lanes 0..7, then the xor tree)
// pre (= pre1 + pre2 as one launch, one workgroup per image; round 4): a (image, channel) pair per 8 lanes, 32 channels per sweep,
// dt kept in LDS for the hidden-unit sums that follow.  Workgroups past the B images reduce the 7x7 convolution's weight-gradient
// partials of the SAME block (cbam_conv7_reduce_kernel's body: independent of everything here, it used to be a launch of its own).
__device__ __forceinline__ void conv7_reduce_body(const float* partial, int nblk, float* dw7, float* db7, int t) {
    // (the first 256 threads of the workgroup, whatever its size: the summation order is part of the result)
    __shared__ double red7[4];
    double s = 0.0;
    if (threadIdx.x < 256)
        for (int k = threadIdx.x; k < nblk; k += 256) s += (double)partial[(size_t)k * 99 + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) red7[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (red7[0] + red7[1]) + (red7[2] + red7[3]);
        if (t < 98) dw7[t] = (float)tot; else db7[0] = (float)tot;
    }
}

// 1024 threads per image: CL channel lanes x KG tile groups, every thread's loads independent of one another and eight in flight (the
// 8-lanes-per-channel form walked C / 32 sweeps of T / 8 dependent round trips: 9-39 us per call for <= 64 KB of partials)
__global__ __launch_bounds__(1024) void cbam_channel_bwd_pre_kernel(const abc_cbam_channel_desc d, const float* c7_partial, int c7_nblk,
                                                                      float* c7_dw, float* c7_db) {
    if ((int)blockIdx.x >= d.B) { conv7_reduce_body(c7_partial, c7_nblk, c7_dw, c7_db, (int)blockIdx.x - d.B); return; }
    extern __shared__ float sm[];      // dt[C]
    __shared__ double sd[1024];
    const int n = blockIdx.x;
    const int T = d.tiles_per_img;
    int CL = 32;
    while (CL < d.C && CL < 512) CL <<= 1;
    const int KG = 1024 / CL;
    const int cl = threadIdx.x & (CL - 1), kg = threadIdx.x / CL;
    for (int c0 = 0; c0 < d.C; c0 += CL) {
        const int c = c0 + cl;
        const bool ok = c < d.C;
        double acc = 0.0;
        if (ok) {
            const float* p = d.partial + (size_t)n * T * d.C + c;
            int k = kg;
            for (; k + 7 * KG < T; k += 8 * KG) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u * KG) * d.C];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (double)v[u];
            }
            for (; k < T; k += KG) acc += (double)p[(size_t)k * d.C];
        }
        sd[kg * CL + cl] = acc;
        __syncthreads();
        if (kg == 0 && ok) {
            double s = sd[cl];
            for (int g = 1; g < KG; ++g) s += sd[g * CL + cl];
            const float ca = d.ca[(size_t)n * d.C + c];
            const float v = (float)s * ca * (1.f - ca);
            d.work[(size_t)n * d.C + c] = v;
            sm[c] = v;
        }
        __syncthreads();
    }
    // hidden unit j: min(1024 / mid, 64) lanes of ONE wave share its dot product
    const int per = 1024 / d.mid < 64 ? 1024 / d.mid : 64;
    const int j = threadIdx.x / per, sub2 = threadIdx.x % per;
    if (j < d.mid) {
        float s = 0.f;
        for (int c = sub2; c < d.C; c += per) s += d.w2[(size_t)c * d.mid + j] * sm[c];
        for (int o = 1; o < per; o <<= 1) s += __shfl_xor(s, o);
        if (sub2 == 0) {
            const int idx = n * d.mid + j;
            d.work[(size_t)d.B * d.C + idx] = d.hid_avg[idx] > 0.f ? s : 0.f;
            d.work[(size_t)d.B * d.C + d.B * d.mid + idx] = d.hid_max[idx] > 0.f ? s : 0.f;
        }
    }
}

// grid = 5 x nb workgroups: the five result groups (dw2 | dw1 | db2 | db1 | the per-image pool gradients) are independent of one another and
// each is a chain of B (or mid) loads per element: run side by side they take one chain, not five (11-15 us per call before), and the
// intermediates come straight from `work` (every workgroup staging all of them in LDS first was a second chain in front of the first)
__global__ __launch_bounds__(256) void cbam_channel_bwd_kernel(const abc_cbam_channel_desc d) {
    const int tid = threadIdx.x;
    const int C_ = d.C, mid = d.mid, B = d.B;
    const float* dt = d.work;                      // [B][C]
    const float* dha = d.work + (size_t)B * C_;    // [B][mid]
    const float* dhm = dha + B * mid;              // [B][mid]
    const int nbp = gridDim.x / 5, phase = blockIdx.x / nbp;
    const int gsz = nbp * 256, gt = (blockIdx.x - phase * nbp) * 256 + tid;
    if (phase == 0) {
        for (int i = gt; i < C_ * mid; i += gsz) {
            const int c = i / mid, j = i - c * mid;
            float s = 0.f;
#pragma unroll 8
            for (int n = 0; n < B; ++n) s += dt[n * C_ + c] * (d.hid_avg[n * mid + j] + d.hid_max[n * mid + j]);
            d.dw2[i] = s;
        }
    } else if (phase == 1) {
        for (int i = gt; i < mid * C_; i += gsz) {
            const int j = i / C_, c = i - j * C_;
            float s = 0.f;
#pragma unroll 8
            for (int n = 0; n < B; ++n) s += dha[n * mid + j] * d.avgz[(size_t)n * C_ + c] + dhm[n * mid + j] * d.maxz[(size_t)n * C_ + c];
            d.dw1[i] = s;
        }
    } else if (phase == 2) {
        for (int c = gt; c < C_; c += gsz) {
            float s = 0.f;
#pragma unroll 8
            for (int n = 0; n < B; ++n) s += 2.f * dt[n * C_ + c];
            d.db2[c] = s;
        }
    } else if (phase == 3) {
        for (int j = gt; j < mid; j += gsz) {
            float s = 0.f;
#pragma unroll 8
            for (int n = 0; n < B; ++n) s += dha[n * mid + j] + dhm[n * mid + j];
            d.db1[j] = s;
        }
    } else {
        for (int idx = gt; idx < B * C_; idx += gsz) {
            const int n = idx / C_, c = idx - n * C_;
            float a = 0.f, m = 0.f;
#pragma unroll 8
            for (int j = 0; j < mid; ++j) { const float w = d.w1[(size_t)j * C_ + c]; a += w * dha[n * mid + j]; m += w * dhm[n * mid + j]; }
            d.d_avgz[idx] = a;
            d.d_maxz[idx] = m;
        }
    }
}

// ------------------------------------------------------------------ backward pass 3
// d_z = d_o1*ca + d_avgz/HW + [pixel == arg-max of z]*d_maxz ; BN partials (sum d_z, sum d_z*xhat) per workgroup; in place
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd3_kernel(const abc_cbam_pix_desc d) {
    constexpr int N = V8<T>::N;
    __shared__ float red[256][2 * N + 1];
    const int ncv = d.C / N;
    const int hw = d.H * d.W;
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int gt = blockIdx.x * 256 + tid;
    const int c = (gt % ncv) * N;
    const int ppass = gridDim.x * (256 / ncv);
    const T* y = (const T*)d.y;
    T* dz = (T*)d.dz;
    float a1[N], a2[N], mu[N], is[N], ca[N], dav[N], dmz[N];
    int fi[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const size_t nc = (size_t)n * d.C + c + j;
        a1[j] = 0.f; a2[j] = 0.f;
        mu[j] = d.mean[c + j]; is[j] = d.invstd[c + j];
        ca[j] = d.ca[nc]; dav[j] = d.d_avgz[nc] / (float)hw; fi[j] = d.first[nc]; dmz[j] = d.d_maxz[nc];
    }
    for (int q = gt / ncv; q < hw; q += ppass) {
        const int64_t p = (int64_t)n * hw + q;
        float t[N], v[N], o[N];
        ld8<T, N>(dz + p * d.ld_dz + c, t);
        ld8<T, N>(y + p * d.ld_y + d.cy_off + c, v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float gz = t[j] * ca[j] + dav[j];
            if (q == fi[j]) gz += dmz[j];      // AdaptiveMaxPool2d(1) backward: the first pixel holding the maximum (torch's choice)
            o[j] = gz;
            a1[j] += gz;
            a2[j] += gz * ((v[j] - mu[j]) * is[j]);
        }
        st8(dz + p * d.ld_dz + c, o);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) { red[tid][j] = a1[j]; red[tid][N + j] = a2[j]; }
    __syncthreads();
    const size_t blk = (size_t)n * gridDim.x + blockIdx.x;
    for (int cc = tid; cc < d.C; cc += 256) {
        const int v = cc / N, j = cc % N;
        float s1 = 0.f, s2 = 0.f;
        for (int t = v; t < 256; t += ncv) { s1 += red[t][j]; s2 += red[t][N + j]; }
        d.partial[(blk * 2 + 0) * d.C + cc] = s1;
        d.partial[(blk * 2 + 1) * d.C + cc] = s2;
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

static int check_channel(const abc_cbam_channel_desc* d) {
    if (d->mid < 1 || d->mid > 32 || (d->mid & (d->mid - 1))) return abc_fail(ABC_EUNSUPPORTED, "cbam_channel: mid must be a power of two <= 32");
    return ABC_OK;
}

extern "C" int abc_cbam_channel_fwd(const abc_cbam_channel_desc* d, abc_stream_t stream) {
    int rc = check_channel(d);
    if (rc) return rc;
    if (d->ext == nullptr || d->first == nullptr) return abc_fail(ABC_EINVAL, "cbam_channel_fwd: ext / first (arg-max of the global max-pool) required");
    const size_t sh = (size_t)(2 * d->C + 2 * d->mid) * sizeof(float);
    hipLaunchKernelGGL(cbam_channel_fwd_kernel, dim3(d->B), dim3(1024), sh, (hipStream_t)stream, *d);
    return abc_check_launch("cbam_channel_fwd");
}

static int channel_bwd_launch(const abc_cbam_channel_desc* d, const abc_cbam_conv7_desc* c7, abc_stream_t stream) {
    int rc = check_channel(d);
    if (rc) return rc;
    if (d->work == nullptr) return abc_fail(ABC_EINVAL, "cbam_channel_bwd: work buffer of B * (C + 2 mid) floats required");
    // one (c, j) weight-gradient element per thread where possible: the per-element loops over the images are chains of
    // dependent global loads, so width beats depth (8 elements per thread cost 50 us per call)
    const int want = abc_cdiv(d->C * d->mid, 256);
    const int nb = want < 1 ? 1 : (want > 64 ? 64 : want);
    if (c7 != nullptr)
        hipLaunchKernelGGL(cbam_channel_bwd_pre_kernel, dim3(d->B + 99), dim3(1024), (size_t)d->C * sizeof(float), (hipStream_t)stream, *d,
                           (const float*)c7->dw_partial, abc_cbam_conv7_blocks(c7), c7->dw7, c7->db7);
    else
        hipLaunchKernelGGL(cbam_channel_bwd_pre_kernel, dim3(d->B), dim3(1024), (size_t)d->C * sizeof(float), (hipStream_t)stream, *d,
                           (const float*)nullptr, 0, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(cbam_channel_bwd_kernel, dim3(5 * nb), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("cbam_channel_bwd");
}

extern "C" int abc_cbam_channel_bwd(const abc_cbam_channel_desc* d, abc_stream_t stream) { return channel_bwd_launch(d, nullptr, stream); }

extern "C" int abc_cbam_channel_bwd_c7(const abc_cbam_channel_desc* d, const abc_cbam_conv7_desc* c7, abc_stream_t stream) {
    if (c7 == nullptr || c7->dw_partial == nullptr || c7->dw7 == nullptr || c7->db7 == nullptr)
        return abc_fail(ABC_EINVAL, "cbam_channel_bwd_c7: the 7x7 convolution's partials and gradient pointers are required");
    return channel_bwd_launch(d, c7, stream);
}

// workgroups per image of the per-pixel passes: enough to fill the chip a few times over, few enough that the
// per-thread coefficient loads amortise (every thread walks >= ~8 items on the big levels)
static int per_image_blocks(const abc_cbam_pix_desc* d, int items_per_thread_min, int cap_total) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    const int64_t items = (int64_t)d->H * d->W * (d->C / N);
    int64_t b = (items + 256 * (int64_t)items_per_thread_min - 1) / (256 * (int64_t)items_per_thread_min);
    const int cap = cap_total / d->B > 1 ? cap_total / d->B : 1;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int abc_cbam_bwd2_blocks(const abc_cbam_pix_desc* d) {
    const int b = per_image_blocks(d, 4, 4096);
    return b > 128 ? 128 : b;
}
extern "C" int abc_cbam_bwd3_blocks(const abc_cbam_pix_desc* d) { return per_image_blocks(d, 4, 4096) * d->B; }

#define ABC_PIX_LAUNCH(KERNEL, GRID)                                                                                 \
    do {                                                                                                             \
        int rc = check_pix(d);                                                                                       \
        if (rc) return rc;                                                                                           \
        if (d->dtype == ABC_BF16) hipLaunchKernelGGL(KERNEL<bf16>, GRID, dim3(256), 0, (hipStream_t)stream, *d);     \
        else hipLaunchKernelGGL(KERNEL<float>, GRID, dim3(256), 0, (hipStream_t)stream, *d);                         \
    } while (0)

// lanes-per-pixel kernels: NVL = vectors per lane (2 only when C / N = 128, i.e. f32 with 512 channels)
#define ABC_PIXGROUP_LAUNCH(KERNEL, GRID)                                                                            \
    do {                                                                                                             \
        int rc = check_pix(d);                                                                                       \
        if (rc) return rc;                                                                                           \
        const int ncv_ = d->C / (d->dtype == ABC_BF16 ? 8 : 4);                                                      \
        if (ncv_ > 128) return abc_fail(ABC_EUNSUPPORTED, "cbam: more than 128 channel vectors per pixel");          \
        if (d->dtype == ABC_BF16) {                                                                                  \
            if (ncv_ > 64) hipLaunchKernelGGL((KERNEL<bf16, 2>), GRID, dim3(256), 0, (hipStream_t)stream, *d);       \
            else hipLaunchKernelGGL((KERNEL<bf16, 1>), GRID, dim3(256), 0, (hipStream_t)stream, *d);                 \
        } else {                                                                                                     \
            if (ncv_ > 64) hipLaunchKernelGGL((KERNEL<float, 2>), GRID, dim3(256), 0, (hipStream_t)stream, *d);      \
            else hipLaunchKernelGGL((KERNEL<float, 1>), GRID, dim3(256), 0, (hipStream_t)stream, *d);                \
        }                                                                                                            \
    } while (0)

static int group_blocks(const abc_cbam_pix_desc* d) {
    const int N = d->dtype == ABC_BF16 ? 8 : 4;
    const int ncv = d->C / N;
    const int cpp = ncv > 64 ? ncv / 2 : ncv;
    const int ppb = 256 / cpp;
    int64_t b = ((int64_t)d->H * d->W + 4 * ppb - 1) / (4 * ppb);   // ~4 pixels per lane group
    const int cap = 4096 / d->B > 1 ? 4096 / d->B : 1;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int abc_cbam_spatial_stats(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    if (d->ext == nullptr || d->first == nullptr) return abc_fail(ABC_EINVAL, "cbam_spatial_stats: ext / first required");
    ABC_PIXGROUP_LAUNCH(cbam_spatial_stats_kernel, dim3(group_blocks(d), d->B));
    return abc_check_launch("cbam_spatial_stats");
}

extern "C" int abc_cbam_apply_fwd(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    ABC_PIX_LAUNCH(cbam_apply_kernel, dim3(per_image_blocks(d, 4, 4096), d->B));
    return abc_check_launch("cbam_apply_fwd");
}

extern "C" int abc_cbam_bwd1(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    ABC_PIXGROUP_LAUNCH(cbam_bwd1_kernel, dim3(group_blocks(d), d->B));
    return abc_check_launch("cbam_bwd1");
}

extern "C" int abc_cbam_bwd2(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    ABC_PIX_LAUNCH(cbam_bwd2_kernel, dim3(abc_cbam_bwd2_blocks(d), d->B));
    return abc_check_launch("cbam_bwd2");
}

extern "C" int abc_cbam_bwd3(const abc_cbam_pix_desc* d, abc_stream_t stream) {
    if (d->first == nullptr) return abc_fail(ABC_EINVAL, "cbam_bwd3: first (arg-max of the global max-pool) required");
    ABC_PIX_LAUNCH(cbam_bwd3_kernel, dim3(abc_cbam_bwd3_blocks(d) / d->B, d->B));
    return abc_check_launch("cbam_bwd3");
}

extern "C" int abc_cbam_conv7_fwd(const abc_cbam_conv7_desc* d, abc_stream_t stream) {
    hipLaunchKernelGGL(cbam_conv7_fwd_kernel, dim3(abc_cdiv(d->W, 16), abc_cdiv(d->H, 16), d->B), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("cbam_conv7_fwd");
}

// four pixels per thread (16 x 64 tiles) from 48 columns up; the narrow maps of the deep levels keep one pixel per thread
static int conv7_px(const abc_cbam_conv7_desc* d) { return d->W >= 48 ? 4 : 1; }

extern "C" int abc_cbam_conv7_blocks(const abc_cbam_conv7_desc* d) {
    const int ntiles = abc_cdiv(d->W, 16 * conv7_px(d)) * abc_cdiv(d->H, 16) * d->B;
    return ntiles < 512 ? ntiles : 512;   // persistent: 2 workgroups per CU (178 registers), all resident
}

static void conv7_bwd_main(const abc_cbam_conv7_desc* d, abc_stream_t stream) {
    if (conv7_px(d) == 4) hipLaunchKernelGGL(cbam_conv7_bwd_kernel<4>, dim3(abc_cbam_conv7_blocks(d)), dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(cbam_conv7_bwd_kernel<1>, dim3(abc_cbam_conv7_blocks(d)), dim3(256), 0, (hipStream_t)stream, *d);
}

extern "C" int abc_cbam_conv7_bwd(const abc_cbam_conv7_desc* d, abc_stream_t stream) {
    conv7_bwd_main(d, stream);
    hipLaunchKernelGGL(cbam_conv7_reduce_kernel, dim3(99), dim3(256), 0, (hipStream_t)stream, (const float*)d->dw_partial,
                       abc_cbam_conv7_blocks(d), d->dw7, d->db7);
    return abc_check_launch("cbam_conv7_bwd");
}

extern "C" int abc_cbam_conv7_bwd_partial(const abc_cbam_conv7_desc* d, abc_stream_t stream) {
    conv7_bwd_main(d, stream);
    return abc_check_launch("cbam_conv7_bwd_partial");
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
