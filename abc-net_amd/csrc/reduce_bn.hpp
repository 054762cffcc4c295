// Bodies shared by the split-K slab reduction (wgrad.hip) and the BatchNorm-backward finaliser (bn_act.hip): each as a device function
// of an explicit block index, so that ONE launch can run both (abc_wgrad_reduce_bn_bwd: the finaliser of layer l is a dependent
// 5 us launch of 128 blocks; the slab reduction of layer l + 1 is independent of it and HBM-bound -- side by side they cost the
// reduction alone).  256-thread blocks.
#pragma once
#include "common.hpp"
#include "../../include/abcnet_hip.h"

__device__ inline double block_sum_f64(double v, double* sm) {
    // 256 threads
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}


__device__ inline void bn_finalize_bwd_body(const abc_bn_bwd_desc& d, int pstride, int c) {
    __shared__ double sm[4];
    if (c >= d.C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < d.nblk; k += 256) {
        s1 += (double)d.partial[((size_t)k * 2 + 0) * pstride + c];
        s2 += (double)d.partial[((size_t)k * 2 + 1) * pstride + c];
    }
    s1 = block_sum_f64(s1, sm);
    s2 = block_sum_f64(s2, sm);
    if (threadIdx.x == 0) {
        const float fin = d.in_scale != nullptr ? *d.in_scale : 1.f;   // the producer's gradient lacked this factor
        s1 *= (double)fin; s2 *= (double)fin;
        if (d.dbeta != nullptr) d.dbeta[c] = (float)s1;
        if (d.dgamma != nullptr) d.dgamma[c] = (float)s2;
        d.k1[c] = (float)(s1 / d.count);
        d.k2[c] = (float)(s2 / d.count);
        const float gs = d.gamma[c] * d.invstd[c];
        d.gscale[c] = gs;
        if (d.ca != nullptr) {
            const float k1 = (float)(s1 / d.count), k2 = (float)(s2 / d.count), is = d.invstd[c];
            d.ca[c] = gs * fin;
            d.cb[c] = -gs * k2 * is;
            d.cc[c] = gs * (d.mean[c] * is * k2 - k1);
        }
    }
}


__device__ inline void wgrad_reduce_body(const abc_wgrad_reduce_desc& d, int blk) {
    const int64_t n = (int64_t)d.ntaps * d.Ca * d.Cb;
    const int64_t idx = (int64_t)blk * 256 + threadIdx.x;
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



// The same sums with more bytes in flight (the element-per-thread form keeps ~2.4 MB outstanding and stays latency-bound at
// ~5.2 TB/s on slabs that still sit in the Infinity Cache): a thread owns FOUR consecutive b (16-byte loads) and one quarter
// of the slabs (wave g of the block: slabs [g*ns/4, (g+1)*ns/4)), eight loads deep; the four quarters meet in LDS and are
// added in order g = 0..3, so the result is reproducible (not bit-equal to the scalar form: different association).
__device__ inline void wgrad_reduce_vec_body(const abc_wgrad_reduce_desc& d, int blk) {
    __shared__ float4 sm[3][64];
    const int cb4 = d.Cb >> 2;
    const int64_t n4 = (int64_t)d.ntaps * d.Ca * cb4;
    const int64_t idx = (int64_t)blk * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const bool live = idx < n4;
    const int bi = live ? (int)(idx % cb4) * 4 : 0;
    const int ai = live ? (int)((idx / cb4) % d.Ca) : 0;
    const int t = live ? (int)(idx / ((int64_t)cb4 * d.Ca)) : 0;
    const size_t slab = (size_t)d.Ca_pad * d.Cb_pad;
    const size_t step = (size_t)d.ntaps * slab;
    const int k0 = (int)((int64_t)d.nsplit * g / 4), k1 = (int)((int64_t)d.nsplit * (g + 1) / 4);
    const float* p = d.partial + (size_t)t * slab + (size_t)ai * d.Cb_pad + bi + (size_t)k0 * step;
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int k = k0;
    if (live) {
        for (; k + 8 <= k1; k += 8, p += 8 * step) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const float4*)(p + (size_t)j * step);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                s0.x += v[j].x; s0.y += v[j].y; s0.z += v[j].z; s0.w += v[j].w;
                s1.x += v[j + 1].x; s1.y += v[j + 1].y; s1.z += v[j + 1].z; s1.w += v[j + 1].w;
            }
        }
        for (; k < k1; ++k, p += step) {
            const float4 v = *(const float4*)p;
            s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        }
    }
    s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
    if (g > 0) sm[g - 1][threadIdx.x & 63] = s0;
    __syncthreads();
    if (g > 0 || !live) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float4 v = sm[w][threadIdx.x];
        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
    }
    float* o = d.dw + ((size_t)ai * d.Cb + bi) * d.ntaps + t;
    const float r[4] = {s0.x, s0.y, s0.z, s0.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(size_t)j * d.ntaps] = d.accumulate ? (o[(size_t)j * d.ntaps] + r[j]) : r[j];
}


