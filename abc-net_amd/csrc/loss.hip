// Fused activation + 8-term loss + d(loss)/d(logits) in one pass over the logits
// and the targets (reference: src/train.py:95-137, K9 of SURVEY.md).
//
// One thread per quarter-resolution pixel.  Logits, dlogits and targets are all NCHW planes
// (the reference's own interface layout), so adjacent lanes = adjacent pixels and every load
// and store of the kernel is a coalesced 256-byte wave access.  rho/omega targets are float64
// as in the reference (utils.py:91-92).
// Normalisers are global sums, so the kernel writes the gradient of each term's
// NUMERATOR; abc_loss_finalize turns the partial sums into the 8 terms, the
// uncertainty-weighted total (train.py:127-137), ds, and a per-channel factor
// weight_i/denominator_i which the heads' backward applies on load.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "loss_math.hpp"

namespace {

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Four threads per pixel: wave `sub` of the workgroup takes omega bins [15 sub, 15 sub + 15) of the workgroup's 64 pixels
// (lane = pixel, so every plane access stays a coalesced 256-byte row) and a share of the small heads.  One thread per
// pixel walking all 60 bins left 9 waves per CU with a 60-deep dependent loop: 247 us for 660 MB.
constexpr int LPX = 64;   // pixels per workgroup

__global__ __launch_bounds__(256) void loss_kernel(const abc_loss_desc d) {
    __shared__ double sm[4][16];
    __shared__ double wsum[4][LPX];
    const int hw = d.h * d.w;
    const int64_t npix = (int64_t)d.B * hw;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * LPX + lane;
    const bool live = p < npix;
    const int b = live ? (int)(p / hw) : 0, yx = live ? (int)(p % hw) : 0;
    double num[8], den[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { num[i] = 0.0; den[i] = 0.0; }
    // plane (b, c) of a head with C channels: base + (b*C + c)*hw + yx
#define PL(ptr, C, c) (ptr)[((size_t)b * (C) + (c)) * hw + yx]
    const int o_lo = 15 * sub, o_hi = o_lo + 15;
    // ---- omega per-pixel weight = sum over the 60 bins of the omega target (train.py:124): partial per wave, then all
    double wpart = 0.0;
    if (live)
        for (int o = o_lo; o < o_hi; ++o) wpart += PL(d.t_omega, 60, o);
    wsum[sub][lane] = wpart;
    __syncthreads();
    const double wpix = (wsum[0][lane] + wsum[1][lane]) + (wsum[2][lane] + wsum[3][lane]);
    if (live) {
        if (sub == 0) {
            // ---- head 0: atom centre, head 4: bond centre
            {
                const float t = PL(d.t_atom, 1, 0);
                float dz;
                num[0] = center_focal(PL(d.logits[0], 1, 0), t, 1.f, &dz);
                den[0] = (t == 1.f) ? 1.0 : 0.0;
                PL(d.dlogits[0], 1, 0) = dz;
            }
            {
                const float t = PL(d.t_bond, 1, 0);
                float dz;
                num[4] = center_focal(PL(d.logits[4], 1, 0), t, 1.f, &dz);
                den[4] = (t == 1.f) ? 1.0 : 0.0;
                PL(d.dlogits[4], 1, 0) = dz;
            }
            den[7] = wpix;
        } else if (sub == 1) {
            // ---- head 1: atom types (14-way softmax, class weights)
            float z[14], t[14], dz[14], dn = 0.f;
#pragma unroll
            for (int k = 0; k < 14; ++k) { z[k] = PL(d.logits[1], 14, k); t[k] = PL(d.t_types, 14, k); }
            num[1] = class_focal<14>(z, t, c_type_w, dz, &dn);
            den[1] = dn;
#pragma unroll
            for (int k = 0; k < 14; ++k) PL(d.dlogits[1], 14, k) = dz[k];
        } else if (sub == 2) {
            // ---- head 2: charges (3-way), head 3: hydrogens (2-way)
            {
                float z[3], t[3], dz[3], dn = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) { z[k] = PL(d.logits[2], 3, k); t[k] = PL(d.t_charges, 3, k); }
                num[2] = class_focal<3>(z, t, nullptr, dz, &dn);
                den[2] = dn;
#pragma unroll
                for (int k = 0; k < 3; ++k) PL(d.dlogits[2], 3, k) = dz[k];
            }
            {
                float z[2], t[2], dz[2], dn = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) { z[k] = PL(d.logits[3], 2, k); t[k] = PL(d.t_hs, 2, k); }
                num[3] = class_focal<2>(z, t, nullptr, dz, &dn);
                den[3] = dn;
#pragma unroll
                for (int k = 0; k < 2; ++k) PL(d.dlogits[3], 2, k) = dz[k];
            }
        }
        // ---- heads 5,6,7 for this wave's omega bins; bond-type channel = type*60 + omega (train.py:101 view)
        double n5 = 0.0, n6 = 0.0, n7 = 0.0, d5 = 0.0;
        for (int o = o_lo; o < o_hi; ++o) {
            float z[6], t[6], dz[6], dn = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) { z[k] = PL(d.logits[5], 360, k * 60 + o); t[k] = PL(d.t_btypes, 360, k * 60 + o); }
            n5 += class_focal<6>(z, t, nullptr, dz, &dn);
            d5 += dn;
#pragma unroll
            for (int k = 0; k < 6; ++k) PL(d.dlogits[5], 360, k * 60 + o) = dz[k];
            // rho: |abs(pred) - rho| * sum_types(t)   (train.py:105,121), f64 like the reference
            {
                const float zr = PL(d.logits[6], 60, o);
                const double tr = PL(d.t_rho, 60, o);
                const double diff = (double)fabsf(zr) - tr;
                n6 += fabs(diff) * (double)dn;
                const float sg = (diff > 0.0) ? 1.f : ((diff < 0.0) ? -1.f : 0.f);
                const float sz = (zr > 0.f) ? 1.f : ((zr < 0.f) ? -1.f : 0.f);
                PL(d.dlogits[6], 60, o) = sg * sz * dn;
            }
            // omega: focal per bin weighted by wpix (train.py:124-125)
            {
                const float to = (float)PL(d.t_omega, 60, o);
                float dz7;
                n7 += center_focal(PL(d.logits[7], 60, o), to, (float)wpix, &dz7);
                PL(d.dlogits[7], 60, o) = dz7;
            }
        }
        num[5] = n5; den[5] = d5; num[6] = n6; den[6] = d5; num[7] = n7;
    }
#undef PL
    // ---- block reduction of the 16 sums
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double a = wave_sum(num[i]), bsum = wave_sum(den[i]);
        if (lane == 0) { sm[sub][i] = a; sm[sub][8 + i] = bsum; }
    }
    __syncthreads();
    if (threadIdx.x < 16)
        d.partial[(size_t)blockIdx.x * 16 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ __launch_bounds__(1024) void loss_finalize_kernel(const abc_loss_fin_desc d) {
    __shared__ double tot[16];
    __shared__ double red[64][16];
    __shared__ float cscale[8];
    const int t = threadIdx.x;
    {
        // 64 lanes per sum (a single 64-thread block walking 576 partial blocks cost 39 us of dependent loads), fixed
        // order -> reproducible
        const int which = t & 15, part = t >> 4;
        double s = 0.0;
        for (int k = part; k < d.nblk; k += 64) s += d.partial[(size_t)k * 16 + which];
        red[part][which] = s;
        __syncthreads();
        if (t < 16) {
            double a = 0.0;
            for (int p = 0; p < 64; ++p) a += red[p][t];
            tot[t] = a;
        }
    }
    __syncthreads();
    if (t == 0) {
        // head -> index into s and the factor in front of exp(-s)   (train.py:127-135)
        const int sidx[8] = {0, 2, 3, 9, 1, 4, 6, 7};
        const double fac[8] = {1, 1, 1, 1, 1, 1, 0.5, 1};
        double total = 0.0;
        for (int i = 0; i < 10; ++i) d.ds[i] = 0.f;
        for (int i = 0; i < 8; ++i) {
            const double den = tot[8 + i] + (i == 3 ? 0.1 : 0.0);  // atom_hs: +0.1 (train.py:114)
            const double term = tot[i] / den;
            const double sv = (double)d.s[sidx[i]];
            const double wgt = fac[i] * exp(-sv) + sv;
            d.out[1 + i] = wgt * term;
            d.out[9 + i] = term;
            total += wgt * term;
            d.ds[sidx[i]] = (float)(term * (1.0 - fac[i] * exp(-sv)) * (double)d.grad_scale);
            cscale[i] = (float)(wgt / den * (double)d.grad_scale);
        }
        d.out[0] = total;
    }
    __syncthreads();
    for (int c = t; c < d.nchan; c += 1024) {
        float v = 0.f;
        for (int i = 0; i < 8; ++i)
            if (c >= d.chan_off[i] && c < d.chan_off[i] + d.head_c[i]) v = cscale[i];
        d.chan_scale[c] = v;
    }
}

}  // namespace

extern "C" int abc_loss_blocks(const abc_loss_desc* d) { return abc_cdiv(d->B * d->h * d->w, LPX); }

extern "C" int abc_loss_fwd_bwd(const abc_loss_desc* d, abc_stream_t stream) {
    if (d->B < 1 || d->h < 1 || d->w < 1) return abc_fail(ABC_EINVAL, "loss: empty");
    hipLaunchKernelGGL(loss_kernel, dim3(abc_loss_blocks(d)), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("loss_fwd_bwd");
}

extern "C" int abc_loss_finalize(const abc_loss_fin_desc* d, abc_stream_t stream) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *d);
    return abc_check_launch("loss_finalize");
}
