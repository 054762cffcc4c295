// The 17 training meters of the reference loop (src/train.py:145-215) on the device.
//
// Every meter there is AverageMeter.update(num / den, den) (src/meter.py:12-16), i.e. sum += num, count += den, and
// every update costs the reference two host round trips (.cpu().detach().numpy()): 34 synchronisations per step.
// Here one call accumulates all (num, den) pairs into a device-resident table that the host reads whenever it wants
// to print (train.py:219 does so every 100 steps).
//
//   pass 1  peaks    : 3x3 local maxima of the activated atom / bond centre maps above 0.25 (train.py:145-151)
//   pass 2  sums     : one thread per quarter-resolution pixel walks the 501 logit and target planes (all NCHW:
//                      adjacent lanes = adjacent pixels, coalesced) and produces the 24 distinct sums the 17 meters
//                      share; the circular 3-tap omega maxima (train.py:189-214) are 60-bit masks per pixel
//   pass 3  finalize : fixed-order reduction of the per-workgroup partials, (num, den) of this batch and the running
//                      totals
// Activations (sigmoid / softmax, clamped to [1e-5, 1-1e-5], train.py:95-105) are recomputed from the logits exactly
// as the loss kernel does; arg-max of a softmax is taken on the logits (first index on ties, as torch.argmax).
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

constexpr float LO = 1e-5f, HI = 1.f - 1e-5f;
constexpr int NSUM = 24;

__device__ inline float act_sig(float z) { return fminf(fmaxf(1.f / (1.f + expf(-z)), LO), HI); }

__global__ __launch_bounds__(256) void metrics_peaks_kernel(const abc_metrics_desc d) {
    const int hw = d.h * d.w;
    const int64_t npix = (int64_t)d.B * hw;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int b = (int)(p / hw), yx = (int)(p % hw);
    const int y = yx / d.w, x = yx % d.w;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float* L = d.logits[which ? 4 : 0] + (size_t)b * hw;
        const float v = act_sig(L[yx]);
        float m = v;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < d.h && xx >= 0 && xx < d.w) m = fmaxf(m, act_sig(L[yy * d.w + xx]));
            }
        d.peaks[(size_t)which * npix + p] = (m == v && v > 0.25f) ? 1 : 0;
    }
}

// (sum_c t) * [argmax t == argmax z], sum_c t      (train.py:165-172, 184-185)
template <int K>
__device__ inline void class_acc(const float* z, const float* t, double* num, double* den) {
    int at = 0, az = 0;
    float st = t[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
        st += t[k];
        if (t[k] > t[at]) at = k;
        if (z[k] > z[az]) az = k;
    }
    *den += (double)st;
    if (at == az) *num += (double)st;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ inline unsigned long long circ3(unsigned long long m) {  // 60-bit circular dilation by one bin either way
    const unsigned long long M60 = (1ull << 60) - 1;
    return (m | ((m << 1) & M60) | (m >> 59) | (m >> 1) | ((m & 1ull) << 59)) & M60;
}

__global__ __launch_bounds__(256) void metrics_sums_kernel(const abc_metrics_desc d) {
    __shared__ double sm[4][NSUM];
    const int hw = d.h * d.w;
    const int64_t npix = (int64_t)d.B * hw;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s[NSUM];
#pragma unroll
    for (int i = 0; i < NSUM; ++i) s[i] = 0.0;
    if (p < npix) {
        const int b = (int)(p / hw), yx = (int)(p % hw);
        const int y = yx / d.w, x = yx % d.w;
#define PL(ptr, C, c) (ptr)[((size_t)b * (C) + (c)) * hw + yx]
        // ---- centre maps: precision / precision3 / recall / recall3 (train.py:153-163, 174-182)
        bool Tb = false;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float* T = (which ? d.t_bond : d.t_atom) + (size_t)b * hw;
            const unsigned char* P = d.peaks + (size_t)which * npix + (size_t)b * hw;
            const bool t = T[yx] == 1.f, pk = P[yx] != 0;
            bool t3 = false, p3 = false;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < d.h && xx >= 0 && xx < d.w) {
                        t3 |= T[yy * d.w + xx] == 1.f;
                        p3 |= P[yy * d.w + xx] != 0;
                    }
                }
            double* o = s + which * 5;
            o[0] = (pk && t) ? 1.0 : 0.0;
            o[1] = (pk && t3) ? 1.0 : 0.0;
            o[2] = pk ? 1.0 : 0.0;
            o[3] = (t && p3) ? 1.0 : 0.0;
            o[4] = t ? 1.0 : 0.0;
            if (which) Tb = t;
        }
        // ---- class accuracies
        {
            float z[14], t[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) { z[k] = PL(d.logits[1], 14, k); t[k] = PL(d.t_types, 14, k); }
            class_acc<14>(z, t, &s[10], &s[11]);
        }
        {
            float z[3], t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { z[k] = PL(d.logits[2], 3, k); t[k] = PL(d.t_charges, 3, k); }
            class_acc<3>(z, t, &s[12], &s[13]);
        }
        {
            float z[2], t[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { z[k] = PL(d.logits[3], 2, k); t[k] = PL(d.t_hs, 2, k); }
            class_acc<2>(z, t, &s[14], &s[15]);
        }
        // ---- per omega bin: bond types (6-way, channel = type*60 + bin), rho MAE, omega peak / target masks
        unsigned long long temp = 0, tom = 0;
        const float p_last = act_sig(PL(d.logits[7], 60, 59)), p_first = act_sig(PL(d.logits[7], 60, 0));
        float prev = p_last, cur = p_first;
        for (int o = 0; o < 60; ++o) {
            float z[6], t[6];
            float st = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) { z[k] = PL(d.logits[5], 360, k * 60 + o); t[k] = PL(d.t_btypes, 360, k * 60 + o); st += t[k]; }
            class_acc<6>(z, t, &s[16], &s[17]);
            s[18] += fabs((double)fabsf(PL(d.logits[6], 60, o)) - PL(d.t_rho, 60, o)) * (double)st;
            const float nxt = (o + 1 < 60) ? act_sig(PL(d.logits[7], 60, o + 1)) : p_first;
            if (Tb && fmaxf(cur, fmaxf(prev, nxt)) == cur && cur > 0.25f) temp |= 1ull << o;
            if (PL(d.t_omega, 60, o) == 1.0) tom |= 1ull << o;
            prev = cur; cur = nxt;
        }
#undef PL
        s[19] = (double)__popcll(tom & temp);
        s[20] = (double)__popcll(temp);
        s[21] = (double)__popcll(tom & circ3(temp));
        s[22] = (double)__popcll(tom);
        s[23] = (double)__popcll(circ3(tom) & temp);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NSUM; ++i) {
        const double a = wave_sum(s[i]);
        if (lane == 0) sm[wave][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < NSUM)
        d.partial[(size_t)blockIdx.x * NSUM + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__global__ __launch_bounds__(1024) void metrics_finalize_kernel(const abc_metrics_desc d, int nblk) {
    __shared__ double red[32][NSUM];
    __shared__ double tot[NSUM];
    const int t = threadIdx.x;
    const int which = t % 32, part = t / 32;   // 32 lanes per sum slot (24 used), 32 parts, fixed order -> reproducible
    if (which < NSUM) {
        double a = 0.0;
        for (int k = part; k < nblk; k += 32) a += d.partial[(size_t)k * NSUM + which];
        red[part][which] = a;
    }
    __syncthreads();
    if (t < NSUM) {
        double a = 0.0;
        for (int q = 0; q < 32; ++q) a += red[q][t];
        tot[t] = a;
    }
    __syncthreads();
    if (t < 17) {
        // meter -> (numerator slot, denominator slot); order = METER_NAMES of oracle/metrics_oracle.py
        const int ni[17] = {0, 1, 0, 3, 10, 12, 14, 5, 6, 5, 8, 16, 18, 19, 21, 19, 23};
        const int di[17] = {2, 2, 4, 4, 11, 13, 15, 7, 7, 9, 9, 17, 17, 20, 22, 22, 20};
        const double num = tot[ni[t]];
        const double den = tot[di[t]] + (t == 6 ? 0.01 : 0.0);   // atom_hs: 0.01 + sum (train.py:171-172)
        d.last[2 * t] = num; d.last[2 * t + 1] = den;
        d.totals[2 * t] += num; d.totals[2 * t + 1] += den;
    }
}

}  // namespace

extern "C" int abc_metrics_blocks(const abc_metrics_desc* d) { return abc_cdiv(d->B * d->h * d->w, 256); }

extern "C" int abc_metrics_update(const abc_metrics_desc* d, abc_stream_t stream) {
    if (d->B < 1 || d->h < 1 || d->w < 1) return abc_fail(ABC_EINVAL, "metrics: empty");
    if (!d->peaks || !d->partial || !d->totals || !d->last) return abc_fail(ABC_EINVAL, "metrics: null workspace");
    const int nb = abc_metrics_blocks(d);
    hipLaunchKernelGGL(metrics_peaks_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
    hipLaunchKernelGGL(metrics_sums_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d);
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *d, nb);
    return abc_check_launch("metrics_update");
}
