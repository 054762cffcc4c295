// Per-element loss terms of train.py:95-125 with their derivatives, shared by the stand-alone loss kernel (loss.hip) and the
// fused heads kernel (heads_fused.hip): the two must agree bit for bit on every term.
#pragma once
#include "common.hpp"

namespace {

constexpr float LO = 1e-5f, HI = 1.f - 1e-5f;

// FAST (the fused bf16 heads kernel only): hardware exp2 / log2 / rcp (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp each, the
// exponent scaling adds |x| * 2^-24 relative) instead of the correctly rounded library routines, which cost 20-40
// instructions apiece -- the results are rounded to bf16 there anyway.  The stand-alone loss kernel keeps the exact forms.
template <bool FAST> __device__ inline float lm_exp(float x) { if constexpr (FAST) return __expf(x); else return expf(x); }
template <bool FAST> __device__ inline float lm_log(float x) { if constexpr (FAST) return __logf(x); else return logf(x); }
template <bool FAST> __device__ inline float lm_div(float a, float b) { if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b); else return a / b; }

template <bool FAST = false> __device__ inline float sigm(float z) { return lm_div<FAST>(1.f, 1.f + lm_exp<FAST>(-z)); }

// penalty-reduced focal on one sigmoid channel (train.py:107-108): returns the loss
// value, writes dL/dz.  `w` multiplies both (omega's per-pixel weight, train.py:124).
template <bool FAST = false>
__device__ inline float center_focal(float z, float t, float w, float* dz) {
    const float ps = sigm<FAST>(z);
    const bool inside = (ps >= LO) && (ps <= HI);
    const float p = fminf(fmaxf(ps, LO), HI);
    const float q = 1.f - p;
    const float lp = lm_log<FAST>(p), lq = lm_log<FAST>(q);
    const float pos = (t == 1.f) ? 1.f : 0.f;
    const float neg = (1.f - t) * (1.f - t); const float neg4 = neg * neg;
    const float loss = -pos * q * q * lp - neg4 * p * p * lq;
    // dL/dp
    const float dLp = -pos * (-2.f * q * lp + lm_div<FAST>(q * q, p)) - neg4 * (2.f * p * lq - lm_div<FAST>(p * p, q));
    *dz = inside ? w * dLp * ps * (1.f - ps) : 0.f;
    return w * loss;
}

// focal cross-entropy over a K-way softmax (train.py:109,111,114,119); z/t/dz are
// register arrays.  Returns the numerator contribution, adds sum(t) to *den.
template <int K, bool FAST = false>
__device__ inline float class_focal(const float* z, const float* t, const float* wk, float* dz, float* den) {
    float m = z[0];
#pragma unroll
    for (int k = 1; k < K; ++k) m = fmaxf(m, z[k]);
    float e[K], se = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { e[k] = lm_exp<FAST>(z[k] - m); se += e[k]; }
    const float inv = lm_div<FAST>(1.f, se);
    float loss = 0.f, dot = 0.f, a[K], q[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        q[k] = e[k] * inv;
        a[k] = 0.f;
        *den += t[k];
        if (t[k] != 0.f) {
            const bool inside = (q[k] >= LO) && (q[k] <= HI);
            const float p = fminf(fmaxf(q[k], LO), HI);
            const float om = 1.f - p;
            const float lp = lm_log<FAST>(p);
            const float w = wk ? wk[k] : 1.f;
            loss += -w * t[k] * om * om * lp;
            if (inside) a[k] = -w * t[k] * (-2.f * om * lp + lm_div<FAST>(om * om, p));
        }
        dot += a[k] * q[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dz[k] = q[k] * (a[k] - dot);
    return loss;
}

__constant__ float c_type_w[14] = {1.f, 0.1f, 0.1f, 0.1f, 1.f, 1.f, 1.f, 1.f, 1.f, 10.f, 10.f, 10.f, 10.f, 10.f};  // train.py:16

}  // namespace
