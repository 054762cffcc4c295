// Per-element loss terms of train.py:95-125 with their derivatives, shared by the stand-alone loss kernel (loss.hip) and the
// fused heads kernel (heads_fused.hip): the two must agree bit for bit on every term.
#pragma once
#include "common.hpp"

namespace {

constexpr float LO = 1e-5f, HI = 1.f - 1e-5f;

__device__ inline float sigm(float z) { return 1.f / (1.f + expf(-z)); }

// penalty-reduced focal on one sigmoid channel (train.py:107-108): returns the loss
// value, writes dL/dz.  `w` multiplies both (omega's per-pixel weight, train.py:124).
__device__ inline float center_focal(float z, float t, float w, float* dz) {
    const float ps = sigm(z);
    const bool inside = (ps >= LO) && (ps <= HI);
    const float p = fminf(fmaxf(ps, LO), HI);
    const float q = 1.f - p;
    const float lp = logf(p), lq = logf(q);
    const float pos = (t == 1.f) ? 1.f : 0.f;
    const float neg = (1.f - t) * (1.f - t); const float neg4 = neg * neg;
    const float loss = -pos * q * q * lp - neg4 * p * p * lq;
    // dL/dp
    const float dLp = -pos * (-2.f * q * lp + q * q / p) - neg4 * (2.f * p * lq - p * p / q);
    *dz = inside ? w * dLp * ps * (1.f - ps) : 0.f;
    return w * loss;
}

// focal cross-entropy over a K-way softmax (train.py:109,111,114,119); z/t/dz are
// register arrays.  Returns the numerator contribution, adds sum(t) to *den.
template <int K>
__device__ inline float class_focal(const float* z, const float* t, const float* wk, float* dz, float* den) {
    float m = z[0];
#pragma unroll
    for (int k = 1; k < K; ++k) m = fmaxf(m, z[k]);
    float e[K], se = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { e[k] = expf(z[k] - m); se += e[k]; }
    const float inv = 1.f / se;
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
            const float lp = logf(p);
            const float w = wk ? wk[k] : 1.f;
            loss += -w * t[k] * om * om * lp;
            if (inside) a[k] = -w * t[k] * (-2.f * om * lp + om * om / p);
        }
        dot += a[k] * q[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dz[k] = q[k] * (a[k] - dot);
    return loss;
}

__constant__ float c_type_w[14] = {1.f, 0.1f, 0.1f, 0.1f, 1.f, 1.f, 1.f, 1.f, 1.f, 10.f, 10.f, 10.f, 10.f, 10.f};  // train.py:16

}  // namespace
