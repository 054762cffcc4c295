// Internal device-side helpers shared by the gfx950 kernels of abcnet_amd.
// (The public C-ABI is include/abcnet_hip.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// OCP e4m3fn (gfx950's fp8; max 448, no infinities): the 8-bit activations / weights of the fp8 inference graph
// (img2smiles2.py:42-59 with BatchNorm folded).  Conversions saturate at +-448.
struct f8 {
    uint8_t v;
    f8() = default;
    __device__ inline explicit f8(float x) {
        x = fminf(fmaxf(x, -448.f), 448.f);
        v = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xFF);
    }
    __device__ inline explicit operator float() const { return __builtin_amdgcn_cvt_f32_fp8((int)v, 0); }
};

// Timing ablations (ABC_*_DBG bit masks that skip phases of a kernel: results invalid) and in-kernel phase timestamps exist
// only in a debug build (ABC_KERNEL_DEBUG=1 ./build_hip.sh); the production library compiles them out.
#ifdef ABC_KERNEL_DEBUG
#define ABC_DBG(x) (x)
#define ABC_PROF(p) (p)
#else
#define ABC_DBG(x) 0
#define ABC_PROF(p) ((long long*)nullptr)
#endif

// Experiment switches (kernel / tile choices of measured A/B runs) are read from the environment by the DEBUG build only:
// the production library never looks at the environment, its kernel choice is a function of the descriptor alone.
#ifdef ABC_KERNEL_DEBUG
#include <stdlib.h>
inline const char* abc_knob(const char* name) { return getenv(name); }
#else
inline const char* abc_knob(const char*) { return nullptr; }
#endif

#define ABC_MAX_TAPS 49
#define ABC_WAVE 64

__host__ __device__ inline int abc_roundup(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int abc_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// 16-byte fragment of the compute type: 4 x f32 or 8 x bf16.
template <typename CT> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; static constexpr int NV = 4; };
template <> struct Frag<bf16>  { typedef bf16x8 type; static constexpr int NV = 8; };
template <> struct Frag<f8>    { typedef u32x4 type;  static constexpr int NV = 16; };

// One 16-byte K-slice of a 32x32 MFMA tile: lane-half h of the wave holds the same
// K-slice of A and of B, so any fixed channel permutation inside a chunk is legal.
__device__ inline void mma16B(f32x16& acc, f32x4 a, f32x4 b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}
__device__ inline void mma16B(f32x16& acc, bf16x8 a, bf16x8 b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
// fp8: ONE 32x32x64 MFMA takes a lane's whole 32 bytes of a 64-byte chunk (the two 16-byte halves a0 | a1, b0 | b1: lane half h
// owns the same 32 k-indices of A and of B).  The block-scaled form with unit scales (E8M0 127 = 2^0): it runs at twice the
// bf16 rate per clock, the unscaled 32x32x16 fp8 form only at the bf16 rate (MI355X_MICROARCH.md, matrix cores).
__device__ inline void mma32B_f8(f32x16& acc, i32x8 A, i32x8 B) {
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// a lane's 32 operand bytes as the instruction wants them: eight consecutive registers, filled by two 16-byte loads
__device__ inline i32x8 abc_join32B(u32x4 lo, u32x4 hi) {
    i32x8 r;
    r[0] = (int)lo[0]; r[1] = (int)lo[1]; r[2] = (int)lo[2]; r[3] = (int)lo[3]; r[4] = (int)hi[0]; r[5] = (int)hi[1]; r[6] = (int)hi[2]; r[7] = (int)hi[3];
    return r;
}

// ---------------------------------------------------------------------------
// activation applied on load: y = scale*x + shift ; out = max(y, slope*y)
//   slope 0 -> ReLU, 0.01 -> LeakyReLU, 1 -> identity
__device__ inline float abc_act(float x, float sc, float sh, float sl) {
    float y = fmaf(x, sc, sh);
    return fmaxf(y, sl * y);
}
// the same for NV values in place, two at a time in packed f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: 4 instead of 6
// instructions per pair; bit-identical -- the same fma, product and maximum per element)
template <int NV> __device__ inline void abc_act_n(float* v, const float* sc, const float* sh, const float* sl) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(NV % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < NV; j += 2) {
        const f32x2 y = __builtin_elementwise_fma((f32x2){v[j], v[j + 1]}, (f32x2){sc[j], sc[j + 1]}, (f32x2){sh[j], sh[j + 1]});
        const f32x2 m = (f32x2){sl[j], sl[j + 1]} * y;
        v[j] = fmaxf(y.x, m.x); v[j + 1] = fmaxf(y.y, m.y);
    }
}
// v = ka * v + (kb * w + kc), NV values in place, packed (the BatchNorm-backward correction applied on load: two fmas per element)
template <int NV> __device__ inline void abc_fma2_n(float* v, const float* w, const float* ka, const float* kb, const float* kc) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(NV % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < NV; j += 2) {
        const f32x2 t = __builtin_elementwise_fma((f32x2){kb[j], kb[j + 1]}, (f32x2){w[j], w[j + 1]}, (f32x2){kc[j], kc[j + 1]});
        const f32x2 r = __builtin_elementwise_fma((f32x2){ka[j], ka[j + 1]}, (f32x2){v[j], v[j + 1]}, t);
        v[j] = r.x; v[j + 1] = r.y;
    }
}

// Counter-based dropout keep-decision (K7).  idx = element index in the tensor the
// mask applies to.  Mirrored bit-for-bit by abcnet_amd.dropout.keep_mask (torch int ops)
// so that the oracle can be driven with the identical mask.
__host__ __device__ inline uint32_t abc_drop_hash24(uint32_t idx, uint32_t seed) {
    uint32_t h = idx * 0x9E3779B1u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h >> 8;
}
__host__ __device__ inline bool abc_drop_keep(uint32_t idx, uint32_t seed, float p) {
    return (float)abc_drop_hash24(idx, seed) * (1.0f / 16777216.0f) >= p;
}
// the same decision as an integer compare: k * 2^-24 >= p  <=>  k >= ceil(p * 2^24)  (k < 2^24 and the scaling are exact)
__host__ inline uint32_t abc_drop_threshold(float p) {
    const double t = (double)p * 16777216.0;
    const uint32_t f = (uint32_t)t;
    return (double)f < t ? f + 1u : f;
}

// Compute units left OUT of the persistent grids (abc_set_reserved_cus; process-wide, 0 by default).  Two 256-VGPR workgroups
// per CU fill the register file of every SIMD, so a kernel of another stream -- RCCL's, during the data-parallel exchange --
// cannot start on a CU before one of them ends; a persistent convolution workgroup lives for up to 12 tiles (~330 us).
// A multiple of 4, so that n workgroups per CU stay a multiple of the 8 XCDs.
inline int& abc_reserved_cus_ref() { static int v = 0; return v; }
inline int abc_wg_slots(int per_cu) { return per_cu * (256 - abc_reserved_cus_ref()); }

// XCD-aware, bijective block remap: blocks with equal (bid % 8) share an XCD (and
// its L2) under round-robin placement, so give each such group a contiguous range of
// logical ids.  Placement only affects speed, never results.
__device__ inline int abc_xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// ---------------------------------------------------------------------------
// load NV consecutive channels of InT as floats
// ---------------------------------------------------------------------------
// Packed f32 FMAs with a broadcast operand.  acc += (a.x, a.x) * b / acc += (a.y, a.y) * b as ONE v_pk_fma_f32 whose first
// source is a half of a register pair picked by op_sel (built from (f32pair){u, u} the compiler spends two moves per pair);
// the plain-FMA kernels (one-channel stem, its weight gradient, CBAM's 7x7) keep sliding windows of consecutive pixels in
// such pairs.  Each half is a fused multiply-add: bit-identical to fmaf.
typedef float f32pair __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void pk_fma_lo(f32pair& acc, const f32pair a, const f32pair b) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ inline void pk_fma_hi(f32pair& acc, const f32pair a, const f32pair b) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(b)); }
template <int CTRL> __device__ inline float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
#else
__device__ inline void pk_fma_lo(f32pair& acc, const f32pair a, const f32pair b) { acc += (f32pair){a[0], a[0]} * b; }
__device__ inline void pk_fma_hi(f32pair& acc, const f32pair a, const f32pair b) { acc += (f32pair){a[1], a[1]} * b; }
template <int CTRL> __device__ inline float dpp_mov(float v) { return v; }
#endif
// element e (compile-time) of a window kept as pairs
template <int E> __device__ inline void pk_fma_el(f32pair& acc, const f32pair* win, const f32pair b) {
    if constexpr (E & 1) pk_fma_hi(acc, win[E >> 1], b); else pk_fma_lo(acc, win[E >> 1], b);
}
// sum over the lanes i, i + m, i + 2m, ... of a 16-lane DPP row, in every lane (rotations by 8 .. m: fixed order); m = 1: the whole row
__device__ inline float row_sum16(float v, int m = 1) {
    v += dpp_mov<0x128>(v);
    if (m <= 4) v += dpp_mov<0x124>(v);
    if (m <= 2) v += dpp_mov<0x122>(v);
    if (m <= 1) v += dpp_mov<0x121>(v);
    return v;
}

__device__ inline float row_max16(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
    return v;
}
__device__ inline float row_min16(float v) {
    v = fminf(v, dpp_mov<0x128>(v)); v = fminf(v, dpp_mov<0x124>(v)); v = fminf(v, dpp_mov<0x122>(v)); v = fminf(v, dpp_mov<0x121>(v));
    return v;
}

template <typename InT, int NV> struct LoadVec;
template <> struct LoadVec<float, 4> {
    __device__ static inline void ld(const float* p, float* v) {
        f32x4 t = *(const f32x4*)p;
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
};
template <> struct LoadVec<float, 8> {
    __device__ static inline void ld(const float* p, float* v) {
        f32x4 t = *(const f32x4*)p, u = *(const f32x4*)(p + 4);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        v[4] = u[0]; v[5] = u[1]; v[6] = u[2]; v[7] = u[3];
    }
};
template <> struct LoadVec<bf16, 8> {
    __device__ static inline void ld(const bf16* p, float* v) {
        bf16x8 t = *(const bf16x8*)p;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
    }
};
template <> struct LoadVec<bf16, 4> {
    __device__ static inline void ld(const bf16* p, float* v) {
        bf16x4 t = *(const bf16x4*)p;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)t[j];
    }
};

// vector load when all NV channels exist, scalar tail (zero-filled) otherwise; the
// scalar path also serves tensors whose pixel stride breaks 16-byte alignment (C = 1 image)
template <typename InT, int NV>
__device__ inline void load_n(const InT* p, float* v, int nval) {
    if (nval >= NV) {
        LoadVec<InT, NV>::ld(p, v);
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = (j < nval) ? (float)p[j] : 0.f;
    }
}

template <typename CT> __device__ inline typename Frag<CT>::type pack_frag(const float* v);
template <> __device__ inline f32x4 pack_frag<float>(const float* v) {
    f32x4 r; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; return r;
}
template <> __device__ inline bf16x8 pack_frag<bf16>(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)v[j];
    return r;
}
template <> __device__ inline u32x4 pack_frag<f8>(const float* v) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[4 * j], -448.f), 448.f), fminf(fmaxf(v[4 * j + 1], -448.f), 448.f), 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[4 * j + 2], -448.f), 448.f), fminf(fmaxf(v[4 * j + 3], -448.f), 448.f), w, true);
        r[j] = (unsigned)w;
    }
    return r;
}

// ---------------------------------------------------------------------------
// Source description of an activation tensor that is consumed "through" the
// previous layer's BN + activation (+ 2x2 max-pool, + dropout): fusion boundary
// "producer writes raw conv output + statistics, consumer normalises on load".
struct ActSrc {
    const void* x;        // NHWC raw tensor, physical dims [B, Hx, Wx, ldx]
    const float* scale;   // per channel (absolute channel index in x), or null = identity
    const float* shift;
    const float* slope;
    int Hx, Wx, ldx;
    int pool;             // 1: logical dims are (Hx/2, Wx/2), value = max over 2x2 of act(x)
    float drop_p;         // >0: multiply by keep/(1-p), keep from abc_drop_keep(idx, seed, p)
    uint32_t drop_seed;
    int planar;           // 1: x is f32 channel-planar [B][ctot][Hx][Wx] (NCHW head maps); no pool / dropout
    int ctot;
    const uint32_t* drop_salt;  // device scalar added to drop_seed (per-step counter), or null
};

// Stage a halo tile [HH][HW] pixels x CK channels (channels c0..c0+CK of src) into LDS in
// the compute type, pixel (hy,hx) at byte hy*RS + hx*PS.  Logical input coords of halo
// pixel (0,0) are (iy0, ix0); everything outside [0,Hin)x[0,Win) is zero (conv padding
// applies to the ACTIVATED tensor).
template <typename InT, typename CT, int CK>
__device__ inline void stage_halo(char* sA, int RS, int PS, int HH, int HW, int b, int iy0, int ix0, int Hin, int Win,
                                  const ActSrc& s, int c0, int tid, int nthreads, int cvalid = 1 << 30) {
    constexpr int NV = Frag<CT>::NV;
    constexpr int SEGS = CK / NV;  // 16-byte LDS segments per pixel
    const int part = tid % SEGS;   // nthreads % SEGS == 0 -> constant per thread
    const int cch = c0 + part * NV;
    float sc[NV], sh[NV], sl[NV];
    const int nval = min(NV, cvalid - part * NV);  // channels beyond the tensor's width are zero-filled
    const bool chan_ok = nval > 0;
    const bool has_t = (s.scale != nullptr) && chan_ok;
    if (has_t) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool ok = j < nval;
            sc[j] = ok ? s.scale[cch + j] : 0.f; sh[j] = ok ? s.shift[cch + j] : 0.f; sl[j] = ok ? s.slope[cch + j] : 0.f;
        }
    }
    const InT* xb = (const InT*)s.x;
    const float dscale = (s.drop_p > 0.f) ? 1.0f / (1.0f - s.drop_p) : 1.0f;
    const uint32_t dseed = s.drop_seed + ((s.drop_p > 0.f && s.drop_salt) ? *s.drop_salt : 0u);
    const int total = HH * HW * SEGS;
    for (int sidx = tid; sidx < total; sidx += nthreads) {
        const int pix = sidx / SEGS;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        float v[NV];
        if (chan_ok && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
            if (s.planar) {
                // consecutive threads of a wave = consecutive channel segments of consecutive pixels: per j the
                // wave reads SEGS planes x (64/SEGS) adjacent pixels, i.e. 64..128-byte runs
                const float* xp = (const float*)s.x + ((size_t)(b * s.ctot + cch) * s.Hx + iy) * s.Wx + ix;
                const size_t plane = (size_t)s.Hx * s.Wx;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float t = (j < nval) ? xp[j * plane] : 0.f;
                    v[j] = has_t ? abc_act(t, sc[j], sh[j], sl[j]) : t;
                }
            } else if (!s.pool) {
                const size_t off = ((size_t)(b * s.Hx + iy) * s.Wx + ix) * s.ldx + cch;
                load_n<InT, NV>(xb + off, v, nval);
                if (has_t) {
                    abc_act_n<NV>(v, sc, sh, sl);
                }
                if (s.drop_p > 0.f) {
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        v[j] = abc_drop_keep((uint32_t)(off + j), dseed, s.drop_p) ? v[j] * dscale : 0.f;
                }
            } else {
                float t[NV];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t off = ((size_t)(b * s.Hx + 2 * iy + (q >> 1)) * s.Wx + 2 * ix + (q & 1)) * s.ldx + cch;
                    load_n<InT, NV>(xb + off, t, nval);
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        float y = has_t ? abc_act(t[j], sc[j], sh[j], sl[j]) : t[j];
                        v[j] = (q == 0) ? y : fmaxf(v[j], y);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = 0.f;
        }
        *(typename Frag<CT>::type*)(sA + hy * RS + hx * PS + part * 16) = pack_frag<CT>(v);
    }
}

// ---------------------------------------------------------------------------
// Split-phase halo staging (cdna_hip_programming.md T14: issue the global loads early, transform and
// write LDS late, so HBM/L2 latency hides under the MFMA block in between).  Fast path only:
// NHWC source, no pool, no dropout, all NV channels of every segment present.
template <typename InT, int NV> struct RawVec;
template <> struct RawVec<bf16, 8> {
    bf16x8 v;
    __device__ inline void ld(const bf16* p) { v = *(const bf16x8*)p; }
    __device__ inline void zero() { for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f; }
    __device__ inline void get(float* o) const { for (int j = 0; j < 8; ++j) o[j] = (float)v[j]; }
};
template <> struct RawVec<float, 4> {
    f32x4 v;
    __device__ inline void ld(const float* p) { v = *(const f32x4*)p; }
    __device__ inline void zero() { v[0] = v[1] = v[2] = v[3] = 0.f; }
    __device__ inline void get(float* o) const { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
};
template <> struct RawVec<float, 8> {
    f32x4 a, b;
    __device__ inline void ld(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    __device__ inline void zero() { a[0] = a[1] = a[2] = a[3] = 0.f; b = a; }
    __device__ inline void get(float* o) const {
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    }
};

template <typename InT, typename CT, int CK, int NMAX, int NTHR>
struct HaloPrefetch {
    static constexpr int NV = Frag<CT>::NV;
    static constexpr int SEGS = CK / NV;
    RawVec<InT, NV> raw[NMAX];
    unsigned inb;  // bit i: item i is inside the image

    // issue the loads of chunk starting at absolute channel c0
    __device__ inline void issue(int HH, int HW, int b, int iy0, int ix0, int Hin, int Win, const ActSrc& s, int c0, int tid,
                                 int cvalid = 1 << 30) {
        const int part = tid % SEGS;
        const int cch = c0 + part * NV;
        const int total = (part * NV < cvalid) ? HH * HW * SEGS : 0;  // segments past the tensor's width stay zero
        const InT* xb = (const InT*)s.x;
        inb = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            if (sidx < total) {
                const int pix = sidx / SEGS;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
                    raw[i].ld(xb + ((size_t)(b * s.Hx + iy) * s.Wx + ix) * s.ldx + cch);
                    inb |= 1u << i;
                }
            }
        }
    }
    // transform + write to LDS.  lcoef = LDS table [3][cstride] of (scale, shift, slope) indexed by the channel
    // RELATIVE to the conv's first input channel (crel0 = first channel of this chunk), or null = identity
    __device__ inline void commit(char* sA, int RS, int PS, int HH, int HW, const float* lcoef, int cstride, int crel0, int tid,
                                  int cvalid = 1 << 30) {
        const int part = tid % SEGS;
        const int cch = crel0 + part * NV;
        const int total = HH * HW * SEGS;
        float sc[NV], sh[NV], sl[NV];
        const bool has_t = lcoef != nullptr && (part * NV < cvalid);
        if (has_t) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { sc[j] = lcoef[cch + j]; sh[j] = lcoef[cstride + cch + j]; sl[j] = lcoef[2 * cstride + cch + j]; }
        }
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            if (sidx < total) {
                const int pix = sidx / SEGS;
                const int hy = pix / HW, hx = pix - hy * HW;
                float v[NV];
                if (inb & (1u << i)) {
                    raw[i].get(v);
                    if (has_t) {
                        abc_act_n<NV>(v, sc, sh, sl);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NV; ++j) v[j] = 0.f;
                }
                *(typename Frag<CT>::type*)(sA + hy * RS + hx * PS + part * 16) = pack_frag<CT>(v);
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Split-phase halo staging, register-lean form (used where the accumulators own the register file).
//  * loads are raw buffer loads (resource in SGPRs, 32-bit byte offset per lane): halo pixels outside the image
//    get an out-of-range offset and the hardware returns zeros -> no divergent branches around the loads;
//  * nothing per-thread survives between patches except the raw data itself: the (row, column) of every segment is
//    recomputed from a laundered thread id with a multiply-shift division (magic = 65536 / HW + 1, exact for
//    pix * HW < 65536), and the BatchNorm coefficients are re-read from the LDS table at commit time.  If such
//    loop invariants are hoisted instead, they spill, and a scratch reload between two loads waits on vmcnt(0)
//    -> every load becomes a serial round trip (measured: 44 of 119 us on the 128x128 3x3 weight gradient).
// Requires: tensor bytes < 2^31 (checked on the host), NHWC, no pool / dropout, whole 16-byte segments.
__device__ inline __amdgpu_buffer_rsrc_t abc_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
// two values to two LDS places in the output type; bf16: ONE packed conversion, the halves written by ds_write_b16 / _d16_hi
template <typename OutT> __device__ inline void abc_put2(char* p0, char* p1, float a, float b) { *(OutT*)p0 = (OutT)a; *(OutT*)p1 = (OutT)b; }
template <> __device__ inline void abc_put2<bf16>(char* p0, char* p1, float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 pk = __builtin_convertvector((f32x2){a, b}, bf16x2);
    *(bf16*)p0 = pk[0]; *(bf16*)p1 = pk[1];
}
template <> __device__ inline void abc_put2<f8>(char* p0, char* p1, float a, float b) {   // (saturating, as f8(float))
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    *(uint8_t*)p0 = (uint8_t)(pk & 0xFF); *(uint8_t*)p1 = (uint8_t)((pk >> 8) & 0xFF);
}
__device__ inline int abc_launder(int x) { asm volatile("" : "+v"(x)); return x; }

template <typename InT, int NV> struct RawBuf;
template <> struct RawBuf<bf16, 8> {
    u32x4 v;
    __device__ inline void ld(__amdgpu_buffer_rsrc_t r, unsigned off) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    __device__ inline void ld2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0); }
    __device__ inline void get(float* o) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(v[j] << 16); o[2 * j + 1] = __uint_as_float(v[j] & 0xFFFF0000u); }
    }
};
template <> struct RawBuf<f8, 16> {
    u32x4 v;
    __device__ inline void ld(__amdgpu_buffer_rsrc_t r, unsigned off) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    __device__ inline void ld2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0); }
    __device__ inline void get(float* o) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[4 * j] = __builtin_amdgcn_cvt_f32_fp8((int)v[j], 0); o[4 * j + 1] = __builtin_amdgcn_cvt_f32_fp8((int)v[j], 1);
            o[4 * j + 2] = __builtin_amdgcn_cvt_f32_fp8((int)v[j], 2); o[4 * j + 3] = __builtin_amdgcn_cvt_f32_fp8((int)v[j], 3);
        }
    }
};
template <> struct RawBuf<float, 4> {
    u32x4 v;
    __device__ inline void ld(__amdgpu_buffer_rsrc_t r, unsigned off) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    __device__ inline void ld2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0); }
    __device__ inline void get(float* o) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(v[j]);
    }
};
template <> struct RawBuf<float, 8> {
    u32x4 a, b;
    __device__ inline void ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
        a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);
    }
    __device__ inline void ld2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        a = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0);
        b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, soff, 0);
    }
    __device__ inline void get(float* o) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = __uint_as_float(a[j]); o[4 + j] = __uint_as_float(b[j]); }
    }
};

struct HaloGeom {      // uniform description of one operand's patch
    int HH, HW, magic;  // halo rows / columns, 65536 / HW + 1
    int Hin, Win;       // logical image (bounds of the zero padding)
    int Hx, Wx, ldx;    // physical tensor
};

template <typename InT, typename CT, int CK, int NMAX, int NTHR>
struct HaloFetch {
    static constexpr int NV = Frag<CT>::NV;
    static constexpr int SEGS = CK / NV;
    RawBuf<InT, NV> raw[NMAX];
    unsigned inb;  // bit i: segment i lies inside the image
    // (cvalid = channels that exist from the tile's first one; 1 << 30 = all)
    __device__ static inline int live_segs(int cvalid) {
        const int n = (cvalid + NV - 1) / NV;
        return (n >= SEGS || (n & (n - 1))) ? SEGS : n;
    }
    // live_segs() is a power of two (SEGS is, and fewer live segments are only used when their count is): every `/ live`
    // and `% live` below is a shift / mask.  As runtime integer divisions (~20 VALU instructions each, per segment, per patch)
    // the address arithmetic of one weight-gradient patch cost 2.4 us per workgroup -- more than its 72 MFMAs per wave.
    static_assert((SEGS & (SEGS - 1)) == 0, "segments per pixel must be a power of two");
    __device__ static inline int live_shift(int live) { return 31 - __builtin_clz((unsigned)live); }

    // Two-phase form of issue(): prepare() computes every segment's byte offset (the address arithmetic, integer divisions
    // included), fire() is NMAX buffer loads and nothing else.  A caller with several fetchers prepares ALL of them, then
    // fires all of them: in the one-phase form hipcc, at ~250 VGPRs, computes the next offset in registers that an
    // already-issued load is going to write and has to put `s_waitcnt vmcnt(N)` -- a full memory round trip -- between
    // the loads of one batch (measured on the weight-gradient kernel: loads and MFMAs did not overlap at all).
    unsigned voff[NMAX];
    __device__ inline void prepare(const HaloGeom& g, int b, int iy0, int ix0, int c0, int tid_, int cvalid) {
        const int tid = abc_launder(tid_);
        const int live = live_segs(cvalid), lsh = live_shift(live);
        const int part = tid & (live - 1);
        const int total = g.HH * g.HW * live;
        const int base = ((b * g.Hx + iy0) * g.Wx + ix0) * g.ldx + c0 + part * NV;
        inb = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            const int pix = sidx >> lsh;
            const int hy = (pix * g.magic) >> 16, hx = pix - hy * g.HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = sidx < total && part * NV < cvalid && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
            voff[i] = ok ? (unsigned)(base + (hy * g.Wx + hx) * g.ldx) * (unsigned)sizeof(InT) : 0x80000000u;
            inb |= ok ? (1u << i) : 0u;
        }
    }
    // prepare() for a patch WITHOUT halo (HW == 16 columns) that lies wholly inside the image: the thread's segments are the same
    // pixel column 2^k rows apart, so their offsets are one base + i x a wave-uniform step -- ~20 instead of ~60 vector
    // instructions per operand and patch (the weight gradient's P operand: g and y_raw)
    __device__ inline void prepare_regular(const HaloGeom& g, int b, int iy0, int ix0, int c0, int tid_, int cvalid) {
        const int tid = abc_launder(tid_);
        const int live = live_segs(cvalid), lsh = live_shift(live);
        const int part = tid & (live - 1);
        const int total = g.HH * 16 * live;
        const int pix0 = tid >> lsh, hy0 = pix0 >> 4, hx0 = pix0 & 15;
        const unsigned v0 = (unsigned)(((b * g.Hx + iy0 + hy0) * g.Wx + ix0 + hx0) * g.ldx + c0 + part * NV) * (unsigned)sizeof(InT);
        const unsigned step = (unsigned)(((NTHR >> lsh) >> 4) * g.Wx * g.ldx) * (unsigned)sizeof(InT);   // (wave-uniform)
        const bool chan = part * NV < cvalid;
        inb = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const bool ok = chan && (tid + i * NTHR) < total;
            voff[i] = ok ? v0 + (unsigned)i * step : 0x80000000u;
            inb |= ok ? (1u << i) : 0u;
        }
    }
    // ---- segment TABLE: the patch-invariant geometry of this thread's segments computed ONCE per kernel (halo patch of whole
    // patches, halo narrower than a patch).  Entry i (LDS, [i][thread]) = (byte offset relative to the patch's first halo pixel) / 16
    // << 12 | (LDS destination) / 16; `mask` (one register) = valid | top << 5 | bottom << 10 | left << 15 | right << 20, bit i of a
    // group set when segment i exists / falls outside the image for a patch on that border.  Per patch what is left is one scalar
    // base, one AND with a scalar border selector and an add + select per segment (~25 vector instructions instead of ~120).
    __device__ inline unsigned table_setup(unsigned* tab, const HaloGeom& g, int RS, int PS, int tid, int cvalid, int prows, int dy_min, int dx_min) {
        const int live = live_segs(cvalid), lsh = live_shift(live);
        const int part = tid & (live - 1);
        const int total = g.HH * g.HW * live;
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            const int pix = sidx >> lsh;
            const int hy = (pix * g.magic) >> 16, hx = pix - hy * g.HW;
            const bool valid = sidx < total && part * NV < cvalid;
            const unsigned rel = (unsigned)((hy * g.Wx + hx) * g.ldx + part * NV) * (unsigned)sizeof(InT);
            const unsigned dst = (unsigned)(hy * RS + hx * PS + part * 16);
            tab[i * NTHR + tid] = valid ? ((rel >> 4) << 12) | (dst >> 4) : 0u;
            mask |= (valid ? 1u : 0u) << i;
            mask |= (hy < -dy_min ? 1u : 0u) << (5 + i);
            mask |= (hy >= prows - dy_min ? 1u : 0u) << (10 + i);
            mask |= (hx < -dx_min ? 1u : 0u) << (15 + i);
            mask |= (hx >= 16 - dx_min ? 1u : 0u) << (20 + i);
        }
        return mask;
    }
    // sbase = byte offset of the patch's first halo pixel + first channel (may wrap below zero for a border patch: the segments that
    // would use it are masked); border = the scalar selector of the patch's borders (0x1F << 5 top, << 10 bottom, << 15 left, << 20 right)
    __device__ inline void prepare_tab(const unsigned* tab, unsigned mask, unsigned sbase, unsigned border, int tid) {
        unsigned o = mask & border;
        o = (o >> 5) | (o >> 10) | (o >> 15) | (o >> 20);
        inb = mask & ~o & 0x1Fu;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const unsigned e = tab[i * NTHR + tid];
            voff[i] = ((inb >> i) & 1u) ? sbase + ((e >> 12) << 4) : 0x80000000u;
        }
    }
    __device__ inline void commit_tab(char* sA, const unsigned* tab, unsigned mask, const float* lcoef, int cstride, int tid, int cvalid) {
        const int live = live_segs(cvalid);
        const int cch = (tid & (live - 1)) * NV;
        float sc[NV], sh[NV], sl[NV];
        const bool has_t = lcoef != nullptr;
        if (has_t) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { sc[j] = lcoef[cch + j]; sh[j] = lcoef[cstride + cch + j]; sl[j] = lcoef[2 * cstride + cch + j]; }
        }
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            if ((mask >> i) & 1u) {
                char* dst = sA + ((tab[i * NTHR + tid] & 0xFFFu) << 4);
                if constexpr (sizeof(InT) == sizeof(CT)) {
                    if (!has_t) { *(u32x4*)dst = raw[i].v; continue; }  // plain copy (zeros outside the image)
                }
                float v[NV];
                raw[i].get(v);  // zeros when outside the image
                if (has_t && (inb & (1u << i))) {
                    abc_act_n<NV>(v, sc, sh, sl);
                }
                *(typename Frag<CT>::type*)dst = pack_frag<CT>(v);
            }
        }
    }
    // the same segments of a SECOND tensor with the same pixel stride (the BatchNorm-fused weight gradient reads g and y_raw
    // of one layer side by side): every offset differs from the other fetcher's by one constant
    template <typename Other>
    __device__ inline void prepare_like(const Other& o, unsigned delta_bytes) {
        inb = o.inb;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) voff[i] = (o.voff[i] >> 31) ? 0x80000000u : o.voff[i] + delta_bytes;
    }
    __device__ inline void fire(__amdgpu_buffer_rsrc_t rs) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) raw[i].ld(rs, voff[i]);
    }
    // the loads whose index (counted from `first` across a caller's fetchers) is congruent to `phase` modulo `nphase`:
    // a batch spread over the K-steps of the MFMA block (all arguments compile-time constants after unrolling)
    __device__ inline void fire_slice(__amdgpu_buffer_rsrc_t rs, int first, int phase, int nphase) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
            if ((first + i) % nphase == phase) raw[i].ld(rs, voff[i]);
    }
    __device__ inline void prepare_none() {
        inb = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) voff[i] = 0x80000000u;   // out of range: the loads return zeros, no memory traffic
    }

    __device__ inline void issue(__amdgpu_buffer_rsrc_t rs, const HaloGeom& g, int b, int iy0, int ix0, int c0, int tid_, int cvalid) {
        const int tid = abc_launder(tid_);
        // live segments per pixel: with fewer valid channels than the tile is wide (and a power-of-two count) the threads
        // are spread over the live segments only -- the padding is zeroed once by the caller and never touched
        const int live = live_segs(cvalid), lsh = live_shift(live);
        const int part = tid & (live - 1);
        const int total = g.HH * g.HW * live;
        const int base = ((b * g.Hx + iy0) * g.Wx + ix0) * g.ldx + c0 + part * NV;
        inb = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            const int pix = sidx >> lsh;
            const int hy = (pix * g.magic) >> 16, hx = pix - hy * g.HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = sidx < total && part * NV < cvalid && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
            const unsigned off = ok ? (unsigned)(base + (hy * g.Wx + hx) * g.ldx) * (unsigned)sizeof(InT) : 0x80000000u;
            raw[i].ld(rs, off);
            inb |= ok ? (1u << i) : 0u;
        }
    }
    // lcoef = LDS table [3][cstride] of (scale, shift, slope), index = channel relative to the workgroup's first one
    __device__ inline void commit(char* sA, int RS, int PS, const HaloGeom& g, const float* lcoef, int cstride, int tid_, int cvalid) {
        const int tid = abc_launder(tid_);
        const int live = live_segs(cvalid), lsh = live_shift(live);
        const int part = tid & (live - 1);
        const int cch = part * NV;
        const int total = g.HH * g.HW * live;
        float sc[NV], sh[NV], sl[NV];
        const bool has_t = lcoef != nullptr;
        if (has_t) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { sc[j] = lcoef[cch + j]; sh[j] = lcoef[cstride + cch + j]; sl[j] = lcoef[2 * cstride + cch + j]; }
        }
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            if (sidx < total) {
                const int pix = sidx >> lsh;
                const int hy = (pix * g.magic) >> 16, hx = pix - hy * g.HW;
                char* dst = sA + hy * RS + hx * PS + part * 16;
                if constexpr (sizeof(InT) == sizeof(CT)) {
                    if (!has_t) { *(u32x4*)dst = raw[i].v; continue; }  // plain copy (zeros outside the image)
                }
                float v[NV];
                raw[i].get(v);  // zeros when outside the image
                if (has_t && (inb & (1u << i))) {
                    abc_act_n<NV>(v, sc, sh, sl);
                }
                *(typename Frag<CT>::type*)dst = pack_frag<CT>(v);
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Halo staging with the per-segment geometry computed ONCE per tile (the kernels that use it are bound by
// instruction issue, not by registers): voff = byte offset of the segment in chunk 0 (out-of-range marker when the
// pixel lies outside the image -> the buffer load returns zeros), dst = LDS byte offset.  Per chunk the issue is
// NMAX buffer loads with the chunk's byte offset in the scalar operand, nothing else.
// NSET register sets for the raw data (issue<S> / commit<S>): with two, the chunk after next can be in flight while the next one waits to
// be committed (the small tiles of the deep levels, whose chunk of matrix work is shorter than a trip to memory).
template <typename InT, typename CT, int CK, int NMAX, int NTHR, int NSET = 1>
struct HaloTile {
    static constexpr int NV = Frag<CT>::NV;
    static constexpr int SEGS = CK / NV;
    RawBuf<InT, NV> raw[NSET][NMAX];
    unsigned voff[NMAX];
    int dst[NMAX];  // < 0: this thread has no such segment

    __device__ inline void setup(const HaloGeom& g, int RS, int PS, int b, int iy0, int ix0, int c0, int tid) {
        const int part = tid % SEGS;
        const int total = g.HH * g.HW * SEGS;
        const int base = ((b * g.Hx + iy0) * g.Wx + ix0) * g.ldx + c0 + part * NV;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            const int sidx = tid + i * NTHR;
            const int pix = sidx / SEGS;
            const int hy = (pix * g.magic) >> 16, hx = pix - hy * g.HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool have = sidx < total;
            const bool ok = have & (iy >= 0) & (iy < g.Hin) & (ix >= 0) & (ix < g.Win);
            voff[i] = ok ? (unsigned)(base + (hy * g.Wx + hx) * g.ldx) * (unsigned)sizeof(InT) : 0x80000000u;
            dst[i] = have ? hy * RS + hx * PS + part * 16 : -1;
        }
    }
    // coff = byte offset of the chunk's first channel relative to chunk 0
    template <int S = 0> __device__ inline void issue(__amdgpu_buffer_rsrc_t rs, unsigned coff) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) raw[S][i].ld2(rs, voff[i], coff);
    }
    // lcoef = LDS table [3][cstride] of (scale, shift, slope) at the chunk's first channel, or null
    template <int S = 0> __device__ inline void commit(char* sA, const float* lcoef, int cstride, int tid) {
        const int cch = (tid % SEGS) * NV;
        float sc[NV], sh[NV], sl[NV];
        if (lcoef != nullptr) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { sc[j] = lcoef[cch + j]; sh[j] = lcoef[cstride + cch + j]; sl[j] = lcoef[2 * cstride + cch + j]; }
        }
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            if (dst[i] >= 0) {
                if constexpr (sizeof(InT) == sizeof(CT)) {
                    if (lcoef == nullptr) { *(u32x4*)(sA + dst[i]) = raw[S][i].v; continue; }
                }
                float v[NV];
                raw[S][i].get(v);
                if (lcoef != nullptr && !(voff[i] >> 31)) {
                    abc_act_n<NV>(v, sc, sh, sl);
                }
                *(typename Frag<CT>::type*)(sA + dst[i]) = pack_frag<CT>(v);
            }
        }
    }
};
