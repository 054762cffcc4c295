// Weight repacking, fused Adam, inference NMS, and the C-ABI error plumbing.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include <string.h>

// ------------------------------------------------------------------ error text
static thread_local char g_err[256] = "";
int abc_fail(int code, const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}
int abc_allow_lds(const void* fn, int bytes, unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < 64 && ((*done >> dev) & 1ull)) return ABC_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) on device %d: %s", bytes, dev, hipGetErrorString(e));
        return ABC_ELAUNCH;
    }
    if (dev >= 0 && dev < 64) *done |= 1ull << dev;
    return ABC_OK;
}
int abc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return ABC_ELAUNCH;
    }
    return ABC_OK;
}
extern "C" const char* abc_last_error(void) { return g_err; }
extern "C" int abc_version(void) { return 101; }
extern "C" int abc_set_reserved_cus(int n) {
    if (n < 0 || n > 128) return abc_fail(ABC_EINVAL, "abc_set_reserved_cus: 0 <= n <= 128");
    abc_reserved_cus_ref() = (n + 3) & ~3;
    return ABC_OK;
}
extern "C" int abc_get_reserved_cus(void) { return abc_reserved_cus_ref(); }

namespace {

// ------------------------------------------------------------------ weight packing
// element offset of (slice = tap * chunks + chunk, row, k) in a packed weight of `rows` rows per slice
__host__ __device__ inline size_t abc_pack_offset(size_t slice, int rows, int row, int CK, int k, int layout) {
    if (layout == 1) {   // 64-byte chunk rows (bf16 CK = 32, fp8 CK = 64): [row / 32][kk][h][row % 32][16 bytes] inside the slice (see abc_pack_desc.layout)
        // elements per 16 bytes: 8 (bf16, CK = 32) or 16 (fp8, CK = 64) -- shifts, not divisions: this runs once per packed
        // element of every weight of the step
        const int sh = CK == 64 ? 4 : 3;
        const int h = k >> (sh + 1), kk = (k >> sh) & 1, e = k & ((1 << sh) - 1);
        return (slice * rows + (size_t)(row & ~31)) * CK + (size_t)((((kk * 2 + h) * 32 + (row & 31)) << sh) + e);
    }
    return (slice * rows + row) * CK + k;
}

// dst[t][chunk][row][k], element type CT; see abc_pack_desc in the public header.
template <typename CT>
__global__ void pack_kernel(const abc_pack_desc d, int CK, int ntaps, int nchunks) {
    // nchunks = chunks of THIS weight; it lands at chunk offset red_off/CK of a dst with red_total/CK chunks
    const int nch_total = (d.red_total + CK - 1) / CK, ch_off = d.red_off / CK;
    const int64_t total = (int64_t)ntaps * nchunks * d.rows_pad * CK;
    CT* dst = (CT*)d.dst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % CK);
        int64_t r = i / CK;
        const int n = (int)(r % d.rows_pad); r /= d.rows_pad;
        const int c = (int)(r % nchunks);
        const int t = (int)(r / nchunks);
        const int rc = c * CK + k;
        float v = 0.f;
        if (d.mode == 0) {  // Conv2d forward: row = cout, reduce over cin
            if (n < d.Cout && rc < d.Cin) v = d.w[((size_t)n * d.Cin + rc) * ntaps + t];
        } else if (d.mode == 1) {  // Conv2d dgrad: row = cin, reduce over cout (host mirrors the tap offsets)
            if (n < d.Cin && rc < d.Cout) v = d.w[((size_t)rc * d.Cin + n) * ntaps + t];
        } else if (d.mode == 2) {  // ConvTranspose2d forward, output parity (py,px): row = cout, reduce over cin
            const int nx = d.px ? 2 : 1;
            const int iy = t / nx, ix = t % nx;
            const int ky = d.py ? (iy == 0 ? 0 : 2) : 1;
            const int kx = d.px ? (ix == 0 ? 0 : 2) : 1;
            if (n < d.Cout && rc < d.Cin) v = d.w[(((size_t)rc * d.Cout + n) * 3 + ky) * 3 + kx];
        } else {  // ConvTranspose2d dgrad: row = cin, reduce over cout, 9 taps
            if (n < d.Cin && rc < d.Cout) v = d.w[((size_t)n * d.Cout + rc) * 9 + t];
        }
        if (d.row_scale != nullptr && (d.mode == 0 || d.mode == 2) && n < d.Cout) v *= d.row_scale[n];
        const int rtot = d.rows_total > 0 ? d.rows_total : d.rows_pad;
        dst[abc_pack_offset((size_t)t * nch_total + ch_off + c, rtot, d.rows_off + n, CK, k, d.layout)] = CT(v);
    }
}

// batched form: one launch over a device-resident table of descriptors (the plan packs ~100 small weights per step)
struct PackItem { abc_pack_desc d; int32_t CK, ntaps, nchunks, dtype; int64_t first; int32_t ntiles, pad_; };
// (ntiles > 0: the item is packed by pack_tiles_kernel -- 32-row x 32-channel source tiles through LDS -- and skipped here)
__device__ inline void pack_one(const PackItem& it, unsigned r) {
    const abc_pack_desc& d = it.d;
    const unsigned CK = it.CK, ntaps = it.ntaps, nchunks = it.nchunks, rows = d.rows_pad;
    const unsigned k = r % CK; r /= CK;
    const unsigned n = r % rows; r /= rows;
    const unsigned c = r % nchunks;
    const unsigned t = r / nchunks;
    const unsigned rc = c * CK + k;
    float v = 0.f;
    if (d.mode == 0) {
        if (n < (unsigned)d.Cout && rc < (unsigned)d.Cin) v = d.w[((size_t)n * d.Cin + rc) * ntaps + t];
    } else if (d.mode == 1) {
        if (n < (unsigned)d.Cin && rc < (unsigned)d.Cout) v = d.w[((size_t)rc * d.Cin + n) * ntaps + t];
    } else if (d.mode == 2) {
        const unsigned nx = d.px ? 2 : 1;
        const unsigned iy = t / nx, ix = t % nx;
        const unsigned ky = d.py ? (iy == 0 ? 0 : 2) : 1;
        const unsigned kx = d.px ? (ix == 0 ? 0 : 2) : 1;
        if (n < (unsigned)d.Cout && rc < (unsigned)d.Cin) v = d.w[(((size_t)rc * d.Cout + n) * 3 + ky) * 3 + kx];
    } else {
        if (n < (unsigned)d.Cin && rc < (unsigned)d.Cout) v = d.w[((size_t)n * d.Cout + rc) * 9 + t];
    }
    if (d.row_scale != nullptr && (d.mode == 0 || d.mode == 2) && n < (unsigned)d.Cout) v *= d.row_scale[n];
    const unsigned nch_total = (d.red_total + CK - 1) / CK, ch_off = d.red_off / CK;
    const unsigned rtot = d.rows_total > 0 ? (unsigned)d.rows_total : rows;
    const size_t o = abc_pack_offset((size_t)t * nch_total + ch_off + c, (int)rtot, (int)((unsigned)d.rows_off + n), (int)CK, (int)k, d.layout);
    if (it.dtype == ABC_BF16) ((bf16*)d.dst)[o] = (bf16)v; else if (it.dtype == ABC_FP8) ((f8*)d.dst)[o] = f8(v); else ((float*)d.dst)[o] = v;
}

// A block owns 2048 consecutive destination elements of the concatenated items.  The item lookup (binary search over
// ~100 entries) is done once per block by thread 0; blocks inside one item -- nearly all -- then run with the item
// in scalar registers and 32-bit index arithmetic; the rare block that straddles items searches per element.
constexpr int PACK_CHUNK = 2048;
__device__ inline int pack_find(const PackItem* items, int nitems, int64_t i) {
    int lo = 0, hi = nitems - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackItem* items, int nitems, int64_t total) {
    __shared__ int s_lo, s_hi;
    const int64_t i0 = (int64_t)blockIdx.x * PACK_CHUNK;
    const int64_t i1 = (i0 + PACK_CHUNK < total) ? i0 + PACK_CHUNK : total;
    if (threadIdx.x == 0) { s_lo = pack_find(items, nitems, i0); s_hi = pack_find(items, nitems, i1 - 1); }
    __syncthreads();
    // (made provably wave-uniform, and the item copied BY VALUE: since the e4m3 packing the kernel contains a one-byte store, which
    //  may alias anything -- through a reference the item's fields were re-read with vector loads after every store, and the
    //  step's packing went 75 -> 139 us)
    const int lo = __builtin_amdgcn_readfirstlane(s_lo), hi = __builtin_amdgcn_readfirstlane(s_hi);
    if (lo == hi) {
        const PackItem it = items[lo];
        if (it.ntiles > 0) return;      // (pack_tiles_kernel's)
        const unsigned base = (unsigned)(i0 - it.first);
        for (unsigned e = threadIdx.x; e < (unsigned)(i1 - i0); e += 256) pack_one(it, base + e);
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const int j = pack_find(items, nitems, i);
            if (items[j].ntiles > 0) continue;
            pack_one(items[j], (unsigned)(i - items[j].first));
        }
    }
}

// The dest-major kernel above gathers every packed element from the [row][channel][tap] source with a stride of `ntaps` floats (and
// three runtime divisions per element): 138 MB at 1.7 TB/s, 81 us per step.  The bulk of the weights -- Conv2d 3x3 / 1x1 with
// channel counts that are multiples of 32, bf16, forward (mode 0) and data-gradient (mode 1) packing -- goes source-major instead:
// a workgroup reads ONE tile of 32 rows x 32 reduction channels x all taps as 32 contiguous runs of 32 * ntaps floats (16-byte
// loads), transposes it through LDS (row stride 32 * ntaps + 1 words: conflict-free both ways) and writes the ntaps destination
// blocks of 32 x 32 elements with 16-byte stores (8 consecutive k: contiguous in both layouts).  Output bit-identical.
constexpr int PT_MAXT = 9;
constexpr int PT_MAXITEMS = 1023;
__global__ __launch_bounds__(256) void pack_tiles_kernel(const PackItem* items, int nitems) {
    __shared__ float tile[32 * (32 * PT_MAXT + 1)];
    __shared__ int s_nt[PT_MAXITEMS + 1];      // exclusive prefix sums of the items' tile counts (one parallel read of the table)
    __shared__ int s_item, s_tile;
    const int tid = threadIdx.x;
    for (int j = tid; j < nitems; j += 256) s_nt[j + 1] = items[j].ntiles;
    __syncthreads();
    if (tid == 0) {
        s_nt[0] = 0;
        for (int j = 0; j < nitems; ++j) s_nt[j + 1] += s_nt[j];
    }
    __syncthreads();
    const int total = s_nt[nitems];
    for (int tg = blockIdx.x; tg < total; tg += gridDim.x) {
        __syncthreads();      // (the previous tile's readers are done with `tile` and s_item)
        if (tid == 0) {
            int lo = 0, hi = nitems - 1;      // last item whose first tile is <= tg and that has tiles
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_nt[mid] <= tg) lo = mid; else hi = mid - 1; }
            s_item = lo; s_tile = tg - s_nt[lo];
        }
        __syncthreads();
        const PackItem it = items[__builtin_amdgcn_readfirstlane(s_item)];
        const abc_pack_desc& d = it.d;
        const int tl = __builtin_amdgcn_readfirstlane(s_tile);
        const int ntaps = it.ntaps, nchunks = it.nchunks;
        const int c = tl % nchunks, rb = tl / nchunks;
        const int n0 = rb * 32, k0 = c * 32;
        const int S = 32 * ntaps + 1;
        const int rows_real = d.mode == 0 ? d.Cout : d.Cin;
        const bool real = n0 < rows_real;      // (whole 32-row blocks are real or padding: channel counts are multiples of 32)
        if (real) {
            // run j (32 of them, 8 threads each): mode 0 = output row n0 + j, its channels k0 .. k0 + 31 x taps;
            //                                     mode 1 = reduction channel (cout) k0 + j, its input channels n0 .. n0 + 31 x taps
            if ((((uintptr_t)d.w) & 15) == 0) {
                // (all loads of a thread issued before the first LDS write: the tile is one memory round trip, not nine)
                const int j = tid >> 3, q0 = tid & 7;
                const float* src = d.mode == 0 ? d.w + ((size_t)(n0 + j) * d.Cin + k0) * ntaps : d.w + ((size_t)(k0 + j) * d.Cin + n0) * ntaps;
                f32x4 v[PT_MAXT];
#pragma unroll
                for (int m = 0; m < PT_MAXT; ++m) {
                    if (m < ntaps) v[m] = *(const f32x4*)(src + 4 * (q0 + 8 * m));
                }
#pragma unroll
                for (int m = 0; m < PT_MAXT; ++m) {
                    if (m < ntaps) {
                        float* o = tile + j * S + 4 * (q0 + 8 * m);
                        o[0] = v[m][0]; o[1] = v[m][1]; o[2] = v[m][2]; o[3] = v[m][3];
                    }
                }
            } else {
                // (a weight is a view at ANY element offset of the flat parameter arena: 4-byte loads, a wave per run -- 256
                //  consecutive bytes per instruction whatever the alignment)
                const int wv = tid >> 6, ln = tid & 63;
                constexpr int NQ = (32 * PT_MAXT + 63) / 64;      // 5
                float v[8][NQ];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = wv * 8 + jj;
                    const float* src = d.mode == 0 ? d.w + ((size_t)(n0 + j) * d.Cin + k0) * ntaps : d.w + ((size_t)(k0 + j) * d.Cin + n0) * ntaps;
#pragma unroll
                    for (int m = 0; m < NQ; ++m) {
                        const int q = ln + 64 * m;
                        v[jj][m] = q < 32 * ntaps ? src[q] : 0.f;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                    for (int m = 0; m < NQ; ++m) {
                        const int q = ln + 64 * m;
                        if (q < 32 * ntaps) tile[(wv * 8 + jj) * S + q] = v[jj][m];
                    }
                }
            }
        }
        __syncthreads();
        const int nch_total = (d.red_total + 31) / 32, ch_off = d.red_off / 32;
        const int rtot = d.rows_total > 0 ? d.rows_total : d.rows_pad;
        const int r = tid & 31, g = (tid >> 5) & 3, th = tid >> 7;      // row, group of 8 reduction channels, tap parity
        const float rs = (real && d.row_scale != nullptr && d.mode == 0) ? d.row_scale[n0 + r] : 1.f;
        for (int t = th; t < ntaps; t += 2) {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = 0.f;
                if (real) v = d.mode == 0 ? tile[r * S + (8 * g + i) * ntaps + t] : tile[(8 * g + i) * S + r * ntaps + t];
                if (d.row_scale != nullptr && d.mode == 0) v *= rs;
                o[i] = (bf16)v;
            }
            const size_t off = abc_pack_offset((size_t)t * nch_total + ch_off + c, rtot, d.rows_off + n0 + r, 32, 8 * g, d.layout);
            *(bf16x8*)((bf16*)d.dst + off) = o;
        }
    }
}

// ------------------------------------------------------------------ Adam
__global__ void step_inc_kernel(int64_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void adam_kernel(const abc_adam_desc d) {
    const double t = (double)*d.step;
    const float bc1 = (float)(1.0 - pow((double)d.beta1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)d.beta2, t));
    const float step_size = d.lr / bc1;
    const int64_t n4 = d.n / 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 p = ((f32x4*)d.p)[i], g = ((const f32x4*)d.g)[i], m = ((f32x4*)d.m)[i], v = ((f32x4*)d.v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = g[j] * d.grad_scale + d.weight_decay * p[j];
            m[j] = m[j] + (gg - m[j]) * (1.f - d.beta1);
            v[j] = v[j] * d.beta2 + (1.f - d.beta2) * gg * gg;
            p[j] -= step_size * m[j] / (sqrtf(v[j]) / bc2s + d.eps);
        }
        ((f32x4*)d.p)[i] = p; ((f32x4*)d.m)[i] = m; ((f32x4*)d.v)[i] = v;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < d.n; i += 256) {
            const float gg = d.g[i] * d.grad_scale + d.weight_decay * d.p[i];
            const float m = d.m[i] + (gg - d.m[i]) * (1.f - d.beta1);
            const float v = d.v[i] * d.beta2 + (1.f - d.beta2) * gg * gg;
            d.m[i] = m; d.v[i] = v;
            d.p[i] -= step_size * m / (sqrtf(v) / bc2s + d.eps);
        }
    }
}

// ------------------------------------------------------------------ NMS (img2smiles2.py:61-79)
__global__ __launch_bounds__(256) void nms_kernel(const abc_nms_desc d) {
    const int hw = d.h * d.w;
    const int64_t npix = (int64_t)d.B * hw;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int b = (int)(p / hw), yx = (int)(p % hw);
    const int y = yx / d.w, x = yx % d.w;
    // 3x3 local maximum with -inf padding, logit > -1
    for (int which = 0; which < 2; ++which) {
        const float* L = (which ? d.bond : d.atom) + (size_t)b * hw;
        const float v = L[yx];
        float m = v;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < d.h && xx >= 0 && xx < d.w) m = fmaxf(m, L[yy * d.w + xx]);
            }
        float* o = which ? d.bond_mask : d.atom_mask;
        o[(size_t)b * hw + yx] = (m == v && v > -1.f) ? 1.f : 0.f;
    }
    const int n = d.n_omega;
    if (n <= 0) return;      // (the channel-axis outputs come from the heads' kernel itself: abc_conv_desc.head_aux)
    const float* R = d.rho + (size_t)b * n * hw + yx;
    const float* O = d.omega + (size_t)b * n * hw + yx;
    float prev = O[(size_t)(n - 1) * hw], cur = O[0];
    const float first = cur;
    for (int k = 0; k < n; ++k) {
        const float nxt = (k + 1 < n) ? O[(size_t)(k + 1) * hw] : first;
        d.rho_abs[((size_t)b * n + k) * hw + yx] = fabsf(R[(size_t)k * hw]);
        const float m = fmaxf(cur, fmaxf(prev, nxt));
        d.omega_mask[((size_t)b * n + k) * hw + yx] = (m == cur && cur > -1.f) ? 1.f : 0.f;
        prev = cur; cur = nxt;
    }
}

// per-channel sum over batch and pixels of planar [B][C][HW] f32: grid (C, NCH) partials, then a tiny reduce
constexpr int PS_CHUNKS = 64;
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* x, int B, int C, int HW, float* work) {
    __shared__ double sm[4];
    const int c = blockIdx.x, ch = blockIdx.y;
    const int64_t n = (int64_t)B * HW;
    const int64_t per = (n + PS_CHUNKS - 1) / PS_CHUNKS;
    const int64_t lo = ch * per, hi = (lo + per < n) ? lo + per : n;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (int64_t)b * HW);
        s += (double)x[((size_t)b * C + c) * HW + p];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) work[(size_t)c * PS_CHUNKS + ch] = (float)(sm[0] + sm[1] + sm[2] + sm[3]);
}
__global__ void plane_sum_reduce_kernel(const float* work, int C, const float* cs, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < PS_CHUNKS; ++k) s += (double)work[(size_t)c * PS_CHUNKS + k];
    out[c] = (float)(s * (double)(cs ? cs[c] : 1.f));
}

// ------------------------------------------------------------------ fp8 inference graph: calibration + weight scales
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* x, int64_t n, float* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf((float)x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // non-negative floats order like their bit patterns: an integer atomic max is exact and order-independent
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned*)out, __float_as_uint(m));
}
template <typename T>
__global__ __launch_bounds__(256) void absmax_cols_kernel(const T* x, int64_t npix, int ld, int c_off, int C, float* out) {
    const int ncv = C / 8;
    const int64_t nitems = npix * ncv;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nitems; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / ncv;
        const int c = (int)(i % ncv) * 8;
        float v[8];
        LoadVec<T, 8>::ld(x + p * ld + c_off + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned*)out, __float_as_uint(m));
}
__global__ void fp8_act_scale_kernel(const float* amax, float margin, float* s_out, float* inv_s_out) {
    const float s = fmaxf(*amax, 1e-12f) * margin / 448.f;
    *s_out = s; *inv_s_out = 1.f / s;
}
// one 64-lane wave per output row
__global__ __launch_bounds__(256) void fp8_weight_scales_kernel(const float* w, int rows, int K, const float* fold, const float* s_in,
                                                               float* qmul, float* deq) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= rows) return;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(w[(size_t)n * K + k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) {
        const float f = fold ? fold[n] : 1.f;
        const float amax = m * fabsf(f);
        const float sw = amax > 0.f ? amax / 448.f : 1.f;
        qmul[n] = f / sw;
        deq[n] = sw * *s_in;
    }
}

}  // namespace

extern "C" int abc_absmax(const void* x, int32_t dtype, int64_t n, float* out, abc_stream_t stream) {
    if (n < 1) return abc_fail(ABC_EINVAL, "absmax: empty");
    int64_t nb = (n + 2047) / 2048;
    if (nb > 2048) nb = 2048;
    if (dtype == ABC_BF16) hipLaunchKernelGGL(absmax_kernel<bf16>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, n, out);
    else if (dtype == ABC_F32) hipLaunchKernelGGL(absmax_kernel<float>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, out);
    else return abc_fail(ABC_EUNSUPPORTED, "absmax: f32 or bf16");
    return abc_check_launch("absmax");
}
extern "C" int abc_absmax_cols(const void* x, int32_t dtype, int64_t npix, int32_t ld, int32_t c_off, int32_t C, float* out, abc_stream_t stream) {
    if (npix < 1 || C < 8 || (C % 8) || (ld % 8) || (c_off % 8)) return abc_fail(ABC_EINVAL, "absmax_cols: shape / alignment");
    int64_t nb = (npix * (C / 8) + 2047) / 2048;
    if (nb > 2048) nb = 2048;
    if (dtype == ABC_BF16) hipLaunchKernelGGL(absmax_cols_kernel<bf16>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, npix, ld, c_off, C, out);
    else if (dtype == ABC_F32) hipLaunchKernelGGL(absmax_cols_kernel<float>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, npix, ld, c_off, C, out);
    else return abc_fail(ABC_EUNSUPPORTED, "absmax_cols: f32 or bf16");
    return abc_check_launch("absmax_cols");
}
extern "C" int abc_fp8_act_scale(const float* amax, float margin, float* s_out, float* inv_s_out, abc_stream_t stream) {
    if (!(margin > 0.f)) return abc_fail(ABC_EINVAL, "fp8_act_scale: margin");
    hipLaunchKernelGGL(fp8_act_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax, margin, s_out, inv_s_out);
    return abc_check_launch("fp8_act_scale");
}
extern "C" int abc_fp8_weight_scales(const float* w, int32_t rows, int32_t K, const float* fold, const float* s_in, float* qmul, float* deq,
                                     abc_stream_t stream) {
    if (rows < 1 || K < 1) return abc_fail(ABC_EINVAL, "fp8_weight_scales: shape");
    hipLaunchKernelGGL(fp8_weight_scales_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, rows, K, fold, s_in, qmul, deq);
    return abc_check_launch("fp8_weight_scales");
}

extern "C" int abc_pack_conv_weights(const abc_pack_desc* d, abc_stream_t stream) {
    int ntaps, red;
    switch (d->mode) {
        case 0: ntaps = d->kh * d->kw; red = d->Cin; break;
        case 1: ntaps = d->kh * d->kw; red = d->Cout; break;
        case 2: ntaps = (d->py ? 2 : 1) * (d->px ? 2 : 1); red = d->Cin; break;
        case 3: ntaps = 9; red = d->Cout; break;
        default: return abc_fail(ABC_EINVAL, "pack: mode");
    }
    const int CK = d->ck;
    if (CK != abc_conv_chunk(d->dtype_c, d->red_total)) return abc_fail(ABC_EINVAL, "pack: ck does not match red_total");
    if (d->layout != 0 && (d->layout != 1 || CK * abc_dsize(d->dtype_c) != 64 || (d->rows_total > 0 ? d->rows_total : d->rows_pad) % 32 || d->rows_off % 32))
        return abc_fail(ABC_EINVAL, "pack: layout 1 needs 64-byte chunks (bf16 CK = 32 / fp8 CK = 64), whole 32-row blocks");
    if (d->red_pad % CK || d->red_pad < red || d->red_off % CK || d->red_off + d->red_pad > abc_roundup(d->red_total, CK))
        return abc_fail(ABC_EINVAL, "pack: red_pad/red_off");
    const int nchunks = d->red_pad / CK;
    const int64_t total = (int64_t)ntaps * nchunks * d->rows_pad * CK;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (d->dtype_c == ABC_BF16) hipLaunchKernelGGL(pack_kernel<bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d, CK, ntaps, nchunks);
    else if (d->dtype_c == ABC_FP8) hipLaunchKernelGGL(pack_kernel<f8>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d, CK, ntaps, nchunks);
    else hipLaunchKernelGGL(pack_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d, CK, ntaps, nchunks);
    return abc_check_launch("pack_conv_weights");
}

static int pack_shape(const abc_pack_desc* d, int* ntaps, int* red) {
    switch (d->mode) {
        case 0: *ntaps = d->kh * d->kw; *red = d->Cin; return 0;
        case 1: *ntaps = d->kh * d->kw; *red = d->Cout; return 0;
        case 2: *ntaps = (d->py ? 2 : 1) * (d->px ? 2 : 1); *red = d->Cin; return 0;
        case 3: *ntaps = 9; *red = d->Cout; return 0;
        default: return -1;
    }
}

extern "C" int abc_pack_item_bytes(void) { return (int)sizeof(PackItem); }

// host: fill one table entry (plain host memory, later copied to the device by the caller); returns the
// number of destination elements this entry covers, or <0
extern "C" int64_t abc_pack_item_fill(void* item, const abc_pack_desc* d, int64_t first) {
    int ntaps, red;
    if (pack_shape(d, &ntaps, &red)) { abc_fail(ABC_EINVAL, "pack: mode"); return -1; }
    const int CK = d->ck;
    if (CK != abc_conv_chunk(d->dtype_c, d->red_total) || d->red_pad % CK || d->red_pad < red || d->red_off % CK ||
        d->red_off + d->red_pad > abc_roundup(d->red_total, CK)) { abc_fail(ABC_EINVAL, "pack: red_pad/red_off/ck"); return -1; }
    if (d->layout != 0 && (d->layout != 1 || CK * abc_dsize(d->dtype_c) != 64 || (d->rows_total > 0 ? d->rows_total : d->rows_pad) % 32 || d->rows_off % 32)) {
        abc_fail(ABC_EINVAL, "pack: layout 1 needs 64-byte chunks (bf16 CK = 32 / fp8 CK = 64), whole 32-row blocks"); return -1;
    }
    PackItem* it = (PackItem*)item;
    it->d = *d; it->CK = CK; it->ntaps = ntaps; it->nchunks = d->red_pad / CK; it->dtype = d->dtype_c; it->first = first;
    // source-major tiles (pack_tiles_kernel): Conv2d forward / data-gradient packing in bf16, whole 32-channel blocks, <= 9 taps
    const bool tiles = (d->mode == 0 || d->mode == 1) && d->dtype_c == ABC_BF16 && CK == 32 && ntaps <= PT_MAXT && d->Cout % 32 == 0 &&
                       d->Cin % 32 == 0 && d->rows_pad % 32 == 0 && d->red_pad == red && ((uintptr_t)d->dst & 15) == 0 &&
                       d->rows_off % 32 == 0;
    it->ntiles = tiles ? (d->rows_pad / 32) * it->nchunks : 0;
    it->pad_ = 0;
    // (a tile item takes NO range of the dest-major kernel's element space: `first` does not advance, so that kernel's search never
    //  lands on it and launches no blocks for it)
    return tiles ? 0 : (int64_t)ntaps * it->nchunks * d->rows_pad * CK;
}

extern "C" int abc_pack_batch(const void* items_dev, int32_t nitems, int64_t total, abc_stream_t stream) {
    if (nitems < 1 || total < 0) return abc_fail(ABC_EINVAL, "pack_batch: empty");
    if (nitems > PT_MAXITEMS) return abc_fail(ABC_EINVAL, "pack_batch: more than 1023 items");
    const int64_t nb = (total + PACK_CHUNK - 1) / PACK_CHUNK;
    if (nb > 0) hipLaunchKernelGGL(pack_batch_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items_dev, nitems, total);
    // (the items flagged for the tile form; a grid-stride loop over their tiles -- the count lives in the device table only)
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items_dev, nitems);
    return abc_check_launch("pack_batch");
}

extern "C" int abc_adam_step(const abc_adam_desc* d, abc_stream_t stream) {
    if (d->n < 1) return abc_fail(ABC_EINVAL, "adam: empty");
    if (((uintptr_t)d->p | (uintptr_t)d->g | (uintptr_t)d->m | (uintptr_t)d->v) & 15) return abc_fail(ABC_EINVAL, "adam: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, d->step);
    int64_t nb = (d->n / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(256), 0, st, *d);
    return abc_check_launch("adam_step");
}

extern "C" int abc_plane_sum_work(int32_t C) { return C * PS_CHUNKS; }

extern "C" int abc_plane_sum(const float* x, int32_t B, int32_t C, int32_t HW, const float* chan_scale, float* work, float* out,
                             abc_stream_t stream) {
    if (C < 1) return abc_fail(ABC_EINVAL, "plane_sum: empty");
    hipLaunchKernelGGL(plane_sum_kernel, dim3(C, PS_CHUNKS), dim3(256), 0, (hipStream_t)stream, x, B, C, HW, work);
    hipLaunchKernelGGL(plane_sum_reduce_kernel, dim3(abc_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, (const float*)work, C,
                       chan_scale, out);
    return abc_check_launch("plane_sum");
}

extern "C" int abc_nms_peaks(const abc_nms_desc* d, abc_stream_t stream) {
    const int64_t npix = (int64_t)d->B * d->h * d->w;
    hipLaunchKernelGGL(nms_kernel, dim3((int)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("nms_peaks");
}

__global__ void counter_add_kernel(uint32_t* p, uint32_t inc) { *p += inc; }
struct ConcatArgs { const float* src[16]; int32_t first[17]; };
__global__ __launch_bounds__(256) void concat_kernel(const ConcatArgs a, int n, float* dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.first[n]) return;
    int j = 0;
    while (j + 1 < n && a.first[j + 1] <= i) ++j;
    dst[i] = a.src[j][i - a.first[j]];
}
extern "C" int abc_concat_f32(const float* const* srcs, const int32_t* counts, int32_t n, float* dst, abc_stream_t stream) {
    if (n < 1 || n > 16) return abc_fail(ABC_EINVAL, "concat_f32: 1..16 arrays");
    ConcatArgs a;
    a.first[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (counts[i] < 0 || srcs[i] == nullptr) return abc_fail(ABC_EINVAL, "concat_f32: null / negative entry");
        a.src[i] = srcs[i]; a.first[i + 1] = a.first[i] + counts[i];
    }
    for (int i = n; i < 16; ++i) { a.src[i] = srcs[0]; a.first[i + 1] = a.first[n]; }
    if (a.first[n] == 0) return ABC_OK;
    hipLaunchKernelGGL(concat_kernel, dim3(abc_cdiv(a.first[n], 256)), dim3(256), 0, (hipStream_t)stream, a, n, dst);
    return abc_check_launch("concat_f32");
}

extern "C" int abc_counter_add_u32(uint32_t* p, uint32_t inc, abc_stream_t stream) {
    if (!p) return abc_fail(ABC_EINVAL, "counter_add: null");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, inc);
    return abc_check_launch("counter_add");
}

// sizeof() of every descriptor, so that the host binding can verify its mirror structs
extern "C" int abc_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(abc_act_src);
        case 1: return (int)sizeof(abc_conv_desc);
        case 2: return (int)sizeof(abc_pack_desc);
        case 3: return (int)sizeof(abc_bn_fwd_desc);
        case 4: return (int)sizeof(abc_act_bwd_desc);
        case 5: return (int)sizeof(abc_bn_bwd_desc);
        case 6: return (int)sizeof(abc_bn_apply_desc);
        case 7: return (int)sizeof(abc_wgrad_desc);
        case 8: return (int)sizeof(abc_wgrad_reduce_desc);
        case 9: return (int)sizeof(abc_loss_desc);
        case 10: return (int)sizeof(abc_loss_fin_desc);
        case 11: return (int)sizeof(abc_adam_desc);
        case 12: return (int)sizeof(abc_nms_desc);
        case 13: return (int)sizeof(abc_cbam_channel_desc);
        case 14: return (int)sizeof(abc_cbam_pix_desc);
        case 15: return (int)sizeof(abc_cbam_conv7_desc);
        case 16: return (int)sizeof(abc_metrics_desc);
        case 17: return (int)sizeof(abc_extract_desc);
        case 18: return (int)sizeof(abc_raster_desc);
        case 19: return (int)sizeof(abc_heads_fused_desc);
        case 20: return (int)sizeof(abc_heads_epi);
        case 21: return (int)sizeof(abc_convt_desc);
        default: return -1;
    }
}
