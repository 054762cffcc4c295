// Generic tap-list convolution as implicit GEMM on the gfx950 matrix cores.
//
// GEMM view: M = output pixels (a (2*MT) x 16 spatial patch of one image per workgroup: 128 or
// 256 pixels), N = output channels (BN per workgroup), K = taps x input channels, walked as
// [Cin chunk of 64 B][tap group].  Per chunk the activated input halo tile is staged ONCE in LDS
// (previous layer's BN + activation (+pool, +dropout) applied on the way in) and reused by every
// tap; the weight slices of a tap group (<= TG taps) sit beside it, double-buffered.
// A wave computes TM x TN tiles of 32x32 with v_mfma_f32_32x32x16_bf16 (bf16 mode) or
// v_mfma_f32_32x32x2_f32 (exact-f32 parity mode): lane-half h of the wave owns bytes
// [32h, 32h+32) of each pixel's/row's 64-byte chunk for BOTH operands, so one LDS image serves
// both dtypes and fragment reads are plain ds_read_b128.
//
// Software pipeline (cdna_hip_programming.md T14): the global loads of stage s+1 (weights, and the
// next chunk's halo when s+1 opens a chunk) are ISSUED before the MFMA block of stage s and
// COMMITTED to LDS after it, so L2/HBM latency hides under the matrix work; one barrier per stage.
//
// LDS images: pixel stride PS = chunk bytes + 16, A row stride a multiple of 256 B: every
// ds_read_b128 lane group then covers 16 distinct 16-byte slots (conflict-free).
//
// Reference ops covered: see include/abcnet_hip.h (abc_conv_desc).
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "conv_fast.hpp"
#include <stdlib.h>

namespace {

constexpr int NTHR = 512;  // 8 waves per workgroup, one workgroup per CU (2 waves per SIMD)
constexpr int NA_MAX = 4;  // halo segments a thread may prefetch (fast path)

struct ConvK {
    ActSrc src;
    const void* w;
    const float* bias;
    void* y;
    float* stats;
    int B, Hin, Win, cin_off, Cin, nchunks;
    int Hg, Wg, Hout, Wout, ldy, cout_off, Cout, Cout_pad;
    int om, oy0, ox0;
    int ntaps, tg, ngroups, dy_min, dx_min, HH, HW, RS;
    int tiles_x, tiles_y, nblocks_n, sA_bytes, a_bufs, sB_off, sB_bytes, tap_off, coef_off, cstride, planar_out, ctot_out, fast_a, dbg, ntiles, b_static, stg_off, stats_rows, accumulate, magic, out_act;
    float out_slope;
    unsigned bytesA, bytesW;
    int8_t ty[ABC_MAX_TAPS], tx[ABC_MAX_TAPS];
};

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT, bool FAST>
__global__ __launch_bounds__(512, 2) void conv_igemm_kernel(const ConvK a) {
    constexpr int CKB = CK * (int)sizeof(CT);
    constexpr int PS = CKB + 16;
    constexpr int LHB = CKB / 2;
    constexpr int NR = LHB / 16;
    constexpr int NV = Frag<CT>::NV;
    constexpr int SEGS = CKB / 16;
    constexpr int NT = BN / 32;
    constexpr int WN = (MT == 6) ? 4 : ((NT >= 2) ? 2 : 1);  // MT = 6 (12-row patch): 2 x 4 waves of 3 x 1 tiles
    constexpr int WM = 8 / WN;
    constexpr int TM = MT / WM;
    constexpr int TN = NT / WN;
    static_assert(TM >= 1 && TM * WM == MT && TN >= 1 && TN * WN == NT, "tile/wave layout");
    constexpr int TGMAX = 3;
    // weight segments a thread prefetches per stage; narrow layers (BN = 32) may hold ALL taps of their single chunk
    // resident in LDS for the whole persistent loop (b_static), loaded in one go
    constexpr int NB = (BN == 32) ? 3 : (TGMAX * BN * SEGS + NTHR - 1) / NTHR;
    typedef typename Frag<CT>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                 // a_bufs buffers of sA_bytes
    char* sB = smem + a.sB_off;      // 2 buffers of sB_bytes
    int* sTap = (int*)(smem + a.tap_off);
    float* sCoef = (float*)(smem + a.coef_off);  // [3][cstride]: scale, shift, slope of the conv's input channels

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- persistent workgroup: tiles id, id + gridDim.x, ...; tile -> (n-block, patch, image), n-block fastest so
    // that neighbouring workgroups share the halo in L2.  The next tile's first halo chunk and weight stage are
    // prefetched into registers before the epilogue of the current tile and committed after it.
    const int ntiles = a.ntiles;
    int tile = abc_xcd_remap(blockIdx.x, gridDim.x);
    int nb, mblock, b, gy0, gx0, n0, iy0, ix0;
    auto decode = [&](int id) {
        nb = id % a.nblocks_n; id /= a.nblocks_n;
        mblock = id;
        const int tx_i = id % a.tiles_x; id /= a.tiles_x;
        const int ty_i = id % a.tiles_y; id /= a.tiles_y;
        b = id;
        gy0 = ty_i * (2 * MT); gx0 = tx_i * 16;
        n0 = nb * BN;
        iy0 = gy0 * STRIDE + a.dy_min; ix0 = gx0 * STRIDE + a.dx_min;
    };
    decode(tile);

    if (tid < a.ntaps) sTap[tid] = a.ty[tid] * a.RS + a.tx[tid] * PS;
    const bool has_coef = a.src.scale != nullptr;
    if (has_coef && FAST) {
        for (int i = tid; i < a.Cin; i += NTHR) {
            sCoef[i] = a.src.scale[a.cin_off + i];
            sCoef[a.cstride + i] = a.src.shift[a.cin_off + i];
            sCoef[2 * a.cstride + i] = a.src.slope[a.cin_off + i];
        }
    }
    const float* lcoef = has_coef ? sCoef : nullptr;

    f32x16 acc[TM][TN];
    int aBase[TM], bBase[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int prow = 2 * (wm * TM + i) + (r >> 4), pcol = r & 15;
        aBase[i] = prow * STRIDE * a.RS + pcol * STRIDE * PS + h * LHB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) bBase[j] = ((wn * TN + j) * 32 + r) * PS + h * LHB;

    const CT* wp = (const CT*)a.w;
    const int nstages = a.nchunks * a.ngroups;

    u32x4 breg[NB];
    // FAST: plain NHWC input -> split-phase prefetch by raw buffer loads; otherwise (pool / dropout / planar / ragged
    // channel tail) the halo is staged synchronously by the general loader.  Separate instantiations: the general
    // loader inlined beside the prefetch registers makes the allocator spill, and a scratch reload waits on vmcnt(0).
    HaloFetch<InT, CT, CK, FAST ? NA_MAX : 1, NTHR> apre;
    const __amdgpu_buffer_rsrc_t rsA = abc_make_rsrc(a.src.x, a.bytesA), rsW = abc_make_rsrc(a.w, a.bytesW);
    const HaloGeom gA = {a.HH, a.HW, a.magic, a.Hin, a.Win, a.src.Hx, a.src.Wx, a.src.ldx};

    // weights of stage (c, g): tcnt x BN rows of CKB bytes, contiguous per tap in the packed layout
    // (thread id laundered: per-segment offsets are recomputed per stage instead of being hoisted and spilled)
    auto b_issue = [&](int c, int g) {
        const int t0 = g * a.tg;
        const int tcnt = min(a.tg, a.ntaps - t0);
        const int total = tcnt * BN * SEGS;
        const int lt = abc_launder(tid);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int s = lt + i * NTHR;
            const int tl = s / (BN * SEGS);
            const int rem = s - tl * (BN * SEGS);
            const int row = rem / SEGS, part = rem - row * SEGS;
            const unsigned off = (unsigned)((((t0 + tl) * a.nchunks + c) * a.Cout_pad + n0 + row) * CK + part * NV) * (unsigned)sizeof(CT);
            breg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, s < total ? off : 0x80000000u, 0, 0);
        }
    };
    auto b_commit = [&](int g, char* dst) {
        const int t0 = g * a.tg;
        const int tcnt = min(a.tg, a.ntaps - t0);
        const int total = tcnt * BN * SEGS;
        const int lt = abc_launder(tid);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int s = lt + i * NTHR;
            if (s < total) {
                const int tl = s / (BN * SEGS);
                const int rem = s - tl * (BN * SEGS);
                const int row = rem / SEGS, part = rem - row * SEGS;
                *(u32x4*)(dst + (tl * BN + row) * PS + part * 16) = breg[i];
            }
        }
    };

    // ---- first tile: chunk 0 halo + stage 0 weights
    b_issue(0, 0);
    if constexpr (FAST) apre.issue(rsA, gA, b, iy0, ix0, a.cin_off, tid, 1 << 30);
    __syncthreads();  // coefficient table + tap offsets visible

  bool first_tile = true;
  for (;;) {
    // ---- commit the prefetched first stage of this tile
    if constexpr (FAST) apre.commit(sA, a.RS, PS, gA, lcoef, a.cstride, tid, 1 << 30);
    else stage_halo<InT, CT, CK>(sA, a.RS, PS, a.HH, a.HW, b, iy0, ix0, a.Hin, a.Win, a.src, a.cin_off, tid, NTHR, a.Cin);
    if (!a.b_static || first_tile) b_commit(0, sB);
    first_tile = false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    __syncthreads();
    // this tile's output coordinates; the next tile (if any) is decoded and its first stage prefetched at the
    // LAST stage of the main loop, before that stage's MFMA block
    const int cur_b = b, cur_gy0 = gy0, cur_gx0 = gx0, cur_n0 = n0, cur_mblock = mblock;
    const int next_tile = tile + gridDim.x;
    const bool more = next_tile < ntiles;

    int c = 0, g = 0;
    for (int s = 0; s < nstages; ++s) {
        int gn = g + 1, cn = c;
        if (gn == a.ngroups) { gn = 0; cn = c + 1; }
        const bool has_next = (s + 1 < nstages);
        const bool new_chunk = has_next && (gn == 0);
        if (has_next && !(ABC_DBG(a.dbg) & 1)) b_issue(cn, gn);
        if (s == nstages - 1 && more) {
            decode(next_tile);
            if (!a.b_static) b_issue(0, 0);
            if constexpr (FAST) apre.issue(rsA, gA, b, iy0, ix0, a.cin_off, tid, 1 << 30);
        }
        if constexpr (FAST) {
            if (new_chunk && !(ABC_DBG(a.dbg) & 2)) apre.issue(rsA, gA, b, iy0, ix0, a.cin_off + cn * CK, tid, 1 << 30);
        }

        // ---- MFMA over the taps of this stage
        {
            const char* sAc = sA + ((a.a_bufs == 2) ? (c & 1) * a.sA_bytes : 0);
            const char* sBc = sB + (s & 1) * a.sB_bytes;
            const int t0 = g * a.tg;
            const int tcnt = min(a.tg, a.ntaps - t0);
            for (int tl = 0; tl < ((ABC_DBG(a.dbg) & 4) ? 0 : tcnt); ++tl) {
                const int aoff = sTap[t0 + tl];
                const int boff = tl * BN * PS;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    frag_t fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = *(const frag_t*)(sAc + aBase[i] + aoff + q * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = *(const frag_t*)(sBc + bBase[j] + boff + q * 16);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) mma16B(acc[i][j], fa[i], fb[j]);
                }
            }
        }

        if (has_next && !(ABC_DBG(a.dbg) & 8)) b_commit(gn, sB + ((s + 1) & 1) * a.sB_bytes);
        if (new_chunk) {
            if (a.a_bufs == 2) {
                // the other halo buffer was last read in chunk c-1: free since the barrier that ended it
                if constexpr (FAST) {
                    if (!(ABC_DBG(a.dbg) & 16)) apre.commit(sA + (cn & 1) * a.sA_bytes, a.RS, PS, gA, lcoef ? lcoef + cn * CK : nullptr, a.cstride, tid, 1 << 30);
                }
            } else {
                __syncthreads();  // every wave is done reading this chunk's halo
                if constexpr (FAST) apre.commit(sA, a.RS, PS, gA, lcoef ? lcoef + cn * CK : nullptr, a.cstride, tid, 1 << 30);
                else
                    stage_halo<InT, CT, CK>(sA, a.RS, PS, a.HH, a.HW, b, iy0, ix0, a.Hin, a.Win, a.src, a.cin_off + cn * CK, tid, NTHR,
                                            a.Cin - cn * CK);
            }
        }
        __syncthreads();
        c = cn; g = gn;
    }

    // ---- epilogue: bias, statistics of the f32 values, store.
    // NHWC outputs go through a per-wave LDS transpose (32 pixels x TN*32 channels at a time) so that every
    // global store is 16 bytes of consecutive channels of one pixel (a lane of the accumulator layout holds ONE
    // channel of 16 pixels: storing from registers would be 2-/4-byte scattered stores, issue-bound).
    OutT* yo = (OutT*)a.y;
    float s1[TN], s2[TN], smx[TN], smn[TN];
    float bv[TN];
    bool nval[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s1[j] = 0.f; s2[j] = 0.f; smx[j] = -3.0e38f; smn[j] = 3.0e38f;
        const int n = cur_n0 + (wn * TN + j) * 32 + r;
        nval[j] = n < a.Cout;
        bv[j] = (a.bias != nullptr && nval[j]) ? a.bias[n] : 0.f;
    }
    const bool planar = (sizeof(OutT) == 4) && a.planar_out;
    if (ABC_DBG(a.dbg) & 64) {
    } else if (!planar) {
        constexpr int TW = TN * 32;                       // channels of this wave's tile row
        constexpr int ROWB = TW * (int)sizeof(OutT) + 16;  // padded LDS row (bytes)
        constexpr int EV = 16 / (int)sizeof(OutT);         // elements per 16-byte store
        constexpr int SEG_PER_ROW = TW / EV;
        __syncthreads();  // main-loop LDS reads finished
        char* stg = smem + a.stg_off + wave * (32 * ROWB);
        const int cbase = cur_n0 + wn * TW;  // first channel of the wave's tile row
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int rit = (k & 3) + 8 * (k >> 2) + 4 * h;
                    const int gy = cur_gy0 + 2 * (wm * TM + i) + (rit >> 4), gx = cur_gx0 + (rit & 15);
                    const float v = acc[i][j][k] + bv[j];
                    if (nval[j] && gy < a.Hg && gx < a.Wg) {
                        // (max / min of the values AS STORED: CBAM's global max-pool and its backward -- unet2.py:10,20 -- see the tensor)
                        const float vr = (float)(OutT)v;
                        s1[j] += v; s2[j] += v * v; smx[j] = fmaxf(smx[j], vr); smn[j] = fminf(smn[j], vr);
                    }
                    *(OutT*)(stg + rit * ROWB + (j * 32 + r) * (int)sizeof(OutT)) = (OutT)(a.out_act ? fmaxf(v, a.out_slope * v) : v);
                }
            }
            __syncthreads();
            // 32 rows x SEG_PER_ROW 16-byte segments, 64 lanes
#pragma unroll
            for (int e = lane; e < 32 * SEG_PER_ROW; e += 64) {
                const int rit = e / SEG_PER_ROW, sg = e - rit * SEG_PER_ROW;
                const int gy = cur_gy0 + 2 * (wm * TM + i) + (rit >> 4), gx = cur_gx0 + (rit & 15);
                const int cch = cbase + sg * EV;
                if (gy < a.Hg && gx < a.Wg && cch < a.Cout) {
                    const size_t o = ((size_t)(cur_b * a.Hout + gy * a.om + a.oy0) * a.Wout + gx * a.om + a.ox0) * a.ldy + a.cout_off + cch;
                    if (a.accumulate) {
                        for (int q = 0; q < EV && cch + q < a.Cout; ++q)
                            yo[o + q] = (OutT)((float)yo[o + q] + (float)*(const OutT*)(stg + rit * ROWB + sg * 16 + q * (int)sizeof(OutT)));
                    } else if (cch + EV <= a.Cout && ((a.ldy | (a.cout_off + cch)) % EV) == 0) {
                        *(f32x4*)(yo + o) = *(const f32x4*)(stg + rit * ROWB + sg * 16);
                    } else {
                        for (int q = 0; q < EV && cch + q < a.Cout; ++q) yo[o + q] = *(const OutT*)(stg + rit * ROWB + sg * 16 + q * (int)sizeof(OutT));
                    }
                }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = cur_n0 + (wn * TN + j) * 32 + r;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int rit = (k & 3) + 8 * (k >> 2) + 4 * h;
                    const int gy = cur_gy0 + 2 * (wm * TM + i) + (rit >> 4), gx = cur_gx0 + (rit & 15);
                    if (nval[j] && gy < a.Hg && gx < a.Wg) {
                        const float v = acc[i][j][k] + bv[j];
                        s1[j] += v; s2[j] += v * v;
                        // NCHW: registers k..k+3 of a lane are 4 consecutive x of one plane -> one 16-byte store
                        const size_t o = ((size_t)(cur_b * a.ctot_out + a.cout_off + n) * a.Hout + gy) * a.Wout + gx;
                        if ((k & 3) == 0 && gx + 3 < a.Wg && (a.Wout & 3) == 0) {
                            f32x4 t;
                            t[0] = v; t[1] = acc[i][j][k + 1] + bv[j]; t[2] = acc[i][j][k + 2] + bv[j]; t[3] = acc[i][j][k + 3] + bv[j];
                            *(f32x4*)((float*)a.y + o) = t;
                        } else if (!(gx - (k & 3) + 3 < a.Wg && (a.Wout & 3) == 0)) {
                            ((float*)a.y)[o] = v;
                        }
                    }
                }
            }
        }
    }
    if (a.stats != nullptr && !(ABC_DBG(a.dbg) & 128)) {
        __syncthreads();  // LDS reuse
        float* red = (float*)(smem + a.stg_off);  // [WM][4][BN]
        const int rows = a.stats_rows == 4 ? 4 : 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float v1 = s1[j] + __shfl_xor(s1[j], 32);
            float v2 = s2[j] + __shfl_xor(s2[j], 32);
            float v3 = fmaxf(smx[j], __shfl_xor(smx[j], 32));
            float v4 = fminf(smn[j], __shfl_xor(smn[j], 32));
            if (h == 0) {
                const int nl = (wn * TN + j) * 32 + r;
                red[(wm * 4 + 0) * BN + nl] = v1;
                red[(wm * 4 + 1) * BN + nl] = v2;
                red[(wm * 4 + 2) * BN + nl] = v3;
                red[(wm * 4 + 3) * BN + nl] = v4;
            }
        }
        __syncthreads();
        if (tid < BN && cur_n0 + tid < a.Cout) {
            float v1 = 0.f, v2 = 0.f, v3 = -3.0e38f, v4 = 3.0e38f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                v1 += red[(w * 4 + 0) * BN + tid]; v2 += red[(w * 4 + 1) * BN + tid];
                v3 = fmaxf(v3, red[(w * 4 + 2) * BN + tid]); v4 = fminf(v4, red[(w * 4 + 3) * BN + tid]);
            }
            a.stats[((size_t)cur_mblock * rows + 0) * a.Cout + cur_n0 + tid] = v1;
            a.stats[((size_t)cur_mblock * rows + 1) * a.Cout + cur_n0 + tid] = v2;
            if (rows == 4) {
                a.stats[((size_t)cur_mblock * rows + 2) * a.Cout + cur_n0 + tid] = v3;
                a.stats[((size_t)cur_mblock * rows + 3) * a.Cout + cur_n0 + tid] = v4;
            }
        }
    }
    if (!more) break;
    tile = next_tile;
    __syncthreads();  // epilogue staging / statistics scratch in LDS is free again
  }
}

struct Geom {
    int CK, BN, MT, dy_min, dx_min, HH, HW, PS, RS, tg, ngroups, sA_bytes, a_bufs, sB_bytes, tap_off, coef_off, cstride, lds, tiles_x,
        tiles_y, nbn, grid, fast_a, b_static, stg_off;
};

static int conv_geom(const abc_conv_desc* d, Geom* g) {
    if (d->ntaps < 1 || d->ntaps > ABC_MAX_TAPS) return abc_fail(ABC_EINVAL, "conv: ntaps out of range");
    if (d->stride != 1 && d->stride != 2) return abc_fail(ABC_EUNSUPPORTED, "conv: stride must be 1 or 2");
    if (d->dtype_c == ABC_FP8 || d->dtype_in == ABC_FP8 || d->dtype_out == ABC_FP8) return abc_fail(ABC_EUNSUPPORTED, "conv: the general kernel has no fp8 form");
    const int csz = d->dtype_c == ABC_BF16 ? 2 : 4;
    g->CK = abc_conv_chunk(d->dtype_c, d->Cin);
    if (g->CK <= 0) return abc_fail(ABC_EINVAL, "conv: Cin must be positive");
    if (d->Cout_pad % 32 || d->Cout_pad < d->Cout) return abc_fail(ABC_EINVAL, "conv: Cout_pad must be a multiple of 32 >= Cout");
    g->BN = (d->Cout_pad % 128 == 0) ? 128 : (d->Cout_pad % 64 == 0 ? 64 : 32);
    g->nbn = d->Cout_pad / g->BN;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    g->dy_min = dymin; g->dx_min = dxmin;
    // patch height (2*MT rows x 16 columns).  Taller patches re-stage the weights less often per output pixel; the
    // persistent grid runs ceil(tiles/256) rounds, so pick the height with the least rounds x tile-work (ties: taller).
    // MT = 6 exists for BN = 128 only (2 x 4 wave layout); BN = 32 needs 8 m-tiles to feed 8 waves; stride 2 keeps
    // the halo image small with MT = 4.
    {
        int best = -1;
        long best_cost = 0;
        const int cand[3] = {8, 6, 4};
        for (int ci = 0; ci < 3; ++ci) {
            const int mt = cand[ci];
            if (g->BN == 32 && mt != 8) continue;
            if (mt == 6 && (g->BN != 128 || d->stride != 1)) continue;
            if (d->stride == 2 && g->BN != 32 && mt != 4) continue;
            const long tiles = (long)g->nbn * abc_cdiv(d->Wg, 16) * abc_cdiv(d->Hg, 2 * mt) * d->B;
            const long cost = ((tiles + 255) / 256) * mt;
            if (best < 0 || cost < best_cost) { best = mt; best_cost = cost; }
        }
        g->MT = best;
    }
    const int prow = 2 * g->MT;
    g->HH = (prow - 1) * d->stride + (dymax - dymin) + 1;
    g->HW = 15 * d->stride + (dxmax - dxmin) + 1;
    const int CKB = g->CK * csz;
    g->PS = CKB + 16;
    g->RS = abc_roundup(g->HW * g->PS, 256);
    g->sA_bytes = abc_roundup(g->HH * g->RS, 256);
    const int segs = CKB / 16;
    const int64_t bytes_a = (int64_t)d->B * d->src.Hx * d->src.Wx * d->src.ldx * (d->dtype_in == ABC_BF16 ? 2 : 4);
    g->fast_a = (!d->src.pool && !d->src.planar && d->src.drop_p <= 0.f && d->Cin % g->CK == 0 && bytes_a < (int64_t(1) << 31) &&
                 abc_cdiv(g->HH * g->HW * segs, NTHR) <= NA_MAX && g->HH * g->HW * g->HW < 65536) ? 1 : 0;
    g->cstride = abc_roundup(d->Cin, 4);
    const int coef_bytes = abc_roundup(3 * g->cstride * 4, 256);
    // one workgroup per CU: spend the LDS on double buffers (halo when it is prefetched, weights always)
    const int budget = 156 * 1024 - coef_bytes - 256;
    g->a_bufs = (g->fast_a && 2 * g->sA_bytes + 2 * g->BN * g->PS <= budget) ? 2 : 1;
    int tg = (budget - g->a_bufs * g->sA_bytes) / 2 / (g->BN * g->PS);
    if (tg < 1) return abc_fail(ABC_EUNSUPPORTED, "conv: LDS tile too large");
    if (tg > 3) tg = 3;
    if (tg > d->ntaps) tg = d->ntaps;
    // narrow layers: one chunk, one n-block -> the whole weight set stays in LDS across the persistent tile loop
    g->b_static = (g->BN == 32 && g->nbn == 1 && abc_cdiv(d->Cin, g->CK) == 1 && d->ntaps * g->BN * segs <= 3 * NTHR) ? 1 : 0;
    if (g->b_static) tg = d->ntaps;
    g->ngroups = abc_cdiv(d->ntaps, tg);
    g->tg = abc_cdiv(d->ntaps, g->ngroups);
    g->sB_bytes = abc_roundup(g->tg * g->BN * g->PS, 256);
    g->tap_off = g->a_bufs * g->sA_bytes + 2 * g->sB_bytes;
    g->coef_off = g->tap_off + 256;
    g->lds = g->coef_off + coef_bytes;
    {   // epilogue transpose staging: 8 waves x 32 rows x (TN*32 channels + 16 B pad); it aliases the halo/weight
        // buffers (dead by then) EXCEPT resident weights, which it must not touch
        const int tn = (g->BN / 32 >= 2) ? g->BN / 64 : 1;
        const int osz = d->dtype_out == ABC_BF16 ? 2 : 4;
        const int stg = 8 * 32 * (tn * 32 * osz + 16) + 8 * 4 * 128 * 4;
        g->stg_off = g->b_static ? abc_roundup(g->lds, 256) : 0;
        if (g->lds < g->stg_off + stg) g->lds = g->stg_off + stg;
    }
    if (g->lds > 160 * 1024) return abc_fail(ABC_EUNSUPPORTED, "conv: LDS tile too large");
    g->tiles_x = abc_cdiv(d->Wg, 16);
    g->tiles_y = abc_cdiv(d->Hg, prow);
    g->grid = g->nbn * g->tiles_x * g->tiles_y * d->B;
    return ABC_OK;
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT, bool FAST>
static int launch_fs(const ConvK& k, const Geom& g, hipStream_t st) {
    auto fn = conv_igemm_kernel<InT, CT, OutT, CK, BN, STRIDE, MT, FAST>;
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)fn, 160 * 1024, &lds_ok)) return rc;
    // persistent: one workgroup per CU walks the tiles
    const int nwg = g.grid < 256 ? g.grid : 256;
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(NTHR), g.lds, st, k);
    return abc_check_launch("conv_igemm");
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT>
static int launch_inst(const ConvK& k, const Geom& g, hipStream_t st) {
    return g.fast_a ? launch_fs<InT, CT, OutT, CK, BN, STRIDE, MT, true>(k, g, st) : launch_fs<InT, CT, OutT, CK, BN, STRIDE, MT, false>(k, g, st);
}

template <typename InT, typename CT, typename OutT, int CK, int BN>
static int launch_mt(const ConvK& k, const Geom& g, int stride, hipStream_t st) {
    if constexpr (BN == 32) {  // 8 waves need 8 m-tiles when there is a single n-tile
        return stride == 2 ? launch_inst<InT, CT, OutT, CK, BN, 2, 8>(k, g, st) : launch_inst<InT, CT, OutT, CK, BN, 1, 8>(k, g, st);
    } else {
        if (stride == 2) return launch_inst<InT, CT, OutT, CK, BN, 2, 4>(k, g, st);
        if constexpr (BN == 128) {
            if (g.MT == 6) return launch_inst<InT, CT, OutT, CK, BN, 1, 6>(k, g, st);
        }
        return g.MT == 8 ? launch_inst<InT, CT, OutT, CK, BN, 1, 8>(k, g, st) : launch_inst<InT, CT, OutT, CK, BN, 1, 4>(k, g, st);
    }
}

template <typename InT, typename CT, typename OutT, int CK>
static int launch_bn(const ConvK& k, const Geom& g, int stride, hipStream_t st) {
    switch (g.BN) {
        case 128: return launch_mt<InT, CT, OutT, CK, 128>(k, g, stride, st);
        case 64: return launch_mt<InT, CT, OutT, CK, 64>(k, g, stride, st);
        default: return launch_mt<InT, CT, OutT, CK, 32>(k, g, stride, st);
    }
}

}  // namespace

extern "C" int abc_conv_chunk(int dtype_c, int Cin) {
    if (Cin <= 0) return -1;
    const int cp = abc_roundup(Cin, 16);
    if (dtype_c == ABC_BF16) return (cp % 32 == 0) ? 32 : 16;
    if (dtype_c == ABC_FP8) return (Cin % 64 == 0) ? 64 : -1;   // 64-byte chunks only (the lean kernel's weights-direct loop)
    return 16;
}

extern "C" int abc_conv_tile(const abc_conv_desc* d, int32_t* bn, int32_t* mt, int32_t* ck) {
    if (abc_head_fwd_ok(d)) { *bn = 32; *mt = 2; *ck = abc_conv_chunk(d->dtype_c, d->Cin); return ABC_OK; }   // (a wave: 32 rows x 64 pixels)
    abc_fast_geom f;
    if (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible) { *bn = f.BN; *mt = f.MT; *ck = f.CK; return ABC_OK; }
    Geom g;
    int rc = conv_geom(d, &g);
    if (rc) return rc;
    *bn = g.BN; *mt = g.MT; *ck = g.CK;
    return ABC_OK;
}

// the heads' epilogue is served by the lean kernel only; act_bwd in the epilogue by the kernel that would run the plain data gradient
// anyway -- the narrow-level kernel for its shapes, else the lean kernel -- and by no other
static bool lean_only(const abc_conv_desc* d) { return d->heads_epi != nullptr; }

// 0: not served; 1: conv_fast.hip; 5: conv_narrow.hip
static int actbwd_server(const abc_conv_desc* d) {
    if (d->actbwd_y == nullptr || d->heads_epi != nullptr) return 0;
    abc_conv_desc p = *d;
    p.actbwd_y = nullptr;
    if (abc_conv_stem_ok(&p, nullptr) || abc_head_fwd_ok(&p) || abc_head_dgrad_ok(&p)) return 0;
    if (abc_conv_narrow_ok(&p)) return abc_conv_narrow_ok(d) ? 5 : 0;
    abc_fast_geom f;
    return (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible) ? 1 : 0;
}

extern "C" int abc_conv_actbwd_ok(const abc_conv_desc* d) { return actbwd_server(d) != 0 ? 1 : 0; }

extern "C" int abc_conv_variant(const abc_conv_desc* d) {
    if (d->actbwd_y != nullptr) return actbwd_server(d);
    if (lean_only(d)) { abc_fast_geom f; return (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible) ? 1 : 0; }
    if (abc_conv_stem_ok(d, nullptr)) return 2;
    if (abc_head_fwd_ok(d)) return 3;
    if (abc_head_dgrad_ok(d)) return 4;
    if (abc_conv_narrow_ok(d)) return 5;
    abc_fast_geom f;
    if (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible) return 1;
    return 0;
}

extern "C" int abc_conv_weight_layout(const abc_conv_desc* d) {
    if (d->actbwd_y != nullptr && actbwd_server(d) == 5) return 0;
    if (lean_only(d) || d->actbwd_y != nullptr) { abc_fast_geom f; return (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible && f.wd) ? 1 : 0; }
    if (abc_conv_stem_ok(d, nullptr) || abc_head_fwd_ok(d) || abc_head_dgrad_ok(d) || abc_conv_narrow_ok(d)) return 0;
    abc_fast_geom f;
    return (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible && f.wd) ? 1 : 0;   // the weights-direct loop of conv_fast.hip
}

extern "C" int abc_conv_stat_blocks(const abc_conv_desc* d) {
    if (d->actbwd_y != nullptr && actbwd_server(d) == 5) return abc_conv_narrow_stat_blocks(d);
    if (!lean_only(d) && d->actbwd_y == nullptr) {
        { int nb = 0; if (abc_conv_stem_ok(d, &nb)) return nb; }
        if (abc_conv_narrow_ok(d)) return abc_conv_narrow_stat_blocks(d);
    }
    abc_fast_geom f;
    if (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible) return (f.b_static && d->stats_rows != 4) ? f.nwg : f.tiles_x * f.tiles_y * d->B * (f.lp ? 2 : 1);   // (resident weights: one row per workgroup; lane = pixel epilogue: one row per wave row, WM = 2)
    Geom g;
    if (conv_geom(d, &g)) return -1;
    return g.tiles_x * g.tiles_y * d->B;
}

extern "C" int abc_conv_fwd(const abc_conv_desc* d, abc_stream_t stream);
static bool batchable(const abc_conv_desc* d, int n) {
    // (only descriptors the lean kernel would run anyway)
    for (int i = 0; i < n; ++i)
        if (d[i].stem_x != nullptr || d[i].pool_y != nullptr || d[i].stats != nullptr || abc_conv_stem_ok(&d[i], nullptr) || abc_head_fwd_ok(&d[i]) ||
            abc_head_dgrad_ok(&d[i]) || abc_conv_narrow_ok(&d[i]) || (d[i].Hg - 1) * d[i].om + d[i].oy0 >= d[i].Hout || (d[i].Wg - 1) * d[i].om + d[i].ox0 >= d[i].Wout)
            return false;
    abc_fast_geom g0;
    return abc_conv_fast_batch_ok(d, n, &g0) != 0;
}

extern "C" int abc_conv_batch_ok(const abc_conv_desc* d, int32_t n) { return (d != nullptr && batchable(d, n)) ? 1 : 0; }

extern "C" int abc_conv_fwd_batch(const abc_conv_desc* d, int32_t n, abc_stream_t stream) {
    if (d == nullptr || n < 1) return abc_fail(ABC_EINVAL, "conv_fwd_batch: no descriptors");
    if (batchable(d, n)) return abc_conv_fast_launch_batch(d, n, stream);
    for (int i = 0; i < n; ++i)
        if (int rc = abc_conv_fwd(&d[i], stream)) return rc;
    return ABC_OK;
}

extern "C" int abc_conv_fwd(const abc_conv_desc* d, abc_stream_t stream) {
    if (d->dtype_c == ABC_FP8 || d->dtype_out == ABC_FP8 || d->dtype_in == ABC_FP8) {
        // the fp8 inference graph: the heads' 1x1 convolution into NCHW f32 (heads.hip), else the lean kernel's weights-direct tile
        if (abc_head_fwd_ok(d)) return abc_head_fwd_launch(d, stream);
        if (d->src.pool || d->src.planar || d->planar_out || d->src.Hx != d->Hin || d->src.Wx != d->Win || d->stem_x != nullptr || d->pool_y != nullptr)
            return abc_fail(ABC_EUNSUPPORTED, "conv: fp8 needs a plain NHWC input and output");
        if ((d->src.ldx * abc_dsize(d->dtype_in)) % 16 || (d->cin_off * abc_dsize(d->dtype_in)) % 16 || (d->ldy * abc_dsize(d->dtype_out)) % 16 || (d->cout_off * abc_dsize(d->dtype_out)) % 16)
            return abc_fail(ABC_EINVAL, "conv: fp8 tensors must keep 16-byte alignment");
        if ((d->Hg - 1) * d->om + d->oy0 >= d->Hout || (d->Wg - 1) * d->om + d->ox0 >= d->Wout) return abc_fail(ABC_EINVAL, "conv: output grid exceeds output tensor");
        if (d->heads_epi != nullptr && d->dtype_out != ABC_FP8) return abc_fail(ABC_EINVAL, "conv: heads_epi with e4m3 operands needs dtype_out = ABC_FP8 (the features' type)");
        abc_fast_geom f;
        if (abc_conv_fast_geom(d, &f) != ABC_OK || !f.eligible) return abc_fail(ABC_EUNSUPPORTED, "conv: fp8 is served for 3x3, stride 1, Cin % 64 == 0, Cout % 128 == 0 only");
        return abc_conv_fast_launch(d, f, stream);
    }
    Geom g;
    int rc = conv_geom(d, &g);  // (also validates the descriptor)
    if (rc) return rc;
    if (d->src.pool && (d->src.Hx / 2 != d->Hin || d->src.Wx / 2 != d->Win))
        return abc_fail(ABC_EINVAL, "conv: pooled dims mismatch");
    if (!d->src.pool && (d->src.Hx != d->Hin || d->src.Wx != d->Win)) return abc_fail(ABC_EINVAL, "conv: dims mismatch");
    if (!d->src.planar && d->Cin % 16 == 0 && ((d->src.ldx * (d->dtype_in == ABC_BF16 ? 2 : 4)) % 16 || (d->cin_off % 8)))
        return abc_fail(ABC_EINVAL, "conv: input stride/offset must keep 16-byte alignment");
    ConvK k;
    k.src.x = d->src.x; k.src.scale = d->src.scale; k.src.shift = d->src.shift; k.src.slope = d->src.slope;
    k.src.Hx = d->src.Hx; k.src.Wx = d->src.Wx; k.src.ldx = d->src.ldx; k.src.pool = d->src.pool;
    k.src.drop_p = d->src.drop_p; k.src.drop_seed = d->src.drop_seed; k.src.drop_salt = d->src.drop_salt;
    k.src.planar = d->src.planar; k.src.ctot = d->src.ctot;
    k.planar_out = d->planar_out; k.ctot_out = d->ctot_out;
    if (d->src.planar && (d->dtype_in != ABC_F32 || d->src.pool || d->src.drop_p > 0.f))
        return abc_fail(ABC_EUNSUPPORTED, "conv: planar input must be f32 without pool/dropout");
    if (d->planar_out && (d->dtype_out != ABC_F32 || d->om != 1 || d->oy0 || d->ox0))
        return abc_fail(ABC_EUNSUPPORTED, "conv: planar output must be f32, unit output stride");
    k.w = d->w; k.bias = d->bias; k.y = d->y; k.stats = d->stats;
    k.B = d->B; k.Hin = d->Hin; k.Win = d->Win; k.cin_off = d->cin_off; k.Cin = d->Cin; k.nchunks = abc_cdiv(d->Cin, g.CK);
    k.Hg = d->Hg; k.Wg = d->Wg; k.Hout = d->Hout; k.Wout = d->Wout; k.ldy = d->ldy; k.cout_off = d->cout_off;
    k.Cout = d->Cout; k.Cout_pad = d->Cout_pad; k.om = d->om; k.oy0 = d->oy0; k.ox0 = d->ox0;
    k.ntaps = d->ntaps; k.tg = g.tg; k.ngroups = g.ngroups; k.dy_min = g.dy_min; k.dx_min = g.dx_min; k.HH = g.HH; k.HW = g.HW; k.RS = g.RS;
    k.tiles_x = g.tiles_x; k.tiles_y = g.tiles_y; k.nblocks_n = g.nbn; k.sA_bytes = g.sA_bytes; k.a_bufs = g.a_bufs;
    k.sB_off = g.a_bufs * g.sA_bytes; k.coef_off = g.coef_off; k.cstride = g.cstride;
    k.sB_bytes = g.sB_bytes; k.tap_off = g.tap_off; k.fast_a = g.fast_a; k.ntiles = g.grid; k.b_static = g.b_static; k.stg_off = g.stg_off;
    k.stats_rows = d->stats_rows; k.accumulate = d->accumulate; k.out_act = d->out_act; k.out_slope = d->out_slope;
    if (d->out_act && d->planar_out) return abc_fail(ABC_EUNSUPPORTED, "conv: out_act needs an NHWC output");
    if (d->accumulate && d->planar_out) return abc_fail(ABC_EUNSUPPORTED, "conv: accumulate needs an NHWC output");
    k.magic = 65536 / g.HW + 1;
    k.bytesA = (unsigned)((int64_t)d->B * d->src.Hx * d->src.Wx * d->src.ldx * (d->dtype_in == ABC_BF16 ? 2 : 4));
    k.bytesW = (unsigned)((int64_t)d->ntaps * k.nchunks * d->Cout_pad * g.CK * (d->dtype_c == ABC_BF16 ? 2 : 4));
    { const char* e = abc_knob("ABC_CONV_DBG"); k.dbg = e ? atoi(e) : 0; }  // timing ablations only (results invalid)
    for (int t = 0; t < d->ntaps; ++t) {
        k.ty[t] = (int8_t)(d->tap_dy[t] - g.dy_min);
        k.tx[t] = (int8_t)(d->tap_dx[t] - g.dx_min);
    }
    // the kernel's grid must address only in-range output pixels
    if ((d->Hg - 1) * d->om + d->oy0 >= d->Hout || (d->Wg - 1) * d->om + d->ox0 >= d->Wout)
        return abc_fail(ABC_EINVAL, "conv: output grid exceeds output tensor");
    if (d->heads_epi != nullptr) {
        abc_fast_geom f;
        if (abc_conv_fast_geom(d, &f) != ABC_OK || !f.eligible) return abc_fail(ABC_EUNSUPPORTED, "conv: heads_epi is served by the 3x3 weights-direct tile only");
        return abc_conv_fast_launch(d, f, stream);
    }
    if (d->actbwd_y != nullptr) {
        const int srv = actbwd_server(d);
        abc_fast_geom f;
        if (srv == 5) return abc_conv_narrow_launch(d, stream);
        if (srv != 1 || abc_conv_fast_geom(d, &f) != ABC_OK || !f.eligible)
            return abc_fail(ABC_EUNSUPPORTED, "conv: actbwd_y is not served for this descriptor (ask abc_conv_actbwd_ok first)");
        return abc_conv_fast_launch(d, f, stream);
    }
    if (d->stem_x != nullptr && !abc_conv_narrow_ok(d)) return abc_fail(ABC_EUNSUPPORTED, "conv: the fused first convolution (stem_x) is served by the narrow-level kernel only");
    if (d->pool_y != nullptr && !abc_conv_narrow_ok(d)) return abc_fail(ABC_EUNSUPPORTED, "conv: pool_y is served by the narrow-level kernel only (abc_conv_variant == 5)");
    if (d->head_aux != nullptr && !abc_head_fwd_ok(d)) return abc_fail(ABC_EUNSUPPORTED, "conv: head_aux is served by the heads' 1x1 kernel only (abc_conv_variant == 3)");
    if (abc_conv_stem_ok(d, nullptr)) return abc_conv_stem_launch(d, stream);
    if (abc_head_fwd_ok(d)) return abc_head_fwd_launch(d, stream);
    if (abc_head_dgrad_ok(d)) return abc_head_dgrad_launch(d, stream);
    if (abc_conv_narrow_ok(d)) return abc_conv_narrow_launch(d, stream);
    {
        abc_fast_geom f;
        if (abc_conv_fast_geom(d, &f) == ABC_OK && f.eligible) return abc_conv_fast_launch(d, f, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const int di = d->dtype_in, dc = d->dtype_c, dout = d->dtype_out;
    if (dc == ABC_F32) {
        if (di != ABC_F32 || dout != ABC_F32) return abc_fail(ABC_EUNSUPPORTED, "conv: f32 compute needs f32 in/out");
        return launch_bn<float, float, float, 16>(k, g, d->stride, st);
    }
    if (di == ABC_BF16 && dout == ABC_BF16)
        return g.CK == 32 ? launch_bn<bf16, bf16, bf16, 32>(k, g, d->stride, st) : launch_bn<bf16, bf16, bf16, 16>(k, g, d->stride, st);
    if (di == ABC_BF16 && dout == ABC_F32)
        return g.CK == 32 ? launch_bn<bf16, bf16, float, 32>(k, g, d->stride, st) : launch_bn<bf16, bf16, float, 16>(k, g, d->stride, st);
    if (di == ABC_F32 && dout == ABC_BF16)
        return g.CK == 32 ? launch_bn<float, bf16, bf16, 32>(k, g, d->stride, st) : launch_bn<float, bf16, bf16, 16>(k, g, d->stride, st);
    return abc_fail(ABC_EUNSUPPORTED, "conv: dtype combination");
}
