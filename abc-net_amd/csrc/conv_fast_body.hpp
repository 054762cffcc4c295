// The lean convolution kernel's device code (see conv_fast.hip for the design notes): shared by conv_fast.hip (4-wave workgroups)
// and conv_fast8.hip (8-wave workgroups on the 192-pixel x 128-channel weights-direct tile).
#pragma once
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "conv_fast.hpp"
#include <stdlib.h>
#include <math.h>
#include <type_traits>

#ifndef ABC_DEEP_TM
#define ABC_DEEP_TM 2      // deep operand pipelining (DEEP below) for wave tiles of up to this many 32-pixel M-tiles x one 32-channel N-tile
#endif

namespace abc_cf {

constexpr int FT = 256;      // threads per workgroup
// halo segments a thread may hold (3x3 taps, 64-byte chunks: (2*MT+2) x 18 pixels x 4 segments over 256 threads;
// MT = 8 also takes unet2's 5x5 taps: 20 x 20 pixels x 4 segments = 6.25 per thread)
__host__ __device__ constexpr int fa_max(int mt) { return mt >= 8 ? 7 : (mt >= 6 ? 4 : (mt >= 4 ? 3 : 2)); }
// stride 2 (the data gradients of the transposed convolutions, unet.py:44 under autograd: a 17 x 33 pixel halo per 8 x 16 tile):
// 9 segments per thread -- these kernels hold 64 accumulator registers, the staging fits
__host__ __device__ constexpr int fa_stride2() { return 9; }
// resident-weight (persistent, narrow-layer) workgroups only ever see 3x3 / 1x1 taps: 6 segments, and three of them per CU
__host__ __device__ constexpr int fa_static(int mt) { return mt >= 8 ? 6 : fa_max(mt); }
constexpr int SR_MAX = 256;  // weight rows per stage (tap group x BN)
constexpr int LDS_WG = 80 * 1024;

struct FastK {
    const void* x;
    const float *scale, *shift, *slope;
    const void* w;
    const float* bias;
    void* y;
    float* stats;
    int B, Hin, Win, Hx, Wx, ldx, cin_off, Cin, nchunks;
    int Hg, Wg, Hout, Wout, ldy, cout_off, Cout, Cout_pad, om, oy0, ox0;
    int ntaps, tg, ngroups, dy_min, dx_min, HH, HW, RS, magic;
    int tiles_x, tiles_y, nblocks_n, ntiles;
    int sA_bytes, a_bufs, sB_off, sB_bytes, tap_off, coef_off, cstride, stats_rows, accumulate, b_static, stg_off, red_off, dbg, stagger;
    int out_act; float out_slope;   // epilogue activation (BatchNorm folded into the weights: eval mode)
    const float* oscale;            // fp8 compute: per output channel, accumulator -> real value (s_in * s_w[n])
    const float* oquant;            // fp8 output: 1 / s_out (device), applied before the rounding to e4m3
    int oq_stride;                  // 0: a scalar; 1: one per n-block (128 output channels)
    const abc_heads_epi* hepi;      // HEPI: per n-block (= head) the 1x1 convolution computed in this tile's epilogue
    // ACTB (abc_conv_desc.actbwd_*): this data gradient is d(activation output) of the producing layer; the epilogue turns it into
    // d(BatchNorm output) and sums that layer's BatchNorm-backward statistics -- bn_act.hip's act_bwd pass, not run
    const void* ab_y; int ab_ld;    // the producer's raw convolution output (already at its channel 0), its row length
    const float *ab_sc, *ab_sh, *ab_sl, *ab_mu, *ab_is;
    int ystg_off;                   // a second staging region (the y_raw tile's way into the accumulator layout)
    int epi_off;                    // LP: the epilogue's per-channel tables, [2 rounds][1 (bias) + 5 (ACTB)][BN] floats
    unsigned bytesA, bytesW;
    long long* prof;  // debugging: per-workgroup phase timestamps (null in production)
    int prof_round;   // debugging: stamp the tile of this round only (< 0: every round, the last one stays)
    int8_t ty[ABC_MAX_TAPS], tx[ABC_MAX_TAPS];
};

__device__ inline void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Sum of v[k] over the 32 lanes of a wave half, for 16 values at once: a halving butterfly (each step a lane keeps half of its values and
// adds its partner's copies of them).  On return the lane with index r (0..31 inside its half) holds the total of v[(r >> 1) & 15]; the
// order of the additions is fixed.  Step 1 pairs lane r with r ^ 16 through v_permlane16_swap (rows 0 <-> 1 of the half: after the swap
// BOTH registers of a pair hold what this lane keeps -- its own and the partner's), steps 2-5 with r ^ 15, r ^ 7, r ^ 3, r ^ 1 through DPP.
__device__ inline float lane_reduce16(const float* v, int r) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float w[8], u[4], x[2];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const u32x2 p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[t]), __float_as_uint(v[t + 8]), false, false);
        w[t] = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    const bool b3 = (r >> 3) & 1, b2 = (r >> 2) & 1, b1 = (r >> 1) & 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) u[t] = (b3 ? w[t + 4] : w[t]) + dpp_mov<0x140>(b3 ? w[t] : w[t + 4]);      // row_mirror: r ^ 15
#pragma unroll
    for (int t = 0; t < 2; ++t) x[t] = (b2 ? u[t + 2] : u[t]) + dpp_mov<0x141>(b2 ? u[t] : u[t + 2]);      // row_half_mirror: r ^ 7
    const float y = (b1 ? x[1] : x[0]) + dpp_mov<0x1B>(b1 ? x[0] : x[1]);                                  // quad_perm [3,2,1,0]: r ^ 3
    return y + dpp_mov<0xB1>(y);                                                                            // quad_perm [1,0,3,2]: r ^ 1
}

// WD ("weights direct", 0 = off, else the tap count 9 or 25): the main loop below that streams the B operand from global
// memory (see there).
// HEPI ("heads in the epilogue", folded inference graph): the tile's 128 output channels are one head's finished features
// (unet.py:66-69 with BatchNorm folded); the head's 1x1 convolution (unet.py:70) is computed from them right here and the f32
// NCHW maps are stored -- the 8 x 128-channel feature tensor is never written or read (2 x 2.1 GB per batch of 64 at 512 x 512)
// M16 (round 5, the bf16 192-pixel x 128-channel weights-direct tile over whole tiles): the SAME tile, halo image, weight packing and
// ring, but multiplied with v_mfma_f32_16x16x32_bf16 -- one MFMA = the 16 pixels of a patch row x 16 output channels x the WHOLE 64-byte
// chunk of a tap (A: lane (m, q) = pixel m's bytes [16 q, 16 q + 16); B: row 16 ci + n's same bytes of the packed weights; four
// accumulator registers: channel n, pixels 4 q .. 4 q + 3 of the row).  The same FLOPs per cycle as the 32x32x16 form on paper; on the chip
// the long launches run power-limited (1.73-1.95 GHz) and the smaller shape costs less energy per FLOP (MI355X_MICROARCH.md: 1.15 x the
// FLOP/s in bare loops): a timing probe that issued two 16x16x32 per 32x32x16 on the same registers read 295 -> 245 us on the heads' conv1
// and 45 -> 41.5 us on a trunk layer before this form existed.  The epilogue differs in its lane mapping only (a lane owns two channels
// x eight pixels of a 32 x 32 block instead of one channel x sixteen pixels); the store sweep is the same.
template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT, bool STATIC, int WD = 0, int EPI = 0, int NW = 4, bool LP = false, int VAR = 0, bool M16 = false>
__device__ __forceinline__ void conv_fast_body(const FastK& a) {
    // NW waves per workgroup.  8 (the 192-pixel x 128-channel weights-direct tile only): the SAME tile, halo and grid, but a wave owns
    // 3 x 1 instead of 3 x 2 MFMA tiles -- 48 accumulator registers, the kernel fits 128 VGPRs, two workgroups = FOUR waves per SIMD:
    // a tile's life (prologue, main loop, epilogue) was what bounded the 4-wave form, whose lone wave per SIMD and workgroup runs
    // its main loop at a third of the matrix pipe's rate (profiles/HISTORY.md section 3)
    constexpr int FT = 64 * NW;
    static_assert(NW == 4 || (NW == 8 && WD == 9 && BN == 128 && MT == 6 && !STATIC), "8-wave workgroups: the 192 x 128 weights-direct tile");
    constexpr int CKB = CK * (int)sizeof(CT);
    constexpr int PS = CKB + 16;
    constexpr int LHB = CKB / 2;
    constexpr int NR = LHB / 16;
    constexpr int SEGS = CKB / 16;
    constexpr int NT = BN / 32;
    // VAR (experiments, measured in profiles/README.md round 5): bit 0 = the four waves side by side along N (1 x 4: a wave owns all of the
    // tile's pixels for 32 channels -- half the weight-fragment loads per MFMA, twice the LDS fragment reads); bit 1 = s_setprio around the MFMA groups
    constexpr bool W14 = (VAR & 1) != 0, SPRIO = (VAR & 2) != 0;
    static_assert(!W14 || (NW == 4 && NT == 4 && WD == 9), "1 x 4 waves: the 128-channel weights-direct tile");
    constexpr int WN = (NW == 8 || W14) ? 4 : ((NT >= 2) ? 2 : 1);
    constexpr int WM = NW / WN;
    constexpr int TM = MT / WM;
    constexpr int TN = NT / WN;
    static_assert(TM >= 1 && TM * WM == MT && TN >= 1 && TN * WN == NT, "tile/wave layout");
    constexpr int NB = (SR_MAX * SEGS + FT - 1) / FT;  // weight segments per thread per stage (a stage = <= SR_MAX weight rows)
    typedef typename Frag<CT>::type frag_t;
    constexpr bool F8C = sizeof(CT) == 1;    // e4m3 operands: one 32x32x64 MFMA per lane-half of a 64-byte chunk (weights-direct loop only)
    constexpr bool F8O = sizeof(OutT) == 1;  // e4m3 output: v * (1 / s_out), saturating
    static_assert(!F8C || WD == 9, "fp8 compute is served by the 9-tap weights-direct loop");
    constexpr bool HEPI = EPI == 1;     // the heads' 1x1 convolutions in the epilogue
    constexpr bool ACTB = EPI == 2;     // the activation / BatchNorm-statistics backward pass of the PRODUCER of this data gradient in the epilogue
    static_assert(!HEPI || (WD == 9 && BN == 128 && MT == 6 && sizeof(CT) <= 2 && NW == 4), "heads epilogue: the 192 x 128 weights-direct tile");
    static_assert(!ACTB || (!STATIC && sizeof(OutT) == 2 && sizeof(CT) == 2), "act_bwd epilogue: bf16 gradients, streamed weights");
    // LP ("lane = pixel"): the MFMA operands swapped -- A = the weight fragment, B = the pixel fragment, the SAME registers -- so that a
    // lane of the accumulator holds 16 CHANNELS of ONE pixel (register k = channel (k & 3) + 8 (k >> 2) + 4 h of the n-tile) instead of 16
    // pixels of one channel.  Four consecutive channels pack into 8 bytes, one v_permlane32_swap per dword pairs them with the other lane
    // half's four (cdna_hip_programming.md T21), and the tile leaves in 16-byte stores straight from registers: NO LDS staging, no wait on
    // an LDS round trip, no workgroup barrier between a tile's last MFMA and the next tile's first halo commit (the staging aliased the
    // halo buffers; the statistics went through an LDS reduction behind a barrier).  Per-channel sums now run ACROSS lanes: a halving
    // butterfly (v_permlane16_swap + DPP mirrors: 38 instructions per 16 channels, fixed order) leaves each channel's total in one lane, and
    // every wave writes its own partial row (rows per tile = WM).  Measured: profiles/README.md round 5.
    static_assert(!LP || (WD == 9 && !STATIC && !HEPI && sizeof(OutT) == 2 && sizeof(CT) == 2 && NW == 4), "lane = pixel: the bf16 weights-direct kernels");
    static_assert(!M16 || (WD == 9 && !STATIC && !HEPI && !LP && VAR == 0 && NW == 4 && STRIDE == 1 && sizeof(InT) == 2 && sizeof(CT) == 2 && sizeof(OutT) == 2 && BN >= 64),
                  "16x16x32 form: the bf16 weights-direct kernels, stride 1");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + a.sB_off;
    int* sTap = (int*)(smem + a.tap_off);
    float* sCoef = (float*)(smem + a.coef_off);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    long long* prof = ABC_PROF(a.prof ? a.prof + (size_t)blockIdx.x * 8 : nullptr);
    if (prof && tid == 0) { prof[0] = wall_clock64(); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); prof[6] = ((long long)xcc << 32) | hw; }

    if (tid < a.ntaps) sTap[tid] = a.ty[tid] * a.RS + a.tx[tid] * PS;
    const bool has_coef = a.scale != nullptr;
    if (has_coef) {
        for (int i = tid; i < a.Cin; i += FT) {
            sCoef[i] = a.scale[a.cin_off + i];
            sCoef[a.cstride + i] = a.shift[a.cin_off + i];
            sCoef[2 * a.cstride + i] = a.slope[a.cin_off + i];
        }
    }
    const float* lcoef = has_coef ? sCoef : nullptr;

    int aBase[TM], bBase[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int prow = 2 * (wm * TM + i) + (r >> 4), pcol = r & 15;
        aBase[i] = prow * STRIDE * a.RS + pcol * STRIDE * PS + h * LHB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) bBase[j] = ((wn * TN + j) * 32 + r) * PS + h * LHB;
    // M16: lane (m = lane % 16, q = lane / 16) reads bytes [16 q, 16 q + 16) of pixel m of patch row 2 (wm TM + i) (+ one halo row: row 1)
    const int m16 = lane & 15, q16 = lane >> 4;
    int aBase16[M16 ? TM : 1];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < TM; ++i) aBase16[i] = 2 * (wm * TM + i) * a.RS + m16 * PS + 16 * q16;
    }

    const __amdgpu_buffer_rsrc_t rsA = abc_make_rsrc(a.x, a.bytesA), rsW = abc_make_rsrc(a.w, a.bytesW);
    const HaloGeom gA = {a.HH, a.HW, a.magic, a.Hin, a.Win, a.Hx, a.Wx, a.ldx};
    const int nstages = a.nchunks * a.ngroups;

    // weight segments of a stage: segment i of a thread is segment 0 plus CONSTANT steps (source: scalar offset,
    // LDS: immediate offset), so one VGPR each for source and destination.  The kernel is bound by instruction issue:
    // the per-stage staging code must be loads / stores and nothing else.
    constexpr int SEGS_TAP = BN * SEGS;                              // 16-byte segments per tap slice
    constexpr int PER = SEGS_TAP >= FT ? SEGS_TAP / FT : 1;          // thread-segments per tap slice
    constexpr int TSTEP = SEGS_TAP >= FT ? 1 : FT / SEGS_TAP;        // tap slices covered by one round of the threads
    const unsigned tap_stride = (unsigned)(a.nchunks * a.Cout_pad * CK) * (unsigned)sizeof(CT);
    const int tl0 = tid / SEGS_TAP, rem0 = tid % SEGS_TAP;           // (tl0 = 0 when a slice has >= FT segments)
    const unsigned bvoff0 = (unsigned)tl0 * tap_stride + (unsigned)rem0 * 16u;
    const int bdst0 = (tl0 * BN + rem0 / SEGS) * PS + (rem0 % SEGS) * 16;
    u32x4 breg[2][NB];
    constexpr bool DEEP = WD == 9 && !F8C && !STATIC && NW == 4 && !LP && !M16 && TN == 1 && TM <= ABC_DEEP_TM;      // (see the weights-direct loop)
    // DEEP2 (measured, off): the halo of chunk c + 2 in flight while chunk c multiplies (two staging register sets, chunk k in set k & 1).
    // It does NOT help -- 24.8 -> 26.2 us on the 24 x 24 layers, 24.3 -> 26.7 us on the 12 x 12 ones: a chunk's end waits for the commit's
    // arithmetic and the barrier, not for the halo loads (profiles/README.md round 5)
    constexpr bool DEEP2 = false;
    HaloTile<InT, CT, CK, STATIC ? fa_static(MT) : (STRIDE == 2 ? fa_stride2() : (fa_max(MT) + NW / 4 - 1) / (NW / 4)), FT, DEEP2 ? 2 : 1> apre;

    unsigned w_n0 = 0;  // byte offset of the n-block's first weight row
    auto b_issue = [&](u32x4* set, int c, int g) {
        const unsigned soff = (unsigned)((g * a.tg * a.nchunks + c) * a.Cout_pad * CK) * (unsigned)sizeof(CT) + w_n0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned step = (unsigned)((i / PER) * TSTEP) * tap_stride + (unsigned)((i % PER) * FT * 16);
            set[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, bvoff0, soff + step, 0);
        }
    };
    auto b_commit = [&](const u32x4* set, int g, char* dst) {
        const int tcnt = min(a.tg, a.ntaps - g * a.tg);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            constexpr int dummy = 0; (void)dummy;
            const int tl = tl0 + (i / PER) * TSTEP;
            const int doff = ((i / PER) * TSTEP * BN + (i % PER) * (FT / SEGS)) * PS;
            if (tl < tcnt) *(u32x4*)(dst + bdst0 + doff) = set[i];
        }
    };

    __syncthreads();  // tap offsets + coefficient table visible
    bool first_tile = true;
    // resident-weight (persistent, one n-block) workgroups keep the statistics of ALL their tiles in registers and write ONE
    // partial row per workgroup: 768 rows for the finaliser instead of one per tile (9216 at 384 x 384, 17 us per finaliser)
    float pst1 = 0.f, pst2 = 0.f;
    // tile of this workgroup in round k (-1: none).  Full rounds: logical id + k * grid (XCD-contiguous ids).  The last,
    // partial round is dealt out in equal contiguous runs per XCD (the grid of a persistent launch is a multiple of 8), so
    // that it keeps all eight XCDs busy instead of filling the first ones.
    const int lid = abc_xcd_remap(blockIdx.x, gridDim.x);
    auto tile_of = [&](int k) -> int {
        const int G = (int)gridDim.x, base = k * G;
        if (base + G <= a.ntiles) return base + lid;
        const int R = a.ntiles - base;
        if (R <= 0) return -1;
        if (G & 7) return lid < R ? base + lid : -1;
        const int per = G >> 3, x = lid / per, i = lid - x * per;
        const int q = R >> 3, rem = R & 7;
        const int cnt = q + (x < rem ? 1 : 0), b0 = x * q + (x < rem ? x : rem);
        return i < cnt ? base + b0 + i : -1;
    };
    // LP with two halo buffers: the buffer parity runs on ACROSS tiles (a tile's first chunk goes into the buffer its predecessor's last
    // chunk did not use), so that a wave may commit the next tile's first chunk while others still read the last chunk of this one
    int cpar = 0;
    float* const sEpi = (float*)(smem + a.epi_off);
    constexpr int NTAB = ACTB ? 6 : 1;      // LP: rows of the epilogue table (bias; act_bwd: scale, shift, slope, mean, 1 / std)
    for (int round = 0, tile = tile_of(0); tile >= 0; tile = tile_of(++round)) {
        const bool pr = prof && tid == 0 && (a.prof_round < 0 || a.prof_round == round);
        if (pr) { prof[5] = wall_clock64(); prof[7] = round; }
        int id = tile;
        const int nb = id % a.nblocks_n; id /= a.nblocks_n;
        const int mblock = id;
        const int tx_i = id % a.tiles_x; id /= a.tiles_x;
        const int ty_i = id % a.tiles_y; id /= a.tiles_y;
        const int b = id;
        const int gy0 = ty_i * (2 * MT), gx0 = tx_i * 16;
        const int n0 = nb * BN;
        w_n0 = (unsigned)(n0 * CK) * (unsigned)sizeof(CT);
        const int iy0 = gy0 * STRIDE + a.dy_min, ix0 = gx0 * STRIDE + a.dx_min;
        float bv[TN];   // bias of this lane's output channels: loaded here, used in the epilogue (latency under the main loop)
        float osc[TN];  // fp8 compute: the lane's dequantisation factors
        bool nval[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + r;
            nval[j] = n < a.Cout;
            bv[j] = (!LP && !M16 && a.bias != nullptr && nval[j]) ? a.bias[n] : 0.f;
            osc[j] = (F8C && nval[j]) ? a.oscale[n] : 1.f;
        }
        // LP: the epilogue reads its per-channel constants from an LDS table of the n-block (a lane owns 16 channels of every n-tile);
        // loaded here, written behind the halo commit, visible after the barrier that opens the main loop.  Two copies (round parity):
        // a wave in the next tile's prologue must not overwrite what a slower wave's epilogue still reads.
        // (M16 too: its lanes own two channels per n-tile -- twice the lane constants of the 32x32 form, and held in registers across the main
        //  loop they pushed its operand addresses into scratch, reloaded between the MFMA groups behind the whole weight ring)
        constexpr bool ETAB = LP || M16;
        float etab[ETAB ? NTAB : 1];
        const bool etab_fill = ETAB && (first_tile || a.nblocks_n > 1) && tid < BN;
        if constexpr (ETAB) {
            if (etab_fill) {
                const int n = n0 + tid;
                const bool ok = n < a.Cout;
                etab[0] = (a.bias != nullptr && ok) ? a.bias[n] : 0.f;
                if constexpr (ACTB) {
                    etab[1] = ok ? a.ab_sc[n] : 0.f; etab[2] = ok ? a.ab_sh[n] : 0.f; etab[3] = ok ? a.ab_sl[n] : 0.f;
                    etab[4] = ok ? a.ab_mu[n] : 0.f; etab[5] = ok ? a.ab_is[n] : 0.f;
                }
            }
        }
        const float oq = F8O ? a.oquant[nb * a.oq_stride] : 1.f;
        float csc[ACTB ? TN : 1], csh[ACTB ? TN : 1], csl[ACTB ? TN : 1], cmu[ACTB ? TN : 1];
        if constexpr (ACTB && !LP && !M16) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + r;
                csc[j] = nval[j] ? a.ab_sc[n] : 0.f; csh[j] = nval[j] ? a.ab_sh[n] : 0.f;
                csl[j] = nval[j] ? a.ab_sl[n] : 0.f; cmu[j] = nval[j] ? a.ab_mu[n] : 0.f;
            }
        }

        // ---- prologue: chunk 0 halo, first two stages of weights (the CU's other workgroup computes meanwhile).
        // Persistent (resident-weight) workgroups prefetched this tile's halo during the previous tile.
        // (WD workgroups are persistent too when there are more tiles than workgroup slots: the next tile's first halo chunk
        //  is issued during the last chunk of this tile, see the main loop)
        if (!(STATIC || WD != 0) || first_tile) {
            apre.setup(gA, a.RS, PS, b, iy0, ix0, a.cin_off, abc_launder(tid));   // (laundered like the next-tile setup: its lane constants were spilled and came back as eight serialised scratch round trips in front of the first halo loads)
            apre.issue(rsA, 0u);
        }
        int ci = 0, gi = 0;  // (chunk, tap group) of the next stage to issue
        auto issue_next = [&](u32x4* set) {
            b_issue(set, ci, gi);
            if (++gi == a.ngroups) { gi = 0; ++ci; }
        };
        if constexpr (STATIC) {
            // resident weights: one chunk, every tap group loaded once per workgroup, straight to its place
            if (first_tile) {
                for (int g = 0; g < a.ngroups; ++g) {
                    b_issue(breg[0], 0, g);
                    b_commit(breg[0], g, sB + g * a.sB_bytes);
                }
            }
        } else if constexpr (WD == 0) {
            issue_next(breg[0]);
            issue_next(breg[1]);
        }
        // ---- WD: the packed weights [tap][chunk][Cout_pad][CK] ARE the MFMA B fragments (row n, bytes 32 h + 16 kk of a
        // 64-byte chunk row), so each lane loads its own fragments straight from global memory (L1 / L2: the CU's waves
        // all stream the same 295 KB) into a ring of 3 taps, 3 taps ahead of their use.  LDS then carries the
        // activation halo only: 3 instead of 5 ds_read_b128 per 6 MFMAs (the LDS pipe was as busy as the matrix
        // pipe), no weight staging writes, and one workgroup barrier per 64-byte chunk instead of one per tap pair.
        constexpr int WNT = WD ? WD : 1;              // taps (static: the loop is fully unrolled)
        // DEEP (small wave tiles: one or two MFMAs per 16-byte K-slice -- the 64-channel blocks of the 24 x 24 and 12 x 12 levels): a tap's
        // matrix work is 64-128 cycles, far less than an LDS round trip (~130) or a trip to the L2 (500-900).  The three-tap weight ring
        // and the half-tap fragment prefetch of the big tile leave every K-slice waiting on its operands (measured: these launches keep
        // the matrix pipe 10-20 % busy).  So: the weight fragments of a WHOLE chunk in flight (ring of 9: TN = 1, 72 registers -- these
        // instantiations use 120-150 of their 256) and the pixel fragments of tap t + 2 read while tap t multiplies.
        // Measured (same box, rocprofv3 of the graph run): the 64-channel blocks at 24 x 24 26.4 -> 24.8 us, at 12 x 12 29.3 -> 24.3 us.
        constexpr int RING = WD == 25 ? 5 : ((DEEP && TM <= 2) ? 9 : 3);        // taps in flight; divides the tap count
        static_assert(WNT % RING == 0 || !WD, "ring must divide the tap count");
        u32x4 bq[(WD && !F8C) ? RING : 1][TN][2];
        i32x8 bq8[F8C ? RING : 1][TN];     // e4m3: a tile's B operand is one 8-register tuple (both 16-byte halves)
        const unsigned chunk_stride = (unsigned)(a.Cout_pad * CK) * (unsigned)sizeof(CT);
        // (the wave's n-offset sits in the VGPR part: everything in the scalar offset must be provably wave-uniform, or
        //  the compiler wraps every load in a readfirstlane loop)
        // (abc_pack_desc.layout 1: a 32-row block is [kk][h][r][16 bytes] -- the 64 lanes of one load read 1 KB back to back)
        // (M16: the same packed block, lane (n, q): row 16 ci + n's piece q = bytes [16 q, 16 q + 16) of its chunk row, which layout 1 keeps at
        //  (q & 1) * 1024 + (q >> 1) * 512 + row * 16; the second index of bq is ci instead of the K half)
        const unsigned bq_voff = M16 ? (unsigned)(wn * TN * 32 * CKB + (q16 & 1) * 1024 + (q16 >> 1) * 512 + m16 * 16) : (unsigned)(wn * TN * 32 * CKB + h * 512 + r * 16);
        constexpr int BQ_STEP = M16 ? 256 : 1024;
        auto bq_load = [&](int slot, int c, int t) {
            const unsigned soff = w_n0 + (unsigned)t * tap_stride + (unsigned)c * chunk_stride;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (F8C) {
                    bq8[slot][j] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsW, bq_voff + (unsigned)(j * 32 * CKB), soff, 0),
                                               __builtin_amdgcn_raw_buffer_load_b128(rsW, bq_voff + (unsigned)(j * 32 * CKB + 1024), soff, 0));
                } else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        bq[slot][j][kk] = __builtin_amdgcn_raw_buffer_load_b128(rsW, bq_voff + (unsigned)(j * 32 * CKB + kk * BQ_STEP), soff, 0);
                }
            }
        };
        if constexpr (WD != 0) {
#pragma unroll
            for (int q = 0; q < RING; ++q) bq_load(q, 0, q);
        }
        if constexpr (DEEP2) {
            if (a.nchunks > 1) apre.template issue<1>(rsA, (unsigned)CK * (unsigned)sizeof(InT));
        }
        f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
        f32x4 acc16[M16 ? TM : 1][2][M16 ? TN : 1][2];      // [M-tile][patch row][n-tile][channel half]
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci) acc16[i][pi][j][ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        }
        if constexpr (!LP) {
            if (!first_tile) __syncthreads();  // previous tile's epilogue staging (aliases the halo buffer) is drained
        }
        apre.commit(sA + ((LP && a.a_bufs == 2) ? (cpar & 1) * a.sA_bytes : 0), lcoef, a.cstride, tid);
        if constexpr (ETAB) {
            if (etab_fill) {
                float* tb = sEpi + ((a.nblocks_n > 1) ? (round & 1) : 0) * (NTAB * BN) + tid;      // (one n-block: written once, never overwritten)
#pragma unroll
                for (int q = 0; q < NTAB; ++q) tb[q * BN] = etab[q];
            }
        }
        if constexpr (!STATIC && WD == 0) b_commit(breg[0], 0, sB);
        // (the epilogue's lane constants are waited for HERE, where the halo's vmcnt(0) has just passed: first used in the
        //  epilogue, their wait is a vmcnt(0) there -- the counter is in order -- which also waits for the NEXT tile's halo
        //  prefetch issued during the last chunk)
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (!M16) {
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bv[j]), "v"(osc[j]));
        }
        asm volatile("" :: "v"(oq));
        if constexpr (ACTB && !LP && !M16) {
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(csc[j]), "v"(csh[j]), "v"(csl[j]), "v"(cmu[j]));
        }
#endif
        first_tile = false;
        __syncthreads();
        if constexpr (STATIC) {
            // narrow layers are bound by the latency of one tile (10 KB in, 8 KB out, 18 MFMAs per wave): start the
            // next tile's halo now, it lands under this tile's MFMAs and stores
            const int nt = tile_of(round + 1);
            if (nt >= 0) {
                int id2 = nt / a.nblocks_n;
                const int tx2 = id2 % a.tiles_x; id2 /= a.tiles_x;
                const int ty2 = id2 % a.tiles_y; id2 /= a.tiles_y;
                apre.setup(gA, a.RS, PS, id2, ty2 * (2 * MT) * STRIDE + a.dy_min, tx2 * 16 * STRIDE + a.dx_min, a.cin_off, tid);
                apre.issue(rsA, 0u);
            }
        }
        if (pr) prof[1] = wall_clock64();

        // ---- main loop, unrolled by 2 so that the two register sets have fixed names.  Stage s: its weights sit in
        // sB[s & 1]; set s & 1 is free (committed at the end of stage s-1) and takes the loads of stage s + 2; the next
        // chunk's halo is issued when a chunk opens and committed when it closes.
        if constexpr (WD != 0) {
            static_assert(NR == 2 && !STATIC, "WD: 64-byte chunks, streamed weights");
            auto chunk = [&](auto PARV, const int c) {
                constexpr int PAR = decltype(PARV)::value;      // DEEP2: c & 1, the staging set that held this chunk and takes chunk c + 2
                const char* sAc = sA + ((a.a_bufs == 2) ? ((c + cpar) & 1) * a.sA_bytes : 0);
                const bool more = c + 1 < a.nchunks;
                if (DEEP2 ? (c + 2 < a.nchunks) : more) {
                    if constexpr (DEEP2) apre.template issue<PAR>(rsA, (unsigned)((c + 2) * CK) * (unsigned)sizeof(InT));
                    else apre.issue(rsA, (unsigned)((c + 1) * CK) * (unsigned)sizeof(InT));
                } else if (!more) {
                    // last chunk: the staging registers are free -> the NEXT tile's first halo chunk lands under this
                    // chunk's MFMAs and the epilogue (a workgroup's prologue measured 5 of its 30 us, with no MFMA issued)
                    const int nt = tile_of(round + 1);
                    if (nt >= 0) {
                        int id2 = nt / a.nblocks_n;
                        const int tx2 = id2 % a.tiles_x; id2 /= a.tiles_x;
                        const int ty2 = id2 % a.tiles_y; id2 /= a.tiles_y;
                        apre.setup(gA, a.RS, PS, id2, ty2 * (2 * MT) * STRIDE + a.dy_min, tx2 * 16 * STRIDE + a.dx_min, a.cin_off, abc_launder(tid));   // (laundered: hoisted out of the tile loop, the lane-constant part of the setup was SPILLED and its reload waited for the whole weight ring)
                        apre.issue(rsA, 0u);
                    }
                }
                if constexpr (F8C) {
                    // e4m3: a tap of a 64-byte chunk is ONE MFMA per tile pair (both 16-byte halves of the lane's 32 bytes at once);
                    // the fragments of tap t + 1 are read while the MFMAs of tap t run (two register sets, static after unrolling)
                    // (one A set: the CU's other waves -- two per SIMD, two workgroups -- cover the LDS latency; a second set of 24
                    //  registers spilled)
#pragma unroll
                    for (int t = 0; t < WNT; ++t) {
                        const int slot = t % RING;
                        const int aoff = a.ty[t] * a.RS + a.tx[t] * PS;
                        i32x8 fq[TM];
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            fq[i] = abc_join32B(*(const u32x4*)(sAc + aBase[i] + aoff), *(const u32x4*)(sAc + aBase[i] + aoff + 16));
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) mma32B_f8(acc[i][j], fq[i], bq8[slot][j]);
                        // hipcc treats the scaled MFMA as freely sinkable: without a use here it moved all 54 of a chunk behind the
                        // halo commit at the end of the chunk, with the nine taps' fragments (216 registers) spilled to scratch on
                        // the way.  An empty asm that READS the accumulators pins each tap's MFMAs to its place (no instruction, no stall).
                        // (device pass only: on the host pass a 64-byte "v" operand is not a valid x86 constraint, and clang then drops
                        //  the whole kernel stub without a diagnostic)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(acc[i][j]));
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        bq_load(slot, t + RING < WNT ? c : c + 1, t + RING < WNT ? t + RING : t + RING - WNT);
                    }
                } else if constexpr (DEEP) {
                    // pixel fragments: a ring of AD taps (both 16-byte halves), AD - 1 taps ahead of the MFMAs
                    constexpr int AD = TM <= 2 ? 3 : 2;
                    frag_t fr[AD][2][TM];
                    auto fr_read = [&](int t) {
                        const int aoff = a.ty[t] * a.RS + a.tx[t] * PS;
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int i = 0; i < TM; ++i) fr[t % AD][kk][i] = *(const frag_t*)(sAc + aBase[i] + aoff + 16 * kk);
                    };
#pragma unroll
                    for (int t = 0; t < AD - 1; ++t) fr_read(t);
#pragma unroll
                    for (int t = 0; t < WNT; ++t) {
                        if (t + AD - 1 < WNT) fr_read(t + AD - 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) mma16B(acc[i][j], fr[t % AD][kk][i], *(const frag_t*)&bq[t % RING][j][kk]);
                        __builtin_amdgcn_sched_barrier(0);
                        // the slot is free: tap t + RING (of this chunk or the next; past the last chunk the offsets run off the buffer and
                        // the loads return zeros -- unconditional, so that vmcnt stays exact)
                        bq_load(t % RING, t + RING < WNT ? c : c + 1, t + RING < WNT ? t + RING : t + RING - WNT);
                    }
                } else if constexpr (M16) {
                    // a step = (tap t, M-tile i): the fragments of the M-tile's two patch rows (read one step ahead: two register sets) times the
                    // tap's 2 TN channel-half fragments: 4 TN MFMAs of 16 cycles
                    frag_t fp[2][2];
                    auto fp_read = [&](int set, int t, int i) {
                        const int aoff = a.ty[t] * a.RS + a.tx[t] * PS;
                        fp[set][0] = *(const frag_t*)(sAc + aBase16[i] + aoff);
                        fp[set][1] = *(const frag_t*)(sAc + aBase16[i] + a.RS + aoff);
                    };
                    fp_read(0, 0, 0);
#pragma unroll
                    for (int t = 0; t < WNT; ++t) {
                        const int slot = t % RING;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int st = (t * TM + i) & 1;
                            if (i + 1 < TM) fp_read(st ^ 1, t, i + 1);
                            else if (t + 1 < WNT) fp_read(st ^ 1, t + 1, 0);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                                for (int j = 0; j < TN; ++j)
#pragma unroll
                                    for (int ci = 0; ci < 2; ++ci)
                                        acc16[i][pi][j][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[st][pi], *(const frag_t*)&bq[slot][j][ci], acc16[i][pi][j][ci], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        bq_load(slot, t + RING < WNT ? c : c + 1, t + RING < WNT ? t + RING : t + RING - WNT);
                    }
                } else {
                frag_t fa0[TM], fa1[TM];
                // (tap offsets from the kernel arguments: scalar registers, no LDS round trip in front of the fragment reads)
                {
                    const int aoff = a.ty[0] * a.RS + a.tx[0] * PS;
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa0[i] = *(const frag_t*)(sAc + aBase[i] + aoff);
                }
#pragma unroll
                for (int t = 0; t < WNT; ++t) {
                    const int slot = t % RING;
                    const int aoff = a.ty[t] * a.RS + a.tx[t] * PS;
                    const int aoff_n = a.ty[t < WNT - 1 ? t + 1 : WNT - 1] * a.RS + a.tx[t < WNT - 1 ? t + 1 : WNT - 1] * PS;
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa1[i] = *(const frag_t*)(sAc + aBase[i] + aoff + 16);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (SPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (LP) mma16B(acc[i][j], *(const frag_t*)&bq[slot][j][0], fa0[i]);
                            else mma16B(acc[i][j], fa0[i], *(const frag_t*)&bq[slot][j][0]);
                        }
                    if constexpr (SPRIO) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t < WNT - 1) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa0[i] = *(const frag_t*)(sAc + aBase[i] + aoff_n);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (SPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (LP) mma16B(acc[i][j], *(const frag_t*)&bq[slot][j][1], fa1[i]);
                            else mma16B(acc[i][j], fa1[i], *(const frag_t*)&bq[slot][j][1]);
                        }
                    if constexpr (SPRIO) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    // the slot is free: tap t + RING (of this chunk or the next; past the last chunk the offsets run off
                    // the buffer and the loads return zeros -- unconditional, so that vmcnt stays exact)
                    bq_load(slot, t + RING < WNT ? c : c + 1, t + RING < WNT ? t + RING : t + RING - WNT);
                }
                }
                if (more) {
                    constexpr int CS = DEEP2 ? 1 - PAR : 0;      // the set that holds chunk c + 1
                    if (a.a_bufs == 2) {
                        apre.template commit<CS>(sA + ((c + 1 + cpar) & 1) * a.sA_bytes, lcoef ? lcoef + (c + 1) * CK : nullptr, a.cstride, tid);
                    } else {
                        __syncthreads();
                        apre.template commit<CS>(sA, lcoef ? lcoef + (c + 1) * CK : nullptr, a.cstride, tid);
                    }
                }
                // next chunk's halo visible; after the last chunk: the halo is dead (the epilogue aliases it) -- LP's epilogue uses no
                // LDS staging, and with two halo buffers the next tile's first chunk goes into the OTHER buffer: no barrier there
                if (!(LP && !more && a.a_bufs == 2)) __syncthreads();
            };
            if constexpr (DEEP2) {
                // (two chunks per trip: the staging set of a chunk is a compile-time index)
                for (int c = 0; c < a.nchunks; c += 2) {
                    chunk(std::integral_constant<int, 0>{}, c);
                    if (c + 1 < a.nchunks) chunk(std::integral_constant<int, 1>{}, c + 1);
                }
            } else {
                for (int c = 0; c < a.nchunks; ++c) chunk(std::integral_constant<int, 0>{}, c);
            }
        } else {
        int c = 0, g = 0;
        for (int s0 = 0; s0 < nstages; s0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int s = s0 + d;
                if (s < nstages) {
                    int gn = g + 1, cn = c;
                    if (gn == a.ngroups) { gn = 0; cn = c + 1; }
                    const bool has_next = (s + 1 < nstages);
                    const bool closes = has_next && (gn == 0);  // last stage of a chunk that has a successor
                    // (unconditional: past the last stage the offsets run off the buffer and the loads return zeros;
                    //  a conditional issue would make the compiler drain ALL loads before every commit)
                    if constexpr (!STATIC) issue_next(breg[d]);
                    if (g == 0 && c + 1 < a.nchunks && !(ABC_DBG(a.dbg) & 2)) apre.issue(rsA, (unsigned)((c + 1) * CK) * (unsigned)sizeof(InT));

                    {
                        const char* sAc = sA + ((a.a_bufs == 2) ? (c & 1) * a.sA_bytes : 0);
                        const char* sBc = STATIC ? sB + g * a.sB_bytes : sB + d * a.sB_bytes;
                        const int t0 = g * a.tg;
                        const int tcnt = min(a.tg, a.ntaps - t0);
                        // fragment reads software-pipelined one K-step ahead of the MFMAs (two named register sets;
                        // every read unconditional -- the step after the last re-reads the last tap -- so that the
                        // compiler can count lgkmcnt exactly instead of draining the LDS queue before each MFMA group)
                        const int ntl = (ABC_DBG(a.dbg) & 4) ? 0 : tcnt;
                        if constexpr (NR == 2) {
                            frag_t fa0[TM], fb0[TN], fa1[TM], fb1[TN];
                            int aoff = sTap[t0];
#pragma unroll
                            for (int i = 0; i < TM; ++i) fa0[i] = *(const frag_t*)(sAc + aBase[i] + aoff);
#pragma unroll
                            for (int j = 0; j < TN; ++j) fb0[j] = *(const frag_t*)(sBc + bBase[j]);
                            for (int tl = 0; tl < ntl; ++tl) {
                                const int tn = min(tl + 1, tcnt - 1);
                                const int boff = tl * BN * PS;
                                const int aoff_n = sTap[t0 + tn];
#pragma unroll
                                for (int i = 0; i < TM; ++i) fa1[i] = *(const frag_t*)(sAc + aBase[i] + aoff + 16);
#pragma unroll
                                for (int j = 0; j < TN; ++j) fb1[j] = *(const frag_t*)(sBc + bBase[j] + boff + 16);
                                __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of the MFMA group (the scheduler sinks them otherwise)
#pragma unroll
                                for (int i = 0; i < TM; ++i)
#pragma unroll
                                    for (int j = 0; j < TN; ++j) mma16B(acc[i][j], fa0[i], fb0[j]);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int i = 0; i < TM; ++i) fa0[i] = *(const frag_t*)(sAc + aBase[i] + aoff_n);
#pragma unroll
                                for (int j = 0; j < TN; ++j) fb0[j] = *(const frag_t*)(sBc + bBase[j] + tn * BN * PS);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int i = 0; i < TM; ++i)
#pragma unroll
                                    for (int j = 0; j < TN; ++j) mma16B(acc[i][j], fa1[i], fb1[j]);
                                __builtin_amdgcn_sched_barrier(0);
                                aoff = aoff_n;
                            }
                        } else {
                            for (int tl = 0; tl < ntl; ++tl) {
                                const int aoff = sTap[t0 + tl];
                                const int boff = tl * BN * PS;
                                frag_t fa[TM], fb[TN];
#pragma unroll
                                for (int i = 0; i < TM; ++i) fa[i] = *(const frag_t*)(sAc + aBase[i] + aoff);
#pragma unroll
                                for (int j = 0; j < TN; ++j) fb[j] = *(const frag_t*)(sBc + bBase[j] + boff);
#pragma unroll
                                for (int i = 0; i < TM; ++i)
#pragma unroll
                                    for (int j = 0; j < TN; ++j) mma16B(acc[i][j], fa[i], fb[j]);
                            }
                        }
                    }

                    if constexpr (!STATIC) {
                        if (has_next && !(ABC_DBG(a.dbg) & 8)) b_commit(breg[1 - d], gn, sB + (1 - d) * a.sB_bytes);
                    }
                    if (closes && !(ABC_DBG(a.dbg) & 16)) {
                        if (a.a_bufs == 2) {
                            apre.commit(sA + (cn & 1) * a.sA_bytes, lcoef ? lcoef + cn * CK : nullptr, a.cstride, tid);
                        } else {
                            __syncthreads();  // every wave is done reading this chunk's halo
                            apre.commit(sA, lcoef ? lcoef + cn * CK : nullptr, a.cstride, tid);
                        }
                    }
                    if constexpr (!STATIC) __syncthreads();  // (resident weights + single halo chunk: nothing changes hands)
                    c = cn; g = gn;
                }
            }
        }
        }

        if (pr) prof[2] = wall_clock64();
        if constexpr (HEPI) {
            // ---- 1. the tile's activated features, [pixel][128 channels] in the compute type, into LDS (the halo buffers are dead)
            constexpr int FROW = 128 * (int)sizeof(CT) + 16;
            char* F = smem;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float v = F8C ? fmaf(acc[i][j][k], osc[j], bv[j]) : acc[i][j][k] + bv[j];
                        float vo = a.out_act ? fmaxf(v, a.out_slope * v) : v;
                        if constexpr (F8C) vo *= oq;
                        const int p = 32 * (wm * TM + i) + (k & 3) + 8 * (k >> 2) + 4 * h;
                        *(CT*)(F + p * FROW + ((wn * TN + j) * 32 + r) * (int)sizeof(CT)) = (CT)vo;
                    }
            __syncthreads();
            // ---- 2. the head's 1x1 convolution: logits[co][p] = b[co] + sum_c W2[co][c] F[p][c].  Work items = (m-tile of 32 output
            // rows) x (one of the tile's six 32-pixel n-tiles), dealt to the four waves: m-tile-major when the head has >= 4 m-tiles
            // (the weight fragments of an m-tile are fetched once and serve its six n-tiles), item by item for the narrow heads (one or
            // two m-tiles: otherwise one wave would do all the work).  Stores: one raw buffer store per accumulator register -- the
            // lane's part of the address (image, its pixel, its row half) is ONE offset per n-tile, the register's output row a
            // scalar offset; rows past Cout and pixels past the map get the out-of-range offset and are dropped by the hardware.
            // MEASURED (b64 at 512 x 512, same box): exact, and slower than conv1 + the separate heads kernel in both forms tried --
            // 64-bit address per store, items dealt round-robin: 7.86 -> 9.02 ms bf16, 6.09 -> 7.68 ms e4m3; this form: 10.4 / 9.6 ms.
            // The epilogue is a serial chain on the tile's critical path (head record -> weight fragments -> MFMAs -> 4-byte-per-lane
            // stores in 64-byte runs) in a kernel with two workgroups per CU and nothing to cover it, and its arrays cost the main
            // loop registers (25 / 92 spilled).  The engine therefore plans it only on request (Engine(heads_epilogue=True)).
            const abc_heads_epi& he = a.hepi[nb];
            const int mtiles = he.Cout_pad >> 5;
            const unsigned HWp = (unsigned)(a.Hg * a.Wg);
            const __amdgpu_buffer_rsrc_t rsW2 = abc_make_rsrc(he.w2, (unsigned)(128 * he.Cout_pad * (int)sizeof(CT)));
            const __amdgpu_buffer_rsrc_t rsY = abc_make_rsrc(he.y, (unsigned)a.B * (unsigned)he.Cout * HWp * 4u);
            const int nitems = mtiles * 6;
            const bool mt_major = mtiles >= 4;
            int last_mt = -1;
            float bvv[16], os2[16];
            i32x8 fa8[F8C ? 2 : 1];
            bf16x8 fa16[F8C ? 1 : 8];
            for (int it = mt_major ? wave * 6 : wave; it < nitems; it += mt_major ? (((it + 1) % 6) ? 1 : 19) : 4) {
                const int mt = it / 6, nt = it - mt * 6;
                if (mt != last_mt) {
                    last_mt = mt;
                    const int co = mt * 32 + r;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int oc = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * h;
                        const bool ok = oc < he.Cout;
                        bvv[k] = (he.bias && ok) ? he.bias[oc] : 0.f;
                        os2[k] = (F8C && ok) ? he.oscale[oc] : 1.f;
                    }
                    if constexpr (F8C) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const unsigned off = (unsigned)((s2 * he.Cout_pad + co) * 64 + 32 * h);
                            fa8[s2] = abc_join32B(__builtin_amdgcn_raw_buffer_load_b128(rsW2, off, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsW2, off + 16, 0, 0));
                        }
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            const unsigned off = (unsigned)((((kk >> 1) * he.Cout_pad + co) * 32 + 16 * (kk & 1) + 8 * h) * 2);
                            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsW2, off, 0, 0);
                            fa16[kk] = *(const bf16x8*)&t;
                        }
                    }
                }
                f32x16 c2;
#pragma unroll
                for (int k = 0; k < 16; ++k) c2[k] = 0.f;
                if constexpr (F8C) {
                    const char* fp = F + (32 * nt + r) * FROW + 32 * h;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        mma32B_f8(c2, fa8[s2], abc_join32B(*(const u32x4*)(fp + 64 * s2), *(const u32x4*)(fp + 64 * s2 + 16)));
                } else {
                    const char* fp = F + (32 * nt + r) * FROW + 16 * h;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa16[kk], *(const bf16x8*)(fp + 32 * kk), c2, 0, 0, 0);
                }
                const int pp = 32 * nt + r, gy = gy0 + (pp >> 4), gx = gx0 + (pp & 15);
                // lane part: image b, output row 4 h of the m-tile's first row, pixel (gy, gx); register part: row (k & 3) + 8 (k >> 2)
                const unsigned vlane = ((unsigned)(b * he.Cout + mt * 32 + 4 * h) * HWp + (unsigned)(gy * a.Wg + gx)) * 4u;
                const bool pix_ok = gy < a.Hg && gx < a.Wg;
                const bool full = mt * 32 + 32 <= he.Cout;     // (wave-uniform: only a head's last m-tile can be partial)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int rowk = (k & 3) + 8 * (k >> 2);
                    const bool ok = pix_ok && (full || mt * 32 + rowk + 4 * h < he.Cout);
                    const float v = F8C ? fmaf(c2[k], os2[k], bvv[k]) : c2[k] + bvv[k];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, ok ? vlane : 0xFFFFFFFCu, (unsigned)rowk * HWp * 4u, 0);
                }
            }
            continue;      // (the next tile's prologue synchronises before it re-uses the LDS; no statistics in the folded graph)
        }
        if constexpr (LP) {
            // ---- lane = pixel epilogue (see LP above): everything from registers.  Lane (r, h) holds pixel r of M-tile i -- patch row
            // 2 (wm TM + i) + (r >> 4), column r & 15 -- and, of n-tile j, channels 8 q + 4 h + e in registers 4 q + e (q, e = 0..3).
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            constexpr int TW = TN * 32;
            const int tl = abc_launder(tid), ll = tl & 63, wl = tl >> 6, rl = ll & 31, hl = ll >> 5;
            const int wml = wl / WN, wnl = wl % WN;
            const float* tab = sEpi + ((a.nblocks_n > 1) ? (round & 1) : 0) * (NTAB * BN) + wnl * TW + 4 * hl;     // + j * 32 + 8 q: the lane's four channels
            OutT* yo = (OutT*)a.y;
            const OutT* ytile = yo + (((size_t)(b * a.Hout + gy0 * a.om + a.oy0) * a.Wout + gx0 * a.om + a.ox0) * a.ldy + a.cout_off + n0);
            const __amdgpu_buffer_rsrc_t rsY = abc_make_rsrc(ytile, 0x80000000u);
            const unsigned istep = (unsigned)(2 * a.om * a.Wout * a.ldy) * (unsigned)sizeof(OutT);   // bytes per M-tile (two pixel rows)
            // a store = 16 bytes per lane: lanes < 32 the channels 16 p .. 16 p + 7 of their pixel, lanes >= 32 the next eight
            const bool col_ok = gx0 + (rl & 15) < a.Wg;
            const int rlim = a.Hg - gy0 - (rl >> 4);          // patch rows 2 m with 2 m < rlim lie inside the map (for this lane's row parity)
            const unsigned voff0 = (unsigned)(wml * TM) * istep +
                                   (unsigned)((((rl >> 4) * a.Wout + (rl & 15)) * a.om * a.ldy + wnl * TW + 8 * hl) * (int)sizeof(OutT));
            const float slope = a.out_act ? a.out_slope : 1.f;
            const bool want_stats = a.stats != nullptr;
            const bool whole_tile = (gy0 + 2 * MT <= a.Hg) && (gx0 + 16 <= a.Wg);      // (wave-uniform)
            // Loop order: (n-tile j, channel octet pair p2) outside, the TM M-tiles inside -- the per-channel constants of 8 channels stay in
            // registers for three stores; the statistics of an n-tile are complete after its two octet pairs.
            // ACTB: y_raw of this tile in the STORE layout (16 bytes = eight channels of the lane's pixel); the TM loads of an octet pair
            // are issued one pair ahead of their use.
            __amdgpu_buffer_rsrc_t rsYR = rsY;
            unsigned voffy = 0, istep_y = 0;
            u32x4 yq[ACTB ? 2 : 1][ACTB ? TM : 1];
            auto yq_load = [&](int set, int jp) {      // jp = 2 j + p2
                if constexpr (ACTB) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        yq[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsYR, voffy, (unsigned)i * istep_y + (unsigned)(jp * 16 * 2), 0);
                }
            };
            if constexpr (ACTB) {
                const bf16* yrt = (const bf16*)a.ab_y + (((size_t)(b * a.Hg + gy0) * a.Wg + gx0) * a.ab_ld + n0);
                rsYR = abc_make_rsrc(yrt, 0x80000000u);
                istep_y = (unsigned)(2 * a.Wg * a.ab_ld) * 2u;
                voffy = (unsigned)(wml * TM) * istep_y + (unsigned)((((rl >> 4) * a.Wg + (rl & 15)) * a.ab_ld + wnl * TW + 8 * hl) * 2);
                yq_load(0, 0);
            }
            float s1[16], s2[16];
            float* srow = want_stats ? a.stats + ((size_t)(mblock * WM + wml) * 2) * a.Cout + n0 + wnl * TW : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int jp = 2 * j + p2;
                    if constexpr (ACTB) {
                        if (jp + 1 < 2 * TN) yq_load((jp + 1) & 1, jp + 1);
                    }
                    // the lane's constants of quads 2 p2, 2 p2 + 1 of n-tile j: 16-byte LDS reads (four consecutive channels each)
                    f32x4 cb[2], c_sc[ACTB ? 2 : 1], c_sh[ACTB ? 2 : 1], c_sl[ACTB ? 2 : 1], c_mu[ACTB ? 2 : 1];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const float* tq = tab + j * 32 + 8 * (2 * p2 + qq);
                        cb[qq] = *(const f32x4*)tq;
                        if constexpr (ACTB) {
                            c_sc[qq] = *(const f32x4*)(tq + 1 * BN); c_sh[qq] = *(const f32x4*)(tq + 2 * BN);
                            c_sl[qq] = *(const f32x4*)(tq + 3 * BN); c_mu[qq] = *(const f32x4*)(tq + 4 * BN);
                        }
                    }
                    const bool chan_ok = n0 + wnl * TW + j * 32 + p2 * 16 + 8 * hl < a.Cout;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const bool pix_ok = col_ok && (2 * (wml * TM + i) < rlim);
                        unsigned xa[ACTB ? 4 : 1];      // y_raw in the accumulator layout: word 2 qq + w = channels 8 (2 p2 + qq) + 4 h + 2 w, + 1
                        if constexpr (ACTB) {
                            const u32x4 yv4 = yq[jp & 1][i];
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                // the inverse of the store's swap (an involution): the store layout's 16 bytes -> quads 2 p2 (lanes' own h) and 2 p2 + 1
                                const u32x2 t = __builtin_amdgcn_permlane32_swap(yv4[w], yv4[2 + w], false, false);
                                xa[w] = t[0]; xa[2 + w] = t[1];
                            }
                        }
                        unsigned d[2][2];
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                const int k = 4 * (2 * p2 + qq) + 2 * w;
                                const f32x2 v = (f32x2){acc[i][j][k], acc[i][j][k + 1]} + (f32x2){cb[qq][2 * w], cb[qq][2 * w + 1]};
                                f32x2 vo;
                                if constexpr (ACTB) {
                                    // g = dA where BatchNorm(y_raw) > 0, slope * dA elsewhere (unet.py:14,17 backward); sums of g and g (y_raw - mean)
                                    const unsigned xw = xa[2 * qq + w];
                                    const f32x2 x = {__uint_as_float(xw << 16), __uint_as_float(xw & 0xFFFF0000u)};
                                    const f32x2 yv = __builtin_elementwise_fma(x, (f32x2){c_sc[qq][2 * w], c_sc[qq][2 * w + 1]}, (f32x2){c_sh[qq][2 * w], c_sh[qq][2 * w + 1]});
                                    const f32x2 f = {yv.x > 0.f ? 1.f : c_sl[qq][2 * w], yv.y > 0.f ? 1.f : c_sl[qq][2 * w + 1]};
                                    vo = v * f;
                                    const f32x2 xm = x - (f32x2){c_mu[qq][2 * w], c_mu[qq][2 * w + 1]};
                                    s1[k] += vo.x; s1[k + 1] += vo.y;
                                    s2[k] = fmaf(vo.x, xm.x, s2[k]); s2[k + 1] = fmaf(vo.y, xm.y, s2[k + 1]);
                                } else {
                                    if (want_stats) {
                                        const f32x2 vs = (whole_tile || pix_ok) ? v : (f32x2){0.f, 0.f};
                                        s1[k] += vs.x; s1[k + 1] += vs.y;
                                        s2[k] = fmaf(vs.x, vs.x, s2[k]); s2[k + 1] = fmaf(vs.y, vs.y, s2[k + 1]);
                                    }
                                    const f32x2 m = v * slope;
                                    vo = (f32x2){fmaxf(v.x, m.x), fmaxf(v.y, m.y)};
                                }
                                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                                const bf16x2 pk = __builtin_convertvector(vo, bf16x2);
                                d[qq][w] = *(const unsigned*)&pk;
                            }
                        // quads (2 p2, 2 p2 + 1) -> eight consecutive channels per lane: lanes < 32 keep their quad 2 p2 and take the upper half's,
                        // lanes >= 32 take the lower half's quad 2 p2 + 1 and keep their own (T21)
                        u32x4 st;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const u32x2 t = __builtin_amdgcn_permlane32_swap(d[0][w], d[1][w], false, false);
                            st[w] = t[0]; st[2 + w] = t[1];
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(st, rsY, (pix_ok && chan_ok && !(ABC_DBG(a.dbg) & 128)) ? voff0 : 0xFFFFFFF0u, (unsigned)i * istep + (unsigned)(jp * 16 * (int)sizeof(OutT)), 0);
                        // (the store-data hazard of the 4-wave epilogue below: keep the data registers live two wait states past the store)
#if defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("s_nop 1" :: "v"(st));
#endif
                    }
                }
                if (want_stats) {
                    // per-channel totals over the wave's 32 x TM pixels: a halving butterfly over the 32 lanes of a half -- after it lane r
                    // holds the total of its half's channel k = r >> 1 (both lanes of a pair the same), summed in a fixed order
                    const int kk = (rl >> 1) & 15;
                    const int cl = (kk & 3) + 8 * (kk >> 2) + 4 * hl;       // channel of the n-tile
                    float t1 = lane_reduce16(s1, rl), t2 = lane_reduce16(s2, rl);
                    if constexpr (ACTB) t2 *= tab[5 * BN - 4 * hl + j * 32 + cl];      // (the row act_bwd writes: sum of g (y_raw - mean) / std)
                    if (!(rl & 1) && n0 + wnl * TW + j * 32 + cl < a.Cout) { srow[j * 32 + cl] = t1; srow[a.Cout + j * 32 + cl] = t2; }
                }
            }
            if (pr) prof[3] = wall_clock64();
            if (pr) prof[4] = wall_clock64();
            if (a.a_bufs == 2) cpar += a.nchunks;
            continue;
        }
        if constexpr (M16) {
            // ---- the epilogue below in the 16x16 accumulator layout: lane (m16, q16) holds, of every 32-pixel x 32-channel block (i, j),
            // channels 16 ci + m16 (ci = 0, 1) at pixels 16 pi + 4 q16 + e (pi = patch row, e = 0..3).  Whole tiles only (host: abc_fast_geom.m16):
            // values -> the wave's staging tile [32 pixels][TW channels], then the SAME 16-byte store sweep.
            OutT* yo = (OutT*)a.y;
            constexpr int TW = TN * 32;
            constexpr int ROWB = TW * (int)sizeof(OutT) + 16;
            constexpr int EV = 16 / (int)sizeof(OutT);
            constexpr int SEG_PER_ROW = TW / EV;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const int tl = abc_launder(tid), ll = tl & 63, wl = tl >> 6;
            const int cl = ll & 15, q4 = ll >> 4;
            char* stgw = smem + a.stg_off + wl * (32 * ROWB);
            char* wbase = stgw + 4 * q4 * ROWB + cl * (int)sizeof(OutT);      // + (16 pi + e) rows, + (32 j + 16 ci) channels
            const int lrow = ll / SEG_PER_ROW, lsg = ll % SEG_PER_ROW;
            constexpr int RSTEP = 64 / SEG_PER_ROW;
            constexpr int NST = 32 / RSTEP;
            const bool seg_ok = n0 + (wl % WN) * TW + lsg * EV < a.Cout;
            const OutT* ytile = yo + (((size_t)(b * a.Hout + gy0 * a.om + a.oy0) * a.Wout + gx0 * a.om + a.ox0) * a.ldy + a.cout_off + n0);
            const __amdgpu_buffer_rsrc_t rsY = abc_make_rsrc(ytile, 0x80000000u);
            const unsigned istep = (unsigned)(2 * a.om * a.Wout * a.ldy) * (unsigned)sizeof(OutT);
            unsigned voff[NST];
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int rit = lrow + RSTEP * st;
                voff[st] = seg_ok ? (unsigned)((wl / WN) * TM) * istep + (unsigned)((((rit >> 4) * a.Wout + (rit & 15)) * a.om * a.ldy + (wl % WN) * TW + lsg * EV) * (int)sizeof(OutT))
                                  : 0xFFFFFFF0u;
            }
            f32x2 s1v[TN][2], s2v[TN][2];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) { s1v[j][ci] = (f32x2){0.f, 0.f}; s2v[j][ci] = (f32x2){0.f, 0.f}; }
            // the lane's per-channel constants: rows of the n-block's LDS table (bias; act_bwd: scale, shift, slope, mean), channel (wn TN + j) 32 + 16 ci + cl
            const float* tab = sEpi + ((a.nblocks_n > 1) ? (round & 1) : 0) * (NTAB * BN) + (wl % WN) * TW + cl;
            float bv16[TN][2], csc16[ACTB ? TN : 1][2], csh16[ACTB ? TN : 1][2], csl16[ACTB ? TN : 1][2], cmu16[ACTB ? TN : 1][2];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const float* tq = tab + j * 32 + 16 * ci;
                    bv16[j][ci] = tq[0];
                    if constexpr (ACTB) { csc16[j][ci] = tq[1 * BN]; csh16[j][ci] = tq[2 * BN]; csl16[j][ci] = tq[3 * BN]; cmu16[j][ci] = tq[4 * BN]; }
                }
            const float slope = a.out_act ? a.out_slope : 1.f;
            const bool rows_ok = gy0 + 2 * MT <= a.Hg;
            const int rlim = (ACTB || rows_ok) ? 0x7FFFFFFF : a.Hg - gy0;
            constexpr int YROW = 32 * 2 + 8;
            static_assert(!ACTB || TW * YROW <= 32 * ROWB, "the transposed y_raw tile fits the staging region");
            char* ystg = smem + a.ystg_off + wl * (32 * ROWB);
            const char* yrd = ystg + cl * YROW + 8 * q4;      // + (32 j + 16 ci) channels, + 32 pi bytes (16 pixels)
            __amdgpu_buffer_rsrc_t rsYR = rsY;
            unsigned voffy[ACTB ? NST : 1], istep_y = 0;
            u32x4 yq[ACTB ? NST : 1];
            if constexpr (ACTB) {
                const bf16* yrt = (const bf16*)a.ab_y + (((size_t)(b * a.Hg + gy0) * a.Wg + gx0) * a.ab_ld + n0);
                rsYR = abc_make_rsrc(yrt, 0x80000000u);
                istep_y = (unsigned)(2 * a.Wg * a.ab_ld) * 2u;
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int rit = lrow + RSTEP * st;
                    voffy[st] = seg_ok ? (unsigned)((wl / WN) * TM) * istep_y + (unsigned)((((rit >> 4) * a.Wg + (rit & 15)) * a.ab_ld + (wl % WN) * TW + lsg * EV) * 2)
                                       : 0xFFFFFFF0u;
                    yq[st] = __builtin_amdgcn_raw_buffer_load_b128(rsYR, voffy[st], 0u, 0);
                }
            }
            // (OACT: an activation in the epilogue -- the folded inference graph; without it the multiply and the two maxima per value pair are
            //  not issued: the epilogue is VALU issue with both workgroups of a CU in the same phase)
            auto store_tiles = [&](auto OACTV, auto WSTATV) {
                constexpr bool OACT = decltype(OACTV)::value;
                constexpr bool WSTAT = decltype(WSTATV)::value;      // the sums and squares are wanted (not in the inference graph)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (ACTB) {
#pragma unroll
                    for (int st = 0; st < NST; ++st) {
                        char* w0 = ystg + (lsg * EV) * YROW + (lrow + RSTEP * st) * 2;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            *(unsigned short*)(w0 + (2 * c) * YROW) = (unsigned short)(yq[st][c] & 0xFFFFu);
                            *(unsigned short*)(w0 + (2 * c + 1) * YROW) = (unsigned short)(yq[st][c] >> 16);
                        }
                    }
                    if (i + 1 < TM) {
#pragma unroll
                        for (int st = 0; st < NST; ++st) yq[st] = __builtin_amdgcn_raw_buffer_load_b128(rsYR, voffy[st], (unsigned)(i + 1) * istep_y, 0);
                    }
                    lds_wave_sync();
                }
                unsigned xq[ACTB ? TN : 1][2][2][ACTB ? 2 : 1];      // [j][ci][pi][pixel pair]: y_raw of the lane's values
                if constexpr (ACTB) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                            for (int pi = 0; pi < 2; ++pi) {
                                const u32x2 t = *(const u32x2*)(yrd + (j * 32 + 16 * ci) * YROW + 32 * pi);
                                xq[j][ci][pi][0] = t[0]; xq[j][ci][pi][1] = t[1];
                            }
                    asm volatile("" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                            for (int pi = 0; pi < 2; ++pi) asm volatile("" : "+v"(xq[j][ci][pi][0]), "+v"(xq[j][ci][pi][1]));
#endif
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) {
                        const f32x2 b2 = {bv16[j][ci], bv16[j][ci]};
#pragma unroll
                        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                const f32x2 v = (f32x2){acc16[i][pi][j][ci][e], acc16[i][pi][j][ci][e + 1]} + b2;
                                char* p = wbase + (16 * pi + e) * ROWB + (j * 32 + 16 * ci) * (int)sizeof(OutT);
                                if constexpr (ACTB) {
                                    const unsigned xw = xq[j][ci][pi][e >> 1];
                                    const f32x2 x = {__uint_as_float(xw << 16), __uint_as_float(xw & 0xFFFF0000u)};
                                    const f32x2 yv = __builtin_elementwise_fma(x, (f32x2){csc16[j][ci], csc16[j][ci]}, (f32x2){csh16[j][ci], csh16[j][ci]});
                                    const f32x2 f = {yv.x > 0.f ? 1.f : csl16[j][ci], yv.y > 0.f ? 1.f : csl16[j][ci]};
                                    const f32x2 gg = v * f;
                                    s1v[j][ci] += gg; s2v[j][ci] = __builtin_elementwise_fma(gg, x - (f32x2){cmu16[j][ci], cmu16[j][ci]}, s2v[j][ci]);
                                    abc_put2<OutT>(p, p + ROWB, gg.x, gg.y);
                                } else {
                                    if constexpr (WSTAT) { s1v[j][ci] += v; s2v[j][ci] = __builtin_elementwise_fma(v, v, s2v[j][ci]); }
                                    if constexpr (OACT) {
                                        const f32x2 m = v * slope;
                                        abc_put2<OutT>(p, p + ROWB, fmaxf(v.x, m.x), fmaxf(v.y, m.y));
                                    } else {
                                        abc_put2<OutT>(p, p + ROWB, v.x, v.y);      // (training: the raw output, the consumer applies the activation on load)
                                    }
                                }
                            }
                    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(s1v[j][0]), "+v"(s2v[j][0]), "+v"(s1v[j][1]), "+v"(s2v[j][1]));
#endif
                lds_wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                u32x4 rd[NST];
#pragma unroll
                for (int st = 0; st < NST; ++st) rd[st] = *(const u32x4*)(stgw + (lrow + RSTEP * st) * ROWB + lsg * 16);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int prow = 2 * ((wl / WN) * TM + i) + ((lrow + RSTEP * st) >> 4);
                    __builtin_amdgcn_raw_buffer_store_b128(rd[st], rsY, prow < rlim ? voff[st] : 0xFFFFFFF0u, (unsigned)i * istep, 0);
                }
                // (the store-data hazard of the epilogue below: keep the data registers live two wait states past the last store)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int st = 0; st < NST; ++st) asm volatile("" :: "v"(rd[st]));
                asm volatile("s_nop 1");
#endif
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            };
            if constexpr (ACTB) store_tiles(std::false_type{}, std::true_type{});
            else if (!a.out_act) store_tiles(std::false_type{}, std::true_type{});
            else if (a.stats == nullptr) store_tiles(std::true_type{}, std::false_type{});
            else store_tiles(std::true_type{}, std::true_type{});
            if (pr) prof[3] = wall_clock64();
            if (a.stats != nullptr) {
                // a channel's sums sit in the four lanes m16 + 16 q: added as (own + lane ^ 16) + (lane ^ 32's same), then the WM waves in order.
                // A halving butterfly on the VALU (v_permlane16_swap, v_permlane32_swap) instead of two ds_bpermute per value: of the lane's
                // 4 TN values, step 1 leaves the pair sums of the even-indexed ones in even rows of 16 lanes and of the odd-indexed ones in odd
                // rows, step 2 the quad sums of half of those in each wave half: a lane ends with TN totals, the four lanes of a channel
                // column with all 4 TN between them.
                float* red = (float*)(smem + a.red_off);  // [WM][4][BN]
                float sv[4 * TN];      // index 4 j + 2 ci + (0: sum, 1: sum of squares)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) {
                        const bool nok = n0 + (wn * TN + j) * 32 + 16 * ci + m16 < a.Cout;      // (padding channels of the last n-block: bias-free zeros)
                        sv[4 * j + 2 * ci + 0] = nok ? s1v[j][ci].x + s1v[j][ci].y : 0.f;
                        sv[4 * j + 2 * ci + 1] = nok ? s2v[j][ci].x + s2v[j][ci].y : 0.f;
                    }
                // step 1: values (2 u, 2 u + 1) -> even rows keep value 2 u, odd rows value 2 u + 1 (own + the lane 16 away)
                float s16[2 * TN];
#pragma unroll
                for (int u = 0; u < 2 * TN; ++u) {
                    const u32x2 pr2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sv[2 * u]), __float_as_uint(sv[2 * u + 1]), false, false);
                    s16[u] = __uint_as_float(pr2[0]) + __uint_as_float(pr2[1]);
                }
                // step 2: values (2 w, 2 w + 1) of step 1 -> the lower wave half keeps 2 w, the upper half 2 w + 1 (own + the lane 32 away)
                float s32[TN];
#pragma unroll
                for (int w2 = 0; w2 < TN; ++w2) {
                    const u32x2 pr2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s16[2 * w2]), __float_as_uint(s16[2 * w2 + 1]), false, false);
                    s32[w2] = __uint_as_float(pr2[0]) + __uint_as_float(pr2[1]);
                }
                // lane (m16, q16) now holds, for w2 = 0 .. TN - 1, the total of value index 4 w2 + 2 (q16 >> 1) + (q16 & 1):
                // j = w2, ci = q16 >> 1, sum or sum of squares = q16 & 1
#pragma unroll
                for (int w2 = 0; w2 < TN; ++w2) {
                    const int nl = (wn * TN + w2) * 32 + 16 * (q16 >> 1) + m16;
                    red[(wm * 4 + (q16 & 1)) * BN + nl] = s32[w2];
                }
                __syncthreads();
                if (tid < BN && n0 + tid < a.Cout) {
                    float v1 = 0.f, v2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) { v1 += red[(w * 4 + 0) * BN + tid]; v2 += red[(w * 4 + 1) * BN + tid]; }
                    if constexpr (ACTB) v2 *= a.ab_is[n0 + tid];
                    a.stats[((size_t)mblock * 2 + 0) * a.Cout + n0 + tid] = v1;
                    a.stats[((size_t)mblock * 2 + 1) * a.Cout + n0 + tid] = v2;
                }
            }
            if (pr) prof[4] = wall_clock64();
            continue;
        }
        // ---- epilogue: bias, statistics of the f32 values, store through a wave-private LDS transpose
        // (a lane of the accumulator layout holds ONE channel of 16 pixels; the transpose turns that into 16-byte
        // stores of consecutive channels of one pixel).  The staging region aliases the halo / weight buffers: every
        // wave passed the barrier that ended the last stage, so they are dead (resident weights are kept clear of it).
        OutT* yo = (OutT*)a.y;
        float s1[TN], s2[TN], smx[TN], smn[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; smx[j] = -3.0e38f; smn[j] = 3.0e38f; }
        constexpr int TW = TN * 32;
        constexpr int ROWB = TW * (int)sizeof(OutT) + 16;
        constexpr int EV = 16 / (int)sizeof(OutT);
        constexpr int SEG_PER_ROW = TW / EV;
        char* stg = smem + a.stg_off + wave * (32 * ROWB);
        const int cbase = n0 + wn * TW;
        const bool vec_ok = ((a.ldy | a.cout_off) % EV) == 0;
        // The epilogue is VALU work per output value (narrow layers are bound by it: 16 instructions per value cost
        // 39 us on the 16-channel 384x384 layers).  Fast path for whole tiles with plain 16-byte stores: add bias,
        // sum, sum of squares, convert, one ds_write per value with an immediate offset; max / min in a second sweep
        // only when unet2's CBAM asks for them; the general path keeps every check.
        // (ACTB launches consist of whole tiles only, checked on the host: the general path below is not compiled into them)
        // (tiles that are short in ROWS only -- the last tile row of a 128 x 128 map under 12-row tiles, one tile in eleven of the
        //  inference graph -- take the fast path too when no statistics are asked for: their missing rows get the dropped offset)
        const bool rows_ok = gy0 + 2 * MT <= a.Hg;
        const bool whole = ACTB || ((rows_ok || a.stats == nullptr) && (gx0 + 16 <= a.Wg) && (a.Cout % EV == 0) && !a.accumulate && vec_ok);
        if (ABC_DBG(a.dbg) & 64) {
        } else if (whole) {
            // (laundered: computed from the plain thread index, the lane parts of the staging addresses and store offsets are hoisted
            //  out of the tile loop, live through the main loop, get spilled, and come back as scratch round trips in front of the
            //  stores -- each a vmcnt(0) wait that also drains the next tile's halo prefetch)
            const int tl = abc_launder(tid), ll = tl & 63, wl = tl >> 6;
            char* stgw = smem + a.stg_off + wl * (32 * ROWB);
            char* wbase = stgw + 4 * (ll >> 5) * ROWB + (ll & 31) * (int)sizeof(OutT);
            const int lrow = ll / SEG_PER_ROW, lsg = ll % SEG_PER_ROW;         // this lane's (pixel row, segment) in the store sweep
            constexpr int RSTEP = 64 / SEG_PER_ROW;                            // pixel rows covered per sweep step
            constexpr int NST = 32 / RSTEP;
            const bool seg_ok = n0 + (wl % WN) * TW + lsg * EV < a.Cout;
            // Stores: the tile's first pixel (wave-uniform) is the base of a buffer descriptor; a lane's part of the address is ONE
            // 32-bit offset per sweep step for the whole tile (segments past Cout get an offset the hardware drops), the M-tile's
            // two rows a scalar offset.  (64-bit address arithmetic and a predicated branch per store kept every
            // ds_read -> global_store pair a serial LDS round trip: 12 per wave and tile.)
            const OutT* ytile = yo + (((size_t)(b * a.Hout + gy0 * a.om + a.oy0) * a.Wout + gx0 * a.om + a.ox0) * a.ldy + a.cout_off + n0);
            const __amdgpu_buffer_rsrc_t rsY = abc_make_rsrc(ytile, 0x80000000u);
            const unsigned istep = (unsigned)(2 * a.om * a.Wout * a.ldy) * (unsigned)sizeof(OutT);   // bytes per M-tile (two pixel rows)
            unsigned voff[NST];
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int rit = lrow + RSTEP * st;       // pixel of the wave's 32: patch row (rit >> 4) of the M-tile, column rit & 15
                voff[st] = seg_ok ? (unsigned)((wl / WN) * TM) * istep + (unsigned)((((rit >> 4) * a.Wout + (rit & 15)) * a.om * a.ldy + (wl % WN) * TW + lsg * EV) * (int)sizeof(OutT))
                                  : 0xFFFFFFF0u;
            }
            // Values two at a time (accumulator registers k, k + 1 = two pixel rows of one channel): packed f32 add / multiply / fma,
            // one conversion per pair.  Per M-tile: 32 writes, a wait for them, then the sweep's reads back to back and its stores
            // behind them (the stores consume the reads, so the next M-tile's writes cannot overtake them).
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 s1v[TN], s2v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) { s1v[j] = (f32x2){0.f, 0.f}; s2v[j] = (f32x2){0.f, 0.f}; }
            const float slope = a.out_act ? a.out_slope : 1.f;      // (max(v, 1 * v) = v: no select per value)
            const int rlim = (ACTB || rows_ok) ? 0x7FFFFFFF : a.Hg - gy0;    // rows of this tile inside the map
            // ACTB: the producer's raw output y_raw at this tile -- per M-tile the sweep's 16-byte loads (the stores' lane mapping),
            // through a second staging region, read back in the accumulator layout (lane = channel, like the sums).
            // The loads of M-tile i + 1 fly under the arithmetic of M-tile i.
            // (the y_raw tile is staged TRANSPOSED, [channel][32 pixels + pad]: sixteen-bit writes need no answer, and a lane reads the four
            //  consecutive pixels of an accumulator-register group with one ds_read_b64 -- 8 reads per M-tile instead of 32 ds_read_u16)
            constexpr int YROW = 32 * 2 + 8;
            static_assert(!ACTB || TW * YROW <= 32 * ROWB, "the transposed y_raw tile fits the staging region");
            char* ystg = smem + a.ystg_off + wl * (32 * ROWB);
            const char* yrd = ystg + (ll & 31) * YROW + 8 * (ll >> 5);
            __amdgpu_buffer_rsrc_t rsYR = rsY;
            unsigned voffy[ACTB ? NST : 1], istep_y = 0;
            u32x4 yq[ACTB ? NST : 1];
            if constexpr (ACTB) {
                const bf16* yrt = (const bf16*)a.ab_y + (((size_t)(b * a.Hg + gy0) * a.Wg + gx0) * a.ab_ld + n0);
                rsYR = abc_make_rsrc(yrt, 0x80000000u);
                istep_y = (unsigned)(2 * a.Wg * a.ab_ld) * 2u;
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int rit = lrow + RSTEP * st;
                    voffy[st] = seg_ok ? (unsigned)((wl / WN) * TM) * istep_y + (unsigned)((((rit >> 4) * a.Wg + (rit & 15)) * a.ab_ld + (wl % WN) * TW + lsg * EV) * 2)
                                       : 0xFFFFFFF0u;
                    yq[st] = __builtin_amdgcn_raw_buffer_load_b128(rsYR, voffy[st], 0u, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (ACTB) {
#pragma unroll
                    for (int st = 0; st < NST; ++st) {
                        char* w0 = ystg + (lsg * EV) * YROW + (lrow + RSTEP * st) * 2;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            *(unsigned short*)(w0 + (2 * c) * YROW) = (unsigned short)(yq[st][c] & 0xFFFFu);
                            *(unsigned short*)(w0 + (2 * c + 1) * YROW) = (unsigned short)(yq[st][c] >> 16);
                        }
                    }
                    if (i + 1 < TM) {
#pragma unroll
                        for (int st = 0; st < NST; ++st) yq[st] = __builtin_amdgcn_raw_buffer_load_b128(rsYR, voffy[st], (unsigned)(i + 1) * istep_y, 0);
                    }
                    lds_wave_sync();      // (the 16-bit writes have landed before other lanes' 8-byte reads)
                }
                // (ACTB: the M-tile's y_raw values read back in ONE batch, two bf16 per register -- read pair by pair between the
                //  staging writes, which the compiler must keep in program order, every pair was a serial LDS round trip)
                unsigned xq[ACTB ? TN : 1][ACTB ? 8 : 1];
                if constexpr (ACTB) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            // pixels 8 g4 + 4 h + (0..3) of channel 32 j + r = accumulator registers 4 g4 .. 4 g4 + 3
                            const u32x2 t = *(const u32x2*)(yrd + j * 32 * YROW + 16 * g4);
                            xq[j][2 * g4] = t[0]; xq[j][2 * g4 + 1] = t[1];
                        }
                    asm volatile("" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
                    // (all eight reads issued back to back, ONE wait: left to itself the compiler sinks each read to its use)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(xq[j][q]));
#endif
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x2 b2 = {bv[j], bv[j]};
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        f32x2 v = {acc[i][j][k], acc[i][j][k + 1]};
                        if constexpr (F8C) v = __builtin_elementwise_fma(v, (f32x2){osc[j], osc[j]}, b2); else v += b2;
                        char* p = wbase + ((k & 3) + 8 * (k >> 2)) * ROWB + j * 32 * (int)sizeof(OutT);
                        if constexpr (ACTB) {
                            // g = dA where BatchNorm(y_raw) > 0, slope * dA elsewhere (unet.py:14,17 backward); sums of g and g (y_raw - mean)
                            const f32x2 x = {__uint_as_float(xq[j][k >> 1] << 16), __uint_as_float(xq[j][k >> 1] & 0xFFFF0000u)};
                            const f32x2 yv = __builtin_elementwise_fma(x, (f32x2){csc[j], csc[j]}, (f32x2){csh[j], csh[j]});
                            const f32x2 f = {yv.x > 0.f ? 1.f : csl[j], yv.y > 0.f ? 1.f : csl[j]};
                            const f32x2 gg = v * f;
                            s1v[j] += gg; s2v[j] = __builtin_elementwise_fma(gg, x - (f32x2){cmu[j], cmu[j]}, s2v[j]);
                            abc_put2<OutT>(p, p + ROWB, gg.x, gg.y);
                        } else {
                            s1v[j] += v; s2v[j] = __builtin_elementwise_fma(v, v, s2v[j]);
                            const f32x2 m = v * slope;
                            f32x2 vo = {fmaxf(v.x, m.x), fmaxf(v.y, m.y)};
                            if constexpr (F8O) vo *= oq;
                            abc_put2<OutT>(p, p + ROWB, vo.x, vo.y);
                        }
                    }
                }
                // (the sums are pinned to their M-tile and the scheduler barriers keep the M-tiles apart: left free, the compiler
                //  sinks the 2 x 96 accumulations behind the last store and hoists the bias adds to the front -- the biased values
                //  of the whole tile live beside the accumulators, ~50 lane constants of the main loop spilled)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(s1v[j]), "+v"(s2v[j]));
#endif
                // (a wait, not program order alone: the rows are read by other lanes than wrote them)
                lds_wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                u32x4 rd[NST];
#pragma unroll
                for (int st = 0; st < NST; ++st) rd[st] = *(const u32x4*)(stgw + (lrow + RSTEP * st) * ROWB + lsg * 16);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int prow = 2 * ((wl / WN) * TM + i) + ((lrow + RSTEP * st) >> 4);      // pixel row of the tile
                    __builtin_amdgcn_raw_buffer_store_b128(rd[st], rsY, prow < rlim ? voff[st] : 0xFFFFFFF0u, (unsigned)i * istep, 0);
                }
                // HAZARD (seen on gfx950, ROCm 7.2): a VALU write to the first data register of a 16-byte buffer store IN THE NEXT
                // INSTRUCTION reaches the stored data -- the compiler had picked that register for the next store's offset select
                // (`buffer_store_dwordx4 v[98:101], ..; v_cndmask_b32 v98, ..`: intermittently two channels of a pixel came out as
                // the bits of an offset); LLVM's hazard recogniser covers this only for stores without a scalar offset register.
                // Keep the data registers live past the last store and two wait states behind it.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int st = 0; st < NST; ++st) asm volatile("" :: "v"(rd[st]));
                asm volatile("s_nop 1");
#endif
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) { s1[j] = s1v[j].x + s1v[j].y; s2[j] = s2v[j].x + s2v[j].y; }
            if (a.stats_rows == 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            // (max / min of the values AS STORED, i.e. after the rounding to OutT: CBAM's global max-pool and its backward,
                            //  unet2.py:10,20, see the tensor -- taken before the rounding, the backward's "is this pixel the maximum"
                            //  never found it in bf16 and the max branch's gradient was lost)
                            const float v = (float)(OutT)(acc[i][j][k] + bv[j]); smx[j] = fmaxf(smx[j], v); smn[j] = fminf(smn[j], v);
                        }
            }
            // channels past Cout (padding lanes of the last n-block) carry bias-free zeros: keep them out of the sums
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (!nval[j]) { s1[j] = 0.f; s2[j] = 0.f; }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int rit = (k & 3) + 8 * (k >> 2) + 4 * h;
                    const int gy = gy0 + 2 * (wm * TM + i) + (rit >> 4), gx = gx0 + (rit & 15);
                    const float v = F8C ? fmaf(acc[i][j][k], osc[j], bv[j]) : acc[i][j][k] + bv[j];
                    if (nval[j] && gy < a.Hg && gx < a.Wg) {
                        const float vr = (float)(OutT)v;
                        s1[j] += v; s2[j] += v * v; smx[j] = fmaxf(smx[j], vr); smn[j] = fminf(smn[j], vr);
                    }
                    float vo = a.out_act ? fmaxf(v, a.out_slope * v) : v;
                    if constexpr (F8O) vo *= oq;
                    *(OutT*)(stg + rit * ROWB + (j * 32 + r) * (int)sizeof(OutT)) = (OutT)vo;
                }
            }
            lds_wave_sync();
#pragma unroll
            for (int e = lane; e < 32 * SEG_PER_ROW; e += 64) {
                const int rit = e / SEG_PER_ROW, sg = e - rit * SEG_PER_ROW;
                const int gy = gy0 + 2 * (wm * TM + i) + (rit >> 4), gx = gx0 + (rit & 15);
                const int cch = cbase + sg * EV;
                if (gy < a.Hg && gx < a.Wg && cch < a.Cout) {
                    const size_t o = ((size_t)(b * a.Hout + gy * a.om + a.oy0) * a.Wout + gx * a.om + a.ox0) * a.ldy + a.cout_off + cch;
                    const char* src = stg + rit * ROWB + sg * 16;
                    if (a.accumulate) {
                        for (int q = 0; q < EV && cch + q < a.Cout; ++q)
                            yo[o + q] = (OutT)((float)yo[o + q] + (float)*(const OutT*)(src + q * (int)sizeof(OutT)));
                    } else if (cch + EV <= a.Cout && vec_ok) {
                        *(f32x4*)(yo + o) = *(const f32x4*)src;
                    } else {
                        for (int q = 0; q < EV && cch + q < a.Cout; ++q) yo[o + q] = *(const OutT*)(src + q * (int)sizeof(OutT));
                    }
                }
            }
            lds_wave_sync();
        }
        }
        if (pr) prof[3] = wall_clock64();
        if (a.stats != nullptr) {
            float* red = (float*)(smem + a.red_off);  // [WM][4][BN], clear of the transpose regions
            const int rows = a.stats_rows == 4 ? 4 : 2;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float v1 = s1[j] + __shfl_xor(s1[j], 32);
                const float v2 = s2[j] + __shfl_xor(s2[j], 32);
                const float v3 = fmaxf(smx[j], __shfl_xor(smx[j], 32));
                const float v4 = fminf(smn[j], __shfl_xor(smn[j], 32));
                if (h == 0) {
                    const int nl = (wn * TN + j) * 32 + r;
                    red[(wm * 4 + 0) * BN + nl] = v1;
                    red[(wm * 4 + 1) * BN + nl] = v2;
                    red[(wm * 4 + 2) * BN + nl] = v3;
                    red[(wm * 4 + 3) * BN + nl] = v4;
                }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < a.Cout) {
                float v1 = 0.f, v2 = 0.f, v3 = -3.0e38f, v4 = 3.0e38f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    v1 += red[(w * 4 + 0) * BN + tid]; v2 += red[(w * 4 + 1) * BN + tid];
                    v3 = fmaxf(v3, red[(w * 4 + 2) * BN + tid]); v4 = fminf(v4, red[(w * 4 + 3) * BN + tid]);
                }
                if constexpr (ACTB) v2 *= a.ab_is[n0 + tid];      // (the row act_bwd writes: sum of g (y_raw - mean) / std)
                if (STATIC && rows == 2) {      // (unet2's CBAM needs its four rows PER IMAGE: those stay per tile)
                    pst1 += v1; pst2 += v2;
                } else {
                    a.stats[((size_t)mblock * rows + 0) * a.Cout + n0 + tid] = v1;
                    a.stats[((size_t)mblock * rows + 1) * a.Cout + n0 + tid] = v2;
                    if (rows == 4) {
                        a.stats[((size_t)mblock * rows + 2) * a.Cout + n0 + tid] = v3;
                        a.stats[((size_t)mblock * rows + 3) * a.Cout + n0 + tid] = v4;
                    }
                }
            }
        }
        if (pr) prof[4] = wall_clock64();
    }
    if constexpr (STATIC) {
        if (a.stats != nullptr && a.stats_rows != 4 && tid < BN && tid < a.Cout) {
            a.stats[((size_t)blockIdx.x * 2 + 0) * a.Cout + tid] = pst1;
            a.stats[((size_t)blockIdx.x * 2 + 1) * a.Cout + tid] = pst2;
        }
    }
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT, bool STATIC, int WD = 0, int EPI = 0, int NW = 4, bool LP = false, int VAR = 0, bool M16 = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 4 : (STATIC ? 3 : 2))) void conv_fast_kernel(const FastK a) {
    conv_fast_body<InT, CT, OutT, CK, BN, STRIDE, MT, STATIC, WD, EPI, NW, LP, VAR, M16>(a);
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT, bool STATIC, int WD = 0, int EPI = 0, int NW = 4, bool LP = false, int VAR = 0, bool M16 = false>
int launch_st(const FastK& k, const abc_fast_geom& g, hipStream_t st) {
    auto fn = conv_fast_kernel<InT, CT, OutT, CK, BN, STRIDE, MT, STATIC, WD, EPI, NW, LP, VAR, M16>;
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)fn, LDS_WG, &lds_ok)) return rc;
    hipLaunchKernelGGL(fn, dim3(g.nwg), dim3(64 * NW), g.lds, st, k);
    return abc_check_launch("conv_fast");
}

}  // namespace abc_cf
