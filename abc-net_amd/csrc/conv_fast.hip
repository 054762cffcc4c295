// Tap-list convolution as implicit GEMM, lean form for plain NHWC inputs (the common case: every 3x3 / 1x1 /
// transposed-phase convolution whose input needs no pooling, dropout or planar read).
//
// Same GEMM view and LDS images as conv_igemm.hip (M = a (2*MT) x 16 pixel patch, N = BN output channels,
// K = [Cin chunk][tap group]; halo staged once per chunk with the previous layer's BN + activation applied on the
// way in, weight slices of a tap group beside it, double-buffered), but organised for OVERLAP BETWEEN WORKGROUPS:
//
//   * 4 waves per workgroup and at most 80 KB of LDS, so TWO workgroups share a CU (2 waves per SIMD, 256 VGPRs
//     each).  Their phases drift apart, so one workgroup's staging, barriers and epilogue run under the other's
//     MFMA block.  (Measured on the 8-wave one-per-CU kernel: MFMA 33 us + staging 24 us + epilogue 38 us = 95 us,
//     i.e. nothing overlapped, because all waves of a CU were always in the same phase.)
//   * a wave owns TM x TN 32x32 tiles with TN = 2 where BN >= 64... (2 x 2 .. 4 x 2): (TM + TN) ds_read_b128 per
//     TM * TN MFMAs keeps the LDS pipe under the matrix pipe.
//   * one tile per workgroup (grid = tiles) unless the whole weight set fits in LDS beside the halo (narrow layers):
//     then the workgroup is persistent and keeps the weights resident.
//   * the epilogue transposes through a wave-private LDS region (no workgroup barriers) into 16-byte stores.
//
// Reference ops covered: nn.Conv2d forward (unet.py:12,15,66,70), its data gradient, nn.ConvTranspose2d (unet.py:44)
// as four output-parity phases; see include/abcnet_hip.h (abc_conv_desc).
#include "conv_fast_body.hpp"

using namespace abc_cf;

// Round-5 experiments on the 192 x 128 weights-direct tile, compiled into the DEBUG flavour only (ABC_KERNEL_DEBUG=1 ./build_hip.sh) and
// selected there by the hooks abc_debug_conv_nw / _lp / _var (profiles/tools/ab_conv128.py).  All three are exact (bit-identical
// outputs); none is faster than the form below (profiles/README.md "Round 5"):
//   conv_fast8.hip    8-wave workgroups (3 x 1 MFMA tiles per wave, 128 VGPRs, four waves per SIMD): 10-17 % SLOWER
//   conv_fast_lp.hip  lane = pixel epilogue (MFMA operands swapped, 16-byte stores from registers, no LDS staging, no barrier at the tile's
//                     end, statistics by a cross-lane butterfly): within 2 % either way; with act_bwd in the epilogue 15 % slower
//                     ... and VAR: the four waves side by side along N (half the weight-fragment loads), s_setprio around the MFMA groups: +-1 %
#ifdef ABC_KERNEL_DEBUG
int abc_conv_fast_launch8(const FastK& k, const abc_fast_geom& g, int epi, hipStream_t st);
int abc_conv_fast_launch_lp(const FastK& k, const abc_fast_geom& g, int epi, hipStream_t st);
#endif

namespace {

static long long* g_prof = nullptr;
static int g_force_var = 0;     // measurement hook (abc_debug_conv_var): VAR of conv_fast_body.hpp for the 192 x 128 weights-direct tile
static int g_force_lp = 0;      // measurement hook (abc_debug_conv_lp, DEBUG flavour): 1 = plan the lane = pixel form (conv_fast_lp.hip) where it applies
static int g_force_nw = 0;     // measurement hook (abc_debug_conv_nw): 8 = plan the 8-wave form (conv_fast8.hip) where it applies

// Up to four convolutions of ONE geometry as one launch, blockIdx.y picks the descriptor: the four output-parity phases of a
// ConvTranspose2d(k3, s2) forward (unet.py:44) -- each a 1 / 2 / 2 / 4-tap convolution over the same input into interleaved output
// pixels, 100-400 workgroups of a few microseconds: as four launches they ran one after the other on a quarter-filled chip
struct FastKB { FastK k[4]; };
template <typename InT, typename CT, typename OutT, int CK, int BN, int MT>
__global__ __launch_bounds__(256, 2) void conv_fast_batch_kernel(const FastKB b) {
    conv_fast_body<InT, CT, OutT, CK, BN, 1, MT, false, 0, 0>(b.k[blockIdx.y]);
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int STRIDE, int MT, int EPI = 0>
int launch_inst(const FastK& k, const abc_fast_geom& g, hipStream_t st) {
    if constexpr (BN == 32 && EPI == 0) {  // resident weights exist for the narrow layers only
        if (g.b_static) return launch_st<InT, CT, OutT, CK, BN, STRIDE, MT, true>(k, g, st);
    }
    if constexpr (BN >= 64 && STRIDE == 1 && CK == 32 && sizeof(CT) == 2) {
        // 3x3 (any 9-tap list) over 64-byte chunks: weights straight from global memory into the MFMA operands
#ifdef ABC_KERNEL_DEBUG
        if constexpr (BN == 128 && MT == 6 && sizeof(InT) == 2 && sizeof(OutT) == 2 && (EPI == 0 || EPI == 2)) {
            if (g.wd == 9 && g.nw == 8) return abc_conv_fast_launch8(k, g, EPI, st);
            if (g.wd == 9 && (g.lp || g.var)) return abc_conv_fast_launch_lp(k, g, EPI, st);
        }
#endif
        if constexpr (((BN == 128 && (MT == 6 || MT == 4)) || (BN == 64 && MT == 8)) && sizeof(InT) == 2 && sizeof(OutT) == 2 && (EPI == 0 || EPI == 2)) {
            // the 16x16x32 form of the same tile (whole tiles, sums and squares only: abc_fast_geom.m16)
            if (g.wd == 9 && g.m16) return launch_st<InT, CT, OutT, CK, BN, STRIDE, MT, false, 9, EPI, 4, false, 0, true>(k, g, st);
        }
        if (g.wd == 9) return launch_st<InT, CT, OutT, CK, BN, STRIDE, MT, false, 9, EPI>(k, g, st);
    }
    if constexpr (BN >= 64 && STRIDE == 2 && MT == 4 && CK == 32 && sizeof(InT) == 2 && sizeof(CT) == 2 && sizeof(OutT) == 2 && EPI == 0) {
        if (g.wd == 9) return launch_st<InT, CT, OutT, CK, BN, STRIDE, MT, false, 9, EPI>(k, g, st);
    }
    if constexpr (BN == 32 && MT == 8 && STRIDE == 1 && CK == 32 && sizeof(CT) == 2 && EPI == 0) {
        // unet2's 5x5 32 -> 32 convolutions: 25 taps, a ring of 5
        if (g.wd == 25) return launch_st<InT, CT, OutT, CK, BN, STRIDE, MT, false, 25>(k, g, st);
    }
    return launch_st<InT, CT, OutT, CK, BN, STRIDE, MT, false, 0, EPI>(k, g, st);
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int EPI = 0>
int launch_mt(const FastK& k, const abc_fast_geom& g, int stride, hipStream_t st) {
    if constexpr (BN == 128) {
        if (stride == 1 && g.MT == 6) return launch_inst<InT, CT, OutT, CK, BN, 1, 6, EPI>(k, g, st);
    }
    if constexpr (EPI == 0) {
        if (stride == 2) return launch_inst<InT, CT, OutT, CK, BN, 2, 4>(k, g, st);
    }
    if constexpr (BN >= 64) {
        if (g.MT == 2) return launch_inst<InT, CT, OutT, CK, BN, 1, 2, EPI>(k, g, st);
    }
    if constexpr (BN != 128) {
        if (g.MT == 8) return launch_inst<InT, CT, OutT, CK, BN, 1, 8, EPI>(k, g, st);
    }
    return launch_inst<InT, CT, OutT, CK, BN, 1, 4, EPI>(k, g, st);
}

template <typename InT, typename CT, typename OutT, int CK, int EPI = 0>
int launch_bn(const FastK& k, const abc_fast_geom& g, int stride, hipStream_t st) {
    switch (g.BN) {
        case 128: return launch_mt<InT, CT, OutT, CK, 128, EPI>(k, g, stride, st);
        case 64: return launch_mt<InT, CT, OutT, CK, 64, EPI>(k, g, stride, st);
        default: return launch_mt<InT, CT, OutT, CK, 32, EPI>(k, g, stride, st);
    }
}

}  // namespace

// debugging hook (not part of the public ABI): device buffer of 8 x int64 per workgroup for phase timestamps
extern "C" void abc_debug_conv_prof(void* p) { g_prof = (long long*)p; }
// measurement hook (not part of the public ABI; profiles/tools/ab_conv128.py): 8 = plan the 8-wave form (conv_fast8.hip: measured SLOWER, profiles/README.md round 5) where it applies, 0 = default.
// Process-wide and read at plan AND launch time: set it before any plan is built, never between a plan and its launches.
#ifdef ABC_KERNEL_DEBUG
extern "C" void abc_debug_conv_nw(int nw) { g_force_nw = nw; }
extern "C" void abc_debug_conv_lp(int lp) { g_force_lp = lp; }
extern "C" void abc_debug_conv_var(int var) { g_force_var = var; }
#endif

// Geometry of the lean kernel for this descriptor, or eligible = 0 (-> the general kernel of conv_igemm.hip).
int abc_conv_fast_geom(const abc_conv_desc* d, abc_fast_geom* g) {
    g->eligible = 0;
    if (abc_knob("ABC_CONV_NOFAST")) return ABC_OK;
    if (d->src.pool || d->src.planar || d->src.drop_p > 0.f || d->planar_out) return ABC_OK;
    const int csz = abc_dsize(d->dtype_c);
    const bool f8 = d->dtype_c == ABC_FP8 || d->dtype_out == ABC_FP8 || d->dtype_in == ABC_FP8;
    // e4m3 (the fp8 inference graph): 3x3 / stride 1 / 128-channel tiles on the weights-direct loop, nothing else
    if (f8 && (d->stride != 1 || d->ntaps != 9 || d->Cout_pad % 128 || d->Cout != d->Cout_pad || d->src.scale != nullptr || d->accumulate || d->stats != nullptr ||
               (d->dtype_c == ABC_FP8) != (d->dtype_in == ABC_FP8) || d->dtype_c == ABC_F32 || d->dtype_out == ABC_F32)) return ABC_OK;
    g->CK = abc_conv_chunk(d->dtype_c, d->Cin);
    if (g->CK <= 0 || d->Cin % g->CK) return ABC_OK;
    // act_bwd in the epilogue (abc_conv_desc.actbwd_y): a plain stride-1 bf16 data gradient over whole tiles, 64-byte chunks
    const bool actb = d->actbwd_y != nullptr;
    if (actb && (d->dtype_in != ABC_BF16 || d->dtype_c != ABC_BF16 || d->dtype_out != ABC_BF16 || g->CK != 32 || d->stride != 1 || d->accumulate ||
                 d->stats == nullptr || d->stats_rows != 2 || d->om != 1 || d->oy0 || d->ox0 || d->Hg != d->Hout || d->Wg != d->Wout || d->out_act ||
                 d->Wg % 16 || d->Cout % 8 || (d->ldy | d->cout_off | d->actbwd_ld) % 8 || d->heads_epi != nullptr ||
                 (int64_t)d->B * d->Hg * d->Wg * d->actbwd_ld * 2 >= (int64_t(1) << 31))) return ABC_OK;
    const int64_t bytes_a = (int64_t)d->B * d->src.Hx * d->src.Wx * d->src.ldx * abc_dsize(d->dtype_in);
    if (bytes_a >= (int64_t(1) << 31)) return ABC_OK;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    g->dy_min = dymin; g->dx_min = dxmin;
    const int CKB = g->CK * csz;
    g->PS = CKB + 16;
    const int segs = CKB / 16;
    g->cstride = abc_roundup(d->Cin, 4);
    const int coef_bytes = abc_roundup(3 * g->cstride * 4, 256);
    const int osz = abc_dsize(d->dtype_out);

    // tile shape (BN output channels x MT*32 pixels) by a small time model calibrated on the round-1 profiles
    // (DESIGN.md section 3): a workgroup walks nchunks x ngroups stages; a stage costs max(0.47 us of staging /
    // barrier latency, its MFMAs at 32 cycles each, twice that when the CU is shared); a round costs ~12 us of
    // prologue + epilogue; rounds = ceil(tiles / 512 slots).  Deep layers (12x12 .. 24x24 pixels, K loops of 80 .. 160
    // stages) thus get narrow tiles with few tap groups, wide layers the biggest tile that fits the registers.
    const int bn_nat = (d->Cout_pad % 128 == 0) ? 128 : (d->Cout_pad % 64 == 0 ? 64 : 32);
    const char* force = abc_knob("ABC_CONV_MT");
    const char* force_bn = abc_knob("ABC_CONV_BN");
    int best = -1, best_bn = 0;
    double best_cost = 0;
    for (int bn = bn_nat; bn >= 32; bn >>= 1) {
        if (bn < 64 && bn != bn_nat) break;          // only 128 -> 64 is offered as a narrower tile
        if (force_bn && atoi(force_bn) != bn) continue;
        const int cand_all[4] = {8, 6, 4, 2};
        for (int ci = 0; ci < 4; ++ci) {
            const int mt = cand_all[ci];
            if (force && atoi(force) != mt) continue;
            if (mt == 6 && (bn != 128 || d->stride != 1)) continue;
            if ((f8 || d->heads_epi != nullptr) && (mt != 6 || bn != 128)) continue;
            if (mt == 8 && bn == 128) continue;  // 4 x 2 tiles per wave + staging registers exceed 256 VGPRs
            if (mt == 2 && bn == 32) continue;   // 4 x 1 wave layout needs 4 m-tiles
            if (d->stride == 2 && mt != 4) continue;
            const int prow = 2 * mt;
            if (actb && d->Hg % prow) continue;      // (whole tiles only)
            const int hh = (prow - 1) * d->stride + (dymax - dymin) + 1, hw = 15 * d->stride + (dxmax - dxmin) + 1;
            if (abc_cdiv(hh * hw * segs, FT) > (d->stride == 2 ? fa_stride2() : fa_max(mt)) || hh * hw * hw >= 65536) continue;
            const int rs = abc_roundup(hw * g->PS, 256);
            const int sa = abc_roundup(hh * rs, 256);
            if (sa + 2 * 128 * g->PS + coef_bytes + 256 > LDS_WG) continue;
            const long tiles = (long)(d->Cout_pad / bn) * abc_cdiv(d->Wg, 16) * abc_cdiv(d->Hg, prow) * d->B;
            int tg = SR_MAX / bn; if (tg > d->ntaps) tg = d->ntaps;
            const int ngroups = abc_cdiv(d->ntaps, tg);
            const int nstages = (d->Cin / g->CK) * ngroups;
            const int wn = bn >= 64 ? 2 : 1, tm = mt / (4 / wn), tnn = (bn / 32) / wn;
            const double mfma_stage = (double)d->ntaps / ngroups * (CKB / 32) * tm * tnn * (csz == 2 ? 1 : 8);
            const double share = tiles > 256 ? 2.0 : 1.0;
            const double t_stage = fmax(0.47, 0.0168 * share * mfma_stage);
            const double rounds = ceil((double)tiles / (double)abc_wg_slots(2));
            // (ties: the shape with more workgroups covers more CUs and wastes fewer rows)
            const double cost = rounds * (nstages * t_stage + 12.0) - 1e-3 * (double)(tiles < abc_wg_slots(2) ? tiles : abc_wg_slots(2));
            if (best < 0 || cost < best_cost) { best = mt; best_bn = bn; best_cost = cost; }
        }
    }
    if (best < 0) return ABC_OK;
    g->MT = best;
    g->BN = best_bn;
    g->nbn = d->Cout_pad / g->BN;
    g->nw = 4;
    int tn = g->BN >= 64 ? g->BN / 64 : 1;
    int stg = 4 * 32 * (tn * 32 * osz + 16);
    int red = 4 * 4 * g->BN * 4;
    const int prow = 2 * g->MT;
    g->HH = (prow - 1) * d->stride + (dymax - dymin) + 1;
    g->HW = 15 * d->stride + (dxmax - dxmin) + 1;
    g->RS = abc_roundup(g->HW * g->PS, 256);
    g->sA_bytes = abc_roundup(g->HH * g->RS, 256);
    const int budget = LDS_WG - coef_bytes - 256;
    const int nchunks = d->Cin / g->CK;
    // a stage = one tap group of <= SR_MAX weight rows; 2 LDS buffers, 2 stages in flight in registers.  Bigger
    // stages amortise the per-stage control code; a second halo buffer hides the chunk turn-over.  Prefer the big
    // stage, then the second halo buffer, as the 80 KB allow.
    int sr = SR_MAX;
    { const char* e = abc_knob("ABC_CONV_SR"); if (e) sr = atoi(e); }
    int tg = 1, abufs = 1;
    for (;;) {
        tg = sr / g->BN; if (tg < 1) tg = 1;
        if (tg > d->ntaps) tg = d->ntaps;
        g->ngroups = abc_cdiv(d->ntaps, tg);
        g->tg = abc_cdiv(d->ntaps, g->ngroups);
        g->sB_bytes = abc_roundup(g->tg * g->BN * g->PS, 256);
        if (g->sA_bytes + 2 * g->sB_bytes <= budget || sr <= 128) break;
        sr /= 2;
    }
    (void)abufs;
    // weights-direct main loop (3x3 over 64-byte bf16 chunks, tiles of >= 64 channels): no weights in LDS at all
    {
        const char* e = abc_knob("ABC_CONV_NOWD");   // "1": never; "2": only the 192 x 128 tile (experiments)
        const int lim = e ? atoi(e) : 0;
        // (stride 2, bf16 in and out: the data gradients of the transposed convolutions -- 9 taps over a 17 x 33-pixel halo per 8 x 16 tile)
        const bool s2 = d->stride == 2 && g->MT == 4 && d->dtype_in == ABC_BF16 && d->dtype_c == ABC_BF16 && d->dtype_out == ABC_BF16 && !abc_knob("ABC_CONV_NOWD_S2");
        g->wd = (csz <= 2 && CKB == 64 && g->BN >= 64 && (d->stride == 1 || s2) && d->ntaps == 9 && lim != 1 && (lim != 2 || (g->BN == 128 && g->MT == 6))) ? 9 : 0;
        if (f8 && (g->wd != 9 || g->BN != 128 || g->MT != 6)) return ABC_OK;
        // (25-tap form for unet2's 5x5 32 -> 32 layers: measured SLOWER than the LDS-staged weights, 136 vs 121 us -- a
        //  32-channel tile has only 4 MFMAs per tap to cover the global-load latency of the ring; opt-in for experiments)
        if (csz == 2 && g->CK == 32 && g->BN == 32 && g->MT == 8 && d->stride == 1 && d->ntaps == 25 && abc_knob("ABC_CONV_WD25")) g->wd = 25;
    }
    if (g->wd) g->sB_bytes = 0;
    // eight waves per workgroup (conv_fast8.hip): the bf16 192 x 128 weights-direct tile, plain or with act_bwd in the epilogue
    if (g->wd == 9 && g->BN == 128 && g->MT == 6 && !f8 && d->dtype_in == ABC_BF16 && d->dtype_c == ABC_BF16 && d->dtype_out == ABC_BF16 &&
        d->heads_epi == nullptr && g_force_nw == 8) {
        g->nw = 8; tn = 1;
        stg = 8 * 32 * (tn * 32 * osz + 16);
    }
    // lane = pixel epilogue (conv_fast_lp.hip): the bf16 192 x 128 weights-direct tile storing whole channel octets, sums and squares only
    g->lp = (g->wd == 9 && g->nw == 4 && g->BN == 128 && g->MT == 6 && !f8 && d->dtype_in == ABC_BF16 && d->dtype_c == ABC_BF16 && d->dtype_out == ABC_BF16 &&
             d->heads_epi == nullptr && d->stats_rows != 4 && !d->accumulate && d->Cout % 8 == 0 && (d->ldy | d->cout_off) % 8 == 0 && g_force_lp == 1) ? 1 : 0;
    if (g->lp) { stg = 0; red = 0; }       // (no staging, no reduction through LDS)
    g->var = 0;
    if (g_force_var && g->wd == 9 && g->nw == 4 && g->BN == 128 && g->MT == 6 && !f8 && d->dtype_in == ABC_BF16 && d->dtype_c == ABC_BF16 && d->dtype_out == ABC_BF16 &&
        d->heads_epi == nullptr) {
        g->var = g_force_var & 3;
        if ((g->var & 1) && !g->lp) stg = 4 * 32 * (32 * osz + 16);      // (1 x 4 waves: 32-channel staging rows)
    }
    // v_mfma_f32_16x16x32_bf16 form (conv_fast_body.hpp M16): the bf16 192 x 128 weights-direct tile where every tile takes the whole-tile epilogue
    // (192 x 128 pixels x channels everywhere; 128 x 128 and 256 x 64 where the epilogue carries no statistics -- the folded inference graph:
    //  +3 % on b64 @ 512 x 512 with them; in the training step, whose epilogues sum and square every value, they measure the same or slightly
    //  slower in this form, 5.924 -> 5.930 ms same box)
    const bool m16_shape = (g->BN == 128 && g->MT == 6) || (d->stats == nullptr && ((g->BN == 128 && g->MT == 4) || (g->BN == 64 && g->MT == 8)));
    g->m16 = (g->wd == 9 && g->nw == 4 && !g->lp && !g->var && m16_shape && d->stride == 1 && !f8 && d->dtype_in == ABC_BF16 && d->dtype_c == ABC_BF16 &&
              d->dtype_out == ABC_BF16 && d->heads_epi == nullptr && d->stats_rows != 4 && !d->accumulate && d->Cout % 8 == 0 && (d->ldy | d->cout_off) % 8 == 0 &&
              d->Wg % 16 == 0 && (d->Hg % (2 * g->MT) == 0 || d->stats == nullptr) && !abc_knob("ABC_CONV_NOM16")) ? 1 : 0;
    // whole weight set resident (narrow layers: one chunk, one n-block): persistent workgroups
    g->b_static = (!actb && !g->wd && g->BN == 32 && nchunks == 1 && g->nbn == 1 && g->sA_bytes + g->ngroups * g->sB_bytes + stg + red <= budget &&
                   abc_cdiv(g->HH * g->HW * segs, FT) <= fa_static(g->MT)) ? 1 : 0;
    if (g->b_static && d->ntaps % 3 == 0 && d->ntaps > 3) {
        // resident weights are loaded once: tap groups of 3 leave no padding rows (9 taps in two groups of 8 cost 24.5 KB of
        // LDS instead of 13.8 KB), which is what lets a third workgroup share the CU
        g->tg = 3; g->ngroups = d->ntaps / 3;
        g->sB_bytes = abc_roundup(g->tg * g->BN * g->PS, 256);
    }
    if (g->b_static) {
        g->a_bufs = 1;
    } else {
        g->a_bufs = (nchunks > 1 && 2 * g->sA_bytes + 2 * g->sB_bytes <= budget) ? 2 : 1;
        if (g->a_bufs * g->sA_bytes + 2 * g->sB_bytes > budget) return ABC_OK;
    }
    const int nbuf_b = g->b_static ? g->ngroups : (g->wd ? 0 : 2);
    g->tap_off = g->a_bufs * g->sA_bytes + nbuf_b * g->sB_bytes;
    g->coef_off = g->tap_off + 256;
    g->lds = g->coef_off + coef_bytes;
    g->stg_off = g->b_static ? abc_roundup(g->lds, 256) : 0;
    g->red_off = g->stg_off + stg;
    if (g->lds < g->red_off + red) g->lds = g->red_off + red;
    g->ystg_off = abc_roundup(g->lds, 256);      // (behind everything: the tap and coefficient tables outlive the tile)
    if (actb) g->lds = g->ystg_off + stg;
    g->epi_off = abc_roundup(g->lds, 256);       // LP, M16: [2][1 or 6][BN] floats
    if (g->lp || g->m16) g->lds = g->epi_off + 2 * (actb ? 6 : 1) * g->BN * 4;
    if (g->lds > LDS_WG) return ABC_OK;
    g->tiles_x = abc_cdiv(d->Wg, 16);
    g->tiles_y = abc_cdiv(d->Hg, prow);
    g->ntiles = g->nbn * g->tiles_x * g->tiles_y * d->B;
    // persistent workgroups: three per CU with resident weights; two per CU (512 slots) on the weights-direct loop when
    // there are more tiles than slots (measured on one box, same run: 6144 tiles 482 -> 463 us, 5632 tiles 473 -> 448 us)
    const int slots3 = abc_wg_slots(3), slots2 = abc_wg_slots(2);      // (768 / 512 unless CUs are reserved, abc_set_reserved_cus)
    g->nwg = g->b_static ? (g->ntiles < slots3 ? g->ntiles : slots3) : ((g->wd && g->ntiles > slots2 && !abc_knob("ABC_CONV_NOPERSIST")) ? slots2 : g->ntiles);
    if (d->heads_epi != nullptr) {
        // heads in the epilogue: the 192 x 128 weights-direct tile, finished input, every 128-channel block a head
        if (!(g->wd == 9 && g->BN == 128 && g->MT == 6) || d->src.scale != nullptr || !d->out_act || d->stats != nullptr || d->accumulate ||
            d->Cout % 128 || d->om != 1 || d->oy0 || d->ox0 || d->Hg != d->Hout || d->Wg != d->Wout) return ABC_OK;
        const int need = 192 * (128 * csz + 16);
        if (g->lds < need) g->lds = need;
        if (g->lds > LDS_WG) return ABC_OK;
    }
    g->eligible = 1;
    return ABC_OK;
}

static void fill_fastk(const abc_conv_desc* d, const abc_fast_geom& g, FastK& k) {
    k.x = d->src.x; k.scale = d->src.scale; k.shift = d->src.shift; k.slope = d->src.slope;
    k.w = d->w; k.bias = d->bias; k.y = d->y; k.stats = d->stats;
    k.B = d->B; k.Hin = d->Hin; k.Win = d->Win; k.Hx = d->src.Hx; k.Wx = d->src.Wx; k.ldx = d->src.ldx;
    k.cin_off = d->cin_off; k.Cin = d->Cin; k.nchunks = d->Cin / g.CK;
    k.Hg = d->Hg; k.Wg = d->Wg; k.Hout = d->Hout; k.Wout = d->Wout; k.ldy = d->ldy; k.cout_off = d->cout_off;
    k.Cout = d->Cout; k.Cout_pad = d->Cout_pad; k.om = d->om; k.oy0 = d->oy0; k.ox0 = d->ox0;
    k.ntaps = d->ntaps; k.tg = g.tg; k.ngroups = g.ngroups; k.dy_min = g.dy_min; k.dx_min = g.dx_min;
    k.HH = g.HH; k.HW = g.HW; k.RS = g.RS; k.magic = 65536 / g.HW + 1;
    k.tiles_x = g.tiles_x; k.tiles_y = g.tiles_y; k.nblocks_n = g.nbn; k.ntiles = g.ntiles;
    k.sA_bytes = g.sA_bytes; k.a_bufs = g.a_bufs; k.sB_off = g.a_bufs * g.sA_bytes; k.sB_bytes = g.sB_bytes;
    k.tap_off = g.tap_off; k.coef_off = g.coef_off; k.cstride = g.cstride; k.stats_rows = d->stats_rows;
    k.accumulate = d->accumulate; k.b_static = g.b_static; k.stg_off = g.stg_off; k.red_off = g.red_off;
    k.out_act = d->out_act; k.out_slope = d->out_slope;
    k.oscale = d->out_scale; k.oquant = d->out_quant; k.oq_stride = d->out_quant_stride ? 1 : 0;
    k.hepi = d->heads_epi;
    k.ab_y = d->actbwd_y ? (const char*)d->actbwd_y + (size_t)d->actbwd_coff * 2 : nullptr; k.ab_ld = d->actbwd_ld;
    k.ab_sc = d->actbwd_scale; k.ab_sh = d->actbwd_shift; k.ab_sl = d->actbwd_slope; k.ab_mu = d->actbwd_mean; k.ab_is = d->actbwd_invstd;
    k.ystg_off = g.ystg_off;
    k.epi_off = g.epi_off;
    { const char* e = abc_knob("ABC_CONV_DBG"); k.dbg = e ? atoi(e) : 0; }  // timing ablations only (results invalid)
    k.prof = g_prof;
    { const char* e = abc_knob("ABC_CONV_PROF_ROUND"); k.prof_round = e ? atoi(e) : -1; }
    { const char* e = abc_knob("ABC_CONV_STAGGER"); k.stagger = e ? atoi(e) : 0; }
    k.bytesA = (unsigned)((int64_t)d->B * d->src.Hx * d->src.Wx * d->src.ldx * abc_dsize(d->dtype_in));
    k.bytesW = (unsigned)((int64_t)d->ntaps * k.nchunks * d->Cout_pad * g.CK * abc_dsize(d->dtype_c));
    for (int t = 0; t < d->ntaps; ++t) {
        k.ty[t] = (int8_t)(d->tap_dy[t] - g.dy_min);
        k.tx[t] = (int8_t)(d->tap_dx[t] - g.dx_min);
    }
}

template <typename InT, typename CT, typename OutT, int CK, int BN, int MT>
static int launch_batch_inst(const FastKB& b, int n, const abc_fast_geom& g, hipStream_t st) {
    auto fn = conv_fast_batch_kernel<InT, CT, OutT, CK, BN, MT>;
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)fn, LDS_WG, &lds_ok)) return rc;
    hipLaunchKernelGGL(fn, dim3(g.nwg, n), dim3(FT), g.lds, st, b);
    return abc_check_launch("conv_fast_batch");
}

// 1: n (2..4) descriptors the batched launch serves -- plain bf16 stride-1 convolutions on the LDS-staged loop with the same tile
// geometry (taps, weights, bias and output placement may differ); 0: launch them one by one
int abc_conv_fast_batch_ok(const abc_conv_desc* d, int n, abc_fast_geom* g0) {
    if (n < 2 || n > 4) return 0;
    abc_fast_geom g[4];
    for (int i = 0; i < n; ++i) {
        if (d[i].heads_epi != nullptr || d[i].actbwd_y != nullptr || d[i].dtype_in != ABC_BF16 || d[i].dtype_c != ABC_BF16 || d[i].dtype_out != ABC_BF16) return 0;
        if (abc_conv_fast_geom(&d[i], &g[i]) != ABC_OK || !g[i].eligible) return 0;
        if (g[i].wd || g[i].b_static || d[i].stride != 1 || g[i].CK != 32) return 0;
        if (i && (g[i].BN != g[0].BN || g[i].MT != g[0].MT || g[i].nwg != g[0].nwg || g[i].ntiles != g[0].ntiles)) return 0;
    }
    const int bn = g[0].BN, mt = g[0].MT;
    if (!((bn == 128 && (mt == 2 || mt == 4 || mt == 6)) || (bn == 64 && (mt == 2 || mt == 4 || mt == 8)) || (bn == 32 && (mt == 4 || mt == 8)))) return 0;
    int lds = 0;
    for (int i = 0; i < n; ++i) lds = g[i].lds > lds ? g[i].lds : lds;
    *g0 = g[0];
    g0->lds = lds;
    return 1;
}

int abc_conv_fast_launch_batch(const abc_conv_desc* d, int n, abc_stream_t stream) {
    abc_fast_geom g0;
    if (!abc_conv_fast_batch_ok(d, n, &g0)) return abc_fail(ABC_EUNSUPPORTED, "conv batch: not one geometry of the lean kernel");
    FastKB b;
    for (int i = 0; i < 4; ++i) {
        abc_fast_geom gi;
        const abc_conv_desc* di = &d[i < n ? i : 0];
        abc_conv_fast_geom(di, &gi);
        fill_fastk(di, gi, b.k[i]);
    }
    hipStream_t st = (hipStream_t)stream;
    const int bn = g0.BN, mt = g0.MT;
    if (bn == 128) {
        if (mt == 2) return launch_batch_inst<bf16, bf16, bf16, 32, 128, 2>(b, n, g0, st);
        if (mt == 4) return launch_batch_inst<bf16, bf16, bf16, 32, 128, 4>(b, n, g0, st);
        return launch_batch_inst<bf16, bf16, bf16, 32, 128, 6>(b, n, g0, st);
    }
    if (bn == 64) {
        if (mt == 2) return launch_batch_inst<bf16, bf16, bf16, 32, 64, 2>(b, n, g0, st);
        if (mt == 4) return launch_batch_inst<bf16, bf16, bf16, 32, 64, 4>(b, n, g0, st);
        return launch_batch_inst<bf16, bf16, bf16, 32, 64, 8>(b, n, g0, st);
    }
    if (mt == 4) return launch_batch_inst<bf16, bf16, bf16, 32, 32, 4>(b, n, g0, st);
    return launch_batch_inst<bf16, bf16, bf16, 32, 32, 8>(b, n, g0, st);
}

int abc_conv_fast_launch(const abc_conv_desc* d, const abc_fast_geom& g, abc_stream_t stream) {
    FastK k;
    fill_fastk(d, g, k);
    hipStream_t st = (hipStream_t)stream;
    const int di = d->dtype_in, dc = d->dtype_c, dout = d->dtype_out;
    if (d->heads_epi != nullptr) {
        // the heads' 1x1 convolutions in the epilogue (geometry checked in abc_conv_fast_geom)
        if (dc == ABC_FP8 && di == ABC_FP8 && d->out_scale != nullptr && d->out_quant != nullptr)
            return launch_st<f8, f8, f8, 64, 128, 1, 6, false, 9, 1>(k, g, st);
        if (dc == ABC_BF16 && di == ABC_BF16 && g.CK == 32) return launch_st<bf16, bf16, bf16, 32, 128, 1, 6, false, 9, 1>(k, g, st);
        return abc_fail(ABC_EUNSUPPORTED, "conv: heads_epi needs bf16 or e4m3 (with out_scale / out_quant) operands");
    }
    if (d->actbwd_y != nullptr) {
        // act_bwd in the epilogue (geometry and types checked in abc_conv_fast_geom)
        if (!(dc == ABC_BF16 && di == ABC_BF16 && dout == ABC_BF16 && g.CK == 32 && d->stride == 1 && !g.b_static) || !d->actbwd_scale || !d->actbwd_shift ||
            !d->actbwd_slope || !d->actbwd_mean || !d->actbwd_invstd)
            return abc_fail(ABC_EINVAL, "conv: actbwd_y needs a bf16 stride-1 data gradient and all five coefficient rows");
        return launch_bn<bf16, bf16, bf16, 32, 2>(k, g, 1, st);
    }
    if (dc == ABC_FP8 || dout == ABC_FP8 || di == ABC_FP8) {
        // the fp8 inference graph (abc_conv_desc.out_scale / out_quant): the 192-pixel x 128-channel weights-direct tile only
        if (!(g.BN == 128 && g.MT == 6 && g.wd == 9 && d->stride == 1)) return abc_fail(ABC_EUNSUPPORTED, "conv: fp8 needs the 3x3 weights-direct tile");
        if (dout == ABC_FP8 && d->out_quant == nullptr) return abc_fail(ABC_EINVAL, "conv: fp8 output needs out_quant");
        if (dc == ABC_FP8) {
            if (di != ABC_FP8 || d->out_scale == nullptr) return abc_fail(ABC_EINVAL, "conv: fp8 compute needs an fp8 input and out_scale");
            if (dout == ABC_FP8) return launch_st<f8, f8, f8, 64, 128, 1, 6, false, 9>(k, g, st);
            if (dout == ABC_BF16) return launch_st<f8, f8, bf16, 64, 128, 1, 6, false, 9>(k, g, st);
        } else if (dc == ABC_BF16 && di == ABC_BF16 && dout == ABC_FP8 && g.CK == 32) {
            return launch_st<bf16, bf16, f8, 32, 128, 1, 6, false, 9>(k, g, st);
        }
        return abc_fail(ABC_EUNSUPPORTED, "conv: fp8 dtype combination");
    }
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
