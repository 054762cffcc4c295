// Weight gradient of a tap-list convolution on the gfx950 matrix cores.
//
//   dW[t][a][b] = sum over pixels p of  P[p][a] * Q[stride*p + d_t][b]
//
// GEMM view per tap: M = a-channels, N = b-channels, K = pixels (split-K over 8x16 spatial
// patches across workgroups; every workgroup writes its own f32 slab and abc_wgrad_reduce sums
// the slabs in a fixed order -> bitwise reproducible, no atomics).
// Both operands have the reduction index (pixel) as the SLOW memory axis (NHWC), so the
// [pixel][channel] LDS images are read column-wise: in bf16 mode with the hardware-transposing
// ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group), in exact-f32 mode with plain
// ds_read_b32 (v_mfma_f32_32x32x2_f32 wants one value per lane).  One K-step = one 16-pixel row
// segment of the patch; the un-shifted operand's fragment is read once per K-step and reused by
// all taps.
//
// Workgroup = 8 waves (one per CU, two per SIMD) arranged AT x BT x RS: AT*BT 32x32 output tile
// pairs (every wave keeps all <= 9 taps of its pair: 144 accumulator VGPRs) and RS-way split of the
// patch rows (folded through LDS at the end).  Patches are double-buffered in LDS and the next patch's
// global loads are issued before / committed after the MFMA block (split-phase, T14), with the
// BatchNorm coefficients of both operands held in an LDS table.
//
// Reference ops covered: autograd of nn.Conv2d / nn.ConvTranspose2d weights
// (unet.py:12,15,44,66,70 under loss.backward(), train.py:140).
#include "common.hpp"
#include <type_traits>
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "reduce_bn.hpp"
#include "heads_fused.hpp"
#include "conv_fast.hpp"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int MAXT_FAST = 9;  // taps per workgroup (accumulator budget: 9 x 16 VGPRs)
constexpr int MAXT_SLOW = 5;  // the general loader needs the registers: fewer taps per workgroup, more tap groups
constexpr int WTHR_HOST = 512;  // 8 waves (the default workgroup)

struct WgK {
    ActSrc p, q;
    float* partial;
    int B, Hg, Wg, Hq, Wq;
    int cp_off, Ca, cq_off, Cb, Ca_pad, Cb_pad;
    int ntaps, tgw, nsplit, npatch, tiles_x, tiles_y;
    int dy_min, dx_min, HH, HW, PSWP, PSWQ, sP_bytes, sQ_bytes, coef_off, cstrP, cstrQ, nta, ntb, fast_p, fast_q, nbuf, dbg, magicQ, k3;
    unsigned mg_tx, mg_ty;   // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y): the patch index is split by two multiply-highs (scalar), not divisions
    int regP;                // P's patches are whole and P has no halo: its segment offsets are affine in the segment index
    int qtab_off;            // > 0: LDS byte offset of Q's segment table (HaloFetch::table_setup): whole patches, 3x3 halo, stride 1
    unsigned bytesP, bytesQ, bytesP2;
    const void* p2; void* p_out; int ld_p2, cp2_off, ld_pout;  // BN-backward correction fused into the load of P (DUAL)
    int8_t ty[ABC_MAX_TAPS], tx[ABC_MAX_TAPS];
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ inline bf16x8 tr_read8(const char* base0, const char* base1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)base0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)base1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// general (pool / dropout / planar / ragged-tail) loader kept out of line: inlined next to the accumulators and the
// prefetch registers it makes the register allocator spill inside the MFMA loop
template <typename T, typename CT, int CW>
__device__ inline void stage_slow(char* dst, int RS, int PS, int HH, int HW, int b, int iy0, int ix0, int Hin, int Win,
                                                     const ActSrc* s, int c0, int tid, int cvalid, int nthr) {
    stage_halo<T, CT, CW>(dst, RS, PS, HH, HW, b, iy0, ix0, Hin, Win, *s, c0, tid, nthr, cvalid);
}

// TS ("tap split", one 32x32 tile pair, more than 9 taps -- unet2's 5x5 convolutions): the 8 waves split the TAPS instead
// of the patch rows (wave w owns taps w, w + 8, w + 16, w + 24: <= 4 accumulators), every wave walks the whole patch, and
// ONE workgroup pass covers all taps.  With the row split 25 taps ran as 3 tap groups (grid.y) that each re-staged both
// operands: 596 MB fetched per launch against 302 MB algorithmic (profiles/r01_f_unet2_pmc_summary.json).
// NW = waves per workgroup: 8 (one workgroup per CU), or 4 with 2 x 2 tile pairs and a single LDS buffer, so that TWO
// workgroups share a CU: more bytes of prefetch in flight per CU (the kernel is bound by memory-level parallelism, DESIGN.md
// section 8) and one workgroup's commit phase under the other's MFMA block.
template <typename PT, typename QT, typename CT, int AT, int BT, int STRIDE, bool FAST, int PM, bool K3, bool DUAL = false, bool TS = false, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void wgrad_kernel(const WgK a) {
    constexpr int WTHR = NW * 64;
    static_assert(!TS || (AT == 1 && BT == 1 && FAST && !K3 && sizeof(CT) == 2), "tap split: one bf16 tile pair on the prefetch path");
    constexpr int MAXT = TS ? 4 : (FAST ? MAXT_FAST : MAXT_SLOW);
    constexpr int RSPLIT = TS ? 1 : NW / (AT * BT);   // waves sharing one tile pair, splitting the patch rows
    constexpr int PROWS = 8 * PM;           // patch rows (x 16 columns)
    constexpr int ROWS = PROWS / RSPLIT;    // patch rows per wave
    constexpr int CWP = AT * 32, CWQ = BT * 32;
    constexpr int NPF_P = (PROWS * 16 * (CWP / Frag<CT>::NV) + WTHR - 1) / WTHR;
    constexpr int NPF_Q = NW == 4 ? 6 : ((PM > 1 || STRIDE == 2) ? 5 : ((sizeof(CT) == 2) ? 4 : 6));  // (stride 2: 17 x 33 halo pixels)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int buf_bytes = a.sP_bytes + a.sQ_bytes;
    float* sCoefP = (float*)(smem + a.coef_off);
    float* sCoefQ = sCoefP + 3 * a.cstrP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // blocks with equal (blockIdx.x % 8) share an XCD and its L2: give each XCD a contiguous range of logical ids, so that
    // the nta x ntb workgroups of one split -- which re-read the same P / Q patches -- hit in L2 instead of going out again
    int id = abc_xcd_remap(blockIdx.x, gridDim.x);
    const int bt = id % a.ntb; id /= a.ntb;
    const int at = id % a.nta; id /= a.nta;
    const int split = id;
    const int t0 = blockIdx.y * a.tgw;
    const int tcnt = min(a.tgw, a.ntaps - t0);

    const int pair = TS ? 0 : wave / RSPLIT, rs = TS ? 0 : wave % RSPLIT;
    // tap owned by accumulator slot j (TS: wave-uniform, kept scalar)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto tap_of = [&](int j) { return TS ? wave_u + 8 * j : t0 + j; };
    auto tap_ok = [&](int j) { return TS ? (wave_u + 8 * j < a.ntaps) : (j < tcnt); };
    const int ai = pair / BT, bi = pair % BT;
    const int row_lo = rs * ROWS;
    const int PSWP = a.PSWP, PSWQ = a.PSWQ;
    const int ca0 = at * CWP, cb0 = bt * CWQ;          // first channel of the workgroup's a / b range
    const int cvalP = a.Ca - ca0, cvalQ = a.Cb - cb0;  // valid channels from there

    // ---- BatchNorm coefficient tables of both operands (relative channel index)
    // FAST: both operands are plain NHWC tensors -> split-phase prefetch; otherwise (pool / dropout / planar /
    // ragged channel tail on either operand) both are staged synchronously by the general loader.  Two separate
    // instantiations: inlining the general loader next to the prefetch registers makes the allocator spill.
    const bool coefP = FAST && a.p.scale != nullptr, coefQ = FAST && a.q.scale != nullptr;
    if (coefP)
        for (int i = tid; i < min(CWP, cvalP); i += WTHR) {
            sCoefP[i] = a.p.scale[a.cp_off + ca0 + i]; sCoefP[a.cstrP + i] = a.p.shift[a.cp_off + ca0 + i];
            sCoefP[2 * a.cstrP + i] = a.p.slope[a.cp_off + ca0 + i];
        }
    if (coefQ)
        for (int i = tid; i < min(CWQ, cvalQ); i += WTHR) {
            sCoefQ[i] = a.q.scale[a.cq_off + cb0 + i]; sCoefQ[a.cstrQ + i] = a.q.shift[a.cq_off + cb0 + i];
            sCoefQ[2 * a.cstrQ + i] = a.q.slope[a.cq_off + cb0 + i];
        }

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    unsigned* const qtab = (unsigned*)(smem + a.qtab_off);

    int tapoff[MAXT];  // LDS byte offset of each tap inside the Q halo
#pragma unroll
    for (int t = 0; t < MAXT; ++t) tapoff[t] = tap_ok(t) ? (a.ty[tap_of(t)] * a.HW + a.tx[tap_of(t)]) * PSWQ : 0;

    // per-lane channel byte offsets inside a pixel
    int pch, qch;
    if constexpr (sizeof(CT) == 2) {
        const int sub = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        pch = (ai * 32 + sub) * 2;
        qch = (bi * 32 + sub) * 2;
    } else {
        pch = (ai * 32 + r) * 4;
        qch = (bi * 32 + r) * 4;
    }

    HaloFetch<PT, CT, CWP, FAST ? NPF_P : 1, WTHR> pp;
    HaloFetch<QT, CT, CWQ, FAST ? NPF_Q : 1, WTHR> pq;
    const __amdgpu_buffer_rsrc_t rsP = abc_make_rsrc(a.p.x, a.bytesP), rsQ = abc_make_rsrc(a.q.x, a.bytesQ);
    // DUAL: P = ca * g + cb * y_raw + cc (the BatchNorm-backward correction, abc_bn_bwd_desc.ca/cb/cc), g and y_raw
    // fetched side by side; the corrected values go to LDS for the MFMAs and (from the b-tile-0 workgroups) to p_out
    // for the data-gradient conv: one pass over g and y instead of abc_bn_apply_bwd's read-modify-write plus a re-read
    HaloFetch<PT, CT, CWP, (FAST && DUAL) ? NPF_P : 1, WTHR> pp2;
    const __amdgpu_buffer_rsrc_t rsP2 = abc_make_rsrc(DUAL ? a.p2 : a.p.x, DUAL ? a.bytesP2 : a.bytesP);
    const HaloGeom gP2 = {PROWS, 16, 4097, a.Hg, a.Wg, a.Hg, a.Wg, a.ld_p2};
    const HaloGeom gP = {PROWS, 16, 4097, a.Hg, a.Wg, a.p.Hx, a.p.Wx, a.p.ldx};
    const HaloGeom gQ = {a.HH, a.HW, a.magicQ, a.Hq, a.Wq, a.q.Hx, a.q.Wx, a.q.ldx};

    // (patch < 2^16 and tiles <= 2^12: the multiply-high by ceil(2^32 / d) is the exact quotient; wave-uniform, stays on the scalar
    //  unit -- the two runtime divisions cost ~50 vector instructions per call, twice per patch)
    // Q's segment geometry once per kernel (FAST && the host found room for the table: a.qtab_off)
    // (compiled into the instantiations whose waves own a tile pair each: the narrow-layer forms are at their register limit)
    constexpr bool QTAB = FAST && K3 && STRIDE == 1 && NW == 8 && !TS && sizeof(CT) == 2 && sizeof(QT) == 2 && AT * BT >= 4;
    unsigned qmask = 0;
    if constexpr (QTAB) {
        if (a.qtab_off) qmask = pq.table_setup(qtab, gQ, a.HW * PSWQ, PSWQ, tid, cvalQ, PROWS, a.dy_min, a.dx_min);
    }
    auto patch_origin = [&](int patch, int& b, int& gy0, int& gx0) {
        const unsigned pid = (unsigned)patch;
        // (a divisor of 1 has no 32-bit magic: ceil(2^32 / 1) = 2^32)
        const unsigned q1 = a.tiles_x == 1 ? pid : __umulhi(pid, a.mg_tx);
        const unsigned tx_i = pid - q1 * (unsigned)a.tiles_x;
        const unsigned q2 = a.tiles_y == 1 ? q1 : __umulhi(q1, a.mg_ty);
        const unsigned ty_i = q1 - q2 * (unsigned)a.tiles_y;
        b = (int)q2; gy0 = (int)ty_i * PROWS; gx0 = (int)tx_i * 16;
    };
    // SPREAD: the next patch's loads are not issued as one batch in front of the MFMA block but a few at a time between
    // its K-steps.  A CU's memory pipeline holds far less than a patch (55 .. 87 KB from 8 waves): the waves of a batch sat
    // in their load instructions until most of the data had come back, so loads and MFMAs never overlapped (ablations on the
    // eight heads' merged weight gradient: loads only 193 us, MFMAs only 214 us, both 329 us, with the commit 571 us).
    constexpr bool SPREAD = FAST && K3 && sizeof(CT) == 2 && !TS && ROWS >= 4;
    auto prepare = [&](int patch, bool live) {
        int b, gy0, gx0;
        patch_origin(patch, b, gy0, gx0);
        if constexpr (FAST) {
            // all address arithmetic first (see HaloFetch::prepare)
            if (live) {
                if (a.regP) pp.prepare_regular(gP, b, gy0, gx0, a.cp_off + ca0, tid, cvalP);
                else pp.prepare(gP, b, gy0, gx0, a.cp_off + ca0, tid, cvalP);
                if constexpr (DUAL) {
                    // (g and y_raw: same patch, same segments; with equal pixel strides the offsets differ by a constant)
                    if (a.ld_p2 == a.p.ldx && a.p.Hx == a.Hg && a.p.Wx == a.Wg)
                        pp2.prepare_like(pp, (unsigned)((a.cp2_off - a.cp_off) * (int)sizeof(PT)));
                    else
                        pp2.prepare(gP2, b, gy0, gx0, a.cp2_off + ca0, tid, cvalP);
                }
                if (QTAB && a.qtab_off) {
                    const unsigned sbase = (unsigned)(((b * a.q.Hx + gy0 + a.dy_min) * a.q.Wx + gx0 + a.dx_min) * a.q.ldx + a.cq_off + cb0) * (unsigned)sizeof(QT);
                    const unsigned border = (gy0 == 0 ? 0x1Fu << 5 : 0u) | (gy0 + PROWS == a.Hq ? 0x1Fu << 10 : 0u) | (gx0 == 0 ? 0x1Fu << 15 : 0u) |
                                            (gx0 + 16 == a.Wq ? 0x1Fu << 20 : 0u);
                    pq.prepare_tab(qtab, qmask, sbase, border, tid);
                } else {
                    pq.prepare(gQ, b, gy0 * STRIDE + a.dy_min, gx0 * STRIDE + a.dx_min, a.cq_off + cb0, tid, cvalQ);
                }
            } else {
                pp.prepare_none();
                if constexpr (DUAL) pp2.prepare_none();
                pq.prepare_none();
            }
        }
    };
    auto fire_all = [&]() {
        if constexpr (FAST) {
            __builtin_amdgcn_sched_barrier(0);
            pp.fire(rsP);
            if constexpr (DUAL) pp2.fire(rsP2);
            pq.fire(rsQ);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (over the FIRST half of the K-steps, so that the last load has the second half to come back in)
    constexpr int NPH = ROWS / 2;
    auto fire_slice = [&](int phase) {   // phase < NPH
        if constexpr (FAST) {
            pp.fire_slice(rsP, 0, phase, NPH);
            if constexpr (DUAL) pp2.fire_slice(rsP2, NPF_P, phase, NPH);
            pq.fire_slice(rsQ, DUAL ? 2 * NPF_P : NPF_P, phase, NPH);
        }
    };
    auto issue = [&](int patch) {
        prepare(patch, true);
        fire_all();
    };
    auto commit = [&](int patch, char* buf) {
        int b, gy0, gx0;
        patch_origin(patch, b, gy0, gx0);
        char* sP = buf;
        char* sQ = buf + a.sP_bytes;
        if constexpr (FAST) {
            if constexpr (DUAL) {
                constexpr int NV = Frag<CT>::NV;
                const int SEGS = pp.live_segs(cvalP);   // same thread -> segment mapping as HaloFetch::issue (a power of two)
                const int ssh = pp.live_shift(SEGS);
                const int lt = abc_launder(tid);
                const int part = lt & (SEGS - 1), cch = part * NV;
                const bool chan = cch < cvalP;
                float ka[NV], kb[NV], kc[NV];
                const float kzero[NV] = {};
#pragma unroll
                for (int j = 0; j < NV; ++j) { ka[j] = sCoefP[cch + j]; kc[j] = sCoefP[a.cstrP + cch + j]; kb[j] = sCoefP[2 * a.cstrP + cch + j]; }
                const bool store = (bt == 0) && (blockIdx.y == 0) && a.p_out != nullptr;
                // dY goes out through a buffer store with a 32-bit offset (an out-of-range offset drops it: no branch, no 64-bit
                // address arithmetic per segment; soffset 0 -- the store form LLVM's hazard recogniser covers)
                const __amdgpu_buffer_rsrc_t rsO = abc_make_rsrc(store ? a.p_out : a.p.x, store ? (unsigned)((size_t)a.B * a.Hg * a.Wg * a.ld_pout * sizeof(CT)) : 0u);
                const int obase = ((b * a.Hg + gy0) * a.Wg + gx0) * a.ld_pout + ca0 + cch;
#pragma unroll
                for (int i = 0; i < NPF_P; ++i) {
                    const int sidx = lt + i * WTHR;
                    if (sidx < PROWS * 16 * SEGS) {
                        const int pix = sidx >> ssh, hy = pix >> 4, hx = pix & 15;
                        float v1[NV], v2[NV];
                        pp.raw[i].get(v1); pp2.raw[i].get(v2);
                        const bool in = chan && ((pp.inb >> i) & 1u);
                        abc_fma2_n<NV>(v1, v2, ka, kb, kc);
                        typename Frag<CT>::type f = pack_frag<CT>(v1);
                        if (!in) f = pack_frag<CT>(kzero);      // (a select on the packed words: 4 instead of 8)
                        *(typename Frag<CT>::type*)(sP + (hy * 16 + hx) * PSWP + part * 16) = f;
                        static_assert(sizeof(f) == 16, "one 16-byte segment");
                        __builtin_amdgcn_raw_buffer_store_b128(*(u32x4*)&f, rsO,
                                                               (store && in) ? (unsigned)(obase + (hy * a.Wg + hx) * a.ld_pout) * (unsigned)sizeof(CT) : 0xFFFFFFF0u, 0, 0);
                    }
                }
            } else {
                pp.commit(sP, 16 * PSWP, PSWP, gP, coefP ? sCoefP : nullptr, a.cstrP, tid, cvalP);
            }
            if (QTAB && a.qtab_off) pq.commit_tab(sQ, qtab, qmask, coefQ ? sCoefQ : nullptr, a.cstrQ, tid, cvalQ);
            else pq.commit(sQ, a.HW * PSWQ, PSWQ, gQ, coefQ ? sCoefQ : nullptr, a.cstrQ, tid, cvalQ);
        } else {
            stage_slow<PT, CT, CWP>(sP, 16 * PSWP, PSWP, PROWS, 16, b, gy0, gx0, a.Hg, a.Wg, &a.p, a.cp_off + ca0, tid, cvalP, WTHR);
            stage_slow<QT, CT, CWQ>(sQ, a.HW * PSWQ, PSWQ, a.HH, a.HW, b, gy0 * STRIDE + a.dy_min, gx0 * STRIDE + a.dx_min, a.Hq, a.Wq,
                                    &a.q, a.cq_off + cb0, tid, cvalQ, WTHR);
        }
    };

    if constexpr (FAST) {
        // zero both patch buffers once: channel padding (Ca / Cb below the 32-wide tile) is never written again
        if (cvalP < CWP || cvalQ < CWQ) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid * 16; i < a.nbuf * buf_bytes; i += WTHR * 16) *(f32x4*)(smem + i) = z;
        }
    }
    __syncthreads();  // coefficient tables visible
    // it = -1 is the prologue (stage the first patch, no compute): one call site for issue / commit
    int patch = split - a.nsplit;
    for (int it = -1; it < 0 || patch < a.npatch; ++it, patch += a.nsplit) {
        const int next = patch + a.nsplit;
        const bool has_next = next < a.npatch;
        if constexpr (SPREAD) {
            prepare(has_next ? next : patch, has_next);
            if (it < 0) fire_all();     // prologue: nothing to hide behind
        } else {
            if (has_next && !(ABC_DBG(a.dbg) & 1)) issue(next);
        }
        if (it >= 0 && !(ABC_DBG(a.dbg) & 4)) {
            const char* sP = smem + ((a.nbuf == 2) ? (it & 1) * buf_bytes : 0);
            const char* sQ = sP + a.sP_bytes;
            // ROT (stride 1, fully unrolled K-steps): tap (dy, dx) of patch row r reads the SAME fragment as tap (dy - 1, dx) of row
            // r + 1 (same halo pixels, same lanes), so only the three fragments of the NEW halo row are read per K-step and the
            // other six are carried in their registers: tap t of K-step rr lives in slot (t + 3 rr) mod 9.  4 instead of 10
            // transposing LDS reads per K-step (SQ_WAIT_INST_LDS 7.9 M -> 2.6 M wave-cycles per launch; the launch itself gains 4 %:
            // the LDS was not what bound the matrix phase -- profiles/r04_wgrad_pingpong.md).
            constexpr bool ROT = SPREAD && STRIDE == 1;
            bf16x8 fbr[9];
#pragma unroll(SPREAD ? ROWS : 1)
            for (int rr = 0; rr < ROWS; ++rr) {
                const int row = row_lo + rr;
                if constexpr (sizeof(CT) == 2 && FAST && K3 && ROT) {
                    constexpr int HW3 = 15 * STRIDE + 3;
                    constexpr int PQ = CWQ == 32 ? 64 : (CWQ == 64 ? 192 : 320);
                    constexpr int PP = CWP == 32 ? 64 : (CWP == 64 ? 192 : 320);
                    const int kq = 8 * h + ((lane & 15) >> 2);
                    const char* pa = sP + (row * 16 + kq) * PP + pch;
                    const char* qb = sQ + (row * HW3 + kq) * PQ + qch;
                    const bf16x8 fa = tr_read8(pa, pa + 4 * PP);
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        if (rr == 0 || t >= 6) {
                            const int off = ((t / 3) * HW3 + (t % 3)) * PQ;
                            fbr[(t + 3 * rr) % 9] = tr_read8(qb + off, qb + off + 4 * PQ);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fbr[(t + 3 * rr) % 9], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (rr < NPH) fire_slice(rr);   // (unconditional: without a next patch the offsets are out of range)
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (sizeof(CT) == 2 && FAST && K3) {
                    {
                        // the usual case, a full 3x3 tap square: tap offsets are immediates, all 10 fragments of the
                        // K-step are read first (one wait), then 9 MFMAs back to back; the generic loop below pays a
                        // branch, two address adds and an exposed LDS round trip per tap
                        constexpr int HW3 = 15 * STRIDE + 3;
                        constexpr int PQ = CWQ == 32 ? 64 : (CWQ == 64 ? 192 : 320);
                        constexpr int PP = CWP == 32 ? 64 : (CWP == 64 ? 192 : 320);
                        const int kq = 8 * h + ((lane & 15) >> 2);
                        const char* pa = sP + (row * 16 + kq) * PP + pch;
                        const char* qb = sQ + ((row * STRIDE) * HW3 + kq * STRIDE) * PQ + qch;
                        const bf16x8 fa = tr_read8(pa, pa + 4 * PP);
                        bf16x8 fb[9];
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            constexpr int dummy = 0; (void)dummy;
                            const int off = ((t / 3) * HW3 + (t % 3)) * PQ;
                            fb[t] = tr_read8(qb + off, qb + off + 4 * STRIDE * PQ);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[t], acc[t], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (SPREAD) {
                            if (rr < NPH) fire_slice(rr);   // (unconditional: without a next patch the offsets are out of range)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else if constexpr (sizeof(CT) == 2) {
                    // lane supplies the address of pixel k = 8h + 4q + ((lane&15)>>2), 4 channels
                    const int kq = 8 * h + ((lane & 15) >> 2);
                    const char* pa = sP + (row * 16 + kq) * PSWP + pch;
                    const bf16x8 fa = tr_read8(pa, pa + 4 * PSWP);
                    const char* qb = sQ + ((row * STRIDE) * a.HW + kq * STRIDE) * PSWQ + qch;
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) {
                        if (tap_ok(t)) {
                            const bf16x8 fb = tr_read8(qb + tapoff[t], qb + tapoff[t] + 4 * STRIDE * PSWQ);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll 2
                    for (int s = 0; s < 8; ++s) {
                        const int k = 2 * s + h;
                        const float fa = *(const float*)(sP + (row * 16 + k) * PSWP + pch);
                        const char* qb = sQ + ((row * STRIDE) * a.HW + k * STRIDE) * PSWQ + qch;
#pragma unroll
                        for (int t = 0; t < MAXT; ++t) {
                            if (t < tcnt) {
                                const float fb = *(const float*)(qb + tapoff[t]);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (a.nbuf == 1) __syncthreads();  // single buffer: everyone is done reading before it is refilled
        }
        if (has_next && !(ABC_DBG(a.dbg) & 2)) commit(next, smem + ((a.nbuf == 2) ? ((it + 1) & 1) * buf_bytes : 0));
        __syncthreads();
    }

    if constexpr (RSPLIT > 1) {
        // the RSPLIT waves of a tile pair hold partial sums over different patch rows: fold into rs == 0
        float* red = (float*)smem;  // [8 waves][16][64] per tap round
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < tcnt) {
                __syncthreads();
                if (rs > 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) red[(wave * 16 + k) * 64 + lane] = acc[t][k];
                }
                __syncthreads();
                if (rs == 0) {
                    for (int w = 1; w < RSPLIT; ++w)
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[t][k] += red[((wave + w) * 16 + k) * 64 + lane];
                }
            }
        }
        if (rs != 0) return;
    }

#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (tap_ok(t)) {
            float* out = a.partial + ((size_t)(split * a.ntaps + tap_of(t)) * a.Ca_pad + ca0 + ai * 32) * a.Cb_pad + cb0 + bi * 32 + r;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int arow = (k & 3) + 8 * (k >> 2) + 4 * h;
                out[(size_t)arow * a.Cb_pad] = acc[t][k];
            }
        }
    }
}

// Sum the split-K slabs.  One thread per (tap, a, b) output element; consecutive threads walk
// b, so every slab read is a coalesced 4-byte stream and the whole reduction is one pass at
// HBM/L2 speed (slab order fixed -> bitwise reproducible).  Output = reference layout [a][b][tap].
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const abc_wgrad_reduce_desc d) { wgrad_reduce_body(d, blockIdx.x); }

__global__ __launch_bounds__(256) void wgrad_reduce_vec_kernel(const abc_wgrad_reduce_desc d) { wgrad_reduce_vec_body(d, blockIdx.x); }

// one launch: the slab reduction of one layer (blocks [0, nr)) beside the BatchNorm-backward finaliser of the next (reduce_bn.hpp)
struct ReduceBn { abc_wgrad_reduce_desc r; abc_bn_bwd_desc f; int nr, vec; };
__global__ __launch_bounds__(256) void wgrad_reduce_bn_kernel(const ReduceBn a) {
    if ((int)blockIdx.x < a.nr) {
        if (a.vec) wgrad_reduce_vec_body(a.r, blockIdx.x); else wgrad_reduce_body(a.r, blockIdx.x);
    } else {
        bn_finalize_bwd_body(a.f, a.f.C, (int)blockIdx.x - a.nr);
    }
}

// ---------------------------------------------------------------------------
// Weight gradient of a head's 1x1 convolution: dW[a][b] = sum_p dL[a][p] * act(H[p][b]) with dL the channel-planar
// f32 gradient map (the reference's NCHW logits layout, unet.py:119) and H the NHWC feature map (BN + LeakyReLU +
// dropout applied on load).  HBM-bound on dL (hc x pixels x 4 B, 212 MB for the 360-channel head): dL is already
// pixel-contiguous per row, i.e. exactly the A-operand layout of the MFMA, so it goes global -> registers with each
// lane reading 256 contiguous bytes per 128-pixel chunk (a whole chunk prefetched ahead); only H is staged and
// transposed through LDS.  Workgroup = 8 waves = 4 m-tiles x 2 halves of the 128 b-channels; grid = m-groups x
// K-splits, slabs reduced by abc_wgrad_reduce like every other weight gradient.
struct HeadK {
    const float* dl;
    const float *psc, *psh, *psl;  // per-row transform of dL (scale = d(loss weight), shift 0, slope 1) or null
    const void* q;
    const float *qsc, *qsh, *qsl;
    float* partial;
    float* rowsum;   // [nsplit][Ca_pad] or null
    int HW, hc, ldq, cq_off, nchunks, nsplit, mtiles, Ca_pad;
    float drop_p;
    uint32_t drop_seed;
    const uint32_t* drop_salt;
    unsigned bytesP, bytesQ;
    int cpad_blk;    // BLK: dl = bf16 [chunk][cpad_blk rows][128 pixels], written by the fused heads kernel (heads_fused.hip)
    const uint8_t* keep;   // BLK: the fused kernel's dropout keep bits, 16 bytes per pixel (byte kk + 8 h = channels 16 kk + 8 h ..), or null
};

constexpr int HQ_PSW = 320;  // pixel stride of the [pixel][128 channel] bf16 LDS image (wgrad Q layout)
constexpr int HP_RSW = 272;  // row stride of the [dL row][128 pixel] bf16 LDS image: 16 consecutive rows = 16 distinct 16-byte bank slots
constexpr int HEAD_PBUF = 128 * HP_RSW, HEAD_QBUF = 128 * HQ_PSW;
constexpr int HEAD_LDS = 2 * (HEAD_PBUF + HEAD_QBUF) + 3 * 128 * 4;

// dL used to go global -> registers with each lane reading ITS row (256 contiguous bytes per chunk): 64 lanes = 64 rows
// 36 KB apart, i.e. 64 cache lines touched per load instruction for 16 useful bytes each -- the kernel ran at the texture
// addresser's line rate, 7.9 us per 128-pixel chunk (1.7 TB/s for all heads together).  Now both operands are loaded
// coalesced (a wave instruction = two whole 512-byte rows of dL) one chunk ahead, transformed, and written to LDS as bf16:
// dL as [row][pixel] (the A fragment of a K-step is one ds_read_b128), the features as [pixel][channel] (read transposed).
template <bool BLK>
__device__ inline void head_wgrad_body(const HeadK& a, const int split, const int mg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mi = wave & 3, nh = wave >> 2;
    if (split >= a.nsplit || mg * 4 >= a.mtiles) return;   // (batched launch: the grid is sized for the largest head)
    const int mt = mg * 4 + mi;
    // (wave-uniform IN A SCALAR REGISTER: an MFMA under a lane-dependent branch is not safe, the instruction ignores EXEC)
    const bool active = __builtin_amdgcn_readfirstlane(mt) < a.mtiles;
    const bool ptrans = a.psc != nullptr;
    const int c0 = (int)((long long)split * a.nchunks / a.nsplit), c1 = (int)((long long)(split + 1) * a.nchunks / a.nsplit);

    const __amdgpu_buffer_rsrc_t rsP = abc_make_rsrc(a.dl, a.bytesP), rsQ = abc_make_rsrc(a.q, a.bytesQ);
    // Q staging: 128 pixels x 16 segments of 8 channels over 512 threads -> 4 per thread, same channel segment always
    const int part = tid & 15, pix0 = tid >> 4;  // segment i: pixel pix0 + 32 i
    const bool qtrans = a.qsc != nullptr;
    float* sCoef = (float*)(smem + 2 * (HEAD_PBUF + HEAD_QBUF));  // [3][128]
    if (qtrans && tid < 128) {
        sCoef[tid] = a.qsc[a.cq_off + tid]; sCoef[128 + tid] = a.qsh[a.cq_off + tid]; sCoef[256 + tid] = a.qsl[a.cq_off + tid];
    }
    // P staging: 128 rows x 32 segments of 4 pixels -> 8 per thread; segment i of a thread: row prow0 + 16 i, pixels 4 pseg ..
    const int pseg = tid & 31, prow0 = tid >> 5;
    float psc[8], psh[8], psl[8], rsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int co = mg * 128 + prow0 + 16 * i;
        const bool ok = co < a.hc;
        psc[i] = (ok && ptrans) ? a.psc[co] : 1.f; psh[i] = (ok && ptrans) ? a.psh[co] : 0.f; psl[i] = (ok && ptrans) ? a.psl[co] : 1.f;
        rsum[i] = 0.f;
    }
    __syncthreads();
    const float dscale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const uint32_t dseed = a.drop_seed + ((a.drop_p > 0.f && a.drop_salt) ? *a.drop_salt : 0u);
    // TWO chunks of prefetch in flight (two named register sets): with one, an iteration was the loaded HBM latency
    // (~4 us for 96 KB per CU) plus the commit -- the 16 MFMAs per wave of a chunk hide nothing.  Loads are issued
    // unconditionally (past the last chunk with an out-of-range offset: zeros, no traffic) so that vmcnt stays exact.
    u32x4 qreg0[4], preg0[8], qreg1[4], preg1[8];
    // (BLK with the fused kernel's keep bits: the four pixels' mask bytes of this thread's channel segment ride in preg[4],
    //  which the blocked form does not use for d(logits))
    const bool kmask = BLK && a.keep != nullptr && a.drop_p > 0.f;
    const __amdgpu_buffer_rsrc_t rsK = abc_make_rsrc(kmask ? a.keep : (const uint8_t*)a.q, kmask ? (unsigned)a.nchunks * 2048u : 0u);
    const unsigned kbyte = (unsigned)((part & 1) * 8 + (part >> 1));
    const int CPI = a.HW / 128;  // chunks per image
    auto issue = [&](int c, u32x4 (&qreg)[4], u32x4 (&preg)[8]) {
        const bool live = c < c1;
        const int b = c / CPI, pp0 = (c - b * CPI) * 128;
        const unsigned qoff = live ? (unsigned)(((unsigned)(c * 128 + pix0) * (unsigned)a.ldq + (unsigned)(a.cq_off + part * 8)) * 2u) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) qreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, live ? qoff + (unsigned)(i * 32 * a.ldq * 2) : qoff, 0, 0);
        if constexpr (BLK) {
            if (kmask) {
                const unsigned koff = live ? (unsigned)(c * 128 + pix0) * 16u + kbyte : 0x80000000u;
#pragma unroll
                for (int i = 0; i < 4; ++i) preg[4][i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsK, live ? koff + (unsigned)(i * 32 * 16) : koff, 0, 0);
            }
            // the chunk's 128 rows are ONE contiguous 32 KB block: 16-byte piece q = tid + 512 i = (row q >> 4, pixels 8 (q & 15) ..)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = mg * 128 + (tid >> 4) + 32 * i;
                const unsigned poff = (live && row < a.cpad_blk) ? (unsigned)((((unsigned)c * (unsigned)a.cpad_blk + (unsigned)row) * 128u + 8u * (tid & 15)) * 2u) : 0x80000000u;
                preg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsP, poff, 0, 0);
            }
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int co = mg * 128 + prow0 + 16 * i;
            const unsigned poff = (live && co < a.hc) ? (unsigned)((((unsigned)(b * a.hc + co)) * (unsigned)a.HW + (unsigned)(pp0 + 4 * pseg)) * 4u) : 0x80000000u;
            preg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsP, poff, 0, 0);
        }
        }
    };
    auto commit = [&](int c, char* sP, char* sQ, const u32x4 (&qreg)[4], const u32x4 (&preg)[8]) {
        float qsc[8], qsh[8], qsl[8];
        if (qtrans) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { qsc[j] = sCoef[part * 8 + j]; qsh[j] = sCoef[128 + part * 8 + j]; qsl[j] = sCoef[256 + part * 8 + j]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = pix0 + 32 * i;
            const uint32_t eoff = (uint32_t)(c * 128 + pix) * (uint32_t)a.ldq + (uint32_t)(a.cq_off + part * 8);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(qreg[i][j] << 16); v[2 * j + 1] = __uint_as_float(qreg[i][j] & 0xFFFF0000u); }
            if (qtrans) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = abc_act(v[j], qsc[j], qsh[j], qsl[j]);
            }
            if (BLK && kmask) {
                const unsigned kb = preg[4][i];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ((kb >> j) & 1u) ? v[j] * dscale : 0.f;
            } else if (a.drop_p > 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = abc_drop_keep(eoff + j, dseed, a.drop_p) ? v[j] * dscale : 0.f;
            }
            *(bf16x8*)(sQ + pix * HQ_PSW + part * 16) = pack_frag<bf16>(v);
        }
        if constexpr (BLK) {
            // already bf16 in the operand layout: a copy, with the row sums (bias gradient) on the way
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float f = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) f += __uint_as_float(preg[i][j] << 16) + __uint_as_float(preg[i][j] & 0xFFFF0000u);
                rsum[i] += f;
                *(u32x4*)(sP + ((tid >> 4) + 32 * i) * HP_RSW + (tid & 15) * 16) = preg[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = mg * 128 + prow0 + 16 * i < a.hc;
            float f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[j] = __uint_as_float(preg[i][j]);
                if (ptrans) f[j] = abc_act(f[j], psc[i], psh[i], psl[i]);
                if (!ok) f[j] = 0.f;
            }
            rsum[i] += (f[0] + f[1]) + (f[2] + f[3]);   // sum over pixels of the transformed dL = the conv's bias gradient
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)f[j];
            *(bf16x4*)(sP + (prow0 + 16 * i) * HP_RSW + pseg * 8) = o;
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;

    const int sub = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int qlane = (64 * h + ((lane & 15) >> 2)) * HQ_PSW + sub * 2;  // + 8 kk pixels + b-tile * 64 bytes
    // K order inside a chunk: MFMA K-step kk covers pixels {64 h + 8 kk + j}: the same pixel set for both operands
    const int plane = (mi * 32 + r) * HP_RSW + (64 * h) * 2;              // + 8 kk pixels

    auto compute = [&](const char* sP, const char* sQ) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 fa = *(const bf16x8*)(sP + plane + kk * 16);
            const char* qb = sQ + qlane + kk * 8 * HQ_PSW + nh * 128;
            const bf16x8 fb0 = tr_read8(qb, qb + 4 * HQ_PSW);
            const bf16x8 fb1 = tr_read8(qb + 64, qb + 64 + 4 * HQ_PSW);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb1, acc[1], 0, 0, 0);
        }
    };

    constexpr int BUF = HEAD_PBUF + HEAD_QBUF;
    char* const b0 = smem;
    char* const b1 = smem + BUF;
    issue(c0, qreg0, preg0);
    issue(c0 + 1, qreg1, preg1);
    if (c0 < c1) commit(c0, b0, b0 + HEAD_PBUF, qreg0, preg0);
    __syncthreads();
    // even chunks (relative) live in LDS buffer 0 and come from register set 0, odd ones buffer 1 / set 1
    for (int c = c0; c < c1; c += 2) {
        issue(c + 2, qreg0, preg0);
        if (active) compute(b0, b0 + HEAD_PBUF);
        if (c + 1 < c1) commit(c + 1, b1, b1 + HEAD_PBUF, qreg1, preg1);
        __syncthreads();
        if (c + 1 < c1) {
            issue(c + 3, qreg1, preg1);
            if (active) compute(b1, b1 + HEAD_PBUF);
            if (c + 2 < c1) commit(c + 2, b0, b0 + HEAD_PBUF, qreg0, preg0);
            __syncthreads();
        }
    }
    if (BLK && a.rowsum != nullptr) {
        // a row's 16 pieces sit in 16 consecutive lanes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rsum[i];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            const int row = mg * 128 + (tid >> 4) + 32 * i;
            if ((tid & 15) == 0 && row < a.Ca_pad) a.rowsum[(size_t)split * a.Ca_pad + row] = v;
        }
    } else if (a.rowsum != nullptr) {
        // a row's 32 segments sit in the 32 lanes of one half-wave
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = rsum[i];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            const int row = mg * 128 + prow0 + 16 * i;
            if (pseg == 0 && row < a.Ca_pad) a.rowsum[(size_t)split * a.Ca_pad + row] = v;
        }
    }
    if (active) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* out = a.partial + ((size_t)split * a.Ca_pad + mt * 32) * 128 + (nh * 2 + j) * 32 + r;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int arow = (k & 3) + 8 * (k >> 2) + 4 * h;
                out[(size_t)arow * 128] = acc[j][k];
            }
        }
    }
}

// dL planar f32 (x) activated NHWC bf16 features, 1x1, 128 b-channels, whole 128-pixel chunks per image
__global__ __launch_bounds__(512, 2) void head_wgrad_kernel(const HeadK a) { head_wgrad_body<false>(a, blockIdx.x, blockIdx.y); }
// all heads in one launch (blockIdx.z = head), as abc_heads_batch does for the forward and the data gradient
struct HeadWgBatch { HeadK k[8]; int first[9]; };
__global__ __launch_bounds__(512, 2) void head_wgrad_batch_kernel(const HeadWgBatch bt) { head_wgrad_body<false>(bt.k[blockIdx.z], blockIdx.x, blockIdx.y); }
// ... with d(logits) from the fused heads kernel's blocked bf16 buffer (abc_heads_fused_wgrad).  A DENSE one-dimensional grid:
// workgroup id -> (head, K-split, m-group) through the prefix table `first` (m-groups of a split adjacent: they stage the same
// feature chunks).  As a (split, m-group, head) box sized for the largest head the grid was half empty workgroups; every one of
// them still claims a CU's 150 KB of LDS for its moment, the dispatcher dealt the ~256 real ones unevenly -- some CUs ran two
// one after the other while others idled: waves alive for 82 us of a 162 us launch (SQ_WAVE_CYCLES against GRBM_GUI_ACTIVE).
__global__ __launch_bounds__(512, 2) void head_wgrad_blocked_kernel(const HeadWgBatch bt) {
    const int id = blockIdx.x;
    int hd = 0;
    while (hd < 7 && id >= bt.first[hd + 1]) ++hd;
    const HeadK& k = bt.k[hd];
    const int units = (k.mtiles + 3) >> 2, local = id - bt.first[hd];
    head_wgrad_body<true>(k, local / units, local % units);
}

static bool head_ok(const abc_wgrad_desc* d) {
    if (abc_knob("ABC_WGRAD_NOHEAD")) return false;
    if (!d->p.planar || d->dtype_p != ABC_F32 || d->dtype_q != ABC_BF16 || d->dtype_c != ABC_BF16) return false;
    if (d->ntaps != 1 || d->tap_dy[0] != 0 || d->tap_dx[0] != 0 || d->stride != 1 || d->Cb != 128 || d->cp_off != 0) return false;
    if (d->q.pool || d->q.planar || d->p.pool || d->p.drop_p > 0.f || (d->Hg * d->Wg) % 128) return false;
    if (d->p.ctot != d->Ca) return false;
    const int64_t bp = (int64_t)d->B * d->Ca * d->Hg * d->Wg * 4, bq = (int64_t)d->B * d->Hg * d->Wg * d->q.ldx * 2;
    return bp < (int64_t(1) << 31) && bq < (int64_t(1) << 31) && (d->q.ldx % 8) == 0 && (d->cq_off % 8) == 0;
}

static void head_fill(HeadK& k, const abc_wgrad_desc* d) {
    k.dl = (const float*)d->p.x; k.psc = d->p.scale; k.psh = d->p.shift; k.psl = d->p.slope;
    k.q = d->q.x; k.qsc = d->q.scale; k.qsh = d->q.shift; k.qsl = d->q.slope;
    k.partial = d->partial; k.rowsum = d->rowsum_partial; k.HW = d->Hg * d->Wg; k.hc = d->Ca; k.ldq = d->q.ldx; k.cq_off = d->cq_off;
    k.nchunks = d->B * k.HW / 128; k.nsplit = d->nsplit; k.mtiles = abc_cdiv(d->Ca, 32); k.Ca_pad = k.mtiles * 32;
    k.drop_p = d->q.drop_p; k.drop_seed = d->q.drop_seed; k.drop_salt = d->q.drop_salt;
    k.bytesP = (unsigned)((int64_t)d->B * d->Ca * k.HW * 4); k.bytesQ = (unsigned)((int64_t)d->B * k.HW * d->q.ldx * 2);
    k.cpad_blk = 0; k.keep = nullptr;
}

static int head_launch(const abc_wgrad_desc* d, hipStream_t st) {
    HeadK k;
    head_fill(k, d);
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)head_wgrad_kernel, 160 * 1024, &lds_ok)) return rc;
    hipLaunchKernelGGL(head_wgrad_kernel, dim3(d->nsplit, abc_cdiv(k.mtiles, 4)), dim3(512), HEAD_LDS, st, k);
    return abc_check_launch("head_wgrad");
}

// ---------------------------------------------------------------------------
// Weight gradient against a ONE-channel operand (the network's first convolution, unet.py:12 with in_channels = 1):
// dW[t][a] = sum_p dY[p][a] * x[p + d_t].  No matrix shape to speak of (N = 1): plain FMAs, bound by reading dY once.
// A thread owns 8 a-channels of one pixel column slot; the three image rows a row of dY needs sit in LDS.
struct C1K {
    const void* p;       // dY, NHWC [B][H][W][ldp]
    const float* x;      // image, [B][H][W] (one channel, f32)
    float* partial;      // [nsplit][ntaps][Ca]
    int B, H, W, ldp, cp_off, Ca, ntaps, nsplit, dy_min, dy_max, dx_min, dx_max;
    // DUAL: P = ca * g + cb * y_raw + cc (the BatchNorm-backward correction, abc_wgrad_desc.p_dual), optionally written out
    const void* p2; const float *ca, *cb, *cc; void* p_out; int ld_p2, cp2_off, ld_pout;
    int8_t ty[25], tx[25];
};

// NT = tap capacity (9: 3x3 stem of unet.py; 25: 5x5 stem of unet2.py:135), CPT = dY channels per thread (NT * CPT
// accumulators live in registers)
template <typename PT, int NT, int CPT, bool DUAL = false>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const C1K a) {
    constexpr int NR = NT > 9 ? 5 : 4;
    __shared__ float sx[NR][512 + 8];    // the image rows a row of dY needs, W <= 512 columns, 4-column halo either side
    __shared__ float red[4][NT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncg = a.Ca / CPT;               // channel groups (a power of two <= 16)
    const int cg = tid % ncg, slot = tid / ncg;
    const int nslot = 256 / ncg;              // pixels per step
    const int nrows = a.B * a.H;
    const int r0 = (int)((long long)blockIdx.x * nrows / a.nsplit), r1 = (int)((long long)(blockIdx.x + 1) * nrows / a.nsplit);
    const int nxr = a.dy_max - a.dy_min + 1;
    float acc[NT][CPT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[t][j] = 0.f;
    float ca[DUAL ? CPT : 1], cb[DUAL ? CPT : 1], cc[DUAL ? CPT : 1];
    if constexpr (DUAL) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) { ca[j] = a.ca[a.cp_off + cg * CPT + j]; cb[j] = a.cb[a.cp_off + cg * CPT + j]; cc[j] = a.cc[a.cp_off + cg * CPT + j]; }
    }
    for (int row = r0; row < r1; ++row) {
        const int b = row / a.H, y = row - b * a.H;
        __syncthreads();
        for (int i = tid; i < nxr * (a.W + 8); i += 256) {
            const int rr = i / (a.W + 8), xx = i - rr * (a.W + 8) - 4;
            const int yy = y + a.dy_min + rr;
            sx[rr][xx + 4] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? a.x[((size_t)b * a.H + yy) * a.W + xx] : 0.f;
        }
        __syncthreads();
        for (int x0 = slot; x0 < a.W; x0 += nslot) {
            float g[CPT];
            const PT* src = (const PT*)a.p + ((size_t)row * a.W + x0) * a.ldp + a.cp_off + cg * CPT;
            LoadVec<PT, CPT>::ld(src, g);
            if constexpr (DUAL) {
                float yv[CPT];
                LoadVec<PT, CPT>::ld((const PT*)a.p2 + ((size_t)row * a.W + x0) * a.ld_p2 + a.cp2_off + cg * CPT, yv);
#pragma unroll
                for (int j = 0; j < CPT; ++j) g[j] = fmaf(ca[j], g[j], fmaf(cb[j], yv[j], cc[j]));
                if (a.p_out != nullptr) {
                    PT* dst = (PT*)a.p_out + ((size_t)row * a.W + x0) * a.ld_pout + cg * CPT;
                    if constexpr (sizeof(PT) == 2 && CPT == 8) *(bf16x8*)dst = pack_frag<bf16>(g);
                    else if constexpr (sizeof(PT) == 2 && CPT == 4) { bf16x4 o; for (int j = 0; j < 4; ++j) o[j] = (bf16)g[j]; *(bf16x4*)dst = o; }
                    else { for (int j = 0; j < CPT; ++j) dst[j] = (PT)g[j]; }
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < a.ntaps) {
                    const float xv = sx[a.ty[t]][x0 + 4 + a.tx[t]];
#pragma unroll
                    for (int j = 0; j < CPT; ++j) acc[t][j] = fmaf(g[j], xv, acc[t][j]);
                }
            }
        }
    }
    // fold the pixel slots: lanes with equal cg inside the wave (ncg divides 64), then the 4 waves through LDS
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            float v = acc[t][j];
            for (int m = ncg; m < 64; m <<= 1) v += __shfl_xor(v, m);
            acc[t][j] = v;
        }
    __syncthreads();
    if (lane < ncg) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < CPT; ++j) red[wave][t * 64 + lane * CPT + j] = acc[t][j];
    }
    __syncthreads();
    for (int i = tid; i < a.ntaps * a.Ca; i += 256) {
        const int t = i / a.Ca, c = i - t * a.Ca;
        a.partial[((size_t)blockIdx.x * a.ntaps + t) * a.Ca + c] = red[0][t * 64 + c] + red[1][t * 64 + c] + red[2][t * 64 + c] + red[3][t * 64 + c];
    }
}

// Four pixels of a row per thread (round 4; the forward twin is stem_conv4_kernel in stem.hip).  The form above stages the image rows
// of ONE row of dY between two barriers and reads one LDS value per CPT FMAs: 172 us for unet2's 25-tap stem (b16 at 384 x 384)
// where the FMAs need ~35.  Here a workgroup stages the image rows of up to 12 rows of dY at once (16-byte loads, four in flight),
// a thread owns CPT channels of FOUR neighbouring pixels, a kernel row's taps read one 8-pixel window kept as register pairs, and an
// FMA is half of a v_pk_fma_f32 over a channel pair of dY with the pixel value broadcast by op_sel (common.hpp).
// KW x KW taps in row-major order (checked on the host), W a multiple of 4.
template <typename PT, int KW, int CPT, bool DUAL>
__global__ __launch_bounds__(256) void wgrad_c1q_kernel(const C1K a) {
    constexpr int NT = KW * KW, R = KW / 2, SROWS = 12, NR = SROWS + 2 * R, NP = CPT / 2;
    constexpr int RC = 16 * 320;                 // floats of the cross-row reduction buffer (it aliases the image rows)
    static_assert(NR * 520 >= RC, "the reduction buffer fits in the image rows");
    __shared__ __attribute__((aligned(16))) float sx[NR][512 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncg = a.Ca / CPT;               // channel groups (a power of two <= 16)
    const int cg = tid % ncg, slot = tid / ncg;
    const int nslot = 256 / ncg;
    const int nrows = a.B * a.H;
    const int r0 = (int)((long long)blockIdx.x * nrows / a.nsplit), r1 = (int)((long long)(blockIdx.x + 1) * nrows / a.nsplit);
    const int NQ = a.W >> 2;
    f32pair acc[NT][NP];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[t][j] = (f32pair){0.f, 0.f};
    float ca[DUAL ? CPT : 1], cb[DUAL ? CPT : 1], cc[DUAL ? CPT : 1];
    if constexpr (DUAL) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) { ca[j] = a.ca[a.cp_off + cg * CPT + j]; cb[j] = a.cb[a.cp_off + cg * CPT + j]; cc[j] = a.cc[a.cp_off + cg * CPT + j]; }
    }
    for (int rc = r0; rc < r1;) {
        const int b = rc / a.H, y0 = rc - b * a.H;
        const int n = min(min(SROWS, r1 - rc), a.H - y0);      // rows of this pass: one image
        const int nload = n + 2 * R, nq = nload * NQ;
        __syncthreads();
        for (int i0 = 0; i0 < nq; i0 += 1024) {
            f32x4 tq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + tid + 256 * u;
                const int rr = i / NQ, q = i - rr * NQ, yy = y0 - R + rr;
                tq[u] = (i < nq && yy >= 0 && yy < a.H) ? *(const f32x4*)(a.x + ((size_t)b * a.H + yy) * a.W + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + tid + 256 * u;
                const int rr = i / NQ, q = i - rr * NQ;
                if (i < nq) *(f32x4*)&sx[rr][4 + 4 * q] = tq[u];
            }
        }
        for (int i = tid; i < nload * 8; i += 256) sx[i >> 3][(i & 7) < 4 ? (i & 7) : a.W + (i & 7)] = 0.f;
        __syncthreads();
        const int nitems = n * NQ;
        // the next item's dY (and y_raw) quads are in flight under this item's FMAs
        static_assert(sizeof(PT) == 2, "bf16 dY");
        typedef typename std::conditional<CPT == 8, bf16x8, bf16x4>::type raw_t;
        raw_t rg[4], ry[DUAL ? 4 : 1];
        auto issue = [&](int it) {
            const int rl = it / NQ, x0 = (it - rl * NQ) << 2;
            const size_t pix = (size_t)(rc + rl) * a.W + x0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                rg[p] = *(const raw_t*)((const PT*)a.p + (pix + p) * a.ldp + a.cp_off + cg * CPT);
                if constexpr (DUAL) ry[p] = *(const raw_t*)((const PT*)a.p2 + (pix + p) * a.ld_p2 + a.cp2_off + cg * CPT);
            }
        };
        if (slot < nitems) issue(slot);
        for (int it = slot; it < nitems; it += nslot) {
            const int rl = it / NQ, x0 = (it - rl * NQ) << 2;
            const size_t pix = (size_t)(rc + rl) * a.W + x0;
            f32pair g[4][NP];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float gv[CPT];
#pragma unroll
                for (int j = 0; j < CPT; ++j) gv[j] = (float)rg[p][j];
                if constexpr (DUAL) {
#pragma unroll
                    for (int j = 0; j < CPT; ++j) gv[j] = fmaf(ca[j], gv[j], fmaf(cb[j], (float)ry[p][j], cc[j]));
                    if (a.p_out != nullptr) {
                        PT* dst = (PT*)a.p_out + (pix + p) * a.ld_pout + cg * CPT;
                        if constexpr (sizeof(PT) == 2 && CPT == 8) *(bf16x8*)dst = pack_frag<bf16>(gv);
                        else if constexpr (sizeof(PT) == 2 && CPT == 4) { bf16x4 o; for (int j = 0; j < 4; ++j) o[j] = (bf16)gv[j]; *(bf16x4*)dst = o; }
                        else { for (int j = 0; j < CPT; ++j) dst[j] = (PT)gv[j]; }
                    }
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) g[p][j] = (f32pair){gv[2 * j], gv[2 * j + 1]};
            }
            if (it + nslot < nitems) issue(it + nslot);
#pragma unroll
            for (int dy = 0; dy < KW; ++dy) {
                const float* rp = &sx[rl + dy][x0 + 2];     // pixels x0 - 2 .. x0 + 5 of image row (dY row + dy - R)
                const f32x4 mid = *(const f32x4*)(rp + 2);
                const f32pair win[4] = {*(const f32pair*)rp, (f32pair){mid[0], mid[1]}, (f32pair){mid[2], mid[3]}, *(const f32pair*)(rp + 6)};
#pragma unroll
                for (int dx = 0; dx < KW; ++dx)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            const int e = p + dx + 2 - R;     // (compile-time after unrolling)
                            if (e & 1) pk_fma_hi(acc[dy * KW + dx][j], win[e >> 1], g[p][j]); else pk_fma_lo(acc[dy * KW + dx][j], win[e >> 1], g[p][j]);
                        }
            }
        }
        rc += n;
    }
    // fold the pixel slots: the lanes of a 16-lane row that share a channel group by DPP rotations, then the 16 rows of the workgroup
    // through LDS in row order, a chunk of taps at a time (the image rows are dead)
    __syncthreads();
    float* redf = &sx[0][0];
    const int tch = 320 / a.Ca;                           // taps per chunk
    for (int t0 = 0; t0 < NT; t0 += tch) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t >= t0 && t < t0 + tch) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    float v = acc[t][j >> 1][j & 1];
                    if (ncg < 16) v = row_sum16(v, ncg);
                    if ((lane & 15) < ncg) redf[(wave * 4 + (lane >> 4)) * 320 + (t - t0) * a.Ca + (lane & 15) * CPT + j] = v;
                }
            }
        }
        __syncthreads();
        const int nval = min(tch, NT - t0) * a.Ca;
        for (int i = tid; i < nval; i += 256) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += redf[r * 320 + i];
            const int t = i / a.Ca, c = i - t * a.Ca;
            a.partial[((size_t)blockIdx.x * a.ntaps + t0 + t) * a.Ca + c] = sum;
        }
        __syncthreads();
    }
}

// the four-pixel form takes: a full 3 x 3 / 5 x 5 square in row-major tap order, whole pixel quads, bf16 dY
static bool c1_quad(const abc_wgrad_desc* d) {
    const int kw = d->ntaps == 9 ? 3 : (d->ntaps == 25 ? 5 : 0);
    bool square = kw != 0 && (d->Wg % 4) == 0 && d->dtype_p == ABC_BF16 && d->Ca <= 32 && d->Ca % (kw == 5 ? 4 : 8) == 0;
    for (int t = 0; square && t < d->ntaps; ++t) square = d->tap_dy[t] == t / kw - kw / 2 && d->tap_dx[t] == t % kw - kw / 2;
    return square;
}

static bool c1_ok(const abc_wgrad_desc* d) {
    if (abc_knob("ABC_WGRAD_NOC1")) return false;
    if (d->Cb != 1 || d->cq_off != 0 || d->q.ldx != 1 || d->dtype_q != ABC_F32 || d->q.scale || d->q.pool || d->q.planar || d->q.drop_p > 0.f) return false;
    // (a transform on P only as the BatchNorm-backward correction of abc_wgrad_desc.p_dual: bf16)
    // (25 taps on the four-pixel form only: the scalar form measured 332 us fused against 157 + 113 us with the separate apply pass)
    if (d->p.scale && !(d->p_dual && d->dtype_p == ABC_BF16 && (d->ntaps <= 9 || c1_quad(d)) && d->p2 != nullptr && (d->ld_p2 % 8) == 0 && (d->cp2_off % 8) == 0)) return false;
    if (d->p.pool || d->p.planar || d->p.drop_p > 0.f || d->stride != 1 || d->ntaps > 25) return false;
    if (d->Ca % 8 || d->Ca > 64 || (d->Ca & (d->Ca - 1)) || d->Wg > 512 || (d->p.ldx % 8) || (d->cp_off % 8)) return false;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    return dymax - dymin <= (d->ntaps > 9 ? 4 : 3) && dxmin >= -4 && dxmax <= 4 && d->Hq == d->Hg && d->Wq == d->Wg;
}

static int c1_launch(const abc_wgrad_desc* d, hipStream_t st) {
    C1K k;
    k.p = d->p.x; k.x = (const float*)d->q.x; k.partial = d->partial;
    k.B = d->B; k.H = d->Hg; k.W = d->Wg; k.ldp = d->p.ldx; k.cp_off = d->cp_off; k.Ca = d->Ca; k.ntaps = d->ntaps; k.nsplit = d->nsplit;
    int dymin = 127, dymax = -127;
    for (int t = 0; t < d->ntaps; ++t) { dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax; }
    k.dy_min = dymin; k.dy_max = dymax; k.dx_min = 0; k.dx_max = 0;
    for (int t = 0; t < d->ntaps; ++t) { k.ty[t] = (int8_t)(d->tap_dy[t] - dymin); k.tx[t] = (int8_t)d->tap_dx[t]; }
    k.p2 = d->p2; k.ld_p2 = d->ld_p2; k.cp2_off = d->cp2_off; k.p_out = d->p_out; k.ld_pout = d->ld_pout;
    k.ca = d->p.scale; k.cc = d->p.shift; k.cb = d->p.slope;     // (abc_act_src of a deferred BatchNorm backward: scale = ca, shift = cc, slope = cb)
    // the four-pixel form
    {
        const int kw = d->ntaps == 9 ? 3 : 5;
        if (c1_quad(d)) {
            const bool dual = d->p_dual && d->p.scale;
            if (kw == 5) {
                if (dual) hipLaunchKernelGGL((wgrad_c1q_kernel<bf16, 5, 4, true>), dim3(d->nsplit), dim3(256), 0, st, k);
                else hipLaunchKernelGGL((wgrad_c1q_kernel<bf16, 5, 4, false>), dim3(d->nsplit), dim3(256), 0, st, k);
            } else if (dual) hipLaunchKernelGGL((wgrad_c1q_kernel<bf16, 3, 8, true>), dim3(d->nsplit), dim3(256), 0, st, k);
            else hipLaunchKernelGGL((wgrad_c1q_kernel<bf16, 3, 8, false>), dim3(d->nsplit), dim3(256), 0, st, k);
            return abc_check_launch("wgrad_c1q");
        }
    }
    if (d->p_dual && d->p.scale) {
        if (d->ntaps > 9) hipLaunchKernelGGL((wgrad_c1_kernel<bf16, 25, 4, true>), dim3(d->nsplit), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((wgrad_c1_kernel<bf16, 9, 8, true>), dim3(d->nsplit), dim3(256), 0, st, k);
        return abc_check_launch("wgrad_c1");
    }
    if (d->ntaps > 9) {
        if (d->dtype_p == ABC_BF16) hipLaunchKernelGGL((wgrad_c1_kernel<bf16, 25, 4>), dim3(d->nsplit), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((wgrad_c1_kernel<float, 25, 4>), dim3(d->nsplit), dim3(256), 0, st, k);
    } else if (d->dtype_p == ABC_BF16) hipLaunchKernelGGL((wgrad_c1_kernel<bf16, 9, 8>), dim3(d->nsplit), dim3(256), 0, st, k);
    else hipLaunchKernelGGL((wgrad_c1_kernel<float, 9, 8>), dim3(d->nsplit), dim3(256), 0, st, k);
    return abc_check_launch("wgrad_c1");
}

// Few outputs, many slabs (the one-channel layer: 288 sums over 1024 slabs): one wave per output element, lanes stride
// the slabs, fixed shuffle tree -> still bitwise reproducible.
__global__ __launch_bounds__(64) void wgrad_reduce_wave_kernel(const abc_wgrad_reduce_desc d) {
    const int idx = blockIdx.x;
    const int bi = idx % d.Cb, ai = (idx / d.Cb) % d.Ca, t = idx / (d.Cb * d.Ca);
    const size_t slab = (size_t)d.Ca_pad * d.Cb_pad;
    const float* p = d.partial + (size_t)t * slab + (size_t)ai * d.Cb_pad + bi;
    const size_t step = (size_t)d.ntaps * slab;
    float s = 0.f;
    for (int k = threadIdx.x; k < d.nsplit; k += 64) s += p[(size_t)k * step];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (threadIdx.x == 0) {
        float* o = d.dw + ((size_t)ai * d.Cb + bi) * d.ntaps + t;
        *o = d.accumulate ? (*o + s) : s;
    }
}

// Several small reductions (the heads' 8 weight gradients and 8 bias row sums) in ONE launch: blockIdx.y = item; an item
// whose output is small against its slab count goes wave-per-output (4 outputs per workgroup), the others thread-per-
// output; summation orders are those of the single-item kernels, so results are bit-identical to them.
constexpr int MAX_RB = 16;
struct ReduceBatch { abc_wgrad_reduce_desc d[MAX_RB]; };
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch bt) {
    const abc_wgrad_reduce_desc& d = bt.d[blockIdx.y];
    const int64_t n = (int64_t)d.ntaps * d.Ca * d.Cb;
    const size_t slab = (size_t)d.Ca_pad * d.Cb_pad;
    const size_t step = (size_t)d.ntaps * slab;
    if (n <= 4096 && d.nsplit >= 128) {
        const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (idx >= n) return;
        const int bi = (int)(idx % d.Cb), ai = (int)((idx / d.Cb) % d.Ca), t = (int)(idx / ((int64_t)d.Cb * d.Ca));
        const float* p = d.partial + (size_t)t * slab + (size_t)ai * d.Cb_pad + bi;
        float s = 0.f;
        for (int k = lane; k < d.nsplit; k += 64) s += p[(size_t)k * step];
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) {
            float* o = d.dw + ((size_t)ai * d.Cb + bi) * d.ntaps + t;
            *o = d.accumulate ? (*o + s) : s;
        }
        return;
    }
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int bi = (int)(idx % d.Cb);
    const int ai = (int)((idx / d.Cb) % d.Ca);
    const int t = (int)(idx / ((int64_t)d.Cb * d.Ca));
    const float* p = d.partial + (size_t)t * slab + (size_t)ai * d.Cb_pad + bi;
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

struct WGeom {
    int AT, BT, dy_min, dx_min, HH, HW, PSWP, PSWQ, sP_bytes, sQ_bytes, coef_off, cstrP, cstrQ, lds, tgw, ngroups, nta, ntb, npatch,
        tiles_x, tiles_y, fast_p, fast_q, nbuf, PM, ts, nw;
};

static int psw_for(int cw, int csz) {
    // bf16: the 64-byte column blocks of 4 consecutive pixels must fall on distinct quarters of the 256-byte bank row
    if (csz == 2) return cw == 32 ? 64 : (cw == 64 ? 192 : 320);
    return cw * 4;
}

static bool fast_ok(const abc_act_src& s, int csz_c, int cvalid_min, int64_t bytes) {
    const int nv = 16 / csz_c;
    return !s.pool && !s.planar && s.drop_p <= 0.f && (cvalid_min % nv) == 0 && bytes < (int64_t(1) << 31);
}

static int wgeom_pm(const abc_wgrad_desc* d, WGeom* g, int pm) {
    const int csz = d->dtype_c == ABC_BF16 ? 2 : 4;
    const int cwp = g->AT * 32, cwq = g->BT * 32;
    g->PM = pm;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    g->dy_min = dymin; g->dx_min = dxmin;
    g->HH = (8 * pm - 1) * d->stride + (dymax - dymin) + 1;
    g->HW = 15 * d->stride + (dxmax - dxmin) + 1;
    g->PSWP = psw_for(cwp, csz); g->PSWQ = psw_for(cwq, csz);
    g->sP_bytes = abc_roundup(128 * pm * g->PSWP, 256);
    g->sQ_bytes = abc_roundup(g->HH * g->HW * g->PSWQ, 256);
    g->cstrP = cwp; g->cstrQ = cwq;
    const int coef_bytes = 3 * (cwp + cwq) * 4 + 256;
    g->nbuf = (2 * (g->sP_bytes + g->sQ_bytes) + coef_bytes <= 160 * 1024) ? 2 : 1;
    g->coef_off = g->nbuf * (g->sP_bytes + g->sQ_bytes);
    g->lds = g->coef_off + coef_bytes;
    if (g->lds < 8 * 16 * 64 * 4) g->lds = 8 * 16 * 64 * 4;
    if (g->lds > 160 * 1024) return abc_fail(ABC_EUNSUPPORTED, "wgrad: LDS tile too large");
    g->nta = abc_cdiv(d->Ca, cwp); g->ntb = abc_cdiv(d->Cb, cwq);
    g->tiles_x = abc_cdiv(d->Wg, 16); g->tiles_y = abc_cdiv(d->Hg, 8 * pm);
    g->npatch = g->tiles_x * g->tiles_y * d->B;
    // prefetch fast path: plain NHWC source whose channel count in the LAST tile is a whole number of 16-byte segments
    const int lastP = d->Ca - (g->nta - 1) * cwp, lastQ = d->Cb - (g->ntb - 1) * cwq;
    const int segq = cwq / (16 / csz);
    const int64_t bytesP = (int64_t)d->B * d->p.Hx * d->p.Wx * d->p.ldx * (d->dtype_p == ABC_BF16 ? 2 : 4);
    const int64_t bytesQ = (int64_t)d->B * d->q.Hx * d->q.Wx * d->q.ldx * (d->dtype_q == ABC_BF16 ? 2 : 4);
    g->fast_p = (fast_ok(d->p, csz, lastP, bytesP) && g->HH * g->HW * g->HW < 65536) ? 1 : 0;
    g->fast_q = (fast_ok(d->q, csz, lastQ, bytesQ) && abc_cdiv(g->HH * g->HW * segq, WTHR_HOST) <= ((pm > 1 || (d->stride == 2 && csz == 2)) ? 5 : (csz == 2 ? 4 : 6))) ? 1 : 0;
    g->ngroups = abc_cdiv(d->ntaps, (g->fast_p && g->fast_q) ? MAXT_FAST : MAXT_SLOW);
    g->tgw = abc_cdiv(d->ntaps, g->ngroups);
    return ABC_OK;
}

// BN-backward correction on load of P: prefetch path for both operands, bf16 everywhere, full 3x3 square, stride 1
static bool dual_ok(const abc_wgrad_desc* d, const WGeom& g) {
    if (d->dtype_p != ABC_BF16 || d->dtype_q != ABC_BF16 || d->dtype_c != ABC_BF16) return false;
    if (d->p.scale == nullptr || d->p2 == nullptr || (d->ld_p2 % 8) || (d->cp2_off % 8) || (d->p_out && (d->ld_pout % 8))) return false;
    if ((int64_t)d->B * d->Hg * d->Wg * d->ld_p2 * 2 >= (int64_t(1) << 31)) return false;
    if (d->p_out && (int64_t)d->B * d->Hg * d->Wg * d->ld_pout * 2 >= (int64_t(1) << 31)) return false;   // (32-bit store offsets)
    // (the correction is applied where P is committed to LDS: independent of the taps, so the tap-split form of the 5x5
    //  layers takes it as well as the static 3x3 K-step)
    if (g.ts) return g.fast_p && g.fast_q && d->stride == 1;
    if (!(g.fast_p && g.fast_q) || d->stride != 1 || d->ntaps != 9 || g.ngroups != 1 || g.HW != 18) return false;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] - g.dy_min != t / 3 || d->tap_dx[t] - g.dx_min != t % 3) return false;
    return true;
}

static int wgeom(const abc_wgrad_desc* d, WGeom* g) {

    if (d->ntaps < 1 || d->ntaps > ABC_MAX_TAPS) return abc_fail(ABC_EINVAL, "wgrad: ntaps");
    if (d->stride != 1 && d->stride != 2) return abc_fail(ABC_EUNSUPPORTED, "wgrad: stride");
    const int csz = d->dtype_c == ABC_BF16 ? 2 : 4;
    const int ta = abc_cdiv(d->Ca, 32), tb = abc_cdiv(d->Cb, 32);
    // 32x32 tile pairs per workgroup (one pair per wave, the remaining waves split the patch rows).  Ragged channel
    // tails are fine (zero-filled): wide tiles are what keeps the operands from being re-staged per pair.
    static const bool no42 = abc_knob("ABC_WGRAD_NO42") != nullptr;  // (experiment switch)
    // 4 x 2 pairs where a workgroup has patches to amortise its 295 KB slab over: with one round of <= 256 workgroups (the engine's
    // split rule: nsplit = min(patches / 2, 256 / tiles)) the 48 x 48 and smaller maps leave 2-5 patches per workgroup, the kernel is
    // prologue + slab store, and 2 x 2 pairs halve the slabs written here and re-read by the reduction (per layer: launch +0..5 us,
    // reduction -5..9 us; at 96 x 96 -- 9 patches per workgroup -- the wide tile wins by 10 us per launch)
    const int patches = d->B * abc_cdiv(d->Hg, 8) * abc_cdiv(d->Wg, 16);
    const int tiles42 = abc_cdiv(d->Ca, 128) * abc_cdiv(d->Cb, 64);
    const int ns42 = std::max(1, std::min(std::max(1, patches / 2), 256 / std::max(1, tiles42)));
    const bool few = patches < 6 * ns42 && !abc_knob("ABC_WGRAD_42_ALWAYS");
    if (!no42 && !few && d->stride == 1 && csz == 2 && ta >= 3 && tb >= 2) { g->AT = 4; g->BT = 2; }
    else if (d->stride == 1 && ta >= 2 && tb >= 2) { g->AT = 2; g->BT = 2; }
    else if (d->stride == 1 && csz == 2 && ta == 1 && tb >= 4) { g->AT = 1; g->BT = 4; }
    // ConvTranspose (stride 2): two P tiles share one staging of Q's 17 x 33-pixel halo (a 32-channel halo is what the prefetch path holds;
    // 64 x 64 pairs fall to the general loader: 150 us against 41).  41 -> 32 us per launch
    else if (d->stride == 2 && csz == 2 && d->dtype_p == ABC_BF16 && d->dtype_q == ABC_BF16 && ta >= 2 && !abc_knob("ABC_WGRAD_S2_11")) { g->AT = 2; g->BT = 1; }
    else { g->AT = 1; g->BT = 1; }
    // narrow layers (one tile pair, 8-way row split) take 32-row patches when both operands can be prefetched:
    // 4 K-steps per wave between barriers instead of 1
    g->ts = 0; g->nw = 8;
    // (experiment, ABC_WGRAD_NW4=1) wide bf16 3x3 layers as 4-wave workgroups on 2 x 2 tile pairs, two per CU
    static const bool nw4 = abc_knob("ABC_WGRAD_NW4") != nullptr;
    if (nw4 && g->AT == 4 && g->BT == 2 && d->ntaps == 9 && d->dtype_p == ABC_BF16 && d->dtype_q == ABC_BF16) {
        g->AT = 2; g->BT = 2;
        int rc = wgeom_pm(d, g, 1);
        const int segq = 64 / 8;
        if (rc == ABC_OK && g->fast_p && fast_ok(d->q, csz, d->Cb - (g->ntb - 1) * 64, (int64_t)d->B * d->q.Hx * d->q.Wx * d->q.ldx * 2) &&
            abc_cdiv(g->HH * g->HW * segq, 256) <= 6) {
            g->fast_q = 1; g->nw = 4; g->nbuf = 1;
            g->coef_off = g->sP_bytes + g->sQ_bytes;
            g->lds = g->coef_off + 3 * (64 + 64) * 4 + 256;
            if (g->lds < 4 * 16 * 64 * 4) g->lds = 4 * 16 * 64 * 4;
            g->ngroups = 1; g->tgw = d->ntaps;
            return ABC_OK;
        }
        g->AT = 4; g->BT = 2;
    }
    if (g->AT == 1 && g->BT == 1 && d->stride == 1 && csz == 2 && d->Hg % 32 == 0) {
        int rc = wgeom_pm(d, g, 4);
        if (rc == ABC_OK && g->fast_p && g->fast_q) return ABC_OK;
    }
    // more than 9 taps on one tile pair (5x5): split the taps over the waves, one pass over the operands
    static const bool nots = abc_knob("ABC_WGRAD_NOTS") != nullptr;  // (experiment switch)
    if (!nots && g->AT == 1 && g->BT == 1 && d->stride == 1 && csz == 2 && d->ntaps > MAXT_FAST && d->ntaps <= 32 &&
        d->dtype_p == ABC_BF16 && d->dtype_q == ABC_BF16) {
        for (int pm = (d->Hg % 16 == 0) ? 2 : 1; pm >= 1; --pm) {
            int rc = wgeom_pm(d, g, pm);
            if (rc == ABC_OK && g->fast_p && g->fast_q) { g->ts = 1; g->ngroups = 1; g->tgw = d->ntaps; return ABC_OK; }
        }
    }
    return wgeom_pm(d, g, 1);
}

template <typename PT, typename QT, typename CT, int AT, int BT, int STRIDE, bool FAST, int PM, bool K3, bool DUAL = false, bool TS = false, int NW = 8>
static int wlaunch3(const WgK& k, const WGeom& g, int nsplit, hipStream_t st) {
    auto fn = wgrad_kernel<PT, QT, CT, AT, BT, STRIDE, FAST, PM, K3, DUAL, TS, NW>;
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)fn, 160 * 1024, &lds_ok)) return rc;
    const int lds = k.qtab_off ? k.qtab_off + (PM > 1 ? 5 : 4) * WTHR_HOST * 4 : g.lds;     // (+ Q's segment table)
    hipLaunchKernelGGL(fn, dim3(g.nta * g.ntb * nsplit, g.ngroups), dim3(NW * 64), lds, st, k);
    return abc_check_launch("wgrad");
}

template <typename PT, typename QT, typename CT, int AT, int BT, int STRIDE, bool FAST, int PM = 1>
static int wlaunch2(const WgK& k, const WGeom& g, int nsplit, hipStream_t st) {
    if constexpr (FAST && sizeof(CT) == 2) {
        if constexpr (sizeof(PT) == 2 && sizeof(QT) == 2 && STRIDE == 1) {
            if (k.k3 && k.p2 != nullptr) return wlaunch3<PT, QT, CT, AT, BT, STRIDE, FAST, PM, true, true>(k, g, nsplit, st);
        }
        if (k.k3) return wlaunch3<PT, QT, CT, AT, BT, STRIDE, FAST, PM, true>(k, g, nsplit, st);
    }
    if (k.p2 != nullptr) return abc_fail(ABC_EUNSUPPORTED, "wgrad: p_dual needs the prefetch path with 3x3 taps in bf16");
    return wlaunch3<PT, QT, CT, AT, BT, STRIDE, FAST, PM, false>(k, g, nsplit, st);
}

template <typename PT, typename QT, typename CT, int AT, int BT, int STRIDE>
static int wlaunch(const WgK& k, const WGeom& g, int nsplit, hipStream_t st) {
    return (g.fast_p && g.fast_q) ? wlaunch2<PT, QT, CT, AT, BT, STRIDE, true>(k, g, nsplit, st)
                                  : wlaunch2<PT, QT, CT, AT, BT, STRIDE, false>(k, g, nsplit, st);
}

template <typename PT, typename QT, typename CT>
static int wdispatch(const WgK& k, const WGeom& g, int stride, int nsplit, hipStream_t st) {
    if constexpr (sizeof(CT) == 2) {
        if (g.AT == 4) return wlaunch<PT, QT, CT, 4, 2, 1>(k, g, nsplit, st);
    }
    if constexpr (sizeof(CT) == 2 && sizeof(PT) == 2 && sizeof(QT) == 2) {
        if (g.AT == 2 && g.nw == 4) {
            if (k.k3 && k.p2 != nullptr) return wlaunch3<PT, QT, CT, 2, 2, 1, true, 1, true, true, false, 4>(k, g, nsplit, st);
            if (k.k3) return wlaunch3<PT, QT, CT, 2, 2, 1, true, 1, true, false, false, 4>(k, g, nsplit, st);
            return wlaunch3<PT, QT, CT, 2, 2, 1, true, 1, false, false, false, 4>(k, g, nsplit, st);
        }
    }
    if constexpr (sizeof(CT) == 2 && sizeof(PT) == 2 && sizeof(QT) == 2) {
        if (g.AT == 2 && g.BT == 1 && stride == 2) return wlaunch<PT, QT, CT, 2, 1, 2>(k, g, nsplit, st);
    }
    if (g.AT == 2) return wlaunch<PT, QT, CT, 2, 2, 1>(k, g, nsplit, st);
    if constexpr (sizeof(CT) == 2) {
        if (g.BT == 4) return wlaunch<PT, QT, CT, 1, 4, 1>(k, g, nsplit, st);
        if constexpr (sizeof(PT) == 2 && sizeof(QT) == 2) {
            if (g.ts && k.p2 != nullptr) {
                if (g.PM == 2) return wlaunch3<PT, QT, CT, 1, 1, 1, true, 2, false, true, true>(k, g, nsplit, st);
                return wlaunch3<PT, QT, CT, 1, 1, 1, true, 1, false, true, true>(k, g, nsplit, st);
            }
            if (g.ts && g.PM == 2) return wlaunch3<PT, QT, CT, 1, 1, 1, true, 2, false, false, true>(k, g, nsplit, st);
            if (g.ts) return wlaunch3<PT, QT, CT, 1, 1, 1, true, 1, false, false, true>(k, g, nsplit, st);
        }
        if (g.PM == 4) return wlaunch2<PT, QT, CT, 1, 1, 1, true, 4>(k, g, nsplit, st);
    }
    if (stride == 1) return wlaunch<PT, QT, CT, 1, 1, 1>(k, g, nsplit, st);
    return wlaunch<PT, QT, CT, 1, 1, 2>(k, g, nsplit, st);
}

}  // namespace

extern "C" int abc_wgrad_rowsum_ok(const abc_wgrad_desc* d) { return head_ok(d) ? 1 : 0; }

// The heads' 1x1 weight gradients (unet.py:70 under autograd) of all heads in one launch: descs[0..n) as abc_wgrad takes
// them one by one (each with its OWN partial / rowsum_partial slabs), n <= 8.  ABC_EUNSUPPORTED unless every one of them
// is served by the heads kernel.
extern "C" int abc_wgrad_heads_batch(const abc_wgrad_desc* descs, int32_t n, abc_stream_t stream) {
    if (n < 1 || n > 8) return abc_fail(ABC_EINVAL, "wgrad_heads_batch: 1..8 heads");
    HeadWgBatch bt;
    int gx = 0, gy = 0;
    for (int i = 0; i < n; ++i) {
        const abc_wgrad_desc* d = descs + i;
        if (d->nsplit < 1 || d->ntaps != 1 || !head_ok(d)) return abc_fail(ABC_EUNSUPPORTED, "wgrad_heads_batch: not a heads' 1x1 weight gradient");
        if (d->p.Hx != d->Hg || d->p.Wx != d->Wg || d->q.Hx != d->Hg || d->q.Wx != d->Wg) return abc_fail(ABC_EINVAL, "wgrad: dims mismatch");
        head_fill(bt.k[i], d);
        gx = bt.k[i].nsplit > gx ? bt.k[i].nsplit : gx;
        gy = abc_cdiv(bt.k[i].mtiles, 4) > gy ? abc_cdiv(bt.k[i].mtiles, 4) : gy;
    }
    for (int i = n; i < 8; ++i) bt.k[i] = bt.k[0];
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)head_wgrad_batch_kernel, 160 * 1024, &lds_ok)) return rc;
    hipLaunchKernelGGL(head_wgrad_batch_kernel, dim3(gx, gy, n), dim3(512), HEAD_LDS, (hipStream_t)stream, bt);
    return abc_check_launch("wgrad_heads_batch");
}

// K-splits of the blocked weight gradient (heads 5, 6, 7: the five small heads' gradients come out of the fused kernel
// itself): ONE round of ~256 workgroups (the kernel holds 150 KB of LDS) shared out over the heads by the cost of a
// 128-pixel chunk (the feature tile is staged and activated once per workgroup, the d(logits) rows on top)
static void hf_splits(int nchunk, int* nsplit) {
    double cost[HF_NH], tot = 0;
    int units[HF_NH];
    for (int i = 5; i < HF_NH; ++i) {
        units[i] = abc_cdiv(hf_tiles(i), 4);
        cost[i] = 4.0 + 1.5 * (double)hf_tiles(i) / units[i] / 4.0;
        tot += units[i] * cost[i];
    }
    for (int i = 0; i < HF_NH; ++i) {
        if (i < 5) { nsplit[i] = 0; continue; }
        int n = (int)(256.0 * cost[i] / tot);
        n = n < 1 ? 1 : n;
        nsplit[i] = n > nchunk ? nchunk : n;
    }
}

extern "C" int64_t abc_heads_fused_wgrad_floats(const abc_heads_fused_desc* d) {
    int ns[HF_NH];
    const int nchunk = d->B * d->h * d->w / 128;
    hf_splits(nchunk, ns);
    int64_t n = (int64_t)(nchunk + 16) * HF_SMALL_ROWS * 129;      // the fused kernel's partials of the small heads + their first reduction
    for (int i = 5; i < HF_NH; ++i) n += (int64_t)ns[i] * hf_tiles(i) * 32 * (128 + 1);
    return n;
}

// slabs -> conv2.weight.grad / conv2.bias.grad: packed rows back to channels (hf_row_of_chan), times the head's loss factor
// (abc_loss_finalize's chan_scale), fixed summation order.  Heads 5-7: the K-split slabs of the blocked kernel; heads 0-4: the
// per-chunk partial rows [chunk][21][128 weights | 1 bias] the fused kernel left.
struct HeadFusedRedK {
    const float* partial[HF_NH]; const float* rowsum[HF_NH];
    float* dw[HF_NH]; float* db[HF_NH];
    const float* chan_scale;
    const float* small2;
    int nsplit[HF_NH], chan_off[HF_NH];
};
// first stage for the small heads: [nchunk][21][129] -> 16 row slices [16][21][129] (grid = 21 x 16: enough loads in flight)
__global__ __launch_bounds__(192) void head_fused_small_reduce_kernel(const float* part, int nchunk, float* out) {
    const int row = blockIdx.x, sl = blockIdx.y, c2 = threadIdx.x;
    if (c2 >= 129) return;
    const float* p = part + (size_t)row * 129 + c2;
    const size_t step = (size_t)HF_SMALL_ROWS * 129;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = sl;
    for (; k + 48 < nchunk; k += 64) {
        s0 += p[(size_t)k * step]; s1 += p[(size_t)(k + 16) * step]; s2 += p[(size_t)(k + 32) * step]; s3 += p[(size_t)(k + 48) * step];
    }
    for (; k < nchunk; k += 16) s0 += p[(size_t)k * step];
    out[((size_t)sl * HF_SMALL_ROWS + row) * 129 + c2] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(128) void head_fused_reduce_kernel(const HeadFusedRedK a) {
    const int head = blockIdx.y, ch = blockIdx.x, ci = threadIdx.x;
    if (ch >= hf_ch(head)) return;
    const float cs = a.chan_scale[a.chan_off[head] + ch];
    if (head < 5) {
        // 16 row slices of the per-chunk partials were summed by head_fused_small_reduce_kernel: [16][21][129]
        const float* p = a.small2 + (size_t)(hf_small_row0(head) + ch) * 129;
        for (int c2 = ci; c2 < 129; c2 += 128) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += p[(size_t)k * HF_SMALL_ROWS * 129 + c2];
            if (c2 < 128) a.dw[head][ch * 128 + c2] = s * cs; else a.db[head][ch] = s * cs;
        }
        return;
    }
    const int cpad = hf_tiles(head) * 32, row = hf_row_of_chan(head, ch);
    const float* p = a.partial[head] + (size_t)row * 128 + ci;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 2 <= a.nsplit[head]; k += 2) { s0 += p[(size_t)k * cpad * 128]; s1 += p[(size_t)(k + 1) * cpad * 128]; }
    if (k < a.nsplit[head]) s0 += p[(size_t)k * cpad * 128];
    a.dw[head][ch * 128 + ci] = (s0 + s1) * cs;
    if (ci == 0) {
        float b = 0.f;
        for (int q = 0; q < a.nsplit[head]; ++q) b += a.rowsum[head][(size_t)q * cpad + row];
        a.db[head][ch] = b * cs;
    }
}

// conv2.weight.grad / conv2.bias.grad of all heads from the fused kernel's outputs (unet.py:70 under autograd):
// dW2[c][ci] = factor_c * sum_p dL[c][p] * act(feat[p][ci]); run after abc_loss_finalize (chan_scale)
extern "C" int abc_heads_fused_wgrad(const abc_heads_fused_desc* d, abc_stream_t stream) {
    const int HW = d->h * d->w, nchunk = d->B * HW / 128;
    if (HW % 128 || d->ld % 8 || (int64_t)d->B * HW * d->ld * 2 >= (int64_t(1) << 31))
        return abc_fail(ABC_EUNSUPPORTED, "heads_fused_wgrad: whole 128-pixel chunks, feature buffer below 2 GB");
    int ns[HF_NH];
    hf_splits(nchunk, ns);
    HeadWgBatch bt;
    HeadFusedRedK rk;
    float* ws = d->wgrad_work + (size_t)nchunk * HF_SMALL_ROWS * 129;
    rk.small2 = ws; ws += 16 * HF_SMALL_ROWS * 129;
    size_t row0 = 0;
    int gx = 0, gy = 0;
    for (int i = 0; i < HF_NH; ++i) {
        const int cpad = hf_tiles(i) * 32;
        rk.dw[i] = d->dw2[i]; rk.db[i] = d->db2[i]; rk.nsplit[i] = ns[i]; rk.chan_off[i] = d->chan_off[i];
        rk.partial[i] = nullptr; rk.rowsum[i] = nullptr;
        if (i >= 5) {
            HeadK& k = bt.k[i - 5];
            k.dl = (const float*)((const bf16*)d->dl + row0 * (size_t)nchunk * 128);
            k.psc = k.psh = k.psl = nullptr;
            k.q = d->feat; k.qsc = d->scale; k.qsh = d->shift; k.qsl = d->slope;
            k.partial = ws; ws += (size_t)ns[i] * cpad * 128;
            k.rowsum = ws; ws += (size_t)ns[i] * cpad;
            k.HW = HW; k.hc = hf_ch(i); k.ldq = d->ld; k.cq_off = 128 * i; k.nchunks = nchunk; k.nsplit = ns[i];
            k.mtiles = hf_tiles(i); k.Ca_pad = cpad;
            k.drop_p = d->drop_p; k.drop_seed = d->drop_seed; k.drop_salt = d->drop_salt;
            k.bytesP = (unsigned)((size_t)nchunk * cpad * 128 * 2); k.bytesQ = (unsigned)((int64_t)d->B * HW * d->ld * 2);
            k.cpad_blk = cpad;
            k.keep = d->keep_mask != nullptr ? (const uint8_t*)d->keep_mask + (size_t)(i - 5) * nchunk * 2048 : nullptr;
            if ((size_t)nchunk * cpad * 128 * 2 >= (size_t(1) << 31)) return abc_fail(ABC_EUNSUPPORTED, "heads_fused_wgrad: d(logits) block above 2 GB");
            gx = ns[i] > gx ? ns[i] : gx;
            gy = abc_cdiv(k.mtiles, 4) > gy ? abc_cdiv(k.mtiles, 4) : gy;
            rk.partial[i] = k.partial; rk.rowsum[i] = k.rowsum;
        }
        row0 += cpad;
    }
    for (int i = 3; i < 8; ++i) bt.k[i] = bt.k[0];
    bt.first[0] = 0;
    for (int i = 0; i < 8; ++i) bt.first[i + 1] = bt.first[i] + (i < 3 ? bt.k[i].nsplit * abc_cdiv(bt.k[i].mtiles, 4) : 0);
    (void)gx; (void)gy;
    rk.chan_scale = d->chan_scale;
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)head_wgrad_blocked_kernel, 160 * 1024, &lds_ok)) return rc;
    hipLaunchKernelGGL(head_wgrad_blocked_kernel, dim3(bt.first[8]), dim3(512), HEAD_LDS, (hipStream_t)stream, bt);
    if (int rc = abc_check_launch("heads_fused_wgrad")) return rc;
    hipLaunchKernelGGL(head_fused_small_reduce_kernel, dim3(HF_SMALL_ROWS, 16), dim3(192), 0, (hipStream_t)stream, (const float*)d->wgrad_work, nchunk, (float*)rk.small2);
    hipLaunchKernelGGL(head_fused_reduce_kernel, dim3(360, HF_NH), dim3(128), 0, (hipStream_t)stream, rk);
    return abc_check_launch("heads_fused_wgrad_reduce");
}

extern "C" int abc_wgrad_fuses_apply(const abc_wgrad_desc* d) {
    if (head_ok(d)) return 0;
    if (c1_ok(d)) return (d->p_dual && d->p.scale) ? 1 : 0;     // the one-channel kernel applies the correction on load too
    if (abc_wgrad_narrow_ok(d) || abc_wgrad_n32r2_ok(d)) return d->p_dual ? 1 : 0;
    WGeom g;
    if (wgeom(d, &g)) return 0;
    return dual_ok(d, g) ? 1 : 0;
}

extern "C" int abc_wgrad_pads(const abc_wgrad_desc* d, int32_t* ca_pad, int32_t* cb_pad) {
    if (head_ok(d)) { *ca_pad = abc_cdiv(d->Ca, 32) * 32; *cb_pad = 128; return ABC_OK; }
    if (c1_ok(d)) { *ca_pad = d->Ca; *cb_pad = 1; return ABC_OK; }
    if (abc_wgrad_narrow_ok(d)) { *ca_pad = 16; *cb_pad = 16; return ABC_OK; }
    if (abc_wgrad_n32r2_ok(d)) { *ca_pad = 32; *cb_pad = 32; return ABC_OK; }
    WGeom g;
    int rc = wgeom(d, &g);
    if (rc) return rc;
    *ca_pad = g.nta * g.AT * 32;
    *cb_pad = g.ntb * g.BT * 32;
    return ABC_OK;
}

extern "C" int abc_wgrad_tile(const abc_wgrad_desc* d, int32_t* at, int32_t* bt) {
    if (head_ok(d)) { *at = 0; *bt = 0; return ABC_OK; }  // (0, 0) = the head kernel
    if (c1_ok(d)) { *at = 0; *bt = 1; return ABC_OK; }    // (0, 1) = the one-channel kernel
    if (abc_wgrad_narrow_ok(d)) { *at = 0; *bt = 2; return ABC_OK; }   // (0, 2) = the 16-channel kernel (wgrad_narrow.hip)
    if (abc_wgrad_n32r2_ok(d)) { *at = 0; *bt = 3; return ABC_OK; }    // (0, 3) = the 5x5 32-channel kernel (wgrad_narrow.hip)
    WGeom g;
    int rc = wgeom(d, &g);
    if (rc) return rc;
    *at = g.AT; *bt = g.BT;
    return ABC_OK;
}

extern "C" int abc_wgrad_blocks(const abc_wgrad_desc* d) {
    if (head_ok(d)) return abc_cdiv(abc_cdiv(d->Ca, 32), 4);
    if (c1_ok(d)) return 1;
    if (abc_wgrad_narrow_ok(d) || abc_wgrad_n32r2_ok(d)) return 1;
    WGeom g;
    if (wgeom(d, &g)) return -1;
    // (4-wave workgroups sit two to a CU: half as many CU-fills per split, so that the caller's nsplit doubles)
    return g.nw == 4 ? (g.nta * g.ntb * g.ngroups + 1) / 2 : g.nta * g.ntb * g.ngroups;
}

extern "C" int abc_wgrad(const abc_wgrad_desc* d, abc_stream_t stream) {
    if (d->nsplit < 1) return abc_fail(ABC_EINVAL, "wgrad: nsplit");
    if (d->ntaps >= 1 && d->ntaps <= ABC_MAX_TAPS && head_ok(d)) {
        if (d->p.Hx != d->Hg || d->p.Wx != d->Wg || d->q.Hx != d->Hg || d->q.Wx != d->Wg) return abc_fail(ABC_EINVAL, "wgrad: dims mismatch");
        return head_launch(d, (hipStream_t)stream);
    }
    if (d->ntaps >= 1 && c1_ok(d)) {
        if (d->p.Hx != d->Hg || d->p.Wx != d->Wg || d->q.Hx != d->Hg || d->q.Wx != d->Wg) return abc_fail(ABC_EINVAL, "wgrad: dims mismatch");
        return c1_launch(d, (hipStream_t)stream);
    }
    if (abc_wgrad_narrow_ok(d)) return abc_wgrad_narrow_launch(d, stream);      // 16 x 16 channels, 3x3: wgrad_narrow.hip
    if (abc_wgrad_n32r2_ok(d)) return abc_wgrad_n32r2_launch(d, stream);        // 32 x 32 channels, 5x5: wgrad_narrow.hip
    WGeom g;
    int rc = wgeom(d, &g);
    if (rc) return rc;
    WgK k;
    auto cp = [](ActSrc& o, const abc_act_src& i) {
        o.x = i.x; o.scale = i.scale; o.shift = i.shift; o.slope = i.slope; o.Hx = i.Hx; o.Wx = i.Wx; o.ldx = i.ldx;
        o.pool = i.pool; o.drop_p = i.drop_p; o.drop_seed = i.drop_seed; o.planar = i.planar; o.ctot = i.ctot; o.drop_salt = i.drop_salt;
    };
    cp(k.p, d->p); cp(k.q, d->q);
    const int php = d->p.pool ? d->p.Hx / 2 : d->p.Hx, pwp = d->p.pool ? d->p.Wx / 2 : d->p.Wx;
    const int qhp = d->q.pool ? d->q.Hx / 2 : d->q.Hx, qwp = d->q.pool ? d->q.Wx / 2 : d->q.Wx;
    if (php != d->Hg || pwp != d->Wg || qhp != d->Hq || qwp != d->Wq) return abc_fail(ABC_EINVAL, "wgrad: dims mismatch");
    k.partial = d->partial; k.B = d->B; k.Hg = d->Hg; k.Wg = d->Wg; k.Hq = d->Hq; k.Wq = d->Wq;
    k.cp_off = d->cp_off; k.Ca = d->Ca; k.cq_off = d->cq_off; k.Cb = d->Cb;
    k.Ca_pad = g.nta * g.AT * 32; k.Cb_pad = g.ntb * g.BT * 32;
    k.ntaps = d->ntaps; k.tgw = g.tgw; k.nsplit = d->nsplit; k.npatch = g.npatch; k.tiles_x = g.tiles_x; k.tiles_y = g.tiles_y;
    k.dy_min = g.dy_min; k.dx_min = g.dx_min; k.HH = g.HH; k.HW = g.HW; k.PSWP = g.PSWP; k.PSWQ = g.PSWQ;
    k.sP_bytes = g.sP_bytes; k.sQ_bytes = g.sQ_bytes; k.coef_off = g.coef_off; k.cstrP = g.cstrP; k.cstrQ = g.cstrQ;
    k.nta = g.nta; k.ntb = g.ntb; k.fast_p = g.fast_p; k.fast_q = g.fast_q; k.nbuf = g.nbuf;
    k.magicQ = 65536 / g.HW + 1;
    if (g.npatch >= 65536 || g.tiles_x > 4096 || g.tiles_y > 4096) return abc_fail(ABC_EUNSUPPORTED, "wgrad: more than 65535 patches");
    k.mg_tx = (unsigned)((0x100000000ull + (unsigned)g.tiles_x - 1) / (unsigned)g.tiles_x);
    k.mg_ty = (unsigned)((0x100000000ull + (unsigned)g.tiles_y - 1) / (unsigned)g.tiles_y);
    k.regP = (d->Hg % (8 * g.PM) == 0 && d->Wg % 16 == 0 && d->p.Hx == d->Hg && d->p.Wx == d->Wg) ? 1 : 0;
    // Q's segment table: the static 3x3 step of the 8-wave bf16 prefetch path over whole patches of a same-size image whose offsets
    // fit the entry (relative offset < 16 MB, LDS image < 64 KB), when the table (segments per thread x 512 x 4 bytes) fits the LDS
    k.qtab_off = 0;
    if (k.k3 && d->stride == 1 && g.fast_p && g.fast_q && g.nw == 8 && !g.ts && g.AT * g.BT >= 4 && d->dtype_c == ABC_BF16 && d->dtype_q == ABC_BF16 &&
        d->Hg % (8 * g.PM) == 0 && d->Wg % 16 == 0 && d->Hq == d->Hg && d->Wq == d->Wg && d->q.Hx == d->Hq && d->q.Wx == d->Wq &&
        (int64_t)(g.HH * d->q.Wx + g.HW) * d->q.ldx * 2 < (int64_t(1) << 24) && g.sQ_bytes < 65536) {
        const int npf_q = g.PM > 1 ? 5 : 4;
        const int off = abc_roundup(g.lds, 16);
        if (off + npf_q * WTHR_HOST * 4 <= 160 * 1024) k.qtab_off = off;
    }
    k.p2 = nullptr; k.p_out = nullptr; k.ld_p2 = 0; k.cp2_off = 0; k.ld_pout = 0; k.bytesP2 = 0;
    if (d->p_dual) {
        if (!dual_ok(d, g)) return abc_fail(ABC_EUNSUPPORTED, "wgrad: p_dual is not served for this descriptor (abc_wgrad_fuses_apply)");
        k.p2 = d->p2; k.p_out = d->p_out; k.ld_p2 = d->ld_p2; k.cp2_off = d->cp2_off; k.ld_pout = d->ld_pout;
        k.bytesP2 = (unsigned)((int64_t)d->B * d->Hg * d->Wg * d->ld_p2 * 2);
    }
    k.k3 = (d->ntaps == 9 && g.ngroups == 1 && g.HW == 15 * d->stride + 3) ? 1 : 0;
    for (int t = 0; t < d->ntaps && k.k3; ++t)
        if (d->tap_dy[t] - g.dy_min != t / 3 || d->tap_dx[t] - g.dx_min != t % 3) k.k3 = 0;
    k.bytesP = (unsigned)((int64_t)d->B * d->p.Hx * d->p.Wx * d->p.ldx * (d->dtype_p == ABC_BF16 ? 2 : 4));
    k.bytesQ = (unsigned)((int64_t)d->B * d->q.Hx * d->q.Wx * d->q.ldx * (d->dtype_q == ABC_BF16 ? 2 : 4));
    { const char* e = abc_knob("ABC_WGRAD_DBG"); k.dbg = e ? atoi(e) : 0; }  // timing ablations only (results invalid)
    for (int t = 0; t < d->ntaps; ++t) { k.ty[t] = (int8_t)(d->tap_dy[t] - g.dy_min); k.tx[t] = (int8_t)(d->tap_dx[t] - g.dx_min); }
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype_c == ABC_F32) {
        if (d->dtype_p != ABC_F32 || d->dtype_q != ABC_F32) return abc_fail(ABC_EUNSUPPORTED, "wgrad: f32 compute needs f32 operands");
        return wdispatch<float, float, float>(k, g, d->stride, d->nsplit, st);
    }
    if (d->dtype_p == ABC_BF16 && d->dtype_q == ABC_BF16) return wdispatch<bf16, bf16, bf16>(k, g, d->stride, d->nsplit, st);
    if (d->dtype_p == ABC_F32 && d->dtype_q == ABC_BF16) return wdispatch<float, bf16, bf16>(k, g, d->stride, d->nsplit, st);
    if (d->dtype_p == ABC_BF16 && d->dtype_q == ABC_F32) return wdispatch<bf16, float, bf16>(k, g, d->stride, d->nsplit, st);
    return abc_fail(ABC_EUNSUPPORTED, "wgrad: dtype combination");
}

extern "C" int abc_wgrad_reduce_batch(const abc_wgrad_reduce_desc* descs, int32_t n, abc_stream_t stream) {
    if (n < 1 || n > MAX_RB) return abc_fail(ABC_EINVAL, "wgrad_reduce_batch: 1..16 items");
    ReduceBatch bt;
    int64_t gx = 1;
    for (int i = 0; i < n; ++i) {
        bt.d[i] = descs[i];
        const int64_t cnt = (int64_t)descs[i].ntaps * descs[i].Ca * descs[i].Cb;
        const int64_t blocks = (cnt <= 4096 && descs[i].nsplit >= 128) ? (cnt + 3) / 4 : (cnt + 255) / 256;
        gx = blocks > gx ? blocks : gx;
    }
    for (int i = n; i < MAX_RB; ++i) bt.d[i] = descs[0];
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((int)gx, n), dim3(256), 0, (hipStream_t)stream, bt);
    return abc_check_launch("wgrad_reduce_batch");
}

extern "C" int abc_wgrad_reduce(const abc_wgrad_reduce_desc* d, abc_stream_t stream);
// The slab reduction of one weight gradient and the BatchNorm-backward finaliser of ANOTHER layer (both inputs complete, neither
// reads the other's output) as one launch; the small-output reduction form (one wave per element) is launched on its own.
extern "C" int abc_wgrad_reduce_bn_bwd(const abc_wgrad_reduce_desc* d, const abc_bn_bwd_desc* f, abc_stream_t stream) {
    if (f->C < 1 || f->nblk < 1) return abc_fail(ABC_EINVAL, "wgrad_reduce_bn_bwd: empty finaliser");
    const int64_t n = (int64_t)d->ntaps * d->Ca * d->Cb;
    if (n <= 4096 && d->nsplit >= 128) {
        if (int rc = abc_wgrad_reduce(d, stream)) return rc;
        return abc_bn_finalize_bwd(f, stream);
    }
    ReduceBn a;
    a.r = *d; a.f = *f;
    a.vec = (d->Cb % 4 == 0 && d->Cb_pad % 4 == 0 && ((uintptr_t)d->partial & 15) == 0 && d->nsplit >= 16 && (n >= 16384 || d->nsplit >= 64)) ? 1 : 0;   // (few outputs over many slabs: the scalar form walks them four at a time, 32 us for 4608 x 256)
    a.nr = a.vec ? (int)((n / 4 + 63) / 64) : (int)((n + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_bn_kernel, dim3(a.nr + f->C), dim3(256), 0, (hipStream_t)stream, a);
    return abc_check_launch("wgrad_reduce_bn_bwd");
}

extern "C" int abc_wgrad_reduce(const abc_wgrad_reduce_desc* d, abc_stream_t stream) {
    const int64_t n = (int64_t)d->ntaps * d->Ca * d->Cb;
    if (n <= 4096 && d->nsplit >= 128) {
        hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3((int)n), dim3(64), 0, (hipStream_t)stream, *d);
        return abc_check_launch("wgrad_reduce");
    }
    static const bool scalar_only = abc_knob("ABC_WGRAD_REDUCE_SCALAR") != nullptr;   // (A/B runs)
    if (!scalar_only && d->Cb % 4 == 0 && d->Cb_pad % 4 == 0 && ((uintptr_t)d->partial & 15) == 0 && d->nsplit >= 16 && (n >= 16384 || d->nsplit >= 64)) {
        hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3((int)((n / 4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, *d);
        return abc_check_launch("wgrad_reduce");
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
    return abc_check_launch("wgrad_reduce");
}
