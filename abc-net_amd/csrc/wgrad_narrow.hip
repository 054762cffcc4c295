// Weight gradient of the 16-channel 3x3 convolutions at full resolution (unet.py:12,15 under autograd, train.py:140: inc1.conv2,
// inc2.conv1, inc2.conv2 -- dW[t][a][b] = sum over pixels of dY[p][a] * act(X)[p + d_t][b], a, b < 16).
//
// These layers are HBM-bound (g + y_raw + X read, dY written: 302 MB per layer at b16, 384 x 384) and the general kernel
// (wgrad.hip) ran them at 3.85 TB/s: its wave holds all nine taps of a 32 x 32 tile pair (144 accumulator registers, a quarter of
// them useful at 16 channels), so 256 registers leave room for ONE patch of prefetch and two waves per SIMD -- 57 KB in flight per
// CU, an iteration = one loaded HBM round trip + the commit; and every workgroup leaves a 32 x 32 x 9 f32 slab (75 MB per layer
// for the reduction to read back).  Nothing of that is needed at 16 channels:
//
//   * v_mfma_f32_16x16x32_bf16: a tap's whole 16 x 16 result is ONE accumulator of four registers, nine taps = 36;
//   * a WAVE owns an 8 x 16 pixel tile and walks a strided run of tiles with its own LDS images -- no workgroup barriers in the
//     loop, only its own LDS queue to wait for; two waves per SIMD (197 registers: at 168 the prefetch registers spill and every
//     scratch reload waits for the loads in flight), each with the next tile's 14 16-byte loads in flight;
//   * both operands go into LDS TRANSPOSED ([channel][pixel], 2-byte writes: 80 per lane and tile), so that every fragment is a
//     plain aligned ds_read_b128: K = 32 pixels = two tile rows; a lane's eight pixels of the shifted operand for the three dx
//     taps come out of ONE 10-pixel window (16 + 4 bytes read, the middle tap by four v_alignbit_b32);
//   * the BatchNorm-backward correction dY = ca g + cb y_raw + cc (abc_wgrad_desc.p_dual) on the way in, dY stored for the data
//     gradient; BatchNorm + activation of X on the way in (zero padding applies to the ACTIVATED tensor);
//   * a workgroup's four waves fold their sums through LDS once: a 9 x 16 x 16 slab per workgroup (9 KB: 7 MB per layer).
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"
#include "conv_fast.hpp"

namespace {

struct WnK {
    const bf16* g; const bf16* y2; const bf16* x; bf16* dy_out; float* partial;
    const float *ca, *cb, *cc;           // dual: per a-channel (already at the first channel), or null
    const float *qsc, *qsh, *qsl;        // transform of X per b-channel (already at the first channel), or null
    int B, H, W, ldg, cg_off, ldy2, cy2_off, ldx, cx_off, ld_out;
    int tiles_x, tiles_y, ntiles;
    unsigned bytesG, bytesY2, bytesX;
};

constexpr int PRS = 128 * 2 + 16;            // P^T row: 128 pixels bf16 + pad (17 16-byte slots: 16 channels on 16 distinct slots)
constexpr int QROW = 24 * 2;                 // Q^T pixel row: columns -1 .. 16 at indices 0 .. 17, padded to 24
constexpr int QCS = 10 * QROW + 16;          // Q^T channel stride (31 slots)
constexpr int WLDS = 16 * PRS + 16 * QCS;    // per wave: 4352 + 7936 = 12288 bytes

template <bool DUAL, bool QT>
__global__ __launch_bounds__(256, 2) void wgrad_narrow16_kernel(const WnK a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * WLDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* sP = smem + wave * WLDS;
    char* sQ = sP + 16 * PRS;
    const int half = lane & 1;                 // this lane always stages channels 8 half .. 8 half + 7
    const __amdgpu_buffer_rsrc_t rsG = abc_make_rsrc(a.g, a.bytesG), rsY = abc_make_rsrc(DUAL ? a.y2 : a.g, DUAL ? a.bytesY2 : 0u),
                                 rsX = abc_make_rsrc(a.x, a.bytesX);
    // coefficient rows [ca | cb | cc | qsc | qsh | qsl][16] in LDS: a lane reads its eight per commit (in registers for the whole kernel
    // they were 48 of 168 and spilled the prefetch)
    __shared__ __attribute__((aligned(16))) float scoef[6][16];
    if (threadIdx.x < 96) {
        const int w = threadIdx.x >> 4, c = threadIdx.x & 15;
        const float* src = w == 0 ? a.ca : (w == 1 ? a.cb : (w == 2 ? a.cc : (w == 3 ? a.qsc : (w == 4 ? a.qsh : a.qsl))));
        scoef[w][c] = src != nullptr ? src[c] : 0.f;
    }
    __syncthreads();
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging roles.  P: segment s = lane + 64 i (i < 4): pixel s >> 1 of the tile (row (s >> 1) >> 4, column (s >> 1) & 15).
    // Q: segment s = lane + 64 i (i < 6, s < 360): halo pixel q = s >> 1: row q / 18 (image row y0 - 1 + that), column q % 18 (x0 - 1 + that)
    u32x4 rg[4], ry[DUAL ? 4 : 1], rx[6];
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    auto issue = [&](int tile) {
        const bool live = tile < a.ntiles;
        int id = tile;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int b = id / a.tiles_y;
        const int y0 = ty * 8, x0 = tx * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = (lane >> 1) + 32 * i;
            const unsigned pix = (unsigned)((b * a.H + y0 + (p >> 4)) * a.W + x0 + (p & 15));
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsG, live ? (pix * (unsigned)a.ldg + (unsigned)(a.cg_off + 8 * half)) * 2u : 0x80000000u, 0, 0);
            if constexpr (DUAL)
                ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, live ? (pix * (unsigned)a.ldy2 + (unsigned)(a.cy2_off + 8 * half)) * 2u : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = (lane >> 1) + 32 * i;
            const int hr = (q * 3641) >> 16, hc = q - 18 * hr;      // (q / 18 for q < 192)
            const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
            const bool ok = live && q < 180 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? ((unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)a.ldx + (unsigned)(a.cx_off + 8 * half)) * 2u : 0x80000000u, 0, 0);
        }
    };
    // fragment addresses: lane l = (channel l & 15, K group l >> 4): tile row 2 ks + (kg >> 1), column half kg & 1
    const int ch = lane & 15, kg = lane >> 4;
    const char* aP = sP + ch * PRS + ((kg >> 1) * 16 + (kg & 1) * 8) * 2;                 // + ks * 64 bytes
    const char* aQ = sQ + ch * QCS + (kg >> 1) * QROW + (kg & 1) * 16;                    // + (2 ks + dy) * QROW, dy = 0 .. 2

    issue(wid);
    for (int tile = wid; tile < a.ntiles; tile += nw) {
        int id = tile;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int b = id / a.tiles_y;
        const int y0 = ty * 8, x0 = tx * 16;
        // ---- commit: registers -> (transform) -> transposed LDS images (the wave's previous fragment reads are ahead of these
        // writes in its LDS queue)
        float ca[DUAL ? 8 : 1], cb[DUAL ? 8 : 1], cc[DUAL ? 8 : 1];
        if constexpr (DUAL) { LoadVec<float, 8>::ld(&scoef[0][8 * half], ca); LoadVec<float, 8>::ld(&scoef[1][8 * half], cb); LoadVec<float, 8>::ld(&scoef[2][8 * half], cc); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = (lane >> 1) + 32 * i;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(rg[i][j] << 16); v[2 * j + 1] = __uint_as_float(rg[i][j] & 0xFFFF0000u); }
            bf16x8 o;
            if constexpr (DUAL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y0v = __uint_as_float(ry[i][j] << 16), y1v = __uint_as_float(ry[i][j] & 0xFFFF0000u);
                    v[2 * j] = fmaf(ca[2 * j], v[2 * j], fmaf(cb[2 * j], y0v, cc[2 * j]));
                    v[2 * j + 1] = fmaf(ca[2 * j + 1], v[2 * j + 1], fmaf(cb[2 * j + 1], y1v, cc[2 * j + 1]));
                }
                o = pack_frag<bf16>(v);
                if (a.dy_out != nullptr) {
                    const size_t pix = (size_t)(b * a.H + y0 + (p >> 4)) * a.W + x0 + (p & 15);
                    *(bf16x8*)(a.dy_out + pix * a.ld_out + 8 * half) = o;
                }
            } else {
                o = *(const bf16x8*)&rg[i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *(bf16*)(sP + (8 * half + j) * PRS + p * 2) = o[j];
        }
        float qsc[QT ? 8 : 1], qsh[QT ? 8 : 1], qsl[QT ? 8 : 1];
        if constexpr (QT) { LoadVec<float, 8>::ld(&scoef[3][8 * half], qsc); LoadVec<float, 8>::ld(&scoef[4][8 * half], qsh); LoadVec<float, 8>::ld(&scoef[5][8 * half], qsl); }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = (lane >> 1) + 32 * i;
            const int hr = (q * 3641) >> 16, hc = q - 18 * hr;
            if (q < 180) {
                bf16x8 o;
                if constexpr (QT) {
                    const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
                    const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(rx[i][j] << 16); v[2 * j + 1] = __uint_as_float(rx[i][j] & 0xFFFF0000u); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = in ? abc_act(v[j], qsc[j], qsh[j], qsl[j]) : 0.f;
                    o = pack_frag<bf16>(v);
                } else {
                    o = *(const bf16x8*)&rx[i];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) *(bf16*)(sQ + (8 * half + j) * QCS + hr * QROW + hc * 2) = o[j];
            }
        }
        issue(tile + nw);      // (the next tile's loads fly under the MFMAs and the next commit waits for them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- 4 K-steps of 32 pixels x 9 taps
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 fa = *(const bf16x8*)(aP + ks * 64);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // columns 8 (kg & 1) - 1 .. + 8 of image row (tile row + dy - 1): indices 8 (kg & 1) .. + 9 of the padded row
                const char* rp = aQ + (2 * ks + dy) * QROW;
                const u32x4 w = *(const u32x4*)rp;
                const unsigned w4 = *(const unsigned*)(rp + 16);
                u32x4 m, r;
                m[0] = __builtin_amdgcn_alignbit(w[1], w[0], 16); m[1] = __builtin_amdgcn_alignbit(w[2], w[1], 16);
                m[2] = __builtin_amdgcn_alignbit(w[3], w[2], 16); m[3] = __builtin_amdgcn_alignbit(w4, w[3], 16);
                r[0] = w[1]; r[1] = w[2]; r[2] = w[3]; r[3] = w4;
                acc[3 * dy + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, *(const bf16x8*)&w, acc[3 * dy + 0], 0, 0, 0);
                acc[3 * dy + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, *(const bf16x8*)&m, acc[3 * dy + 1], 0, 0, 0);
                acc[3 * dy + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, *(const bf16x8*)&r, acc[3 * dy + 2], 0, 0, 0);
                // (one window at a time: hoisted, the 28 fragment reads of a tile took 100 registers and spilled the prefetch)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- fold the four waves: accumulator register i of lane l = dW[t][a = 4 (l >> 4) + i][b = l & 15]
    __syncthreads();
    float* red = (float*)smem;     // [4 waves][9][256]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * 9 + t) * 256 + (4 * (lane >> 4) + i) * 16 + (lane & 15)] = acc[t][i];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 256; i += 256)
        a.partial[(size_t)blockIdx.x * 9 * 256 + i] = (red[i] + red[9 * 256 + i]) + (red[2 * 9 * 256 + i] + red[3 * 9 * 256 + i]);
}

}  // namespace

// 1 when the 16-channel kernel takes this descriptor
int abc_wgrad_narrow_ok(const abc_wgrad_desc* d) {
    if (abc_knob("ABC_WGRAD_NONARROW")) return 0;
    if (d->Ca != 16 || d->Cb != 16 || d->ntaps != 9 || d->stride != 1) return 0;
    if (d->dtype_p != ABC_BF16 || d->dtype_q != ABC_BF16 || d->dtype_c != ABC_BF16) return 0;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) return 0;
    if (d->p.pool || d->q.pool || d->p.planar || d->q.planar || d->p.drop_p > 0.f || d->q.drop_p > 0.f) return 0;
    if (d->Hg % 8 || d->Wg % 16 || d->Hq != d->Hg || d->Wq != d->Wg || d->p.Hx != d->Hg || d->p.Wx != d->Wg || d->q.Hx != d->Hg || d->q.Wx != d->Wg) return 0;
    if ((d->p.ldx % 8) || (d->cp_off % 8) || (d->q.ldx % 8) || (d->cq_off % 8)) return 0;
    if (d->p_dual) {
        if (d->p.scale == nullptr || d->p2 == nullptr || (d->ld_p2 % 8) || (d->cp2_off % 8) || (d->p_out && (d->ld_pout % 8))) return 0;
        if ((int64_t)d->B * d->Hg * d->Wg * d->ld_p2 * 2 >= (int64_t(1) << 31)) return 0;
    } else if (d->p.scale != nullptr) return 0;      // (a transform on P only as the BatchNorm-backward correction)
    if ((int64_t)d->B * d->Hg * d->Wg * d->p.ldx * 2 >= (int64_t(1) << 31) || (int64_t)d->B * d->Hg * d->Wg * d->q.ldx * 2 >= (int64_t(1) << 31)) return 0;
    return 1;
}

int abc_wgrad_narrow_launch(const abc_wgrad_desc* d, abc_stream_t stream) {
    WnK k;
    k.g = (const bf16*)d->p.x; k.y2 = (const bf16*)d->p2; k.x = (const bf16*)d->q.x; k.dy_out = (bf16*)d->p_out; k.partial = d->partial;
    const bool dual = d->p_dual != 0, qt = d->q.scale != nullptr;
    k.ca = dual ? d->p.scale + d->cp_off : nullptr; k.cc = dual ? d->p.shift + d->cp_off : nullptr; k.cb = dual ? d->p.slope + d->cp_off : nullptr;
    k.qsc = qt ? d->q.scale + d->cq_off : nullptr; k.qsh = qt ? d->q.shift + d->cq_off : nullptr; k.qsl = qt ? d->q.slope + d->cq_off : nullptr;
    k.B = d->B; k.H = d->Hg; k.W = d->Wg; k.ldg = d->p.ldx; k.cg_off = d->cp_off; k.ldy2 = d->ld_p2; k.cy2_off = d->cp2_off;
    k.ldx = d->q.ldx; k.cx_off = d->cq_off; k.ld_out = d->ld_pout;
    k.tiles_x = d->Wg / 16; k.tiles_y = d->Hg / 8; k.ntiles = k.tiles_x * k.tiles_y * d->B;
    k.bytesG = (unsigned)((int64_t)d->B * d->Hg * d->Wg * d->p.ldx * 2);
    k.bytesY2 = dual ? (unsigned)((int64_t)d->B * d->Hg * d->Wg * d->ld_p2 * 2) : 0u;
    k.bytesX = (unsigned)((int64_t)d->B * d->Hg * d->Wg * d->q.ldx * 2);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(d->nsplit), blk(256);
    if (dual && qt) hipLaunchKernelGGL((wgrad_narrow16_kernel<true, true>), grid, blk, 0, st, k);
    else if (dual) hipLaunchKernelGGL((wgrad_narrow16_kernel<true, false>), grid, blk, 0, st, k);
    else if (qt) hipLaunchKernelGGL((wgrad_narrow16_kernel<false, true>), grid, blk, 0, st, k);
    else hipLaunchKernelGGL((wgrad_narrow16_kernel<false, false>), grid, blk, 0, st, k);
    return abc_check_launch("wgrad_narrow16");
}
