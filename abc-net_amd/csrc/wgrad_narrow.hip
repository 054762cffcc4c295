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
    unsigned bytesG, bytesY2, bytesX, bytesOut;
    int dbg;                             // debug build: phase-skipping ablations (bit 0 loads, 1 commit, 2 MFMA phase, 3 dY store)
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

// ---------------------------------------------------------------------------
// Weight gradient of unet2's 5x5 32 -> 32 convolutions at full resolution (unet2.py:52-58 under autograd): 25 taps x 32 x 32.
// The general kernel ran them tap-split over the eight waves of one 32 x 32 tile pair (236 us per layer for 604 MB of g + y_raw + X in,
// dY out: 2.6 TB/s, the matrix pipe at 0.22).  Same ideas as the 16-channel kernel above on WORKGROUP-shared images of one 8 x 16
// tile: four waves, wave (at, bt) owns the 16 x 16 channel block of all 25 taps (100 accumulator registers; one wave per SIMD, two
// workgroups per CU), every wave stores its own block of the slab, nothing to fold.
// The K slots of a lane are pixel PAIRS (column c, column c + 8) of one tile row: one dword of the images holds
//   P^T[ch][row][i] = (P[row][i], P[row][i + 8]), i < 8          Q^T[ch][halo row][i] = (Q[i], Q[i + 8]), i < 12 (halo columns -2 .. 17)
// so that the B fragment of tap dx is the dwords 4 h + dx + 2 .. + 3 of ONE 8-dword window of the lane (h = its column half): no
// half-dword shifts; and the window of halo row w + (kg >> 1) serves every (K step ks, kernel row dy) with 2 ks + dy = w: 11 windows
// per tile instead of 20.  Staging: a thread owns a pixel pair x 4 channels, (even, odd) go to bf16 together (v_cvt_pk_bf16_f32 =
// the image's dword) and the LDS writes are conflict-free dwords (channels 16 .. 31 sit 32 bytes further); the tile's position is
// the SCALAR offset of the buffer loads and stores, the lane's part a loop invariant, and only border tiles test pixels.
constexpr int P32RS = 8 * 8 * 4 + 16;          // P^T channel stride: 8 rows x 8 dwords + pad (68 dwords: 4 channels = 16 banks)
constexpr int P32HI = 16 * P32RS + 32;         // channels 16 .. 31 start 8 banks further
constexpr int Q32ROW = 12 * 4;                 // Q^T halo row: 12 dwords
constexpr int Q32CS = 12 * Q32ROW + 16;        // Q^T channel stride (148 dwords: 4 channels = 16 banks)
constexpr int Q32HI = 16 * Q32CS + 32;
constexpr int W32_P = 2 * P32HI, W32_Q = 2 * Q32HI;

__device__ inline unsigned pk2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t r; r[0] = (bf16)lo; r[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned, r);
}
__device__ inline float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

template <bool DUAL, bool QT>
__global__ __launch_bounds__(256, 2) void wgrad_n32r2_kernel(const WnK a) {
    __shared__ __attribute__((aligned(16))) char smem[W32_P + W32_Q];
    __shared__ __attribute__((aligned(16))) float scoef[6][32];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int at = wave >> 1, bt = wave & 1;
    char* sP = smem;
    char* sQ = smem + W32_P;
    const int q4 = lane >> 3, i8 = lane & 7;   // this thread always stages channels 4 q4 .. 4 q4 + 3; entry i8 of its wave's group of 8
    // (X's resource starts two rows and two pixels BEFORE the tensor: halo offsets are never negative; what lies outside an image is masked)
    const int xshift = (2 * a.W + 2) * a.ldx;
    const __amdgpu_buffer_rsrc_t rsG = abc_make_rsrc(a.g, a.bytesG), rsY = abc_make_rsrc(DUAL ? a.y2 : a.g, DUAL ? a.bytesY2 : 0u),
                                 rsX = abc_make_rsrc(a.x - xshift, a.bytesX + 2u * (unsigned)xshift),
                                 rsO = abc_make_rsrc(DUAL && a.dy_out ? a.dy_out : (bf16*)a.g, DUAL && a.dy_out ? a.bytesOut : 0u);
    if (tid < 192) {
        const int w = tid >> 5, c = tid & 31;
        const float* src = w == 0 ? a.ca : (w == 1 ? a.cb : (w == 2 ? a.cc : (w == 3 ? a.qsc : (w == 4 ? a.qsh : a.qsl))));
        scoef[w][c] = src != nullptr ? src[c] : 0.f;
    }
    __syncthreads();
    f32x4 acc[5][5];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) acc[dy][dx] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging.  P: pass it < 2, group gw = wave + 4 it = the tile row, entry i8: pixels (row, i8) and (row, i8 + 8).
    //           Q: pass it < 5, group gw < 18: flat entry L = 8 gw + i8 of the 12 x 12 (halo row, i) table: halo pixels (hr, i) and (hr, i + 8)
    const unsigned voG = (unsigned)(i8 * a.ldg + a.cg_off + 4 * q4) * 2u, voY = (unsigned)(i8 * a.ldy2 + a.cy2_off + 4 * q4) * 2u,
                   voO = (unsigned)(i8 * a.ld_out + 4 * q4) * 2u;
    unsigned voX[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int L = 8 * (wave + 4 * it) + i8;
        const int hr = (L * 5462) >> 16, hi = L - 12 * hr;      // (L / 12 for L < 160)
        voX[it] = (unsigned)((hr * a.W + hi) * a.ldx + a.cx_off + 4 * q4) * 2u;
    }
    u32x2 rg[2][2], ry[DUAL ? 2 : 1][2], rx[5][2];
    // bit 0 / 1: the tile touches the top / bottom of its image, bit 2 / 3: the left / right
    auto border = [&](int tx, int ty) { return (ty == 0 ? 1 : 0) | (ty == a.tiles_y - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == a.tiles_x - 1 ? 8 : 0); };
    auto outside = [&](int m, int it, int e) {
        const int L = 8 * (wave + 4 * it) + i8;
        const int hr = (L * 5462) >> 16, c = L - 12 * hr + 8 * e;
        return ((m & 1) && hr < 2) || ((m & 2) && hr >= 10) || ((m & 4) && c < 2) || ((m & 8) && c >= 18);
    };
    auto issue = [&](int tile) {
        int id = tile;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int b = id / a.tiles_y;
        const unsigned pixbase = (unsigned)((b * a.H + ty * 8) * a.W + tx * 16);
        const int m = border(tx, ty);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const unsigned pix = pixbase + (unsigned)((wave + 4 * it) * a.W);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                rg[it][e] = __builtin_amdgcn_raw_buffer_load_b64(rsG, voG, (pix + 8 * e) * (unsigned)a.ldg * 2u, 0);
                if constexpr (DUAL) ry[it][e] = __builtin_amdgcn_raw_buffer_load_b64(rsY, voY, (pix + 8 * e) * (unsigned)a.ldy2 * 2u, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            if (wave + 4 * it < 18) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const unsigned so = (pixbase + 8 * e) * (unsigned)a.ldx * 2u;
                    if (m == 0) rx[it][e] = __builtin_amdgcn_raw_buffer_load_b64(rsX, voX[it], so, 0);      // an interior tile (nine in ten): nothing to test
                    else rx[it][e] = __builtin_amdgcn_raw_buffer_load_b64(rsX, outside(m, it, e) ? 0x80000000u : voX[it], so, 0);
                }
            }
        }
    };
    // fragment addresses: lane l = (channel l & 15 of a channel tile, K group kg = l >> 4): tile row 2 ks + (kg >> 1), column half kg & 1
    const int ch = lane & 15, kg = lane >> 4;
    const char* aP = sP + at * P32HI + ch * P32RS + kg * 16;                                     // + ks * 64
    const char* aQ = sQ + bt * Q32HI + ch * Q32CS + (kg >> 1) * Q32ROW + (kg & 1) * 16;          // + w halo rows
    char* wP = sP + 4 * q4 * P32RS + (q4 >> 2) * 32 + i8 * 4;                                    // + row * 32, + c * P32RS
    char* wQ = sQ + 4 * q4 * Q32CS + (q4 >> 2) * 32 + i8 * 4;                                    // + 32 gw, + c * Q32CS

    const int first = blockIdx.x, step = gridDim.x;
    if (first < a.ntiles && !(ABC_DBG(a.dbg) & 1)) issue(first);
    for (int tile = first; tile < a.ntiles; tile += step) {
        int id = tile;
        const int tx = id % a.tiles_x; id /= a.tiles_x;
        const int ty = id % a.tiles_y;
        const int b = id / a.tiles_y;
        const unsigned pixbase = (unsigned)((b * a.H + ty * 8) * a.W + tx * 16);
        const int m = border(tx, ty);
        if (!(ABC_DBG(a.dbg) & 2)) {
            float ca[DUAL ? 4 : 1], cb[DUAL ? 4 : 1], cc[DUAL ? 4 : 1];
            if constexpr (DUAL) { LoadVec<float, 4>::ld(&scoef[0][4 * q4], ca); LoadVec<float, 4>::ld(&scoef[1][4 * q4], cb); LoadVec<float, 4>::ld(&scoef[2][4 * q4], cc); }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = wave + 4 * it;
                unsigned o[4];
                if constexpr (DUAL) {
                    float v[2][4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        v[e][0] = fmaf(ca[0], bf_lo(rg[it][e].x), fmaf(cb[0], bf_lo(ry[it][e].x), cc[0]));
                        v[e][1] = fmaf(ca[1], bf_hi(rg[it][e].x), fmaf(cb[1], bf_hi(ry[it][e].x), cc[1]));
                        v[e][2] = fmaf(ca[2], bf_lo(rg[it][e].y), fmaf(cb[2], bf_lo(ry[it][e].y), cc[2]));
                        v[e][3] = fmaf(ca[3], bf_hi(rg[it][e].y), fmaf(cb[3], bf_hi(ry[it][e].y), cc[3]));
                        if (a.dy_out != nullptr && !(ABC_DBG(a.dbg) & 8))
                            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pk2(v[e][0], v[e][1]), pk2(v[e][2], v[e][3])}, rsO, voO,
                                                                  (pixbase + (unsigned)(r * a.W) + 8 * e) * (unsigned)a.ld_out * 2u, 0);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = pk2(v[0][c], v[1][c]);
                } else {
                    o[0] = __builtin_amdgcn_perm(rg[it][1].x, rg[it][0].x, 0x05040100u); o[1] = __builtin_amdgcn_perm(rg[it][1].x, rg[it][0].x, 0x07060302u);
                    o[2] = __builtin_amdgcn_perm(rg[it][1].y, rg[it][0].y, 0x05040100u); o[3] = __builtin_amdgcn_perm(rg[it][1].y, rg[it][0].y, 0x07060302u);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) *(unsigned*)(wP + r * 32 + c * P32RS) = o[c];
            }
        }
        if (!(ABC_DBG(a.dbg) & 2)) {
            float qsc[QT ? 4 : 1], qsh[QT ? 4 : 1], qsl[QT ? 4 : 1];
            if constexpr (QT) { LoadVec<float, 4>::ld(&scoef[3][4 * q4], qsc); LoadVec<float, 4>::ld(&scoef[4][4 * q4], qsh); LoadVec<float, 4>::ld(&scoef[5][4 * q4], qsl); }
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int gw = wave + 4 * it;
                if (gw < 18) {
                    unsigned o[4];
                    if constexpr (QT) {
                        float v[2][4];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            v[e][0] = abc_act(bf_lo(rx[it][e].x), qsc[0], qsh[0], qsl[0]);
                            v[e][1] = abc_act(bf_hi(rx[it][e].x), qsc[1], qsh[1], qsl[1]);
                            v[e][2] = abc_act(bf_lo(rx[it][e].y), qsc[2], qsh[2], qsl[2]);
                            v[e][3] = abc_act(bf_hi(rx[it][e].y), qsc[3], qsh[3], qsl[3]);
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = pk2(v[0][c], v[1][c]);
                        if (m != 0) {              // (the padding of the convolution is zero AFTER the activation)
                            const unsigned keep = (outside(m, it, 0) ? 0u : 0x0000FFFFu) | (outside(m, it, 1) ? 0u : 0xFFFF0000u);
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] &= keep;
                        }
                    } else {
                        o[0] = __builtin_amdgcn_perm(rx[it][1].x, rx[it][0].x, 0x05040100u); o[1] = __builtin_amdgcn_perm(rx[it][1].x, rx[it][0].x, 0x07060302u);
                        o[2] = __builtin_amdgcn_perm(rx[it][1].y, rx[it][0].y, 0x05040100u); o[3] = __builtin_amdgcn_perm(rx[it][1].y, rx[it][0].y, 0x07060302u);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) *(unsigned*)(wQ + gw * 32 + c * Q32CS) = o[c];
                }
            }
        }
        __syncthreads();               // the tile's images are complete
        if (tile + step < a.ntiles && !(ABC_DBG(a.dbg) & 1)) issue(tile + step);      // (the next tile's loads fly under the MFMAs)
        if (!(ABC_DBG(a.dbg) & 4)) {
            bf16x8 fa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ks] = *(const bf16x8*)(aP + ks * 64);
#pragma unroll
            for (int w = 0; w < 11; ++w) {
                const char* rp = aQ + w * Q32ROW;
                const u32x4 wl = *(const u32x4*)rp, wh = *(const u32x4*)(rp + 16);
                typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
                const u32x8 t = __builtin_shufflevector(wl, wh, 0, 1, 2, 3, 4, 5, 6, 7);
                const u32x4 f1 = __builtin_shufflevector(t, t, 1, 2, 3, 4), f2 = __builtin_shufflevector(t, t, 2, 3, 4, 5),
                            f3 = __builtin_shufflevector(t, t, 3, 4, 5, 6);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int dy = w - 2 * ks;
                    if (dy >= 0 && dy < 5) {
                        acc[dy][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks], *(const bf16x8*)&wl, acc[dy][0], 0, 0, 0);
                        acc[dy][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks], *(const bf16x8*)&f1, acc[dy][1], 0, 0, 0);
                        acc[dy][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks], *(const bf16x8*)&f2, acc[dy][2], 0, 0, 0);
                        acc[dy][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks], *(const bf16x8*)&f3, acc[dy][3], 0, 0, 0);
                        acc[dy][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks], *(const bf16x8*)&wh, acc[dy][4], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();               // every wave is done reading before the next commit overwrites the images
    }
    // accumulator register i of lane l = dW[tap 5 dy + dx][a = 16 at + 4 (l >> 4) + i][b = 16 bt + (l & 15)]
    float* out = a.partial + (size_t)blockIdx.x * 25 * 1024;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(5 * dy + dx) * 1024 + (16 * at + 4 * kg + i) * 32 + 16 * bt + ch] = acc[dy][dx][i];
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
    k.bytesOut = 0u; k.dbg = 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(d->nsplit), blk(256);
    if (dual && qt) hipLaunchKernelGGL((wgrad_narrow16_kernel<true, true>), grid, blk, 0, st, k);
    else if (dual) hipLaunchKernelGGL((wgrad_narrow16_kernel<true, false>), grid, blk, 0, st, k);
    else if (qt) hipLaunchKernelGGL((wgrad_narrow16_kernel<false, true>), grid, blk, 0, st, k);
    else hipLaunchKernelGGL((wgrad_narrow16_kernel<false, false>), grid, blk, 0, st, k);
    return abc_check_launch("wgrad_narrow16");
}

// 1 when the 5x5 32-channel kernel takes this descriptor
int abc_wgrad_n32r2_ok(const abc_wgrad_desc* d) {
    if (abc_knob("ABC_WGRAD_NON32R2")) return 0;
    if (d->Ca != 32 || d->Cb != 32 || d->ntaps != 25 || d->stride != 1) return 0;
    if (d->dtype_p != ABC_BF16 || d->dtype_q != ABC_BF16 || d->dtype_c != ABC_BF16) return 0;
    for (int t = 0; t < 25; ++t)
        if (d->tap_dy[t] != t / 5 - 2 || d->tap_dx[t] != t % 5 - 2) return 0;
    if (d->p.pool || d->q.pool || d->p.planar || d->q.planar || d->p.drop_p > 0.f || d->q.drop_p > 0.f) return 0;
    if (d->Hg % 8 || d->Wg % 16 || d->Hq != d->Hg || d->Wq != d->Wg || d->p.Hx != d->Hg || d->p.Wx != d->Wg || d->q.Hx != d->Hg || d->q.Wx != d->Wg) return 0;
    if ((d->p.ldx % 8) || (d->cp_off % 8) || (d->q.ldx % 8) || (d->cq_off % 8)) return 0;
    if (d->p_dual) {
        if (d->p.scale == nullptr || d->p2 == nullptr || (d->ld_p2 % 8) || (d->cp2_off % 8) || (d->p_out && (d->ld_pout % 8))) return 0;
        if ((int64_t)d->B * d->Hg * d->Wg * d->ld_p2 * 2 >= (int64_t(1) << 31)) return 0;
    } else if (d->p.scale != nullptr) return 0;
    if ((int64_t)d->B * d->Hg * d->Wg * d->p.ldx * 2 >= (int64_t(1) << 31) || (int64_t)d->B * d->Hg * d->Wg * d->q.ldx * 2 >= (int64_t(1) << 31)) return 0;
    return 1;
}

int abc_wgrad_n32r2_launch(const abc_wgrad_desc* d, abc_stream_t stream) {
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
    k.bytesOut = dual && d->p_out ? (unsigned)((int64_t)d->B * d->Hg * d->Wg * d->ld_pout * 2) : 0u;
    k.dbg = abc_knob("ABC_W32_DBG") ? atoi(abc_knob("ABC_W32_DBG")) : 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(d->nsplit), blk(256);
    if (dual && qt) hipLaunchKernelGGL((wgrad_n32r2_kernel<true, true>), grid, blk, 0, st, k);
    else if (dual) hipLaunchKernelGGL((wgrad_n32r2_kernel<true, false>), grid, blk, 0, st, k);
    else if (qt) hipLaunchKernelGGL((wgrad_n32r2_kernel<false, true>), grid, blk, 0, st, k);
    else hipLaunchKernelGGL((wgrad_n32r2_kernel<false, false>), grid, blk, 0, st, k);
    return abc_check_launch("wgrad_n32r2");
}
