// ConvTranspose2d(C -> C / 2, k3, s2) + the reference's crop of the first row / column (unet.py:44,51-56), forward, ALL FOUR output-parity
// phases in one pass over the input.
//
// The lean convolution kernel runs the layer as four small convolutions (1 / 2 / 2 / 4 taps over the same input into interleaved output
// pixels, one batched launch: conv_fast.hip, abc_conv_fwd_batch).  On the three up-layers of unet.py (512 -> 256 at 12 x 12, 256 -> 128 at
// 24 x 24, 128 -> 64 at 48 x 48: 5.4 GFLOP each) that is 768 - 1536 workgroups which each stage the input tile and synchronise once per
// 64-byte channel chunk for 2 - 8 MFMAs per wave: 30 - 45 us per layer, 0.05 - 0.07 of the matrix peak, all of it per-chunk overhead.
//
// Here a workgroup owns a 4-row x 16-column INPUT tile and 64 output channels and produces the 8 x 32 output pixels of all four phases:
//   * the halo (5 x 17 input pixels, the producer's BatchNorm + ReLU applied on the way in) is staged ONCE per chunk for the nine
//     (phase, tap) products -- they read only four pixel positions (dy, dx in {0, 1}), eight fragment reads for 18 MFMAs per wave;
//   * four accumulator tiles per wave (one per phase; wave = 32 input pixels x 32 channels);
//   * the nine weight slices of a chunk are fetched by each lane straight from global memory a whole chunk ahead (the packed
//     weights ARE the fragments: abc_pack_desc.layout 1), like the deep-pipelined form of the lean kernel;
//   * MFMA operands swapped (A = weights, B = pixels): a lane holds 16 channels of ONE input pixel per phase, four consecutive channels
//     pack into 8 bytes, one v_permlane32_swap per dword pairs them with the other lane half's -> 16-byte stores, no LDS staging.
// Quarter the workgroups, quarter the halo stagings and barriers, 18 instead of 2 - 8 MFMAs between them.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

constexpr int CK = 32;                 // channels per chunk (64 bytes of bf16)
constexpr int PS = CK * 2 + 16;        // LDS bytes per halo pixel (the lean kernel's padding: 16 pixels = 16 distinct 16-byte bank slots)
constexpr int TROWS = 4, HH = TROWS + 1, HW = 17;
constexpr int RS = 1536;               // LDS bytes per halo row (>= HW * PS = 1360, a multiple of 256)
constexpr int SA = HH * RS;            // one halo buffer
constexpr int NSEG = (HH * HW * 4 + 255) / 256;     // 16-byte segments per thread and chunk (340 / 256 -> 2)

struct CTK {
    const void* x;
    const float *scale, *shift, *slope;
    const void* w;            // nine packed slices [slice][chunk][Cout_pad][32], layout 1
    const float* bias;
    void* y;
    int B, Hin, Win, Hx, Wx, ldx, cin_off, Cin, nchunks;
    int Hout, Wout, ldy, cout_off, Cout, Cout_pad;
    int tiles_x, tiles_y, nbn, ntiles, cstride;
    unsigned bytesA, bytesW;
};

// slice s = (phase, input offset): phase (0,0): (0,0) | (0,1): (0,1) (0,0) | (1,0): (1,0) (0,0) | (1,1): (1,1) (1,0) (0,1) (0,0)
// (engine.convT_phase_taps with both crops; abc_pack_conv_weights mode 2 packs a phase's taps in this order)
__device__ constexpr int ph_of(int s) { return s == 0 ? 0 : (s <= 2 ? 1 : (s <= 4 ? 2 : 3)); }
__device__ constexpr int dy_of(int s) { return (s == 3 || s == 5 || s == 6) ? 1 : 0; }
__device__ constexpr int dx_of(int s) { return (s == 1 || s == 5 || s == 7) ? 1 : 0; }

__global__ __launch_bounds__(256, 2) void convt_fused_kernel(const CTK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    float* sCoef = (float*)(smem + 2 * SA);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;       // wave = input rows 2 wm, 2 wm + 1 of the tile x channels 32 wn .. of the 64-channel block

    const bool has_coef = a.scale != nullptr;
    if (has_coef) {
        for (int i = tid; i < a.Cin; i += 256) {
            sCoef[i] = a.scale[a.cin_off + i]; sCoef[a.cstride + i] = a.shift[a.cin_off + i]; sCoef[2 * a.cstride + i] = a.slope[a.cin_off + i];
        }
    }
    const float* lcoef = has_coef ? sCoef : nullptr;

    int id = abc_xcd_remap(blockIdx.x, gridDim.x);
    const int nb = id % a.nbn; id /= a.nbn;
    const int tx_i = id % a.tiles_x; id /= a.tiles_x;
    const int ty_i = id % a.tiles_y; id /= a.tiles_y;
    const int b = id;
    const int iy0 = ty_i * TROWS, ix0 = tx_i * 16, n0 = nb * 64;

    const __amdgpu_buffer_rsrc_t rsA = abc_make_rsrc(a.x, a.bytesA), rsW = abc_make_rsrc(a.w, a.bytesW);
    const HaloGeom gA = {HH, HW, 65536 / HW + 1, a.Hin, a.Win, a.Hx, a.Wx, a.ldx};
    HaloTile<bf16, bf16, CK, NSEG, 256> apre;
    apre.setup(gA, RS, PS, b, iy0, ix0, a.cin_off, tid);
    apre.issue(rsA, 0u);

    // weight fragments: slice s of chunk c for this wave's 32 channels, 16-byte half kk
    const unsigned slice_stride = (unsigned)(a.nchunks * a.Cout_pad * CK) * 2u, chunk_stride = (unsigned)(a.Cout_pad * CK) * 2u;
    const unsigned bq_voff = (unsigned)((n0 + wn * 32) * CK * 2 + h * 512 + r * 16);
    u32x4 bq[9][2];
    auto bq_load = [&](int s, int c) {
        const unsigned soff = (unsigned)s * slice_stride + (unsigned)c * chunk_stride;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) bq[s][kk] = __builtin_amdgcn_raw_buffer_load_b128(rsW, bq_voff + (unsigned)(kk * 1024), soff, 0);
    };
#pragma unroll
    for (int s = 0; s < 9; ++s) bq_load(s, 0);

    // the lane's bias: channels 8 q + 4 h + e of the wave's 32 (register 4 q + e of an accumulator)
    f32x4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = n0 + wn * 32 + 8 * q + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[q][e] = (a.bias != nullptr && c + e < a.Cout) ? a.bias[c + e] : 0.f;
    }

    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[p][k] = 0.f;

    // this lane's pixel: tile row 2 wm + (r >> 4), column r & 15
    const int prow = 2 * wm + (r >> 4), pcol = r & 15;
    const int aBase = prow * RS + pcol * PS + h * 32;

    __syncthreads();      // coefficient table visible
    apre.commit(sA, lcoef, a.cstride, tid);
    __syncthreads();

    typedef Frag<bf16>::type frag_t;
    for (int c = 0; c < a.nchunks; ++c) {
        const char* sAc = sA + (c & 1) * SA;
        const bool more = c + 1 < a.nchunks;
        if (more) apre.issue(rsA, (unsigned)((c + 1) * CK) * 2u);
        // the four pixel positions (dy, dx), both 16-byte halves
        frag_t fr[4][2];
#pragma unroll
        for (int pos = 0; pos < 4; ++pos)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fr[pos][kk] = *(const frag_t*)(sAc + aBase + (pos >> 1) * RS + (pos & 1) * PS + 16 * kk);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int pos = dy_of(s) * 2 + dx_of(s);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) acc[ph_of(s)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const frag_t*)&bq[s][kk], fr[pos][kk], acc[ph_of(s)], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the slot is free: the same slice of the next chunk (past the last chunk the offsets run off the buffer: zeros, no traffic)
            bq_load(s, c + 1);
        }
        if (more) apre.commit(sA + ((c + 1) & 1) * SA, lcoef ? lcoef + (c + 1) * CK : nullptr, a.cstride, tid);
        __syncthreads();
    }

    // ---- epilogue: register k of phase p = channel (k & 3) + 8 (k >> 2) + 4 h of the wave's 32, at output pixel (2 iy + py, 2 ix + px)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const int iy = iy0 + prow, ix = ix0 + pcol;
    const bool pix_in = iy < a.Hin && ix < a.Win;
    bf16* yo = (bf16*)a.y;
    const bf16* ybase = yo + ((size_t)b * a.Hout * a.Wout) * a.ldy + a.cout_off + n0 + wn * 32;
    const __amdgpu_buffer_rsrc_t rsY = abc_make_rsrc(ybase, 0x80000000u);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int oy = 2 * iy + (p >> 1), ox = 2 * ix + (p & 1);
        const bool ok_px = pix_in && oy < a.Hout && ox < a.Wout;
        const unsigned vpix = (unsigned)((oy * a.Wout + ox) * a.ldy + 8 * h) * 2u;
        unsigned d[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int k = 4 * q + 2 * w;
                const f32x2 v = (f32x2){acc[p][k], acc[p][k + 1]} + (f32x2){bias4[q][2 * w], bias4[q][2 * w + 1]};
                const bf16x2 pk = __builtin_convertvector(v, bf16x2);
                d[q][w] = *(const unsigned*)&pk;
            }
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            u32x4 st;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const u32x2 t = __builtin_amdgcn_permlane32_swap(d[2 * p2][w], d[2 * p2 + 1][w], false, false);
                st[w] = t[0]; st[2 + w] = t[1];
            }
            const bool ok = ok_px && (n0 + wn * 32 + 16 * p2 + 8 * h < a.Cout);
            __builtin_amdgcn_raw_buffer_store_b128(st, rsY, ok ? vpix : 0xFFFFFFF0u, (unsigned)(16 * p2 * 2), 0);
            // (the store-data hazard of conv_fast_body.hpp: keep the data registers live two wait states past the store)
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_nop 1" :: "v"(st));
#endif
        }
    }
}

int ct_check(const abc_convt_desc* d) {
    if (d->dtype != ABC_BF16) return 0;
    if (d->src.pool || d->src.planar || d->src.drop_p > 0.f) return 0;
    if (d->Cin % CK || d->Cin > 1024 || d->Cout % 8 || d->Cout_pad % 64 || d->Cout > d->Cout_pad) return 0;
    if (d->Hout != 2 * d->Hin || d->Wout != 2 * d->Win) return 0;                      // (both axes cropped: unet.py:51-56 with a 2n skip tensor)
    if ((d->ldy | d->cout_off) % 8 || d->src.ldx % 8 || d->cin_off % 8) return 0;
    if ((int64_t)d->B * d->src.Hx * d->src.Wx * d->src.ldx * 2 >= (int64_t(1) << 31)) return 0;
    if ((int64_t)d->Hout * d->Wout * d->ldy * 2 >= (int64_t(1) << 31)) return 0;     // (per-image output offsets are 32-bit)
    return 1;
}

}  // namespace

extern "C" int abc_convt_fused_ok(const abc_convt_desc* d) { return ct_check(d); }

extern "C" int abc_convt_fused_fwd(const abc_convt_desc* d, abc_stream_t stream) {
    if (!ct_check(d)) return abc_fail(ABC_EUNSUPPORTED, "convt_fused: bf16, both axes cropped, Cin a multiple of 32, Cout_pad of 64 (abc_convt_fused_ok)");
    CTK k;
    k.x = d->src.x; k.scale = d->src.scale; k.shift = d->src.shift; k.slope = d->src.slope;
    k.w = d->w; k.bias = d->bias; k.y = d->y;
    k.B = d->B; k.Hin = d->Hin; k.Win = d->Win; k.Hx = d->src.Hx; k.Wx = d->src.Wx; k.ldx = d->src.ldx; k.cin_off = d->cin_off; k.Cin = d->Cin;
    k.nchunks = d->Cin / CK;
    k.Hout = d->Hout; k.Wout = d->Wout; k.ldy = d->ldy; k.cout_off = d->cout_off; k.Cout = d->Cout; k.Cout_pad = d->Cout_pad;
    k.tiles_x = abc_cdiv(d->Win, 16); k.tiles_y = abc_cdiv(d->Hin, TROWS); k.nbn = d->Cout_pad / 64;
    k.ntiles = k.nbn * k.tiles_x * k.tiles_y * d->B;
    k.cstride = abc_roundup(d->Cin, 4);
    k.bytesA = (unsigned)((int64_t)d->B * d->src.Hx * d->src.Wx * d->src.ldx * 2);
    k.bytesW = (unsigned)((int64_t)9 * k.nchunks * d->Cout_pad * CK * 2);
    const int lds = 2 * SA + abc_roundup(3 * k.cstride * 4, 256);
    static unsigned long long lds_ok = 0;
    if (int rc = abc_allow_lds((const void*)convt_fused_kernel, 64 * 1024, &lds_ok)) return rc;
    hipLaunchKernelGGL(convt_fused_kernel, dim3(k.ntiles), dim3(256), lds, (hipStream_t)stream, k);
    return abc_check_launch("convt_fused_fwd");
}
