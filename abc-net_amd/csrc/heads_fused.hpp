// Row order of the fused heads kernels (heads_fused.hip, and the blocked weight gradient in wgrad.hip).
//
// The 1x1 convolution's output rows are PACKED per head so that the MFMA accumulator layout (a lane = one pixel; register k
// of lane half h = row (k & 3) + 8 (k >> 2) + 4 h of a 32-row tile) keeps everything a loss term needs inside one lane:
//   bond types (360 = 6 types x 60 omega bins, channel = type * 60 + bin): registers 8 gi .. 8 gi + 5 of tile mt = the six
//       types of bin 30 h + 2 mt + gi (two padding rows per group: 15 tiles);
//   rho / omega (60 bins): register k of tile mt = bin 30 h + 16 mt + k (the same half as the bond-type group of that bin);
//   the small heads: all channels in half 0 of one tile.
#pragma once
#include <stdint.h>

constexpr int HF_NH = 8;
constexpr int HF_GROUPS = 7;   // work types of the fused kernel: {bond types, rho}, omega, and the five small heads one by one
__host__ __device__ constexpr int hf_ch(int head) { return head == 0 ? 1 : head == 1 ? 14 : head == 2 ? 3 : head == 3 ? 2 : head == 4 ? 1 : head == 5 ? 360 : 60; }
__host__ __device__ constexpr int hf_tiles(int head) { return head == 5 ? 15 : (head >= 6 ? 2 : 1); }
// channel of packed row m of a head (-1: padding)
__host__ __device__ inline int hf_chan_of_row(int head, int m) {
    const int h = (m >> 2) & 1, k = (m & 3) + 4 * ((m >> 3) & 3), mt = m >> 5;
    if (head == 5) { const int kk = k & 7; return kk < 6 ? kk * 60 + 30 * h + 2 * mt + (k >> 3) : -1; }
    if (head >= 6) { const int j = 16 * mt + k; return j < 30 ? 30 * h + j : -1; }
    return (h == 0 && mt == 0 && k < hf_ch(head)) ? k : -1;
}
// packed row of a head's channel (inverse of hf_chan_of_row)
__host__ __device__ inline int hf_row_of_chan(int head, int ch) {
    int h = 0, mt = 0, k = ch;
    if (head == 5) { const int kk = ch / 60, o = ch % 60, g = o % 30; h = o / 30; mt = g >> 1; k = 8 * (g & 1) + kk; }
    else if (head >= 6) { const int j = ch % 30; h = ch / 30; mt = j >> 4; k = j & 15; }
    return 32 * mt + (k & 3) + 8 * (k >> 2) + 4 * h;
}
// the five small heads' conv2 gradients leave the fused kernel as per-chunk partial rows: head i's channels start at row
constexpr int HF_SMALL_ROWS = 1 + 14 + 3 + 2 + 1;
__host__ __device__ constexpr int hf_small_row0(int head) { return head == 0 ? 0 : head == 1 ? 1 : head == 2 ? 15 : head == 3 ? 18 : 20; }
// byte offset of a head's block in the packed-weights workspace: [forward Cpad x 128 bf16][data-gradient Cpad x 128 bf16][bias Cpad f32]
__host__ __device__ inline int64_t hf_pack_off(int head) {
    int64_t off = 0;
    for (int i = 0; i < head; ++i) off += (int64_t)hf_tiles(i) * 32 * (512 + 4);
    return off;
}
__host__ __device__ inline int hf_rows_total() {
    int n = 0;
    for (int i = 0; i < HF_NH; ++i) n += hf_tiles(i) * 32;
    return n;
}
