// Candidate extraction for the SMILES decoder: img2smiles2.py:113-191 on the device.
//
// The reference walks the NMS masks with per-pixel `.cpu().item()` calls (hundreds of host round trips per image).
// Here one workgroup per image turns the head maps into two compact, ORDERED lists -- the wire format into the
// unchanged CPU graph-assembly / RDKit stage (img2smiles2.py:193-344):
//   atoms : (x, y, type, charge, hs)           raster order, greedy suppression of peaks within squared distance < 4
//                                              of an already accepted atom (img2smiles2.py:171-191)
//   bonds : (x, y, omega bin, type) + |rho|    raster order of the bond peaks, bins ascending; a bin survives unless the
//                                              opposite direction wins (img2smiles2.py:128-169; the rule is applied to
//                                              every bin whose RAW omega logit is non-zero, as the reference does)
// x = row, y = column, as in the reference.  Order and content are bit-exact with the reference lists (integer work;
// |rho| is the f32 the reference reads with .item()).
//
// Phases (1024 threads = 16 waves): (A) ordered compaction of both peak masks by block-wide prefix sums over 1024-pixel
// chunks; (B) wave 0 runs the sequential greedy suppression over the compacted atom peaks, all waves then fill in the
// arg-max classes; (C) one wave per bond peak evaluates the 60 bins (lane = bin, opposite bins by shuffles), a block
// scan of the per-peak counts gives every peak its output range, a second sweep writes the candidates.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

constexpr int XT = 1024;          // threads per workgroup
constexpr int MAX_BPEAKS = 4096;  // bond peaks per image held for phase C (more are counted, not expanded)

// exclusive prefix sum of v over the workgroup (v may pack two 16-bit counters); wt = LDS scratch [XT / 64 + 1]
__device__ inline unsigned block_excl_scan(unsigned v, unsigned* wt, unsigned* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();  // wt may still be read from the previous call
    if (lane == 63) wt[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const unsigned x = lane < XT / 64 ? wt[lane] : 0u;
        unsigned s = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(s, o);
            if (lane >= o) s += t;
        }
        if (lane < XT / 64) wt[lane] = s - x;
        if (lane == XT / 64 - 1) wt[XT / 64] = s;
    }
    __syncthreads();
    *total = wt[XT / 64];
    return wt[wave] + inc - v;
}

template <int K>
__device__ inline int argmax_plane(const float* p, size_t stride) {  // first maximum, as torch.argmax
    int best = 0;
    float bv = p[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
        const float v = p[(size_t)k * stride];
        if (v > bv) { bv = v; best = k; }
    }
    return best;
}

__global__ __launch_bounds__(XT) void extract_kernel(const abc_extract_desc d) {
    __shared__ unsigned wt[XT / 64 + 1];
    __shared__ int cnt[MAX_BPEAKS];
    __shared__ int acc_xy[2048];   // accepted atoms (x << 16 | y), cap_atoms <= 2048
    __shared__ int n_acc_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = d.h * d.w;
    const float* am = d.atom_mask + (size_t)b * hw;
    const float* bm = d.bond_mask + (size_t)b * hw;
    int* atom_px = d.work + (size_t)b * (d.cap_atoms + MAX_BPEAKS);
    int* bond_px = atom_px + d.cap_atoms;
    unsigned long long* masks = (unsigned long long*)d.work_masks + (size_t)b * MAX_BPEAKS;

    // ---- (A) ordered compaction of the two peak masks
    int na = 0, nb = 0;   // running totals (per-chunk counts travel packed, 16 bits each, through one scan)
    for (int p0 = 0; p0 < hw; p0 += XT) {
        const int p = p0 + tid;
        const bool fa = p < hw && am[p] != 0.f, fb = p < hw && bm[p] != 0.f;
        unsigned tot;
        const unsigned ex = block_excl_scan((fa ? 1u : 0u) | (fb ? 0x10000u : 0u), wt, &tot);
        const int ia = na + (int)(ex & 0xFFFFu), ib = nb + (int)(ex >> 16);
        if (fa && ia < d.cap_atoms) atom_px[ia] = p;
        if (fb && ib < MAX_BPEAKS) bond_px[ib] = p;
        na += (int)(tot & 0xFFFFu);
        nb += (int)(tot >> 16);
    }
    __syncthreads();   // lists visible to the whole workgroup (global writes by this workgroup, read back below)
    __threadfence_block();
    const int na_l = min(na, d.cap_atoms), nb_l = min(nb, MAX_BPEAKS);

    // ---- (B) greedy suppression, sequential in raster order (wave 0), lanes over the accepted list
    if (wave == 0) {
        int nacc = 0;
        for (int i = 0; i < na_l; ++i) {
            const int p = atom_px[i];
            const int x = p / d.w, y = p - x * d.w;
            bool close = false;
            for (int j0 = 0; j0 < nacc; j0 += 64) {
                const int j = j0 + lane;
                if (j < nacc) {
                    const int q = acc_xy[j];
                    const int dx = x - (q >> 16), dy = y - (q & 0xFFFF);
                    close |= dx * dx + dy * dy < 4;
                }
            }
            if (__ballot(close) == 0ull) {
                if (lane == 0) acc_xy[nacc] = (x << 16) | y;
                ++nacc;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
        if (lane == 0) n_acc_s = nacc;
    }
    __syncthreads();
    const int nacc = n_acc_s;
    for (int i = tid; i < nacc; i += XT) {
        const int q = acc_xy[i];
        const int x = q >> 16, y = q & 0xFFFF;
        const size_t px = (size_t)x * d.w + y;
        int* o = d.atoms + ((size_t)b * d.cap_atoms + i) * 5;
        o[0] = x; o[1] = y;
        o[2] = argmax_plane<14>(d.types + (size_t)b * 14 * hw + px, hw);
        o[3] = argmax_plane<3>(d.charges + (size_t)b * 3 * hw + px, hw);
        o[4] = argmax_plane<2>(d.hs + (size_t)b * 2 * hw + px, hw);
    }

    // ---- (C) bonds: per peak, the surviving omega bins
    const float* om = d.omega + (size_t)b * 60 * hw;
    for (int i = wave; i < nb_l; i += XT / 64) {
        const int p = bond_px[i];
        const float v = lane < 60 ? om[(size_t)lane * hw + p] : 0.f;
        const int k = lane;
        // opposite bins (img2smiles2.py:141-157)
        int i1, i2;
        if (k <= 28) { i1 = k + 29; i2 = k + 30; }
        else if (k == 29) { i1 = 58; i2 = 0; }
        else if (k == 30) { i1 = 0; i2 = 59; }
        else { i1 = k - 31; i2 = k - 30; }
        const float o1 = __shfl(v, i1 & 63), o2 = __shfl(v, i2 & 63);
        const float mo = fmaxf(o1, o2);
        const bool drop = (k <= 29) ? (v < mo) : (v <= mo);
        const bool keep = k < 60 && v != 0.f && !drop;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) { masks[i] = m; cnt[i] = __popcll(m); }
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. nb_l): 4 consecutive entries per thread
    int c4[4], s4 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int i = tid * 4 + q; c4[q] = i < nb_l ? cnt[i] : 0; s4 += c4[q]; }
    unsigned tot_b;
    // (totals can exceed 16 bits: plain 32-bit scan, nothing packed)
    unsigned ex = block_excl_scan((unsigned)s4, wt, &tot_b);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int i = tid * 4 + q; if (i < nb_l) cnt[i] = (int)ex; ex += (unsigned)c4[q]; }
    __syncthreads();
    for (int i = wave; i < nb_l; i += XT / 64) {
        const unsigned long long m = masks[i];
        const int p = bond_px[i];
        const int x = p / d.w, y = p - x * d.w;
        if (lane < 60 && ((m >> lane) & 1ull)) {
            const int slot = cnt[i] + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < d.cap_bonds) {
                int* o = d.bonds + ((size_t)b * d.cap_bonds + slot) * 4;
                o[0] = x; o[1] = y; o[2] = lane;
                o[3] = d.btype_idx != nullptr ? (int)d.btype_idx[((size_t)b * 60 + lane) * hw + p]
                                              : argmax_plane<6>(d.btypes + ((size_t)b * 360 + lane) * hw + p, (size_t)60 * hw);
                d.bond_rho[(size_t)b * d.cap_bonds + slot] = fabsf(d.rho[((size_t)b * 60 + lane) * hw + p]);
            }
        }
    }
    if (tid == 0) {
        int* c = d.counts + (size_t)b * 4;
        c[0] = na; c[1] = nacc; c[2] = nb; c[3] = (int)tot_b;
    }
}

}  // namespace

extern "C" int64_t abc_extract_work_ints(const abc_extract_desc* d) { return (int64_t)d->B * (d->cap_atoms + MAX_BPEAKS); }
extern "C" int64_t abc_extract_work_masks(const abc_extract_desc* d) { return (int64_t)d->B * MAX_BPEAKS; }

extern "C" int abc_extract_peaks(const abc_extract_desc* d, abc_stream_t stream) {
    if (d->B < 1 || d->h < 1 || d->w < 1) return abc_fail(ABC_EINVAL, "extract: empty");
    if (d->cap_atoms < 1 || d->cap_atoms > 2048 || d->cap_bonds < 1) return abc_fail(ABC_EINVAL, "extract: cap_atoms must be 1..2048, cap_bonds >= 1");
    if (d->h >= 65536 || d->w >= 65536) return abc_fail(ABC_EUNSUPPORTED, "extract: map too large");
    if (!d->work || !d->work_masks || !d->counts || !d->atoms || !d->bonds || !d->bond_rho) return abc_fail(ABC_EINVAL, "extract: null buffer");
    hipLaunchKernelGGL(extract_kernel, dim3(d->B), dim3(XT), 0, (hipStream_t)stream, *d);
    return abc_check_launch("extract_peaks");
}
