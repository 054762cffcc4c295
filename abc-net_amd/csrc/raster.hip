// Target rasteriser on the device: src/utils.py:83-228 (MolecularImageDataset.__getitem__) from compact records.
//
// The reference builds the 8 target maps of a molecule on the host (22.9 MB per image at 128 x 128, almost all zeros)
// and ships them through the DataLoader and PCIe every step (376 MB per batch-16 step).  Here the host parses the
// annotation strings into a few hundred bytes of records (abcnet_amd/raster.py: the string handling, vocabulary
// look-ups and the atan stay in Python, bit-identical to the reference's float64 arithmetic) and the maps are
// rasterised where they are consumed:
//   zero pass   : the 8 maps, 16-byte stores
//   raster pass : ONE WAVE PER IMAGE walks the atoms, then the bonds, IN ORDER -- the reference's slice assignments are
//                 order dependent (a later item's 3x3 ring overwrites an earlier item's centre) -- with the lanes spread
//                 over the <= 3 bins x 3 x 3 pixels an item touches.  Every pixel an item touches is written once with
//                 its final value (centre 1, ring 0.8 / 0.5), and a fence separates consecutive items, so the result is
//                 the reference's, bit for bit.
// Record formats: atoms[b][i] = (x, y, type, charge, hs) with hs in {0, 1} or -1; bonds[b][i] = (x, y, type, omega bin,
// single) where single = 1 for stereo bonds (types 4, 5: one direction, utils.py:165-185) and 0 for the rest (both bin k
// and k + 30, utils.py:187-221); rho[b][i] float64.  x = row, y = column; 0 <= x < h, 0 <= y < w.
#include "common.hpp"
#include "../../include/abcnet_hip.h"
#include "capi_util.hpp"

namespace {

__global__ __launch_bounds__(256) void raster_zero_kernel(f32x4* p, int64_t n16) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
}

// what the maps hold of ONE item, written (val = true: the reference's values) or erased (false: zeros, the incremental form's first phase)
template <bool DRAW>
__device__ inline void put_atom(const abc_raster_desc& d, int b, int lane, const int* a, uint32_t* flags) {
    const int h = d.h, w = d.w;
    const size_t hw = (size_t)h * w;
    const int x = a[0], y = a[1], ty = a[2], ch = a[3], hsv = a[4];
    const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
    const int px = xb + (lane % 9) / 3, py = yb + lane % 3;
    if (lane < 9 && px < min(x + 2, h) && py < min(y + 2, w)) {
        const bool c = px == x && py == y;
        const size_t o = (size_t)px * w + py;
        d.t_atom[(size_t)b * hw + o] = DRAW ? (c ? 1.f : 0.8f) : 0.f;
        d.t_types[((size_t)b * 14 + ty) * hw + o] = DRAW ? (c ? 1.f : 0.5f) : 0.f;
        d.t_charges[((size_t)b * 3 + ch) * hw + o] = DRAW ? (c ? 1.f : 0.5f) : 0.f;
        if (hsv == 0 || hsv == 1) d.t_hs[((size_t)b * 2 + hsv) * hw + o] = DRAW ? (c ? 1.f : 0.5f) : 0.f;
        if (DRAW && flags) atomicOr(flags + ((size_t)b * hw + o) / 32, 0x0Fu);
    }
}
template <bool DRAW>
__device__ inline void put_bond(const abc_raster_desc& d, int b, int lane, const int* q, double rho, uint32_t* flags) {
    const int h = d.h, w = d.w;
    const size_t hw = (size_t)h * w;
    const int x = q[0], y = q[1], ty = q[2], k0 = q[3], single = q[4];
    const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
    const int px = xb + (lane % 9) / 3, py = yb + lane % 3, slot = lane / 9;
    const bool inbox = px < min(x + 2, h) && py < min(y + 2, w);
    const bool c = px == x && py == y;
    const size_t o = (size_t)px * w + py;
    if (lane < 9 && inbox) {
        d.t_bond[(size_t)b * hw + o] = DRAW ? (c ? 1.f : 0.8f) : 0.f;
        if (DRAW && flags) atomicOr(flags + ((size_t)b * hw + o) / 32, 0xF0u);
    }
    // one or two directions; the bins of the two directions never overlap (30 apart, 3 wide), so they share a round
    for (int dir = 0; dir < (single ? 1 : 2); ++dir) {
        const int k = k0 + 30 * dir;
        const bool wrap_lo = single || dir == 0, wrap_hi = single || dir == 1;   // utils.py:179-185, 201-204, 218-221
        const int kb = k == 0 ? 0 : k - 1;
        int bin = -1;
        if (slot < 3) { if (kb + slot <= min(k + 1, 59)) bin = kb + slot; }
        else if (slot == 3) { if (wrap_lo && k == 0) bin = 59; else if (wrap_hi && k == 59) bin = 0; }
        if (bin >= 0 && inbox && lane < 36) {
            const bool cc = c && bin == k;
            d.t_rho[((size_t)b * 60 + bin) * hw + o] = DRAW ? rho : 0.0;
            d.t_omega[((size_t)b * 60 + bin) * hw + o] = DRAW ? (cc ? 1.0 : 0.8) : 0.0;
            d.t_btypes[((size_t)b * 360 + (size_t)ty * 60 + bin) * hw + o] = DRAW ? (cc ? 1.f : 0.5f) : 0.f;
        }
    }
}

// The sparse form (abc_raster_desc.group_flags / prev_* / incremental): one wave per image erases what the PREVIOUS records drew (the maps
// are zero everywhere else), clears the image's group flags, draws the new records in order exactly as raster_kernel does, raises the
// flags of the 32-pixel groups it touched and keeps the new records for the next call.
__global__ __launch_bounds__(64) void raster_sparse_kernel(const abc_raster_desc d) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t hw = (size_t)d.h * d.w;
    uint32_t* flags = d.group_flags + (size_t)b * (hw / 32);
    if (d.incremental) {
        const int pa = min(d.prev_counts[b], d.max_atoms), pb = min(d.prev_counts[d.B + b], d.max_bonds);
        for (int i = 0; i < pa; ++i) put_atom<false>(d, b, lane, d.prev_atoms + ((size_t)b * d.max_atoms + i) * 5, nullptr);
        for (int i = 0; i < pb; ++i) put_bond<false>(d, b, lane, d.prev_bonds + ((size_t)b * d.max_bonds + i) * 5, 0.0, nullptr);
    }
    for (size_t i = lane; i < hw / 32; i += 64) flags[i] = 0u;
    __threadfence();      // zeros first: the drawing below may land on the same pixels
    const int na = min(d.n_atoms[b], d.max_atoms), nb = min(d.n_bonds[b], d.max_bonds);
    for (int i = 0; i < na; ++i) {
        put_atom<true>(d, b, lane, d.atoms + ((size_t)b * d.max_atoms + i) * 5, d.group_flags);
        __threadfence();   // the next item may overwrite these pixels: keep the reference's order
    }
    for (int i = 0; i < nb; ++i) {
        put_bond<true>(d, b, lane, d.bonds + ((size_t)b * d.max_bonds + i) * 5, d.rho[(size_t)b * d.max_bonds + i], d.group_flags);
        __threadfence();
    }
    // the records the maps now hold
    for (int i = lane; i < na * 5; i += 64) d.prev_atoms[(size_t)b * d.max_atoms * 5 + i] = d.atoms[(size_t)b * d.max_atoms * 5 + i];
    for (int i = lane; i < nb * 5; i += 64) d.prev_bonds[(size_t)b * d.max_bonds * 5 + i] = d.bonds[(size_t)b * d.max_bonds * 5 + i];
    for (int i = lane; i < nb; i += 64) d.prev_rho[(size_t)b * d.max_bonds + i] = d.rho[(size_t)b * d.max_bonds + i];
    if (lane == 0) { d.prev_counts[b] = na; d.prev_counts[d.B + b] = nb; }
}

__global__ __launch_bounds__(64) void raster_kernel(const abc_raster_desc d) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int h = d.h, w = d.w;
    const size_t hw = (size_t)h * w;
    float* atom_t = d.t_atom + (size_t)b * hw;
    float* types = d.t_types + (size_t)b * 14 * hw;
    float* charges = d.t_charges + (size_t)b * 3 * hw;
    float* hs_map = d.t_hs + (size_t)b * 2 * hw;
    float* bond_t = d.t_bond + (size_t)b * hw;
    float* btypes = d.t_btypes + (size_t)b * 360 * hw;
    double* rho_map = d.t_rho + (size_t)b * 60 * hw;
    double* om_map = d.t_omega + (size_t)b * 60 * hw;
    const int na = min(d.n_atoms[b], d.max_atoms), nb = min(d.n_bonds[b], d.max_bonds);
    // this lane's pixel of the 3x3 box and its bin slot (0..2 = bins kb..kb+2, 3 = the wrap-around bin)
    const int bi = (lane % 9) / 3, bj = lane % 3, slot = lane / 9;

    for (int i = 0; i < na; ++i) {
        const int* a = d.atoms + ((size_t)b * d.max_atoms + i) * 5;
        const int x = a[0], y = a[1], ty = a[2], ch = a[3], hsv = a[4];
        const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
        const int px = xb + bi, py = yb + bj;
        if (lane < 9 && px < min(x + 2, h) && py < min(y + 2, w)) {
            const bool c = px == x && py == y;
            const size_t o = (size_t)px * w + py;
            atom_t[o] = c ? 1.f : 0.8f;
            types[(size_t)ty * hw + o] = c ? 1.f : 0.5f;
            charges[(size_t)ch * hw + o] = c ? 1.f : 0.5f;
            if (hsv == 0 || hsv == 1) hs_map[(size_t)hsv * hw + o] = c ? 1.f : 0.5f;
        }
        __threadfence();   // the next item may overwrite these pixels: keep the reference's order
    }
    for (int i = 0; i < nb; ++i) {
        const int* q = d.bonds + ((size_t)b * d.max_bonds + i) * 5;
        const int x = q[0], y = q[1], ty = q[2], k0 = q[3], single = q[4];
        const double rho = d.rho[(size_t)b * d.max_bonds + i];
        const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
        const int px = xb + bi, py = yb + bj;
        const bool inbox = px < min(x + 2, h) && py < min(y + 2, w);
        const bool c = px == x && py == y;
        const size_t o = (size_t)px * w + py;
        if (lane < 9 && inbox) bond_t[o] = c ? 1.f : 0.8f;
        // one or two directions; the bins of the two directions never overlap (30 apart, 3 wide), so they share a round
        for (int dir = 0; dir < (single ? 1 : 2); ++dir) {
            const int k = k0 + 30 * dir;
            const bool wrap_lo = single || dir == 0, wrap_hi = single || dir == 1;   // utils.py:179-185, 201-204, 218-221
            const int kb = k == 0 ? 0 : k - 1;
            int bin = -1;
            if (slot < 3) { if (kb + slot <= min(k + 1, 59)) bin = kb + slot; }
            else if (slot == 3) { if (wrap_lo && k == 0) bin = 59; else if (wrap_hi && k == 59) bin = 0; }
            if (bin >= 0 && inbox && lane < 36) {
                const bool cc = c && bin == k;
                rho_map[(size_t)bin * hw + o] = rho;
                om_map[(size_t)bin * hw + o] = cc ? 1.0 : 0.8;
                btypes[((size_t)ty * 60 + bin) * hw + o] = cc ? 1.f : 0.5f;
            }
        }
        __threadfence();
    }
}

}  // namespace

extern "C" int abc_rasterize_targets(const abc_raster_desc* d, abc_stream_t stream) {
    if (d->B < 1 || d->h < 1 || d->w < 1 || d->max_atoms < 0 || d->max_bonds < 0) return abc_fail(ABC_EINVAL, "raster: dims");
    if (((size_t)d->h * d->w) % 4) return abc_fail(ABC_EUNSUPPORTED, "raster: h * w must be a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const size_t hw = (size_t)d->h * d->w;
    const bool sparse = d->group_flags != nullptr;
    if (sparse && (!d->prev_atoms || !d->prev_bonds || !d->prev_rho || !d->prev_counts || hw % 32))
        return abc_fail(ABC_EINVAL, "raster: group_flags needs the prev_* buffers and h * w a multiple of 32");
    struct { void* p; size_t bytes; } maps[8] = {
        {d->t_atom, hw * 4}, {d->t_types, 14 * hw * 4}, {d->t_charges, 3 * hw * 4}, {d->t_hs, 2 * hw * 4}, {d->t_bond, hw * 4},
        {d->t_btypes, 360 * hw * 4}, {d->t_rho, 60 * hw * 8}, {d->t_omega, 60 * hw * 8}};
    for (int i = 0; i < 8; ++i) {
        if (!maps[i].p) return abc_fail(ABC_EINVAL, "raster: null map");
        if (sparse && d->incremental) continue;      // (the sparse kernel erases what the previous records drew)
        const int64_t n16 = (int64_t)(maps[i].bytes * d->B / 16);
        int64_t nb = (n16 + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(raster_zero_kernel, dim3((int)nb), dim3(256), 0, st, (f32x4*)maps[i].p, n16);
    }
    if (sparse) {
        abc_raster_desc k = *d;
        // (the first call: nothing to erase -- the zero pass above cleared the maps)
        hipLaunchKernelGGL(raster_sparse_kernel, dim3(d->B), dim3(64), 0, st, k);
    } else {
        hipLaunchKernelGGL(raster_kernel, dim3(d->B), dim3(64), 0, st, *d);
    }
    return abc_check_launch("rasterize_targets");
}
