// Target rasteriser on the device: src/utils.py:83-228 (MolecularImageDataset.__getitem__) from compact records.
//
// The reference builds the 8 target maps of a molecule on the host (22.9 MB per image at 128 x 128, almost all zeros)
// and ships them through the DataLoader and PCIe every step (376 MB per batch-16 step).  Here the host parses the
// annotation strings into a few hundred bytes of records (abcnet_amd/raster.py: the string handling, vocabulary
// look-ups and the atan stay in Python, bit-identical to the reference's float64 arithmetic) and the maps are
// rasterised where they are consumed:
//   zero pass   : the 8 maps, 16-byte stores
//   raster pass : one workgroup per image; a unit of work = (item, pixel of its 3x3 box, bin slot).  The reference's slice
//                 assignments are order dependent (a later item's 3x3 ring overwrites an earlier item's centre): a unit writes
//                 a cell only if no LATER item writes it (records compared in LDS), so every cell is written once with the
//                 value the reference's in-order walk leaves (centre 1, ring 0.8 / 0.5) -- bit for bit, without a fence per item.
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

// ---- the items of ONE image, in parallel but with the reference's order semantics
//
// The reference's slice assignments are order dependent (a later item's 3x3 ring overwrites an earlier item's centre).  Walking the
// items one by one with a memory fence between them (the round-1 form: one wave per image) costs a store round trip per item: 85 us for
// 62 items.  Here every (item, pixel, bin slot) of an image is one unit of work of a 256-thread workgroup, and a unit writes a cell
// (plane, pixel) only if NO LATER item writes the same cell -- "last writer wins" decided by comparing records (n <= a few hundred
// items, in LDS), not by ordering stores.  Every cell is therefore written exactly once, with the value the sequential walk leaves.
struct ImgRec {
    const int* atoms;   // [na][5] in LDS
    const int* bonds;   // [nb][5]
    const double* rho;  // [nb]
    int na, nb;
};

__device__ inline bool in_box(int x, int y, int px, int py, int h, int w) {
    const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
    return px >= xb && px < min(x + 2, h) && py >= yb && py < min(y + 2, w);
}
// does a bond (first bin k0, one direction or two) write omega bin `bin`?  (utils.py:165-221: bins k-1 .. k+1 clipped to 0 .. 59, the
// wrap-around bin of the direction that owns it, both directions 30 bins apart for the non-stereo bonds)
__device__ inline bool bond_has_bin(int k0, int single, int bin) {
    for (int dir = 0; dir < (single ? 1 : 2); ++dir) {
        const int k = k0 + 30 * dir;
        const bool wrap_lo = single || dir == 0, wrap_hi = single || dir == 1;
        if (bin >= (k == 0 ? 0 : k - 1) && bin <= min(k + 1, 59)) return true;
        if ((wrap_lo && k == 0 && bin == 59) || (wrap_hi && k == 59 && bin == 0)) return true;
    }
    return false;
}

// unit `sub` (0 .. 8: the pixel of the 3x3 box) of atom i.  DRAW: the reference's values where no later atom overwrites; else zeros
// everywhere the atom drew (the incremental form's first phase: order does not matter for zeros)
template <bool DRAW>
__device__ inline void put_atom(const abc_raster_desc& d, int b, const ImgRec& R, int i, int sub, uint32_t* flags) {
    const int h = d.h, w = d.w;
    const size_t hw = (size_t)h * w;
    const int* a = R.atoms + 5 * i;
    const int x = a[0], y = a[1], ty = a[2], ch = a[3], hsv = a[4];
    const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
    const int px = xb + sub / 3, py = yb + sub % 3;
    if (!(px < min(x + 2, h) && py < min(y + 2, w))) return;
    bool o_atom = false, o_ty = false, o_ch = false, o_hs = false;
    if (DRAW) {
        for (int j = i + 1; j < R.na; ++j) {
            const int* q = R.atoms + 5 * j;
            if (in_box(q[0], q[1], px, py, h, w)) {
                o_atom = true;
                o_ty |= q[2] == ty; o_ch |= q[3] == ch; o_hs |= q[4] == hsv;
            }
        }
    }
    const bool c = px == x && py == y;
    const size_t o = (size_t)px * w + py;
    if (!o_atom) d.t_atom[(size_t)b * hw + o] = DRAW ? (c ? 1.f : 0.8f) : 0.f;
    if (!o_ty) d.t_types[((size_t)b * 14 + ty) * hw + o] = DRAW ? (c ? 1.f : 0.5f) : 0.f;
    if (!o_ch) d.t_charges[((size_t)b * 3 + ch) * hw + o] = DRAW ? (c ? 1.f : 0.5f) : 0.f;
    if ((hsv == 0 || hsv == 1) && !o_hs) d.t_hs[((size_t)b * 2 + hsv) * hw + o] = DRAW ? (c ? 1.f : 0.5f) : 0.f;
    if (DRAW && flags) atomicOr(flags + ((size_t)b * hw + o) / 32, 0x0Fu);
}
// unit `sub` (0 .. 35: pixel sub % 9 of the box, bin slot sub / 9: 0 .. 2 = bins kb .. kb + 2, 3 = the wrap-around bin) of bond i
template <bool DRAW>
__device__ inline void put_bond(const abc_raster_desc& d, int b, const ImgRec& R, int i, int sub, uint32_t* flags) {
    const int h = d.h, w = d.w;
    const size_t hw = (size_t)h * w;
    const int* q = R.bonds + 5 * i;
    const int x = q[0], y = q[1], ty = q[2], k0 = q[3], single = q[4];
    const double rho = DRAW ? R.rho[i] : 0.0;
    const int xb = x > 0 ? x - 1 : 0, yb = y > 0 ? y - 1 : 0;
    const int px = xb + (sub % 9) / 3, py = yb + sub % 3, slot = sub / 9;
    if (!(px < min(x + 2, h) && py < min(y + 2, w))) return;
    const bool c = px == x && py == y;
    const size_t o = (size_t)px * w + py;
    if (slot == 0) {
        bool over = false;
        if (DRAW)
            for (int j = i + 1; j < R.nb && !over; ++j) over = in_box(R.bonds[5 * j], R.bonds[5 * j + 1], px, py, h, w);
        if (!over) d.t_bond[(size_t)b * hw + o] = DRAW ? (c ? 1.f : 0.8f) : 0.f;
        if (DRAW && flags) atomicOr(flags + ((size_t)b * hw + o) / 32, 0xF0u);
    }
    // one or two directions; the bins of the two directions never overlap (30 apart, 3 wide)
    for (int dir = 0; dir < (single ? 1 : 2); ++dir) {
        const int k = k0 + 30 * dir;
        const bool wrap_lo = single || dir == 0, wrap_hi = single || dir == 1;   // utils.py:179-185, 201-204, 218-221
        const int kb = k == 0 ? 0 : k - 1;
        int bin = -1;
        if (slot < 3) { if (kb + slot <= min(k + 1, 59)) bin = kb + slot; }
        else { if (wrap_lo && k == 0) bin = 59; else if (wrap_hi && k == 59) bin = 0; }
        if (bin < 0) continue;
        bool o_bin = false, o_ty = false;
        if (DRAW) {
            for (int j = i + 1; j < R.nb; ++j) {
                const int* u = R.bonds + 5 * j;
                if (in_box(u[0], u[1], px, py, h, w) && bond_has_bin(u[3], u[4], bin)) { o_bin = true; o_ty |= u[2] == ty; }
            }
        }
        const bool cc = c && bin == k;
        if (!o_bin) {
            d.t_rho[((size_t)b * 60 + bin) * hw + o] = rho;
            d.t_omega[((size_t)b * 60 + bin) * hw + o] = DRAW ? (cc ? 1.0 : 0.8) : 0.0;
        }
        if (!o_ty) d.t_btypes[((size_t)b * 360 + (size_t)ty * 60 + bin) * hw + o] = DRAW ? (cc ? 1.f : 0.5f) : 0.f;
    }
}

constexpr int RTHR = 256;

// records of image b from global memory into LDS ([max_atoms][5] | [max_bonds][5] | rho[max_bonds], rho 8-byte aligned)
__device__ inline ImgRec load_records(char* smem, const int* atoms, const int* bonds, const double* rho, int na, int nb, int max_atoms, int max_bonds, int b) {
    int* sa = (int*)smem;
    int* sb = sa + 5 * max_atoms;
    double* sr = (double*)(smem + (((size_t)5 * (max_atoms + max_bonds) * 4 + 7) & ~(size_t)7));
    for (int i = threadIdx.x; i < 5 * na; i += RTHR) sa[i] = atoms[(size_t)b * max_atoms * 5 + i];
    for (int i = threadIdx.x; i < 5 * nb; i += RTHR) sb[i] = bonds[(size_t)b * max_bonds * 5 + i];
    for (int i = threadIdx.x; i < nb; i += RTHR) sr[i] = rho[(size_t)b * max_bonds + i];
    __syncthreads();
    return ImgRec{sa, sb, sr, na, nb};
}
template <bool DRAW>
__device__ inline void put_image(const abc_raster_desc& d, int b, const ImgRec& R, uint32_t* flags) {
    for (int u = threadIdx.x; u < 9 * R.na; u += RTHR) put_atom<DRAW>(d, b, R, u / 9, u % 9, flags);
    for (int u = threadIdx.x; u < 36 * R.nb; u += RTHR) put_bond<DRAW>(d, b, R, u / 36, u % 36, flags);
}

// The sparse form (abc_raster_desc.group_flags / prev_* / incremental): one workgroup per image erases what the PREVIOUS records drew
// (the maps are zero everywhere else), clears the image's group flags, draws the new records, raises the flags of the 32-pixel groups
// it touched and keeps the new records for the next call.
__global__ __launch_bounds__(RTHR) void raster_sparse_kernel(const abc_raster_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const size_t hw = (size_t)d.h * d.w;
    uint32_t* flags = d.group_flags + (size_t)b * (hw / 32);
    if (d.incremental) {
        const int pa = min(d.prev_counts[b], d.max_atoms), pb = min(d.prev_counts[d.B + b], d.max_bonds);
        const ImgRec P = load_records(smem, d.prev_atoms, d.prev_bonds, d.prev_rho, pa, pb, d.max_atoms, d.max_bonds, b);
        put_image<false>(d, b, P, nullptr);
    }
    for (size_t i = threadIdx.x; i < hw / 32; i += RTHR) flags[i] = 0u;
    __threadfence();
    __syncthreads();      // zeros (and the cleared flags) first: the drawing below may land on the same pixels
    const int na = min(d.n_atoms[b], d.max_atoms), nb = min(d.n_bonds[b], d.max_bonds);
    const ImgRec R = load_records(smem, d.atoms, d.bonds, d.rho, na, nb, d.max_atoms, d.max_bonds, b);
    put_image<true>(d, b, R, d.group_flags);
    // the records the maps now hold
    for (int i = threadIdx.x; i < na * 5; i += RTHR) d.prev_atoms[(size_t)b * d.max_atoms * 5 + i] = R.atoms[i];
    for (int i = threadIdx.x; i < nb * 5; i += RTHR) d.prev_bonds[(size_t)b * d.max_bonds * 5 + i] = R.bonds[i];
    for (int i = threadIdx.x; i < nb; i += RTHR) d.prev_rho[(size_t)b * d.max_bonds + i] = R.rho[i];
    if (threadIdx.x == 0) { d.prev_counts[b] = na; d.prev_counts[d.B + b] = nb; }
}

// the dense form: the maps were zeroed by raster_zero_kernel launches in front (stream order)
__global__ __launch_bounds__(RTHR) void raster_kernel(const abc_raster_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int na = min(d.n_atoms[b], d.max_atoms), nb = min(d.n_bonds[b], d.max_bonds);
    const ImgRec R = load_records(smem, d.atoms, d.bonds, d.rho, na, nb, d.max_atoms, d.max_bonds, b);
    put_image<true>(d, b, R, nullptr);
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
    const size_t lds = (((size_t)5 * (d->max_atoms + d->max_bonds) * 4 + 7) & ~(size_t)7) + (size_t)d->max_bonds * 8;
    if (lds > 60000) return abc_fail(ABC_EUNSUPPORTED, "raster: max_atoms + max_bonds above ~2000");
    if (sparse) {
        abc_raster_desc k = *d;
        // (the first call: nothing to erase -- the zero pass above cleared the maps)
        hipLaunchKernelGGL(raster_sparse_kernel, dim3(d->B), dim3(RTHR), lds, st, k);
    } else {
        hipLaunchKernelGGL(raster_kernel, dim3(d->B), dim3(RTHR), lds, st, *d);
    }
    return abc_check_launch("rasterize_targets");
}
