"""ctypes binding of libabcnet_hip.so (the C ABI of include/abcnet_hip.h).

This is the reference-side stub in full: plain pointers and sizes go in, an int
status comes out.  There is NO CPU fallback: when the library is missing or a call
fails, an exception is raised (the product path must fail loudly).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libabcnet_hip.so")

F32, BF16, FP8 = 0, 1, 2
MAX_TAPS = 49

vp, i32, u32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64, C.c_float, C.c_double


class ActSrc(C.Structure):
    _fields_ = [("x", vp), ("scale", vp), ("shift", vp), ("slope", vp), ("Hx", i32), ("Wx", i32), ("ldx", i32),
                ("pool", i32), ("drop_p", f32), ("drop_seed", u32), ("planar", i32), ("ctot", i32), ("drop_salt", vp)]


class ConvDesc(C.Structure):
    _fields_ = [("src", ActSrc), ("w", vp), ("bias", vp), ("y", vp), ("stats", vp),
                ("dtype_in", i32), ("dtype_c", i32), ("dtype_out", i32),
                ("B", i32), ("Hin", i32), ("Win", i32), ("cin_off", i32), ("Cin", i32), ("Hg", i32), ("Wg", i32),
                ("Hout", i32), ("Wout", i32), ("ldy", i32), ("cout_off", i32), ("Cout", i32), ("Cout_pad", i32),
                ("stride", i32), ("om", i32), ("oy0", i32), ("ox0", i32), ("ntaps", i32),
                ("tap_dy", C.c_int8 * MAX_TAPS), ("tap_dx", C.c_int8 * MAX_TAPS), ("stats_rows", i32), ("accumulate", i32),
                ("planar_out", i32), ("ctot_out", i32), ("out_act", i32), ("out_slope", f32), ("pool_y", vp), ("ld_pool", i32),
                ("stem_x", vp), ("stem_w", vp), ("stem_scale", vp), ("stem_bias", vp), ("stem_slope", f32),
                ("out_scale", vp), ("out_quant", vp), ("out_quant_stride", i32), ("heads_epi", vp),
                ("actbwd_y", vp), ("actbwd_ld", i32), ("actbwd_coff", i32), ("actbwd_scale", vp), ("actbwd_shift", vp), ("actbwd_slope", vp),
                ("actbwd_mean", vp), ("actbwd_invstd", vp), ("head_aux", vp), ("head_aux_mode", i32)]


class HeadsEpi(C.Structure):
    _fields_ = [("w2", vp), ("bias", vp), ("oscale", vp), ("y", vp), ("Cout", i32), ("Cout_pad", i32)]


class PackDesc(C.Structure):
    _fields_ = [("w", vp), ("dst", vp), ("mode", i32), ("dtype_c", i32), ("Cout", i32), ("Cin", i32), ("kh", i32),
                ("kw", i32), ("py", i32), ("px", i32), ("rows_pad", i32), ("red_pad", i32), ("red_total", i32),
                ("red_off", i32), ("ck", i32), ("rows_total", i32), ("rows_off", i32), ("layout", i32), ("row_scale", vp)]


class BnFwdDesc(C.Structure):
    _fields_ = [("partial", vp), ("nblk", i32), ("C", i32), ("count", f64), ("rows", i32), ("gamma", vp), ("beta", vp),
                ("scale", vp), ("shift", vp), ("mean", vp), ("invstd", vp), ("running_mean", vp), ("running_var", vp),
                ("num_batches_tracked", vp), ("eps", f32), ("momentum", f32)]


class ActBwdDesc(C.Structure):
    _fields_ = [("y_raw", vp), ("ld_y", i32), ("dA_same", vp), ("ld_same", i32), ("dA_pool", vp), ("ld_pool", i32),
                ("g", vp), ("ld_g", i32), ("partial", vp), ("scale", vp), ("shift", vp), ("slope", vp), ("mean", vp),
                ("invstd", vp), ("dtype", i32), ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("cy_off", i32),
                ("csame_off", i32), ("cpool_off", i32), ("drop_p", f32), ("drop_seed", u32), ("drop_ld", i32), ("drop_salt", vp)]


class BnBwdDesc(C.Structure):
    _fields_ = [("partial", vp), ("nblk", i32), ("C", i32), ("count", f64), ("gamma", vp), ("invstd", vp),
                ("dgamma", vp), ("dbeta", vp), ("k1", vp), ("k2", vp), ("gscale", vp),
                ("mean", vp), ("ca", vp), ("cb", vp), ("cc", vp), ("in_scale", vp)]


class BnApplyDesc(C.Structure):
    _fields_ = [("g", vp), ("ld_g", i32), ("y_raw", vp), ("ld_y", i32), ("cy_off", i32), ("mean", vp), ("invstd", vp),
                ("k1", vp), ("k2", vp), ("gscale", vp), ("dtype", i32), ("C", i32), ("npix", i64), ("out", vp), ("ld_out", i32)]


class WgradDesc(C.Structure):
    _fields_ = [("p", ActSrc), ("q", ActSrc), ("partial", vp), ("dtype_p", i32), ("dtype_q", i32), ("dtype_c", i32),
                ("B", i32), ("Hg", i32), ("Wg", i32), ("Hq", i32), ("Wq", i32), ("cp_off", i32), ("Ca", i32),
                ("cq_off", i32), ("Cb", i32), ("stride", i32), ("ntaps", i32), ("nsplit", i32),
                ("tap_dy", C.c_int8 * MAX_TAPS), ("tap_dx", C.c_int8 * MAX_TAPS),
                ("p2", vp), ("ld_p2", i32), ("cp2_off", i32), ("p_dual", i32), ("p_out", vp), ("ld_pout", i32),
                ("rowsum_partial", vp)]


class WgradReduceDesc(C.Structure):
    _fields_ = [("partial", vp), ("nsplit", i32), ("ntaps", i32), ("Ca", i32), ("Cb", i32), ("Ca_pad", i32),
                ("Cb_pad", i32), ("dw", vp), ("accumulate", i32)]


class LossDesc(C.Structure):
    _fields_ = [("logits", vp * 8), ("dlogits", vp * 8), ("t_atom", vp), ("t_types", vp), ("t_charges", vp),
                ("t_hs", vp), ("t_bond", vp), ("t_btypes", vp), ("t_rho", vp), ("t_omega", vp), ("B", i32), ("h", i32),
                ("w", i32), ("partial", vp)]


class LossFinDesc(C.Structure):
    _fields_ = [("partial", vp), ("nblk", i32), ("s", vp), ("ds", vp), ("out", vp), ("chan_scale", vp), ("nchan", i32),
                ("chan_off", i32 * 8), ("head_c", i32 * 8), ("grad_scale", f32)]


class AdamDesc(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("step", vp), ("lr", f32), ("beta1", f32),
                ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("grad_scale", f32)]


class NmsDesc(C.Structure):
    _fields_ = [("atom", vp), ("bond", vp), ("rho", vp), ("omega", vp), ("B", i32), ("h", i32), ("w", i32),
                ("n_omega", i32), ("atom_mask", vp), ("bond_mask", vp), ("rho_abs", vp), ("omega_mask", vp)]


class CbamChannelDesc(C.Structure):
    _fields_ = [("partial", vp), ("tiles_per_img", i32), ("B", i32), ("C", i32), ("mid", i32), ("HW", f64), ("scale", vp),
                ("shift", vp), ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("ca", vp), ("avgz", vp), ("maxz", vp),
                ("hid_avg", vp), ("hid_max", vp), ("dw1", vp), ("db1", vp), ("dw2", vp), ("db2", vp), ("d_avgz", vp),
                ("d_maxz", vp), ("work", vp), ("ext", vp), ("first", vp)]


class CbamPixDesc(C.Structure):
    _fields_ = [("y", vp), ("ld_y", i32), ("cy_off", i32), ("scale", vp), ("shift", vp), ("mean", vp), ("invstd", vp),
                ("ca", vp), ("maxz", vp), ("d_avgz", vp), ("d_maxz", vp), ("sa", vp), ("st", vp), ("amax", vp), ("du", vp),
                ("dst", vp), ("res", vp), ("ld_res", i32), ("cres_off", i32), ("res_pool", i32), ("out", vp), ("ld_out", i32),
                ("cout_off", i32), ("d_same", vp), ("ld_same", i32), ("csame_off", i32), ("d_pool", vp), ("ld_pool", i32),
                ("cpool_off", i32), ("g", vp), ("ld_g", i32), ("dz", vp), ("ld_dz", i32), ("partial", vp), ("dtype", i32),
                ("B", i32), ("H", i32), ("W", i32), ("C", i32), ("ext", vp), ("first", vp)]


class CbamConv7Desc(C.Structure):
    _fields_ = [("st", vp), ("w7", vp), ("b7", vp), ("sa", vp), ("du", vp), ("dst", vp), ("dw_partial", vp), ("dw7", vp),
                ("db7", vp), ("B", i32), ("H", i32), ("W", i32)]


class MetricsDesc(C.Structure):
    _fields_ = [("logits", vp * 8), ("t_atom", vp), ("t_types", vp), ("t_charges", vp), ("t_hs", vp), ("t_bond", vp),
                ("t_btypes", vp), ("t_rho", vp), ("t_omega", vp), ("B", i32), ("h", i32), ("w", i32), ("peaks", vp),
                ("partial", vp), ("totals", vp), ("last", vp)]


class ExtractDesc(C.Structure):
    _fields_ = [("atom_mask", vp), ("bond_mask", vp), ("types", vp), ("charges", vp), ("hs", vp), ("btypes", vp), ("rho", vp),
                ("omega", vp), ("B", i32), ("h", i32), ("w", i32), ("cap_atoms", i32), ("cap_bonds", i32), ("counts", vp),
                ("atoms", vp), ("bonds", vp), ("bond_rho", vp), ("work", vp), ("work_masks", vp), ("btype_idx", vp)]


class RasterDesc(C.Structure):
    _fields_ = [("t_atom", vp), ("t_types", vp), ("t_charges", vp), ("t_hs", vp), ("t_bond", vp), ("t_btypes", vp), ("t_rho", vp),
                ("t_omega", vp), ("B", i32), ("h", i32), ("w", i32), ("max_atoms", i32), ("max_bonds", i32), ("atoms", vp),
                ("n_atoms", vp), ("bonds", vp), ("n_bonds", vp), ("rho", vp), ("group_flags", vp), ("prev_atoms", vp), ("prev_bonds", vp),
                ("prev_rho", vp), ("prev_counts", vp), ("incremental", i32)]


class HeadsFusedDesc(C.Structure):
    _fields_ = [("feat", vp), ("ld", i32), ("scale", vp), ("shift", vp), ("slope", vp), ("mean", vp), ("invstd", vp),
                ("drop_p", f32), ("drop_seed", u32), ("drop_salt", vp), ("w2", vp * 8), ("b2", vp * 8), ("w2_pack", vp),
                ("logits", vp * 8), ("t_atom", vp), ("t_types", vp), ("t_charges", vp), ("t_hs", vp), ("t_bond", vp),
                ("t_btypes", vp), ("t_rho", vp), ("t_omega", vp), ("dl", vp), ("g", vp), ("bn_partial", vp), ("loss_partial", vp),
                ("B", i32), ("h", i32), ("w", i32), ("chan_scale", vp), ("chan_off", i32 * 8), ("dw2", vp * 8), ("db2", vp * 8),
                ("wgrad_work", vp), ("keep_mask", vp), ("target_flags", vp), ("zero_bytes", vp)]


class ConvTDesc(C.Structure):
    _fields_ = [("src", ActSrc), ("w", vp), ("bias", vp), ("y", vp), ("dtype", i32), ("B", i32), ("Hin", i32), ("Win", i32), ("cin_off", i32),
                ("Cin", i32), ("Hout", i32), ("Wout", i32), ("ldy", i32), ("cout_off", i32), ("Cout", i32), ("Cout_pad", i32)]


_STRUCTS = [ActSrc, ConvDesc, PackDesc, BnFwdDesc, ActBwdDesc, BnBwdDesc, BnApplyDesc, WgradDesc, WgradReduceDesc,
            LossDesc, LossFinDesc, AdamDesc, NmsDesc, CbamChannelDesc, CbamPixDesc, CbamConv7Desc, MetricsDesc, ExtractDesc, RasterDesc,
            HeadsFusedDesc, HeadsEpi, ConvTDesc]

# every symbol include/abcnet_hip.h declares: name -> (restype, argtypes)
P = C.POINTER
SYMBOLS = {
    "abc_conv_stat_blocks": (C.c_int, [P(ConvDesc)]),
    "abc_convt_fused_ok": (C.c_int, [P(ConvTDesc)]),
    "abc_convt_fused_fwd": (C.c_int, [P(ConvTDesc), vp]),
    "abc_conv_actbwd_ok": (C.c_int, [P(ConvDesc)]),
    "abc_conv_fwd_batch": (C.c_int, [P(ConvDesc), C.c_int32, C.c_void_p]),
    "abc_conv_batch_ok": (C.c_int, [P(ConvDesc), C.c_int32]),
    "abc_conv_variant": (C.c_int, [vp]),
    "abc_conv_weight_layout": (C.c_int, [vp]),
    "abc_conv_fwd": (C.c_int, [P(ConvDesc), vp]),
    "abc_conv_chunk": (C.c_int, [C.c_int, C.c_int]),
    "abc_heads_batch": (C.c_int, [vp, i32, i32, vp]),
    "abc_heads_fused_pack_bytes": (i64, []),
    "abc_heads_fused_chunks": (C.c_int, [P(HeadsFusedDesc)]),
    "abc_heads_fused_loss_blocks": (C.c_int, [P(HeadsFusedDesc)]),
    "abc_heads_fused_dl_elems": (i64, [P(HeadsFusedDesc)]),
    "abc_heads_fused_wgrad_floats": (i64, [P(HeadsFusedDesc)]),
    "abc_heads_fused_rows": (C.c_int, [i32]),
    "abc_heads_fused_chan_of_row": (C.c_int, [i32, i32]),
    "abc_heads_fused_pack": (C.c_int, [P(HeadsFusedDesc), vp]),
    "abc_heads_fused_fwd_bwd": (C.c_int, [P(HeadsFusedDesc), vp]),
    "abc_heads_fused_wgrad": (C.c_int, [P(HeadsFusedDesc), vp]),
    "abc_conv_tile": (C.c_int, [P(ConvDesc), P(i32), P(i32), P(i32)]),
    "abc_pack_conv_weights": (C.c_int, [P(PackDesc), vp]),
    "abc_pack_item_bytes": (C.c_int, []),
    "abc_pack_item_fill": (i64, [vp, P(PackDesc), i64]),
    "abc_pack_batch": (C.c_int, [vp, i32, i64, vp]),
    "abc_bn_finalize_fwd": (C.c_int, [P(BnFwdDesc), vp]),
    "abc_bn_finalize_fwd_batch": (C.c_int, [vp, i32, i32, vp]),
    "abc_bn_finalize_bwd_batch": (C.c_int, [vp, i32, i32, vp]),
    "abc_bn_eval_coeffs": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, f32, vp]),
    "abc_bn_eval_fold": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, f32, vp]),
    "abc_act_bwd_blocks": (C.c_int, [P(ActBwdDesc)]),
    "abc_act_bwd": (C.c_int, [P(ActBwdDesc), vp]),
    "abc_bn_finalize_bwd": (C.c_int, [P(BnBwdDesc), vp]),
    "abc_bn_apply_bwd": (C.c_int, [P(BnApplyDesc), vp]),
    "abc_wgrad_fuses_apply": (C.c_int, [vp]),
    "abc_wgrad_rowsum_ok": (C.c_int, [vp]),
    "abc_wgrad_pads": (C.c_int, [P(WgradDesc), P(i32), P(i32)]),
    "abc_wgrad_blocks": (C.c_int, [P(WgradDesc)]),
    "abc_wgrad_tile": (C.c_int, [P(WgradDesc), P(i32), P(i32)]),
    "abc_wgrad": (C.c_int, [P(WgradDesc), vp]),
    "abc_wgrad_reduce": (C.c_int, [P(WgradReduceDesc), vp]),
    "abc_wgrad_reduce_bn_bwd": (C.c_int, [P(WgradReduceDesc), P(BnBwdDesc), vp]),
    "abc_wgrad_heads_batch": (C.c_int, [vp, i32, vp]),
    "abc_wgrad_reduce_batch": (C.c_int, [vp, i32, vp]),
    "abc_colsum_blocks": (C.c_int, [i64]),
    "abc_colsum": (C.c_int, [vp, i32, i64, i32, i32, i32, vp, vp, vp, vp]),
    "abc_colsum_w1": (C.c_int, [vp, i32, i64, i32, i32, i32, vp, vp, vp, vp, vp]),
    "abc_loss_blocks": (C.c_int, [P(LossDesc)]),
    "abc_loss_fwd_bwd": (C.c_int, [P(LossDesc), vp]),
    "abc_loss_finalize": (C.c_int, [P(LossFinDesc), vp]),
    "abc_adam_step": (C.c_int, [P(AdamDesc), vp]),
    "abc_nms_peaks": (C.c_int, [P(NmsDesc), vp]),
    "abc_extract_work_ints": (i64, [P(ExtractDesc)]),
    "abc_extract_work_masks": (i64, [P(ExtractDesc)]),
    "abc_extract_peaks": (C.c_int, [P(ExtractDesc), vp]),
    "abc_rasterize_targets": (C.c_int, [P(RasterDesc), vp]),
    "abc_metrics_blocks": (C.c_int, [P(MetricsDesc)]),
    "abc_metrics_update": (C.c_int, [P(MetricsDesc), vp]),
    "abc_plane_sum": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "abc_plane_sum_work": (C.c_int, [i32]),
    "abc_cbam_channel_fwd": (C.c_int, [P(CbamChannelDesc), vp]),
    "abc_cbam_channel_bwd": (C.c_int, [P(CbamChannelDesc), vp]),
    "abc_cbam_spatial_stats": (C.c_int, [P(CbamPixDesc), vp]),
    "abc_cbam_apply_fwd": (C.c_int, [P(CbamPixDesc), vp]),
    "abc_cbam_bwd1": (C.c_int, [P(CbamPixDesc), vp]),
    "abc_cbam_bwd2_blocks": (C.c_int, [P(CbamPixDesc)]),
    "abc_cbam_bwd2": (C.c_int, [P(CbamPixDesc), vp]),
    "abc_cbam_bwd3_blocks": (C.c_int, [P(CbamPixDesc)]),
    "abc_cbam_bwd3": (C.c_int, [P(CbamPixDesc), vp]),
    "abc_cbam_conv7_fwd": (C.c_int, [P(CbamConv7Desc), vp]),
    "abc_cbam_conv7_blocks": (C.c_int, [P(CbamConv7Desc)]),
    "abc_cbam_conv7_bwd": (C.c_int, [P(CbamConv7Desc), vp]),
    "abc_cbam_conv7_bwd_partial": (C.c_int, [P(CbamConv7Desc), vp]),
    "abc_cbam_channel_bwd_c7": (C.c_int, [P(CbamChannelDesc), P(CbamConv7Desc), vp]),
    "abc_add_into": (C.c_int, [vp, i32, i32, vp, i32, i32, i32, i64, i32, vp]),
    "abc_nhwc_to_nchw_f32": (C.c_int, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "abc_nchw_to_nhwc_f32": (C.c_int, [vp, i32, i32, i32, i32, vp, i32, i32, vp]),
    "abc_fill_f32": (C.c_int, [vp, f32, i64, vp]),
    "abc_absmax": (C.c_int, [vp, i32, i64, vp, vp]),
    "abc_absmax_cols": (C.c_int, [vp, i32, i64, i32, i32, i32, vp, vp]),
    "abc_fp8_act_scale": (C.c_int, [vp, f32, vp, vp, vp]),
    "abc_fp8_weight_scales": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "abc_concat_f32": (C.c_int, [vp, vp, i32, vp, vp]),
    "abc_counter_add_u32": (C.c_int, [vp, u32, vp]),
    "abc_pool_act": (C.c_int, [vp, i32, i32, i32, i32, vp, i32, i32, vp]),
    "abc_sizeof": (C.c_int, [C.c_int]),
    "abc_last_error": (C.c_char_p, []),
    "abc_version": (C.c_int, []),
    "abc_set_reserved_cus": (C.c_int, [i32]),
    "abc_get_reserved_cus": (C.c_int, []),
}

_lib = None


class AbcNetHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it is missing or does not match this binding."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AbcNetHipError(
            "libabcnet_hip.so not found at %s -- build it with ./build_hip.sh (or __graft_entry__.build()); "
            "abcnet_amd has no CPU fallback" % LIB_PATH)
    # On a GPU box the HIP runtime has to be up (through torch, whose bundled runtime this library binds to) BEFORE the library is
    # mapped: loaded first -- __graft_entry__.build() followed by smoke() in one process did that -- its kernels' first launch fails
    # with "no ROCm-capable device is detected" (measured, round 4).  No-op without a GPU.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:      # noqa: BLE001  (the check below still reports a missing / mismatched library)
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for i, st in enumerate(_STRUCTS):
        n = lib.abc_sizeof(i)
        if n != C.sizeof(st):
            raise AbcNetHipError("struct #%d (%s): binding has %d bytes, library %d" % (i, st.__name__, C.sizeof(st), n))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise AbcNetHipError("%s failed (%d): %s" % (what, rc, load().abc_last_error().decode()))


def ptr(t):
    """raw device/host address of a torch tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def set_taps(desc, taps):
    desc.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        desc.tap_dy[i] = dy
        desc.tap_dx[i] = dx
