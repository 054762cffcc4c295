"""Launch plan of the U-Net hot path on one MI355X.

The engine lowers the topology of the reference networks (unet.py:100-119,
unet2.py:152-173) to a static list of C-ABI calls into libabcnet_hip.so for a fixed
(batch, height, width, dtype): weight packing, forward, backward.  All activations
are NHWC in HBM; every conv writes its RAW output plus BatchNorm partial statistics
and every consumer applies BN + activation (+pool, +dropout) on load, so an
activation is written once and read once per consumer.  The skip concatenations are
never materialised by a copy: the encoder conv and the transposed conv write the two
channel halves of one buffer in place (unet.py:51-59).

PyTorch is used here only to own device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib as L
from . import arch

BN_EPS = 1e-5
BN_MOM = 0.1
DROP_STEP = 0x9E3779B9  # odd: the salted seed walks all 2^32 values
def head_offsets(heads):
    """offset of each head inside the flat per-channel loss-scale array"""
    offs, o = [], 0
    for h in heads:
        offs.append(o)
        o += h
    return offs


def taps_square(k):
    p = (k - 1) // 2
    return [(ky - p, kx - p) for ky in range(k) for kx in range(k)]


def taps_mirror(taps):
    return [(-dy, -dx) for dy, dx in taps]


def convT_phase_taps(py, px, crop_y=True, crop_x=True):
    """input offsets of the taps of output parity (py,px) of ConvTranspose2d(k3,s2) after the reference's pad / crop
    (unet.py:51-57); order matches abc_pack_conv_weights mode 2 for the pack parity convT_pack_parity() names.

    Per axis: the 2n+1 outputs o = 2i + k (k = 0..2) meet a skip tensor of size s.  s = 2n (every level when the input size
    is a multiple of 32): diff = -1, F.pad crops the FIRST row -> output o' = o - 1, parity 0 takes k = 1 at i = a, parity 1
    takes k = 0 at i = a + 1 and k = 2 at i = a.  s = 2n + 1 (odd levels of other input sizes): nothing is cropped ->
    parity 0 takes k = 0 at i = a and k = 2 at i = a - 1, parity 1 takes k = 1 at i = a."""
    def axis(p, crop):
        if crop:
            return [0] if p == 0 else [1, 0]
        return [0, -1] if p == 0 else [0]
    return [(dy, dx) for dy in axis(py, crop_y) for dx in axis(px, crop_x)]


def convT_pack_parity(p, crop):
    """which kernel rows abc_pack_conv_weights mode 2 has to pick for output parity p: pack parity 1 = (k = 0, k = 2),
    pack parity 0 = (k = 1)"""
    return p if crop else 1 - p


def convT_dgrad_taps(crop_y=True, crop_x=True):
    """dX[i] = sum_k dOut[2 i + k - c] W[k] with c = 1 where the first row / column was cropped, 0 where not"""
    cy, cx = (1 if crop_y else 0), (1 if crop_x else 0)
    return [(ky - cy, kx - cx) for ky in range(3) for kx in range(3)]


TAPS_CONVT_DGRAD = convT_dgrad_taps()


class Src:
    """an activation as a consumer sees it: raw tensor + on-load transform"""

    def __init__(self, t, dt, H, W, ld, coff, C, coef=None, pool=False, drop_p=0.0, drop_seed=0, producer=None, planar=False):
        self.t, self.dt, self.H, self.W, self.ld, self.coff, self.C = t, dt, H, W, ld, coff, C
        self.coef, self.pool, self.drop_p, self.drop_seed, self.producer = coef, pool, drop_p, drop_seed, producer
        self.drop_salt = None  # device uint32 scalar added to drop_seed (the engine's per-step counter)
        self.planar = planar  # NCHW f32 [B][C][H][W] (head maps / their gradients)
        self.q = None         # fp8 tensors: (amax, s, 1 / s) device scalars of the per-tensor scale

    def lh(self):  # logical dims
        return (self.H // 2, self.W // 2) if self.pool else (self.H, self.W)

    def fill(self, a: L.ActSrc):
        a.x = self.t.data_ptr()
        a.planar, a.ctot = (1, self.C) if self.planar else (0, 0)
        if self.coef is not None:
            a.scale, a.shift, a.slope = (c.data_ptr() for c in self.coef)
        else:
            a.scale = a.shift = a.slope = None
        a.Hx, a.Wx, a.ldx, a.pool = self.H, self.W, self.ld, 1 if self.pool else 0
        a.drop_p, a.drop_seed = self.drop_p, self.drop_seed
        a.drop_salt = self.drop_salt.data_ptr() if (self.drop_salt is not None and self.drop_p > 0) else None


class Rec:
    """one conv (+BN) layer: what forward produced and what backward needs"""

    def __init__(self, **kw):
        self.grad_same = None  # (tensor, ld, coff): grad wrt the activated output, full resolution
        self.grad_pool = None  # (tensor, ld, coff): grad wrt the pooled activated output
        self.__dict__.update(kw)


def set_reserved_cus(n):
    """abc_set_reserved_cus for Python callers: leave n of the 256 CUs out of the persistent convolution grids (PROCESS-wide).  The plans
    of this process bake the value into grid sizes and BatchNorm partial-row buffers, so it can only change while no plan is alive:
    raises otherwise.  Setting the value it already has is always fine."""
    lib = L.load()
    n = (int(n) + 3) & ~3
    if n == lib.abc_get_reserved_cus():
        return n
    alive = [e for e in Engine._live]
    if alive:
        raise L.AbcNetHipError("cannot change the reserved-CU count from %d to %d: %d plan(s) built under the current value are alive "
                               "(it is a process-wide setting baked into their grids; build every Trainer / InferenceRunner of this "
                               "process with the same reserve_cus, or drop the old ones first)" % (lib.abc_get_reserved_cus(), n, len(alive)))
    L.check(lib.abc_set_reserved_cus(n), "set_reserved_cus")
    return n


class Engine:
    # every plan alive in this process (weak references): the reserved-CU count is a PROCESS-wide setting of the library that plans bake
    # into their grids and buffer sizes, so it may only change while no plan exists (set_reserved_cus below)
    _live = weakref.WeakSet()

    def __init__(self, variant, in_channels, heads, params, grads, buffers, counters, layout, B, H, W, dtype, train,
                 dropout_p=0.2, device="cuda", drop_seed=0x1234ABCD, fold_bn=False, fused_heads=False, batched_heads=True, fp8=False,
                 guards=False, heads_epilogue=False, actbwd_epilogue=True, merge_reduce=True, nms_heads=False, decode=False,
                 dual_wgrad=True, fused_convt=True):
        """merge_reduce=False: every slab reduction and every BatchNorm-backward finaliser a launch of its own;
        dual_wgrad=False: the BatchNorm-backward apply as a pass of its own instead of on the weight gradient's load (the A/B of that
        fusion; part of the engine cache key like the other plan switches);
        actbwd_epilogue=False: every act_bwd pass as a launch of its own (the form the fused epilogue is tested against);
        batched_heads=False: one launch per head instead of the batched / merged heads launches (kept as the plain form the
        batched one is tested against, tests/test_gpu_model.py::test_batched_heads_equal_one_by_one_launches)"""
        if variant not in ("unet", "unet2"):
            raise NotImplementedError("variant %r" % variant)
        if H < 32 or W < 32:
            raise ValueError("H and W must be at least 32 (five 2x2 poolings; got %dx%d)" % (H, W))
        if in_channels < 1:
            raise ValueError("in_channels must be positive")
        self.in_channels = in_channels
        self.lib = L.load()
        # the persistent convolution grids -- and the BatchNorm partial-row buffers sized after them -- follow the library's
        # process-wide reserved-CU count AT PLAN TIME; the launches re-derive their grids from the current value, so it must
        # not move while this plan lives (checked before every run_*; set_reserved_cus refuses to move it)
        self.reserved_cus = self.lib.abc_get_reserved_cus()
        self.dual_wgrad = bool(dual_wgrad)
        self.fused_convt = bool(fused_convt)      # False: the ConvTranspose forward as four batched phase convolutions (the A/B of convt_fused.hip)
        self.variant, self.heads, self.B, self.H, self.W = variant, list(heads), B, H, W
        self.train = train
        # eval-mode graph with every BatchNorm folded into the convolution in front of it: the weights are
        # packed times gamma / sqrt(running_var + eps), the bias becomes (b - running_mean) * that + beta, the conv's epilogue
        # applies the activation, and every consumer loads a finished tensor with the identity transform
        self.fold = bool(fold_bn)
        if self.fold and (train or variant != "unet"):
            raise ValueError("fold_bn is the eval-mode graph of unet.py")
        # fp8 (e4m3) form of that graph (BASELINE config 5): the 128-channel 3x3 convolutions at the output
        # resolution -- up3.conv's second convolution, dconv1, dconv2 and the eight heads' conv1, 13 of the graph's 15 "128 -> 128"
        # units -- run on the block-scaled MFMA (2 x the bf16 rate) over e4m3 activations (one calibrated scale per tensor) and
        # e4m3 weights (one scale per output row); everything else stays bf16.  calibrate_fp8() sets the activation scales.
        self.fp8 = bool(fp8)
        if self.fp8 and not (self.fold and dtype == "bf16"):
            raise ValueError("fp8 is a form of the BatchNorm-folded bf16 inference graph (fold_bn=True, dtype='bf16')")
        self.fp8_recs = []
        self.hfeat_q = None
        self.fp8_calibrated = False
        self.guards = bool(guards)
        self._guarded = []
        # folded inference graph, opt-in: the heads' 1x1 convolutions in the epilogue of the merged conv1 (abc_conv_desc.heads_epi) --
        # the 8 x 128-channel feature tensor is never written.  Bit-identical to the default plan (the separate heads kernel) and
        # measured SLOWER than it (b64 at 512 x 512: 7.86 -> 9.0-10.4 ms bf16, 6.09 -> 7.7-9.6 ms e4m3), hence off
        self.heads_epilogue = bool(heads_epilogue)
        # eval graph: |rho| and the omega-bin NMS mask (img2smiles2.py:73-79) as second outputs of the heads' 1x1 kernel
        # (abc_conv_desc.head_aux) -- nms_rho / nms_omega; the NMS kernel then reads the two one-channel centre maps only
        self.nms_heads = bool(nms_heads) and not train
        self.nms_rho = self.nms_omega = None
        # decode (with nms_heads): the maps the decoder of img2smiles2.py:104-191 never reads as such are not stored -- the raw rho map
        # (only |rho| is used, :73) and the 360 bond-type planes (only their six-way arg max per omega bin, :71,112: a uint8 map,
        # abc_conv_desc.head_aux_mode 3); logits[5] / logits[6] are then None
        self.decode = bool(decode) and self.nms_heads
        self.btype_idx = None
        # training: the activation / BatchNorm-statistics backward pass of a layer in the epilogue of the data gradient that produces
        # its input (abc_conv_desc.actbwd_*), where the layer has that one reader (_actbwd_target)
        self.actbwd_epilogue = bool(actbwd_epilogue)
        self.merge_reduce = bool(merge_reduce)
        self._pending_reduce = None
        self.dt = L.BF16 if dtype == "bf16" else L.F32
        self.tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        # the fused train step's heads (csrc/heads_fused.hip): conv2 forward + loss + the way back to the BatchNorm outputs as
        # one pass.  Needs the loss in the plan (so: the Trainer asks for it; a module-style forward() cannot use it), bf16,
        # the reference's eight heads and whole 128-pixel chunks; settled in _build_heads
        self.batched_heads = bool(batched_heads)
        self.want_fused_heads = bool(fused_heads) and train and dtype == "bf16" and self.batched_heads
        self.hf = None
        self.dev = device
        self.params, self.grads, self.buffers, self.counters = params, grads, buffers, counters
        self.lay_p, self.lay_b, self.lay_c = layout
        self.drop_p = dropout_p if (train and variant == "unet") else 0.0
        self.drop_seed = drop_seed & 0xFFFFFFFF
        # nn.Dropout draws a fresh mask every forward (unet.py:69).  The launch descriptors -- and a captured hipGraph --
        # are static, so the step-dependence lives in HBM: a counter the forward plan bumps first thing (DROP_STEP per
        # step), added to drop_seed by every kernel that applies or replays the mask
        self.drop_salt = torch.zeros(1, dtype=torch.int32, device=device)
        self.head_off = head_offsets(self.heads)
        self.keep = []  # ctypes descriptors and tensors referenced by raw pointer
        self.pack_ops, self.fwd_ops, self.bwd_ops = [], [], []
        self._pack_descs = []
        self._w_layout = {}   # packed weight buffer -> abc_pack_desc.layout its consuming conv wants
        self.recs = []
        self._ws_need = 0
        self._ws_users = []
        self._colsum_need = 0
        self._colsum_users = []
        self._build()
        Engine._live.add(self)      # (a plan that was built: half-constructed ones hold no launches)

    def dropout_seed(self, nth_forward):
        """the hash seed the nth train-mode forward of this engine (1-based) draws its mask with (tests mirror the mask
        with abcnet_amd.dropout.keep_mask)"""
        return (self.drop_seed + nth_forward * DROP_STEP) & 0xFFFFFFFF

    # ------------------------------------------------------------------ memory helpers
    def P(self, name):
        off, n = self.lay_p[name]
        return self.params.data_ptr() + 4 * off

    def G(self, name):
        off, n = self.lay_p[name]
        return self.grads.data_ptr() + 4 * off

    def Bf(self, name):
        off, n = self.lay_b[name]
        return self.buffers.data_ptr() + 4 * off

    def Cn(self, name):
        return self.counters.data_ptr() + 8 * self.lay_c[name]

    GUARD_BYTES = 4096

    def new(self, shape, dtype=None, fill=0.0):
        """a device buffer of the plan.  Debug plans (Engine(guards=True), the bounds-check build of SURVEY section 5): the buffer
        sits between two 4 KB guard bands of 0xA5 bytes that check_guards() inspects -- the kernels take raw pointers, loads
        are range-checked by the buffer descriptors, stores are not"""
        dtype = dtype or self.tdt
        if not self.guards:
            t = torch.full(shape, fill, dtype=dtype, device=self.dev)
            self.keep.append(t)
            return t
        n = 1
        for s_ in shape:
            n *= s_
        esz = torch.empty((), dtype=dtype).element_size()
        body = -(-n * esz // 256) * 256                   # (keeps every buffer 256-byte aligned, as the caching allocator does)
        raw = torch.full((body + 2 * self.GUARD_BYTES,), 0xA5, dtype=torch.uint8, device=self.dev)
        t = raw[self.GUARD_BYTES:self.GUARD_BYTES + n * esz].view(dtype).view(shape)
        t.fill_(fill)
        self.keep.append(t)
        self._guarded.append((raw, n * esz, tuple(shape), str(dtype)))
        return t

    def check_guards(self):
        """debug plans: every guard band (and the alignment slack behind each buffer) still holds its pattern; returns the number
        of buffers checked, raises with the offenders otherwise (host sync)"""
        if not self.guards:
            raise RuntimeError("this plan was built without guards (Engine(guards=True) / Trainer(guards=True))")
        bad = []
        for i, (raw, nbytes, shape, dt) in enumerate(self._guarded):
            G = self.GUARD_BYTES
            lo_ok = bool((raw[:G] == 0xA5).all())
            hi_ok = bool((raw[G + nbytes:] == 0xA5).all())
            if not (lo_ok and hi_ok):
                bad.append((i, shape, dt, "below" if not lo_ok else "above"))
        if bad:
            raise RuntimeError("out-of-bounds stores next to %d of %d plan buffers: %s" % (len(bad), len(self._guarded), bad[:8]))
        return len(self._guarded)

    def act_buf(self, H, W, C, dt=None):
        t = self.new((self.B, H, W, C), self._tdt(dt)) if dt is not None else self.new((self.B, H, W, C))
        coef = (self.new((C,), torch.float32, 1.0), self.new((C,), torch.float32, 0.0), self.new((C,), torch.float32, 1.0))
        return t, coef

    # ------------------------------------------------------------------ op emitters
    def _emit(self, ops, fn, desc, what, writes=(), meta=None):
        """writes = names of parameters whose GRADIENT is final once this op has run (bucketed all-reduce);
        meta = {kernel, flops, bytes}: algorithmic work of this launch (for the roofline report)"""
        self.keep.append(desc)
        ref = C.byref(desc)
        ops.append((fn, ref, what, tuple(writes), meta or {"kernel": what.split(" ")[0], "flops": 0, "bytes": 0}))

    # ---- a weight gradient's slab reduction is independent of everything until the optimiser: it waits (_pending_reduce) for the
    # next BatchNorm-backward finaliser of the plan -- a dependent ~5 us launch of C blocks -- and rides in ITS launch
    # (abc_wgrad_reduce_bn_bwd); flushed on its own before the next weight gradient (the slabs share one workspace) and at the end
    def _flush_reduce(self, ops):
        pr = getattr(self, "_pending_reduce", None)
        if pr is not None:
            self._pending_reduce = None
            pops, rd, what, writes, meta = pr
            self._emit(pops, self.lib.abc_wgrad_reduce, rd, what, writes=writes, meta=meta)

    def _emit_bn_bwd(self, ops, f, what, writes):
        pr = getattr(self, "_pending_reduce", None)
        if pr is None or pr[0] is not ops or not self.merge_reduce:
            self._flush_reduce(ops)
            self._emit(ops, self.lib.abc_bn_finalize_bwd, f, what, writes=writes)
            return
        self._pending_reduce = None
        _pops, rd, rwhat, rwrites, rmeta = pr
        self.keep += [rd, f]
        lib = self.lib
        meta = {"ws": rmeta.get("ws", 0), "kernel": "wgrad_reduce+bn_bwd", "flops": 0, "bytes": rmeta["bytes"]}
        ops.append((lambda _r, st, a=(rd, f): lib.abc_wgrad_reduce_bn_bwd(C.byref(a[0]), C.byref(a[1]), st), None,
                    rwhat + " + " + what, tuple(rwrites) + tuple(writes), meta))

    def _dn(self, dt):
        return {L.BF16: "bf16", L.F32: "f32", L.FP8: "fp8"}[dt]

    def _esz(self, dt):
        return {L.BF16: 2, L.F32: 4, L.FP8: 1}[dt]

    def _tdt(self, dt):
        return {L.BF16: torch.bfloat16, L.F32: torch.float32, L.FP8: torch.float8_e4m3fn}[dt]

    # fp8 inference graph: which convolutions write e4m3 (their consumers then compute in e4m3)
    FP8_OUT = ("up3.conv.double_conv.0", "up3.conv.double_conv.3", "dconv1.double_conv.0", "dconv1.double_conv.3",
               "dconv2.double_conv.0", "dconv2.double_conv.3")

    def _out_dt(self, cname):
        return L.FP8 if (self.fp8 and cname in self.FP8_OUT) else self.dt

    def emit_pack(self, wname, dst, mode, Cout, Cin, k, rows_pad, red_real, red_total=None, red_off=0, py=0, px=0, rows_total=0,
                  rows_off=0, row_scale=None, cdt=None):
        cdt = self.dt if cdt is None else cdt
        d = L.PackDesc()
        d.rows_total, d.rows_off = rows_total, rows_off
        d.row_scale = row_scale
        d.w, d.dst, d.mode, d.dtype_c = self.P(wname), dst.data_ptr(), mode, cdt
        d.Cout, d.Cin, d.kh, d.kw, d.py, d.px = Cout, Cin, k, k, py, px
        total = red_real if red_total is None else red_total
        ck = self.lib.abc_conv_chunk(cdt, total)
        d.rows_pad, d.red_pad, d.red_total, d.red_off, d.ck = rows_pad, -(-red_real // ck) * ck, total, red_off, ck
        self.keep.append(d)
        self._pack_descs.append(d)

    def packed(self, ntaps, red, rows_pad, cdt=None):
        cdt = self.dt if cdt is None else cdt
        ck = self.lib.abc_conv_chunk(cdt, red)
        if ck <= 0:
            raise ValueError("no K-chunk for %d reduction channels in %s" % (red, self._dn(cdt)))
        return self.new((ntaps * (-(-red // ck)) * rows_pad * ck,), self._tdt(cdt))

    def emit_conv(self, ops, src: Src, w, bias, y, y_dt, Hout, Wout, ldy, cout_off, Cout, taps, stats=None, stride=1,
                  grid=None, om=1, oy0=0, ox0=0, cin_off=None, Cin=None, what="conv", planar_out=False, stats_rows=2,
                  accumulate=False, collect=None, out_slope=None, cdt=None, out_scale=None, out_quant=None, out_quant_stride=0, heads_epi=None,
                  actbwd=None, only_variant=None):
        """collect: a list -- the launch is not emitted but appended as (desc, what, meta) for emit_heads_batch;
        only_variant: emit only if abc_conv_variant says this kernel family serves the descriptor (else return None, emitting nothing);
        actbwd: the Rec of the layer whose activation output this data gradient differentiates -> abc_conv_desc.actbwd_* (the
        epilogue stores d(BatchNorm output) and that layer's BatchNorm-backward partial sums); returns None, emitting nothing, when
        the library does not serve it for this shape;
        out_slope: not None -> the epilogue stores max(v, out_slope * v) (folded-BatchNorm eval graph);
        cdt / out_scale / out_quant: fp8 inference graph (abc_conv_desc.out_scale, .out_quant)"""
        cdt = self.dt if cdt is None else cdt
        d = L.ConvDesc()
        if out_slope is not None:
            d.out_act, d.out_slope = 1, out_slope
        src.fill(d.src)
        d.w, d.bias, d.y = w.data_ptr(), bias, y.data_ptr()
        d.stats = None
        d.out_scale = None if out_scale is None else out_scale.data_ptr()
        d.out_quant = None if out_quant is None else out_quant.data_ptr()
        d.out_quant_stride = out_quant_stride
        d.heads_epi = None if heads_epi is None else heads_epi.data_ptr()
        d.dtype_in, d.dtype_c, d.dtype_out = src.dt, cdt, y_dt
        lh, lw = src.lh()
        d.B, d.Hin, d.Win = self.B, lh, lw
        d.cin_off = src.coff if cin_off is None else cin_off
        d.Cin = src.C if Cin is None else Cin
        gh, gw = grid if grid is not None else (Hout, Wout)
        d.Hg, d.Wg, d.Hout, d.Wout, d.ldy, d.cout_off, d.Cout, d.Cout_pad = gh, gw, Hout, Wout, ldy, cout_off, Cout, -(-Cout // 32) * 32
        d.stride, d.om, d.oy0, d.ox0 = stride, om, oy0, ox0
        d.planar_out, d.ctot_out = (1, Cout) if planar_out else (0, 0)
        d.stats_rows, d.accumulate = stats_rows, 1 if accumulate else 0
        L.set_taps(d, taps)
        self._last_conv_desc = d
        if actbwd is not None:
            p = actbwd
            d.actbwd_y, d.actbwd_ld, d.actbwd_coff = p.y.data_ptr(), p.ld, p.coff
            d.actbwd_scale, d.actbwd_shift, d.actbwd_slope = p.scale.data_ptr(), p.shift.data_ptr(), p.slopes.data_ptr()
            d.actbwd_mean, d.actbwd_invstd = p.mean.data_ptr(), p.invstd.data_ptr()
            d.stats = y.data_ptr()      # (placeholder for the query: the rows are allocated below)
            stats = True
            if not self.lib.abc_conv_actbwd_ok(C.byref(d)):
                return None
        if only_variant is not None and self.lib.abc_conv_variant(C.byref(d)) != only_variant:
            return None
        nblk = self.lib.abc_conv_stat_blocks(C.byref(d))
        st = None
        if stats:
            st = self.new((nblk, stats_rows, Cout), torch.float32)
            d.stats = st.data_ptr()
        self._w_layout[d.w] = self.lib.abc_conv_weight_layout(C.byref(d))
        bn_, mt_, ck_ = L.i32(), L.i32(), L.i32()
        L.check(self.lib.abc_conv_tile(C.byref(d), C.byref(bn_), C.byref(mt_), C.byref(ck_)), "conv_tile")
        bn, mt, ck = bn_.value, mt_.value, ck_.value
        npx = self.B * gh * gw
        in_px = self.B * lh * lw * (4 if src.pool else 1)
        kname = ("conv_igemm", "conv_fast", "stem_conv", "head_fwd", "head_dgrad", "conv_narrow")[self.lib.abc_conv_variant(C.byref(d))]
        if actbwd is not None:
            kname += "+act_bwd"
        meta = {"kernel": "%s<%s,%s,%s,CK%d,BN%d,S%d,MT%d>" % (kname, self._dn(src.dt), self._dn(cdt), self._dn(y_dt), ck, bn, stride, mt),
                "flops": 2.0 * npx * Cout * len(taps) * d.Cin,
                "bytes": float(in_px * d.Cin * self._esz(src.dt) + npx * Cout * self._esz(y_dt))}
        if actbwd is not None:
            meta["bytes"] += float(npx * Cout * self._esz(self.dt))      # (the producer's y_raw, read by the epilogue)
        if collect is not None:
            collect.append((d, what, meta))
        else:
            self._emit(ops, self.lib.abc_conv_fwd, d, what, meta=meta)
        return st, nblk

    def emit_conv_batch(self, ops, items, what):
        """convolutions collected by emit_conv(collect=...) as ONE launch where the library serves them so (abc_conv_fwd_batch: the four
        output-parity phases of a ConvTranspose2d forward, unet.py:44, share a tile geometry of the lean kernel), else one launch each"""
        arr = (L.ConvDesc * len(items))()
        for i, (d, _w, _m) in enumerate(items):
            arr[i] = d
        if not self.lib.abc_conv_batch_ok(arr, len(items)):
            for d, w, m in items:
                self._emit(ops, self.lib.abc_conv_fwd, d, w, meta=m)
            return
        self.keep.append(arr)
        self.keep.append([d for d, _w, _m in items])
        lib, n = self.lib, len(items)
        k0 = items[0][2]["kernel"]
        meta = {"kernel": k0.replace("<", "_x%d<" % n, 1), "flops": sum(m["flops"] for _d, _w, m in items),
                "bytes": sum(m["bytes"] for _d, _w, m in items)}
        ops.append((lambda _r, st, a=arr: lib.abc_conv_fwd_batch(a, n, st), None, what, (), meta))

    def emit_heads_batch(self, ops, items, which, what):
        """the heads' 1x1 convolutions collected by emit_conv(collect=...) as ONE launch (abc_heads_batch) when every one of
        them is served by the dedicated heads kernel (which = 0 forward / 1 data gradient), else one launch each"""
        want = 3 if which == 0 else 4
        ok = 1 <= len(items) <= 8 and all(self.lib.abc_conv_variant(C.byref(d)) == want for d, _w, _m in items) and self.batched_heads
        if not ok:
            for d, w, m in items:
                self._emit(ops, self.lib.abc_conv_fwd, d, w, meta=m)
            return
        arr = (L.ConvDesc * len(items))()
        for i, (d, _w, _m) in enumerate(items):
            arr[i] = d
        self.keep.append(arr)
        self.keep.append([d for d, _w, _m in items])
        lib, n = self.lib, len(items)
        meta = {"kernel": "heads_%s_batch" % ("fwd" if which == 0 else "dgrad"), "flops": sum(m["flops"] for _d, _w, m in items),
                "bytes": sum(m["bytes"] for _d, _w, m in items)}
        ops.append((lambda _r, st, a=arr: lib.abc_heads_batch(a, n, which, st), None, what, (), meta))

    def emit_wgrad(self, ops, p: Src, q: Src, Ca, Cb, taps, stride, wname, what, cp_off=None, cq_off=None, dual=None, rowsum_to=None,
                   collect=None, nsplit=None):
        """dual = (y_raw tensor, ld, channel offset, dY pointer, dY pixel stride): fuse the BatchNorm-backward correction into the load of P
        (p = act_bwd output with coef = (ca, cc, cb)); returns False without emitting anything when the library does not
        serve this descriptor that way.
        collect: a list -- nothing is emitted; the launch and its reductions are appended as a dict for
        emit_wgrad_heads_batch, and the split-K slabs get their OWN buffer (the batched heads run concurrently)"""
        dw_ptr = self.G(wname)
        d = L.WgradDesc()
        p.fill(d.p)
        q.fill(d.q)
        d.dtype_p, d.dtype_q, d.dtype_c = p.dt, q.dt, self.dt
        gh, gw = p.lh()
        qh, qw = q.lh()
        d.B, d.Hg, d.Wg, d.Hq, d.Wq = self.B, gh, gw, qh, qw
        d.cp_off = p.coff if cp_off is None else cp_off
        d.cq_off = q.coff if cq_off is None else cq_off
        d.Ca, d.Cb, d.stride = Ca, Cb, stride
        L.set_taps(d, taps)
        if dual is not None:
            y2, ld2, c2, out_ptr, ld_out = dual
            d.p2, d.ld_p2, d.cp2_off, d.p_dual, d.p_out, d.ld_pout = y2.data_ptr(), ld2, c2, 1, out_ptr, ld_out
            if not self.lib.abc_wgrad_fuses_apply(C.byref(d)):
                return False
        ca_pad, cb_pad = L.i32(), L.i32()
        L.check(self.lib.abc_wgrad_pads(C.byref(d), C.byref(ca_pad), C.byref(cb_pad)), "wgrad_pads")
        ca_pad, cb_pad = ca_pad.value, cb_pad.value
        per_split = self.lib.abc_wgrad_blocks(C.byref(d))
        npatch = self.B * (-(-gh // 8)) * (-(-gw // 16))
        nsplit_arg = nsplit
        # one 8-wave workgroup per CU is resident: a single round of ~256 workgroups keeps the split-K slabs small
        nsplit = max(1, min(max(1, npatch // 2), 256 // per_split))
        at_, bt_ = L.i32(), L.i32()
        L.check(self.lib.abc_wgrad_tile(C.byref(d), C.byref(at_), C.byref(bt_)), "wgrad_tile")
        if (at_.value, bt_.value) == (0, 1):   # one-channel kernel: 256-thread workgroups streaming dY, two per CU (184-200 registers
            # with the BatchNorm-backward apply fused): ONE round of <= 512, 12 rows per pass (768 workgroups of 8 rows ran 1.5 rounds)
            nsplit = min(-(-self.B * gh // 12), 512)
        elif (at_.value, bt_.value) == (0, 2):   # 16-channel kernel: a wave per 8 x 16 tile run, 256-thread workgroups two to a CU (197 registers), 9 KB slabs
            nsplit = max(1, min(512, self.B * (gh // 8) * (gw // 16) // 8))
        elif (at_.value, bt_.value) == (0, 3):   # 5x5 32-channel kernel: five-wave workgroups two to a CU, one 8 x 16 tile at a time, 100 KB slabs
            nsplit = max(1, min(512, self.B * (gh // 8) * (gw // 16) // 4))
        elif nsplit_arg is not None and (at_.value, bt_.value) == (0, 0):
            nsplit = max(1, min(nsplit_arg, self.B * gh * gw // 128))   # the heads' kernel splits whole 128-pixel chunks
        d.nsplit = nsplit
        need = nsplit * len(taps) * ca_pad * cb_pad
        self._ws_need = max(self._ws_need, need)
        # (one workspace for the split-K slabs: a layer's reduction runs before the next layer's weight gradient, in stream order)
        wsk = 0
        r = L.WgradReduceDesc()
        r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad, r.dw, r.accumulate = nsplit, len(taps), Ca, Cb, ca_pad, cb_pad, dw_ptr, 0
        if collect is not None:
            own = self.new((need,), torch.float32)
            d.partial = r.partial = own.data_ptr()
        else:
            self._ws_users += [(d, wsk), (r, wsk)]
        meta = {"ws": wsk, "kernel": "wgrad<%s,%s,%s,%dx%d,S%d>" % (self._dn(p.dt), self._dn(q.dt), self._dn(self.dt), at_.value, bt_.value, stride),
                "flops": 2.0 * self.B * gh * gw * Ca * Cb * len(taps),
                # both operands once, the split-K slabs this launch writes, and (DUAL) the y_raw it reads + the dY it writes
                "bytes": float(self.B * gh * gw * Ca * self._esz(p.dt) + self.B * qh * qw * (4 if q.pool else 1) * Cb * self._esz(q.dt)
                               + need * 4 + (2 * self.B * gh * gw * Ca * self._esz(self.dt) if dual is not None else 0)),
                # the OPERANDS alone: dY + X read once, dW written once (what a weight gradient has to move whatever its algorithm)
                "operand_bytes": float(self.B * gh * gw * Ca * self._esz(p.dt) + self.B * qh * qw * (4 if q.pool else 1) * Cb * self._esz(q.dt)
                                       + Ca * Cb * len(taps) * 4)}
        post = []
        if collect is None:
            self._flush_reduce(ops)      # (the previous layer's slabs sit in the workspace this launch overwrites)
            self._emit(ops, self.lib.abc_wgrad, d, what, meta=meta)
        fused_rowsum = False
        if rowsum_to is not None and self.lib.abc_wgrad_rowsum_ok(C.byref(d)):
            # the heads' kernel also leaves per-split row sums of P = the conv's bias gradient (no separate pass over dL)
            rs = self.new((nsplit, ca_pad), torch.float32)
            d.rowsum_partial = rs.data_ptr()
            r2 = L.WgradReduceDesc()
            r2.partial, r2.nsplit, r2.ntaps, r2.Ca, r2.Cb, r2.Ca_pad, r2.Cb_pad, r2.dw, r2.accumulate = rs.data_ptr(), nsplit, 1, Ca, 1, ca_pad, 1, self.G(rowsum_to), 0
            fused_rowsum = True
        post.append((r, what + " reduce", (wname,) if wname else (),
                     {"ws": wsk, "kernel": "wgrad_reduce", "flops": 0, "bytes": float(need * 4 + Ca * Cb * len(taps) * 4)}))
        if fused_rowsum:
            post.append((r2, "dbias " + what[6:] + " reduce", (rowsum_to,),
                         {"kernel": "wgrad_reduce", "flops": 0, "bytes": float(nsplit * ca_pad * 4)}))
        if collect is not None:
            collect.append({"d": d, "what": what, "meta": meta, "post": post})
        else:
            for k, (rd, w, wr, m) in enumerate(post):
                if k == 0 and self.merge_reduce and self.train:
                    self.keep.append(rd)
                    self._pending_reduce = (ops, rd, w, wr, m)
                else:
                    self._emit(ops, self.lib.abc_wgrad_reduce, rd, w, writes=wr, meta=m)
        return "rowsum" if fused_rowsum else True

    def emit_wgrad_heads_batch(self, ops, items, what):
        """the heads' 1x1 weight gradients collected by emit_wgrad(collect=...) as ONE launch (abc_wgrad_heads_batch) when the
        heads kernel serves every one of them ((0, 0) tile), else one launch each; their slab reductions follow"""
        def tile(d):
            at_, bt_ = L.i32(), L.i32()
            L.check(self.lib.abc_wgrad_tile(C.byref(d), C.byref(at_), C.byref(bt_)), "wgrad_tile")
            return at_.value, bt_.value
        ok = 1 <= len(items) <= 8 and all(tile(it["d"]) == (0, 0) for it in items) and self.batched_heads
        if ok:
            arr = (L.WgradDesc * len(items))()
            for i, it in enumerate(items):
                arr[i] = it["d"]
            self.keep.append(arr)
            lib, n = self.lib, len(items)
            meta = {"kernel": "heads_wgrad_batch", "flops": sum(it["meta"]["flops"] for it in items), "bytes": sum(it["meta"]["bytes"] for it in items)}
            ops.append((lambda _r, st, a=arr: lib.abc_wgrad_heads_batch(a, n, st), None, what, (), meta))
        else:
            for it in items:
                self._emit(ops, self.lib.abc_wgrad, it["d"], it["what"], meta=it["meta"])
        posts = [p for it in items for p in it["post"]]
        self.keep.append([it["d"] for it in items])
        if ok and len(posts) <= 16:
            # the 8 weight-gradient and 8 bias row-sum reductions as one launch too
            rarr = (L.WgradReduceDesc * len(posts))()
            for i, (rd, _w, _wr, _m) in enumerate(posts):
                rarr[i] = rd
            self.keep.append(rarr)
            self.keep.append([rd for rd, _w, _wr, _m in posts])
            lib, nr = self.lib, len(posts)
            writes = tuple(w for _rd, _w, wr, _m in posts for w in wr)
            ops.append((lambda _r, st, a=rarr: lib.abc_wgrad_reduce_batch(a, nr, st), None, what + " reduce", writes,
                        {"kernel": "wgrad_reduce_batch", "flops": 0, "bytes": sum(m["bytes"] for _rd, _w, _wr, m in posts)}))
        else:
            for rd, w, wr, m in posts:
                self._emit(ops, self.lib.abc_wgrad_reduce, rd, w, writes=wr, meta=m)

    def emit_colsum(self, ops, t, dt, npix, ld, c_off, Cn, chan_scale, bname, what):
        out_ptr = self.G(bname)
        nb = self.lib.abc_colsum_blocks(npix)
        self._colsum_need = max(self._colsum_need, nb * Cn)
        args = [t.data_ptr(), dt, npix, ld, c_off, Cn, None if chan_scale is None else chan_scale.data_ptr(), None, out_ptr]
        self._colsum_users.append(args)
        lib = self.lib

        def fn(_ref, stream, a=args):
            return lib.abc_colsum(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], stream)

        ops.append((fn, None, what, (bname,), {"kernel": "colsum", "flops": 0, "bytes": float(npix * Cn * self._esz(dt))}))

    def emit_colsum_w1(self, ops, t, dt, npix, ld, c_off, Cn, img, bname, wname, what):
        """bias AND weight gradient of a 1x1 convolution over a one-channel f32 image in one pass over d(out) (abc_colsum_w1):
        unet2's first res_conv (unet2.py:62,135)"""
        nb = self.lib.abc_colsum_blocks(npix)
        self._colsum_need = max(self._colsum_need, 2 * nb * Cn)
        args = [t.data_ptr(), dt, npix, ld, c_off, Cn, img.data_ptr(), None, self.G(bname), self.G(wname)]
        self._colsum_users.append(args)
        lib = self.lib

        def fn(_ref, stream, a=args):
            return lib.abc_colsum_w1(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], stream)

        ops.append((fn, None, what, (bname, wname), {"kernel": "colsum_w1", "flops": 0, "bytes": float(npix * (Cn * self._esz(dt) + 4))}))

    # ------------------------------------------------------------------ layers
    def conv_bn(self, cname, bname, src: Src, cout, k, dst, slope, recname=None, stats_rows=2, force_stats=False, stat_out=None,
                collect_fin=None, shared=None):
        """conv (+bias) -> raw output into dst=(tensor, coef, H, W, ld, coff); BN stats/coefficients.
        stat_out = (mean, invstd) tensors to use instead of fresh ones (slices of a shared array: the heads);
        collect_fin: a list -- the train-mode finalisation is appended as (desc, what) instead of being emitted;
        shared = (stats, nblk): the convolution itself was emitted by the caller as part of a wider one whose statistics
        partials are [nblk][rows][ld] -- this layer's columns start at coff"""
        yt, coef, H, W, ld, coff = dst
        cin = src.C
        taps = taps_square(k)
        rows_pad = -(-cout // 32) * 32
        if self.fold:
            return self._conv_bn_folded(cname, bname, src, cout, k, dst, slope, shared)
        if shared is not None:
            stats, nblk = shared
        else:
            wf = self.packed(len(taps), cin, rows_pad)
            self.emit_pack(cname + ".weight", wf, 0, cout, cin, k, rows_pad, cin)
            stats, nblk = self.emit_conv(self.fwd_ops, src, wf, self.P(cname + ".bias"), yt, self.dt, H, W, ld, coff, cout, taps,
                                         stats=self.train or force_stats, what="fwd " + cname, stats_rows=stats_rows)
        rec = Rec(kind="conv", cname=cname, bname=bname, src=src, cin=cin, cout=cout, k=k, taps=taps, y=yt, H=H, W=W, ld=ld,
                  coff=coff, coef=coef, slope=slope)
        sc, sh, sl = coef
        sl[coff:coff + cout] = slope
        rec.scale, rec.shift, rec.slopes = sc[coff:coff + cout], sh[coff:coff + cout], sl[coff:coff + cout]
        if stat_out is not None:
            rec.mean, rec.invstd = stat_out
        else:
            rec.mean, rec.invstd = self.new((cout,), torch.float32), self.new((cout,), torch.float32, 1.0)
        if self.train:
            d = L.BnFwdDesc()
            d.partial, d.nblk, d.C, d.count, d.rows = stats.data_ptr() + (4 * coff if shared is not None else 0), nblk, cout, float(self.B * H * W), stats_rows
            d.gamma, d.beta = self.P(bname + ".weight"), self.P(bname + ".bias")
            d.scale, d.shift, d.mean, d.invstd = rec.scale.data_ptr(), rec.shift.data_ptr(), rec.mean.data_ptr(), rec.invstd.data_ptr()
            d.running_mean, d.running_var = self.Bf(bname + ".running_mean"), self.Bf(bname + ".running_var")
            d.num_batches_tracked = self.Cn(bname + ".num_batches_tracked")
            d.eps, d.momentum = BN_EPS, BN_MOM
            if collect_fin is not None:
                self.keep.append(d)
                collect_fin.append((d, "bn " + bname))
            else:
                self._emit(self.fwd_ops, self.lib.abc_bn_finalize_fwd, d, "bn " + bname)
        else:
            lib = self.lib
            a = (self.P(bname + ".weight"), self.P(bname + ".bias"), self.Bf(bname + ".running_mean"),
                 self.Bf(bname + ".running_var"), rec.scale.data_ptr(), rec.shift.data_ptr(), cout, BN_EPS)
            # eval-mode coefficients are functions of the parameters alone: they are refreshed with the weight packing
            # (every call on the module path, once per weight load in InferenceRunner), not inside the forward plan
            self.pack_ops.append((lambda _r, st, a=a: lib.abc_bn_eval_coeffs(*a, st), None, "bn-eval " + bname, (), {"kernel": "bn_eval", "flops": 0, "bytes": 0}))
        rec.stats, rec.nblk = stats, nblk
        self.recs.append(rec)
        out = Src(yt, self.dt, H, W, ld, coff, cout, coef=coef, producer=rec)
        return rec, out

    def _fold_coeffs(self, cname, bname, cout, scale_t, bias_t):
        """pack-time op: scale_t = gamma / sqrt(running_var + eps), bias_t = (conv bias - running_mean) * scale_t + beta"""
        lib = self.lib
        a = (self.P(bname + ".weight"), self.P(bname + ".bias"), self.Bf(bname + ".running_mean"), self.Bf(bname + ".running_var"),
             self.P(cname + ".bias"), scale_t.data_ptr(), bias_t.data_ptr(), cout, BN_EPS)
        self.pack_ops.append((lambda _r, st, a=a: lib.abc_bn_eval_fold(*a, st), None, "bn-fold " + bname, (),
                              {"kernel": "bn_fold", "flops": 0, "bytes": 0}))

    def _conv_bn_folded(self, cname, bname, src, cout, k, dst, slope, shared):
        """eval-mode conv + BatchNorm + activation as ONE convolution (see __init__): returns (rec, Src with no transform)"""
        yt, coef, H, W, ld, coff = dst
        cin = src.C
        taps = taps_square(k)
        rows_pad = -(-cout // 32) * 32
        y_dt = self._out_dt(cname) if shared is None else self.dt
        q_out = None
        if shared is None:
            fs, fb = self.new((cout,), torch.float32, 1.0), self.new((cout,), torch.float32)
            self._fold_coeffs(cname, bname, cout, fs, fb)
            in_f8 = src.dt == L.FP8
            if y_dt == L.FP8:
                # per-tensor scale of this convolution's e4m3 output: (amax, s, 1 / s) on the device, set by calibrate_fp8()
                q_out = (self.new((1,), torch.float32, 0.0), self.new((1,), torch.float32, 1.0), self.new((1,), torch.float32, 1.0))
            if in_f8:
                qmul, deq = self._fp8_weight_scales(cname + ".weight", cout, cin * k * k, fs, src.q[1], self.new((cout,), torch.float32, 1.0),
                                                    self.new((cout,), torch.float32, 1.0))
                wf = self.packed(len(taps), cin, rows_pad, cdt=L.FP8)
                self.emit_pack(cname + ".weight", wf, 0, cout, cin, k, rows_pad, cin, row_scale=qmul.data_ptr(), cdt=L.FP8)
                self.emit_conv(self.fwd_ops, src, wf, fb.data_ptr(), yt, y_dt, H, W, ld, coff, cout, taps, what="fwd " + cname,
                               out_slope=slope, cdt=L.FP8, out_scale=deq, out_quant=None if q_out is None else q_out[2])
            else:
                wf = self.packed(len(taps), cin, rows_pad)
                self.emit_pack(cname + ".weight", wf, 0, cout, cin, k, rows_pad, cin, row_scale=fs.data_ptr())
                self.emit_conv(self.fwd_ops, src, wf, fb.data_ptr(), yt, y_dt, H, W, ld, coff, cout, taps, what="fwd " + cname,
                               out_slope=slope, out_quant=None if q_out is None else q_out[2])
        rec = Rec(kind="conv", cname=cname, bname=bname, src=src, cin=cin, cout=cout, k=k, taps=taps, y=yt, H=H, W=W, ld=ld,
                  coff=coff, coef=None, slope=slope)
        rec.fold_desc = self._last_conv_desc if shared is None else None
        rec.q = q_out
        if q_out is not None:
            self.fp8_recs.append(rec)
        self.recs.append(rec)
        out = Src(yt, y_dt, H, W, ld, coff, cout, coef=None, producer=rec)
        out.q = q_out
        return rec, out

    def _fp8_weight_scales(self, wname, rows, K, fold, s_in, qmul, deq):
        """pack-time op (before the weight packing): per output row the e4m3 scale of the BatchNorm-folded weight --
        qmul = fold / s_w (what the packing multiplies the master weight with), deq = s_w * s_in (the convolution's out_scale)"""
        lib = self.lib
        a = (self.P(wname), rows, K, None if fold is None else fold.data_ptr(), s_in.data_ptr(), qmul.data_ptr(), deq.data_ptr())
        self.keep += [fold, s_in, qmul, deq]
        self.pack_ops.append((lambda _r, st, a=a: lib.abc_fp8_weight_scales(*a, st), None, "fp8 scales " + wname, (),
                              {"kernel": "fp8_weight_scales", "flops": 0, "bytes": 0}))
        return qmul, deq

    def calibrate_fp8(self, ref, stream, margin=1.0):
        """per-tensor activation scales of the e4m3 tensors from the bf16 folded graph `ref` (same model, same batch, forward
        already run on `stream`): s = max|x| * margin / 448.  All on the device; call run_pack() afterwards (the weight
        scales fold s_in in)."""
        lib = self.lib
        mine = {r.cname: r for r in self.fp8_recs}
        theirs = {r.cname: r for r in ref.recs if r.kind == "conv" and r.cname in mine}
        if set(mine) != set(theirs):
            raise ValueError("calibrate_fp8: the reference graph lacks %s" % sorted(set(mine) - set(theirs)))
        for cname, r in mine.items():
            t = theirs[cname]
            if t.y.dtype != torch.bfloat16 or (t.ld, t.coff) != (t.cout, 0):
                raise ValueError("calibrate_fp8: %s of the reference graph is not a plain bf16 tensor" % cname)
            amax, s, inv_s = r.q
            L.check(lib.abc_fill_f32(amax.data_ptr(), 0.0, 1, stream), "fill")
            L.check(lib.abc_absmax(t.y.data_ptr(), L.BF16, t.y.numel(), amax.data_ptr(), stream), "absmax")
            L.check(lib.abc_fp8_act_scale(amax.data_ptr(), margin, s.data_ptr(), inv_s.data_ptr(), stream), "fp8_act_scale")
        if self.hfeat_q is not None:      # the eight heads' features: 128 columns each of one tensor, each head its own scale
            amax, s, inv_s = self.hfeat_q
            nh = amax.numel()
            npx = ref.hfeat.numel() // (128 * nh)
            L.check(lib.abc_fill_f32(amax.data_ptr(), 0.0, nh, stream), "fill")
            for i in range(nh):
                L.check(lib.abc_absmax_cols(ref.hfeat.data_ptr(), L.BF16, npx, 128 * nh, 128 * i, 128, amax.data_ptr() + 4 * i, stream), "absmax_cols")
                L.check(lib.abc_fp8_act_scale(amax.data_ptr() + 4 * i, margin, s.data_ptr() + 4 * i, inv_s.data_ptr() + 4 * i, stream), "fp8_act_scale")
        self.fp8_calibrated = True

    def double_conv(self, prefix, src, cout, k, dst_b=None):
        H, W = src.lh()
        p = prefix + ".double_conv"
        ta, ca = self.act_buf(H, W, cout, self._out_dt(p + ".0") if self.fold else None)
        _, a = self.conv_bn(p + ".0", p + ".1", src, cout, k, (ta, ca, H, W, cout, 0), 0.0)
        if dst_b is None:
            tb, cb = self.act_buf(H, W, cout, self._out_dt(p + ".3") if self.fold else None)
            dst_b = (tb, cb, H, W, cout, 0)
        _, b = self.conv_bn(p + ".3", p + ".4", a, cout, k, dst_b, 0.0)
        return b

    def _stem_fused_double_conv(self, prefix, img: Src, cout):
        """folded inference graph, one input channel: DoubleConv's first convolution (+ BatchNorm + ReLU, unet.py:12-14) is
        computed inside the halo staging of its second one (abc_conv_desc.stem_*): the full-resolution 16-channel tensor
        between them is never written.  Returns None (and leaves the plan untouched) when the narrow-level kernel does not
        serve the descriptor (e.g. B*H*W*16*2 bytes >= 2^31): the caller then emits the ordinary folded DoubleConv"""
        H, W = img.H, img.W
        p = prefix + ".double_conv"
        undo = [(lst, len(lst)) for lst in (self.pack_ops, self._pack_descs, self.fwd_ops, self.recs)]
        fs0, fb0 = self.new((cout,), torch.float32, 1.0), self.new((cout,), torch.float32)
        self._fold_coeffs(p + ".0", p + ".1", cout, fs0, fb0)
        tb, cb = self.act_buf(H, W, cout)
        virt = Src(self.img, self.dt, H, W, cout, 0, cout, coef=None)      # (never read: the kernel computes it from the image)
        rec, b = self._conv_bn_folded(p + ".3", p + ".4", virt, cout, 3, (tb, cb, H, W, cout, 0), 0.0, None)
        d = rec.fold_desc
        d.stem_x, d.stem_w = self.img.data_ptr(), self.P(p + ".0.weight")
        d.stem_scale, d.stem_bias, d.stem_slope = fs0.data_ptr(), fb0.data_ptr(), 0.0
        if self.lib.abc_conv_variant(C.byref(d)) != 5:
            for lst, n in undo:
                del lst[n:]
            return None
        return b

    def pooled(self, s: Src):
        """nn.MaxPool2d(2) (unet.py:30, unet2.py:83): materialised once by abc_pool_act, so that the level's first conv, its
        weight gradient (and, in unet2, the block's residual branch) read a plain tensor on their prefetch paths; the
        gradient still routes to the producer as a pooled one (via_pool).  (unet2 pooled on load at first: its first
        conv of every level and that conv's weight gradient then ran on the general loaders -- 0.37 ms for down1 alone.)"""
        Ho, Wo = s.H // 2, s.W // 2
        out = self.new((self.B, Ho, Wo, s.C))
        # folded inference graph: a producer on the narrow-level kernel writes the pooled tensor as a second output
        # (abc_conv_desc.pool_y) -- no separate pass over the full-resolution tensor
        d = getattr(s.producer, "fold_desc", None) if self.fold else None
        if d is not None and s.coef is None and s.C % 8 == 0 \
                and self.lib.abc_conv_variant(C.byref(d)) == 5 and d.y == s.t.data_ptr() and d.cout_off == s.coff and d.Cout == s.C:
            d.pool_y, d.ld_pool = out.data_ptr(), s.C
            r = Src(out, self.dt, Ho, Wo, s.C, 0, s.C, coef=None, pool=False, producer=s.producer)
            r.via_pool = True
            return r
        a = L.ActSrc()
        s.fill(a)
        lib = self.lib
        args = (a, s.dt, s.coff, s.C, self.B, out.data_ptr(), self.dt, s.C)
        self.fwd_ops.append((lambda _r, st, g=args: lib.abc_pool_act(C.byref(g[0]), *g[1:], st), None, "pool", (),
                             {"kernel": "pool_act", "flops": 0, "bytes": float(self.B * s.H * s.W * s.C * self._esz(s.dt) * 1.25)}))
        r = Src(out, self.dt, Ho, Wo, s.C, 0, s.C, coef=None, pool=False, producer=s.producer)
        r.via_pool = True
        return r

    def up(self, name, low: Src, cat, cat_coef, Hs, Ws, Ctot, cout, skip_producer=None):
        """ConvTranspose2d(Ctot -> Ctot/2, k3, s2) of `low` into cat[..., Ctot/2:], then DoubleConv(Ctot -> cout)"""
        half = Ctot // 2
        cin = low.C
        lh, lw = low.lh()
        # unet.py:51-56: the 2n+1 outputs are padded by diff = s - (2n+1), i.e. the first row / column is cropped where the
        # skip tensor has 2n rows (always, for input sizes that are multiples of 32) and nothing where it has 2n+1
        if Hs - 2 * lh not in (0, 1) or Ws - 2 * lw not in (0, 1):
            raise ValueError("skip tensor %dx%d does not fit the transposed conv of %dx%d" % (Hs, Ws, lh, lw))
        crop_y, crop_x = Hs == 2 * lh, Ws == 2 * lw
        rows_pad = -(-half // 32) * 32
        phases, items = [], []
        # all four phases in ONE pass over the input where the library serves it (abc_convt_fused_fwd: bf16, both axes cropped): the nine
        # weight slices of the phases one after the other in one buffer, fragment-contiguous (abc_pack_desc.layout 1)
        fused = None
        if self.fused_convt and crop_y and crop_x and self.dt == L.BF16 and low.dt == L.BF16:
            dT = L.ConvTDesc()
            low.fill(dT.src)
            dT.bias, dT.y, dT.dtype = self.P(name + ".up.bias"), cat.data_ptr(), L.BF16
            dT.B, dT.Hin, dT.Win, dT.cin_off, dT.Cin = self.B, lh, lw, low.coff, cin
            dT.Hout, dT.Wout, dT.ldy, dT.cout_off, dT.Cout, dT.Cout_pad = Hs, Ws, Ctot, half, half, rows_pad
            if self.lib.abc_convt_fused_ok(C.byref(dT)):
                fused = dT
                wall = self.packed(9, cin, rows_pad)
                dT.w = wall.data_ptr()
                slice_elems = wall.numel() // 9
                first_slice = {(0, 0): 0, (0, 1): 1, (1, 0): 3, (1, 1): 5}
        for py in (0, 1):
            for px in (0, 1):
                taps = convT_phase_taps(py, px, crop_y, crop_x)
                if fused is not None:
                    wp = wall[first_slice[(py, px)] * slice_elems:(first_slice[(py, px)] + len(taps)) * slice_elems]
                    self._w_layout[wp.data_ptr()] = 1
                    self.emit_pack(name + ".up.weight", wp, 2, half, cin, 3, rows_pad, cin, py=convT_pack_parity(py, crop_y),
                                   px=convT_pack_parity(px, crop_x))
                    phases.append(wp)
                    continue
                wp = self.packed(len(taps), cin, rows_pad)
                self.emit_pack(name + ".up.weight", wp, 2, half, cin, 3, rows_pad, cin, py=convT_pack_parity(py, crop_y),
                               px=convT_pack_parity(px, crop_x))
                # output rows 2a + py < Hs: without the crop the even parity has one row more than the input
                gh, gw = (Hs - py + 1) // 2, (Ws - px + 1) // 2
                self.emit_conv(self.fwd_ops, low, wp, self.P(name + ".up.bias"), cat, self.dt, Hs, Ws, Ctot, half, half, taps,
                               grid=(gh, gw), om=2, oy0=py, ox0=px, what="fwd %s.up phase %d%d" % (name, py, px), collect=items)
                phases.append(wp)
        if fused is not None:
            self.keep.append(fused)
            lib = self.lib
            meta = {"kernel": "convt_fused<bf16>", "flops": 2.0 * self.B * lh * lw * cin * half * 9,
                    "bytes": float(self.B * lh * lw * cin * 2 + self.B * Hs * Ws * half * 2)}
            self.fwd_ops.append((lambda _r, st, d=fused: lib.abc_convt_fused_fwd(C.byref(d), st), None, "fwd %s.up (4 phases, one pass)" % name, (), meta))
        else:
            self.emit_conv_batch(self.fwd_ops, items, "fwd %s.up (4 phases)" % name)
        rec = Rec(kind="convT", cname=name + ".up", src=low, cin=cin, cout=half, H=Hs, W=Ws, ld=Ctot, coff=half, y=cat,
                  taps_bwd=convT_dgrad_taps(crop_y, crop_x))
        self.recs.append(rec)
        cat_src = Src(cat, self.dt, Hs, Ws, Ctot, 0, Ctot, coef=None if self.fold else cat_coef, producer=("cat", None))
        if self.variant == "unet2":
            cat_src.cat = (skip_producer, rec)
            self.units2.append(("convT", rec))
            out = self.block2(name + ".conv", cat_src, cout, 3)
            return out, rec
        out = self.double_conv(name + ".conv", cat_src, cout, 3)
        # remember who receives the two halves of d(cat)
        first = [r for r in self.recs if r.kind == "conv" and r.cname == name + ".conv.double_conv.0"][0]
        first.cat_upper = rec
        return out, rec

    # ------------------------------------------------------------------ build
    def _image_src(self):
        """the input image, NCHW f32 as the reference passes it (unet.py:100).  One channel (train.py:47) is at the same
        time NHWC with a pixel stride of 1 and takes the dedicated first-layer kernels; more channels (unet.py:122-134's
        self-check uses 3) are read channel-planar by the general loaders"""
        cin = self.in_channels
        self.img = self.new((self.B, cin, self.H, self.W), torch.float32)
        if cin == 1:
            return Src(self.img, L.F32, self.H, self.W, 1, 0, 1)
        return Src(self.img, L.F32, self.H, self.W, 0, 0, cin, planar=True)

    def _build(self):
        if self.variant == "unet2":
            return self._build2()
        B, H, W = self.B, self.H, self.W
        S = [(H >> i, W >> i) for i in range(6)]
        img_src = self._image_src()
        if self.drop_p > 0:
            lib, sp = self.lib, self.drop_salt.data_ptr()
            self.fwd_ops.append((lambda _r, st: lib.abc_counter_add_u32(sp, DROP_STEP, st), None, "dropout step", (),
                                 {"kernel": "counter_add", "flops": 0, "bytes": 0}))
        x = None
        if self.fold and self.dt == L.BF16 and self.in_channels == 1:
            x = self._stem_fused_double_conv("inc1", img_src, 16)
        if x is None:
            x = self.double_conv("inc1", img_src, 16, 3)
        x1 = self.double_conv("inc2", x, 16, 3)
        x2 = self.double_conv("down1.maxpool_conv.1", self.pooled(x1), 32, 3)
        x = self.double_conv("down2.maxpool_conv.1", self.pooled(x2), 64, 3)
        cat3, cc3 = self.act_buf(S[2][0], S[2][1], 128)
        cat2, cc2 = self.act_buf(S[3][0], S[3][1], 256)
        cat1, cc1 = self.act_buf(S[4][0], S[4][1], 512)
        x3 = self.double_conv("inc3", x, 64, 3, dst_b=(cat3, cc3, S[2][0], S[2][1], 128, 0))
        x4 = self.double_conv("down3.maxpool_conv.1", self.pooled(x3), 128, 3, dst_b=(cat2, cc2, S[3][0], S[3][1], 256, 0))
        x5 = self.double_conv("down4.maxpool_conv.1", self.pooled(x4), 256, 3, dst_b=(cat1, cc1, S[4][0], S[4][1], 512, 0))
        x6 = self.double_conv("down5.maxpool_conv.1", self.pooled(x5), 512, 3)
        u, _ = self.up("up1", x6, cat1, cc1, S[4][0], S[4][1], 512, 256)
        u, _ = self.up("up2", u, cat2, cc2, S[3][0], S[3][1], 256, 128)
        u, _ = self.up("up3", u, cat3, cc3, S[2][0], S[2][1], 128, 128)
        u = self.double_conv("dconv1", u, 128, 3)
        trunk = self.double_conv("dconv2", u, 128, 3)
        self.trunk = trunk
        self._build_heads(trunk)
        if self.train:
            self._build_backward()
            self._flush_reduce(self.bwd_ops)
        self._finish_build()

    def _finish_build(self):
        # all weight re-packing of a step as ONE table-driven launch
        lib = self.lib
        isz = lib.abc_pack_item_bytes()
        host = (C.c_char * (isz * len(self._pack_descs)))()
        first, elems = 0, 0
        for i, pd in enumerate(self._pack_descs):
            pd.layout = self._w_layout.get(pd.dst, 0)
            n = lib.abc_pack_item_fill(C.addressof(host) + i * isz, C.byref(pd), first)
            if n < 0:
                L.check(-1, "pack_item_fill")
            first += n      # (0 for the items the source-major tile kernel packs: they take no range of the dest-major kernel)
            ntaps = {0: pd.kh * pd.kw, 1: pd.kh * pd.kw, 2: (2 if pd.py else 1) * (2 if pd.px else 1), 3: 9}[pd.mode]
            elems += ntaps * pd.red_pad * pd.rows_pad
        table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.dev)
        self.keep.append(table)
        a = (table.data_ptr(), len(self._pack_descs), first)
        self.pack_ops.append((lambda _r, st, a=a: lib.abc_pack_batch(a[0], a[1], a[2], st), None, "pack weights", (),
                              {"kernel": "pack_batch", "flops": 0, "bytes": float(elems * (2 if self.dt == L.BF16 else 4) + elems * 4)}))
        # shared workspaces
        self.ws = [self.new((max(self._ws_need, 4),), torch.float32)]
        for d, k in self._ws_users:
            d.partial = self.ws[k].data_ptr()
        self.cs_ws = self.new((max(self._colsum_need, 4),), torch.float32)
        for a in self._colsum_users:
            a[7] = self.cs_ws.data_ptr()

    def _build_heads(self, trunk: Src):
        h, w = trunk.H, trunk.W
        nh = len(self.heads)
        self.h, self.w = h, w
        # (fp8 graph: the heads' features are e4m3 too when the merged conv1 and the heads' own 1x1 kernel serve them)
        f8_feat = self.fp8 and trunk.dt == L.FP8 and trunk.C == 128 and nh <= 8 and self.batched_heads and (h * w) % 64 == 0
        # folded graph: the heads' 1x1 convolutions in the epilogue of the merged conv1 (abc_conv_desc.heads_epi): no feature tensor
        self.hepi = None
        if self.fold and self.heads_epilogue and self.batched_heads and self.dt == L.BF16 and trunk.C == 128 and nh <= 8 and \
                (trunk.dt == L.BF16 or f8_feat):
            self.hepi = torch.zeros(nh * C.sizeof(L.HeadsEpi), dtype=torch.uint8, device=self.dev)
            self.keep.append(self.hepi)
        if self.hepi is not None:
            # (a placeholder: with heads_epi the convolution does not write its y)
            self.hfeat = self.new((1, 1, 1, 128 * nh), self._tdt(L.FP8) if f8_feat else None)
            self.hcoef = None
        else:
            self.hfeat, self.hcoef = self.act_buf(h, w, 128 * nh, L.FP8 if f8_feat else None)
        if f8_feat:
            self.hfeat_q = (self.new((nh,), torch.float32, 0.0), self.new((nh,), torch.float32, 1.0), self.new((nh,), torch.float32, 1.0))
        epi_items = []
        # the list forward() returns: one contiguous NCHW f32 map per head (unet.py:119), written directly
        self.logits = [self.new((self.B, hc, h, w), torch.float32) for hc in self.heads]
        self.head_recs, self.head2 = [], []
        head_convs, head_fins = [], []
        # batch statistics of the eight heads' BatchNorms side by side (one act_bwd pass over all 8 x 128 channels)
        self.hmean, self.hinvstd = self.new((128 * nh,), torch.float32), self.new((128 * nh,), torch.float32, 1.0)
        batch_fin = nh <= 8 and self.batched_heads
        # the eight conv1's (unet.py:66,116-118: the same 128-channel trunk into 8 x 128 channels) as ONE 128 -> 8 x 128
        # convolution: the input halo tile is shared by the 8 n-blocks of a pixel tile (same XCD, its L2), and 8 x 768 tiles
        # fill the 512 workgroup slots 12.0 times instead of 8 x 1.5
        shared = None
        if batch_fin and self.dt == L.BF16 and trunk.C == 128:
            shared = self._heads_conv1_merged(trunk, h, w)
        fused = self.want_fused_heads and shared is not None and self.heads == [1, 14, 3, 2, 1, 360, 60, 60] and (h * w) % 128 == 0 \
            and self.B * h * w * 128 * nh < (1 << 30)
        for i, hc in enumerate(self.heads):
            p = "out_modules.%d" % i
            rec, f = self.conv_bn(p + ".conv1", p + ".bn", trunk, 128, 3, (self.hfeat, self.hcoef, h, w, 128 * nh, 128 * i), 0.01,
                                  stat_out=(self.hmean[128 * i:128 * (i + 1)], self.hinvstd[128 * i:128 * (i + 1)]),
                                  collect_fin=head_fins if batch_fin else None, shared=shared)
            rec.is_head = True
            self.head_recs.append(rec)
            f.drop_p, f.drop_seed, f.drop_salt = self.drop_p, self.drop_seed, self.drop_salt
            if not fused:
                rows_pad = -(-hc // 32) * 32
                if self.hfeat_q is not None:
                    f.dt, f.q = L.FP8, tuple(t[i:i + 1] for t in self.hfeat_q)
                    qmul, deq = self._fp8_weight_scales(p + ".conv2.weight", hc, 128, None, f.q[1], self.new((hc,), torch.float32, 1.0),
                                                        self.new((hc,), torch.float32, 1.0))
                    w2 = self.packed(1, 128, rows_pad, cdt=L.FP8)
                    self.emit_pack(p + ".conv2.weight", w2, 0, hc, 128, 1, rows_pad, 128, row_scale=qmul.data_ptr(), cdt=L.FP8)
                    if self.hepi is not None:
                        epi_items.append((w2, self.P(p + ".conv2.bias"), deq, self.logits[i], hc, rows_pad))
                    else:
                        self.emit_conv(self.fwd_ops, f, w2, self.P(p + ".conv2.bias"), self.logits[i], L.F32, h, w, hc, 0, hc,
                                       [(0, 0)], what="fwd %s.conv2" % p, planar_out=True, collect=head_convs, cdt=L.FP8, out_scale=deq)
                    self.head2.append(Rec(kind="head2", cname=p + ".conv2", src=f, cout=hc, idx=i))
                    continue
                w2 = self.packed(1, 128, rows_pad)
                self.emit_pack(p + ".conv2.weight", w2, 0, hc, 128, 1, rows_pad, 128)
                if self.hepi is not None:
                    epi_items.append((w2, self.P(p + ".conv2.bias"), None, self.logits[i], hc, rows_pad))
                else:
                    self.emit_conv(self.fwd_ops, f, w2, self.P(p + ".conv2.bias"), self.logits[i], L.F32, h, w, hc, 0, hc,
                                   [(0, 0)], what="fwd %s.conv2" % p, planar_out=True, collect=head_convs)
            self.head2.append(Rec(kind="head2", cname=p + ".conv2", src=f, cout=hc, idx=i))
        if head_fins:
            arr = (L.BnFwdDesc * len(head_fins))()
            for i, (d, _w) in enumerate(head_fins):
                arr[i] = d
            self.keep.append(arr)
            lib, n = self.lib, len(head_fins)
            pstride = 128 * nh if shared is not None else 0
            self.fwd_ops.append((lambda _r, st, a=arr: lib.abc_bn_finalize_fwd_batch(a, n, pstride, st), None, "bn out_modules.*.bn", (),
                                 {"kernel": "bn", "flops": 0, "bytes": 0}))
        if fused:
            self._heads_fused_setup()
            return
        if self.hepi is not None:
            # the table the merged conv1 reads in its epilogue: static pointers, written once
            host = (L.HeadsEpi * nh)()
            for i, (w2, bias_ptr, deq, lg, hc, rows_pad) in enumerate(epi_items):
                host[i].w2, host[i].bias, host[i].oscale = w2.data_ptr(), bias_ptr, (None if deq is None else deq.data_ptr())
                host[i].y, host[i].Cout, host[i].Cout_pad = lg.data_ptr(), hc, rows_pad
            self.hepi.copy_(torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8))
            self.keep.append(epi_items)
            return
        # (the eight conv1 launches above write the eight slices of hfeat; the eight 1x1 convolutions go as one launch)
        if self.nms_heads and self.heads == [1, 14, 3, 2, 1, 360, 60, 60] and len(head_convs) == 8:
            d6, d7 = head_convs[6][0], head_convs[7][0]
            rho, om = self.new((self.B, 60, h, w), torch.float32), self.new((self.B, 60, h, w), torch.float32)
            d6.head_aux, d6.head_aux_mode = rho.data_ptr(), 1
            d7.head_aux, d7.head_aux_mode = om.data_ptr(), 2
            if self.lib.abc_conv_variant(C.byref(d6)) == 3 and self.lib.abc_conv_variant(C.byref(d7)) == 3:
                self.nms_rho, self.nms_omega = rho, om
                head_convs[7][2]["bytes"] += float(self.B * 60 * h * w * 4)
                head_convs[6][2]["bytes"] += float(self.B * 60 * h * w * 4)
                if self.decode:
                    d5 = head_convs[5][0]
                    idx = self.new((self.B, 60, h, w), torch.uint8)
                    y5, y6 = d5.y, d6.y
                    d5.head_aux, d5.head_aux_mode, d5.y, d6.y = idx.data_ptr(), 3, None, None
                    if self.lib.abc_conv_variant(C.byref(d5)) == 3 and self.lib.abc_conv_variant(C.byref(d6)) == 3:
                        self.btype_idx = idx
                        head_convs[5][2]["bytes"] -= float(self.B * 360 * h * w * 4 - self.B * 60 * h * w)
                        head_convs[6][2]["bytes"] -= float(self.B * 60 * h * w * 4)
                        # (the bond-type and rho maps are neither written nor read in decode mode: release them -- 1.7 GB at b64 @ 512 x 512)
                        for dead in (self.logits[5], self.logits[6]):
                            self.keep[:] = [t for t in self.keep if t is not dead]
                        self.logits[5] = self.logits[6] = None
                    else:
                        d5.head_aux, d5.head_aux_mode, d5.y, d6.y = None, 0, y5, y6
                        self.decode = False
            else:
                d6.head_aux = d7.head_aux = None
                d6.head_aux_mode = d7.head_aux_mode = 0
        self.emit_heads_batch(self.fwd_ops, head_convs, 0, "fwd out_modules.*.conv2")

    def _heads_fused_setup(self):
        """descriptor + buffers of the fused heads pass (abc_heads_fused_*): the conv2 forward leaves the forward plan -- the
        Trainer runs abc_heads_fused_fwd_bwd + abc_loss_finalize between forward and backward (ops.FusedHeadsLoss, which also
        binds the targets) -- and the backward plan starts from its outputs (_heads_backward)"""
        lib, nh, B, h, w = self.lib, len(self.heads), self.B, self.h, self.w
        Ct = 128 * nh
        d = L.HeadsFusedDesc()
        sc, sh, sl = self.hcoef
        d.feat, d.ld = self.hfeat.data_ptr(), Ct
        d.scale, d.shift, d.slope = sc.data_ptr(), sh.data_ptr(), sl.data_ptr()
        d.mean, d.invstd = self.hmean.data_ptr(), self.hinvstd.data_ptr()
        d.drop_p, d.drop_seed = self.drop_p, self.drop_seed
        d.drop_salt = self.drop_salt.data_ptr() if self.drop_p > 0 else None
        d.B, d.h, d.w = B, h, w
        nchunk = lib.abc_heads_fused_chunks(C.byref(d))
        self.hf_pack = self.new((lib.abc_heads_fused_pack_bytes() // 4 + 4,), torch.float32)
        self.hf_dl = self.new((lib.abc_heads_fused_dl_elems(C.byref(d)),), torch.bfloat16)
        self.hf_g = self.new((B, h, w, Ct))
        self.hf_bnpart = self.new((nchunk, 2, Ct), torch.float32)
        self.hf_lossblocks = lib.abc_heads_fused_loss_blocks(C.byref(d))
        self.hf_losspart = torch.zeros((self.hf_lossblocks, 16), dtype=torch.float64, device=self.dev)
        self.chan_scale = self.new((sum(self.heads),), torch.float32, 0.0)
        self.hf_work = self.new((lib.abc_heads_fused_wgrad_floats(C.byref(d)),), torch.float32)
        d.w2_pack, d.dl, d.g = self.hf_pack.data_ptr(), self.hf_dl.data_ptr(), self.hf_g.data_ptr()
        d.bn_partial, d.loss_partial = self.hf_bnpart.data_ptr(), self.hf_losspart.data_ptr()
        d.chan_scale, d.wgrad_work = self.chan_scale.data_ptr(), self.hf_work.data_ptr()
        if self.drop_p > 0:
            # the dropout keep bits of the three wide heads' features, from the fused pass to its conv2 weight gradient
            self.hf_keep = self.new((3 * B * h * w * 16,), torch.uint8, 0)
            d.keep_mask = self.hf_keep.data_ptr()
        for i in range(nh):
            p = "out_modules.%d.conv2" % i
            d.w2[i], d.b2[i], d.logits[i] = self.P(p + ".weight"), self.P(p + ".bias"), self.logits[i].data_ptr()
            d.dw2[i], d.db2[i], d.chan_off[i] = self.G(p + ".weight"), self.G(p + ".bias"), self.head_off[i]
        self.hf, self.hf_chunks = d, nchunk
        self.pack_ops.append((lambda _r, st: lib.abc_heads_fused_pack(C.byref(d), st), None, "pack out_modules.*.conv2", (),
                              {"kernel": "heads_fused_pack", "flops": 0, "bytes": 0}))

    def _heads_conv1_merged(self, trunk: Src, h, w):
        """pack the eight conv1 weights one below the other ([tap][chunk][8 x 128][CK]), gather their biases, emit the one
        convolution into hfeat; returns (stats partials [nblk][2][8 x 128] or None, nblk)"""
        nh = len(self.heads)
        Ct = 128 * nh
        taps = taps_square(3)
        wf = self.packed(len(taps), 128, Ct)
        bias_all = self.new((Ct,), torch.float32)
        if self.fold:
            scale_all = self.new((Ct,), torch.float32, 1.0)
            f8 = trunk.dt == L.FP8      # fp8 graph: e4m3 trunk and weights, bf16 features for the heads' 1x1 convolutions
            if f8:
                wf = self.packed(len(taps), 128, Ct, cdt=L.FP8)
                qmul_all, deq_all = self.new((Ct,), torch.float32, 1.0), self.new((Ct,), torch.float32, 1.0)
            for i in range(nh):
                p = "out_modules.%d" % i
                sl = slice(128 * i, 128 * (i + 1))
                self._fold_coeffs(p + ".conv1", p + ".bn", 128, scale_all[sl], bias_all[sl])
                if f8:
                    self._fp8_weight_scales(p + ".conv1.weight", 128, 128 * 9, scale_all[sl], trunk.q[1], qmul_all[sl], deq_all[sl])
                    self.emit_pack(p + ".conv1.weight", wf, 0, 128, 128, 3, 128, 128, rows_total=Ct, rows_off=128 * i,
                                   row_scale=qmul_all[sl].data_ptr(), cdt=L.FP8)
                else:
                    self.emit_pack(p + ".conv1.weight", wf, 0, 128, 128, 3, 128, 128, rows_total=Ct, rows_off=128 * i,
                                   row_scale=scale_all[sl].data_ptr())
            self.emit_conv(self.fwd_ops, trunk, wf, bias_all.data_ptr(), self.hfeat, L.FP8 if self.hfeat_q is not None else self.dt, h, w, Ct, 0, Ct, taps,
                           what="fwd out_modules.*.conv1", out_slope=0.01, cdt=L.FP8 if f8 else None, out_scale=deq_all if f8 else None,
                           out_quant=None if self.hfeat_q is None else self.hfeat_q[2], out_quant_stride=0 if self.hfeat_q is None else 1,
                           heads_epi=self.hepi)
            return None, 0
        for i in range(nh):
            self.emit_pack("out_modules.%d.conv1.weight" % i, wf, 0, 128, 128, 3, 128, 128, rows_total=Ct, rows_off=128 * i)
        srcs = (C.c_void_p * nh)(*[self.P("out_modules.%d.conv1.bias" % i) for i in range(nh)])
        counts = (C.c_int32 * nh)(*([128] * nh))
        self.keep += [srcs, counts]
        lib, bp = self.lib, bias_all.data_ptr()
        self.pack_ops.append((lambda _r, st: lib.abc_concat_f32(srcs, counts, nh, bp, st), None, "gather out_modules.*.conv1.bias", (),
                              {"kernel": "concat", "flops": 0, "bytes": 0}))
        return self.emit_conv(self.fwd_ops, trunk, wf, bp, self.hfeat, self.dt, h, w, Ct, 0, Ct, taps, stats=self.train,
                              what="fwd out_modules.*.conv1")

    # ------------------------------------------------------------------ backward plan
    def _bn_backward(self, ops, rec, same, pool, drop=None, defer=False):
        """act_bwd + bn finalize + apply for rec; returns Src of dY (plain).
        defer=True: the apply pass is NOT emitted; returns (Src of g with coef = (ca, cc, cb) for a consumer that applies
        dY = ca*g + cb*y_raw + cc on load, emit_apply) where emit_apply() emits the classic in-place pass and returns
        the plain Src -- the caller picks one"""
        fg = getattr(rec, "fused_g", None)
        if fg is not None:
            # the data gradient that produced d(activation output) already stored g and the partial sums (emit_conv(actbwd=rec))
            assert pool is None and drop is None
            return self._bn_finish(ops, rec, fg[1], fg[2], fg[0], defer=defer, keep_g=True)
        C_ = rec.cout
        g = self.new((self.B, rec.H, rec.W, C_))
        rec.g = g      # (handle for the in-situ parity tests)
        d = L.ActBwdDesc()
        d.y_raw, d.ld_y = rec.y.data_ptr(), rec.ld
        if same is not None:
            d.dA_same, d.ld_same, d.csame_off = same[0].data_ptr(), same[1], same[2]
        if pool is not None:
            d.dA_pool, d.ld_pool, d.cpool_off = pool[0].data_ptr(), pool[1], pool[2]
        d.g, d.ld_g = g.data_ptr(), C_
        d.scale, d.shift, d.slope = rec.scale.data_ptr(), rec.shift.data_ptr(), rec.slopes.data_ptr()
        d.mean, d.invstd = rec.mean.data_ptr(), rec.invstd.data_ptr()
        d.dtype, d.B, d.H, d.W, d.C, d.cy_off = self.dt, self.B, rec.H, rec.W, C_, rec.coff
        if drop is not None:
            d.drop_p, d.drop_seed, d.drop_ld, d.drop_salt = drop[0], drop[1], rec.ld, self.drop_salt.data_ptr()
        nblk = self.lib.abc_act_bwd_blocks(C.byref(d))
        part = self.new((nblk, 2, C_), torch.float32)
        d.partial = part.data_ptr()
        nsrc = (1 if d.dA_same else 0) + (0.25 if d.dA_pool else 0)
        self._emit(ops, self.lib.abc_act_bwd, d, "act_bwd " + rec.bname,
                   meta={"kernel": "act_bwd", "flops": 0, "bytes": float(self.B * rec.H * rec.W * C_ * self._esz(self.dt) * (2 + nsrc))})
        k1, k2, gs = (self.new((C_,), torch.float32) for _ in range(3))
        f = L.BnBwdDesc()
        f.partial, f.nblk, f.C, f.count = part.data_ptr(), nblk, C_, float(self.B * rec.H * rec.W)
        f.gamma, f.invstd = self.P(rec.bname + ".weight"), rec.invstd.data_ptr()
        f.dgamma, f.dbeta = self.G(rec.bname + ".weight"), self.G(rec.bname + ".bias")
        f.k1, f.k2, f.gscale = k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
        if defer:
            ca, cb, cc = (self.new((C_,), torch.float32) for _ in range(3))
            f.mean, f.ca, f.cb, f.cc = rec.mean.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
        self._emit_bn_bwd(ops, f, "bn_bwd " + rec.bname, (rec.bname + ".weight", rec.bname + ".bias"))
        a = L.BnApplyDesc()
        a.g, a.ld_g, a.y_raw, a.ld_y, a.cy_off = g.data_ptr(), C_, rec.y.data_ptr(), rec.ld, rec.coff
        a.mean, a.invstd, a.k1, a.k2, a.gscale = rec.mean.data_ptr(), rec.invstd.data_ptr(), k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
        a.dtype, a.C, a.npix = self.dt, C_, self.B * rec.H * rec.W

        def emit_apply():
            dy = self.new((self.B, rec.H, rec.W, C_))      # (g is kept: the in-situ parity tests read it)
            a.out, a.ld_out = dy.data_ptr(), C_
            self._emit(ops, self.lib.abc_bn_apply_bwd, a, "bn_apply " + rec.bname,
                       meta={"kernel": "bn_apply", "flops": 0, "bytes": float(self.B * rec.H * rec.W * C_ * self._esz(self.dt) * 3)})
            rec.dY = dy
            return Src(dy, self.dt, rec.H, rec.W, C_, 0, C_)

        if defer:
            return Src(g, self.dt, rec.H, rec.W, C_, 0, C_, coef=(ca, cc, cb)), emit_apply
        return emit_apply()

    def _conv_backward(self, ops, rec, dY, want_dgrad=True, into=None):
        """wgrad (+ dgrad into a fresh buffer registered with the producer of rec.src).
        into: a [B, H, W, cin] tensor the data gradient may be ADDED to instead (abc_conv_desc.accumulate, where the narrow-level
        kernel serves that: rec.dsrc_accumulated tells) -- unet2's identity residual, whose gradient is the block's d(out) itself.
        dY is a plain Src, or the deferred pair of _bn_backward(defer=True): then the weight-gradient kernel applies the
        BatchNorm-backward correction on load and writes dY for the data-gradient conv (one pass less over g and y);
        layers whose weight gradient runs on another kernel fall back to the separate apply pass."""
        if isinstance(dY, tuple) and not self.dual_wgrad:      # (Engine(dual_wgrad=False): the separate apply pass everywhere)
            dY = dY[1]()
        if isinstance(dY, tuple):
            gsrc, emit_apply = dY
            out = self.new((self.B, rec.H, rec.W, rec.cout))
            if self.emit_wgrad(ops, gsrc, rec.src, rec.cout, rec.cin, rec.taps, 1, rec.cname + ".weight", "wgrad " + rec.cname,
                               dual=(rec.y, rec.ld, rec.coff, out.data_ptr(), rec.cout)):
                rec.dY = out
                dY = Src(out, self.dt, rec.H, rec.W, rec.cout, 0, rec.cout)
            else:
                dY = emit_apply()
                self.emit_wgrad(ops, dY, rec.src, rec.cout, rec.cin, rec.taps, 1, rec.cname + ".weight", "wgrad " + rec.cname)
        else:
            self.emit_wgrad(ops, dY, rec.src, rec.cout, rec.cin, rec.taps, 1, rec.cname + ".weight", "wgrad " + rec.cname)
        prod = rec.src.producer
        if not want_dgrad or prod is None:
            return None
        lh, lw = rec.src.lh()
        rows_pad = -(-rec.cin // 32) * 32
        wd = self.packed(len(rec.taps), rec.cout, rows_pad)
        self.emit_pack(rec.cname + ".weight", wd, 1, rec.cout, rec.cin, rec.k, rows_pad, rec.cout)
        rec.dsrc_accumulated = False
        if into is not None and self.dt == L.BF16 and tuple(into.shape) == (self.B, lh, lw, rec.cin):
            got = self.emit_conv(ops, dY, wd, None, into, self.dt, lh, lw, rec.cin, 0, rec.cin, taps_mirror(rec.taps),
                                 what="dgrad " + rec.cname + " (+= d(out) of the identity residual)", accumulate=True, only_variant=5)
            if got is not None:
                rec.dsrc, rec.dsrc_accumulated = into, True
                return into
        dsrc = self.new((self.B, lh, lw, rec.cin))
        rec.dsrc = dsrc
        tgt = self._actbwd_target(rec)
        if tgt is not None:
            # the producer's act_bwd pass in this convolution's epilogue: dsrc holds d(BatchNorm output) of `tgt`, not d(src)
            got = self.emit_conv(ops, dY, wd, None, dsrc, self.dt, lh, lw, rec.cin, 0, rec.cin, taps_mirror(rec.taps),
                                 what="dgrad " + rec.cname + " + act_bwd " + tgt.bname, actbwd=tgt)
            if got is not None:
                tgt.fused_g = (dsrc, got[0], got[1])
                tgt.g = dsrc
                rec.dsrc_is_g = True
                return dsrc
        self.emit_conv(ops, dY, wd, None, dsrc, self.dt, lh, lw, rec.cin, 0, rec.cin, taps_mirror(rec.taps), what="dgrad " + rec.cname)
        return dsrc

    def _actbwd_target(self, rec):
        """the layer whose act_bwd pass can ride in the epilogue of rec's data gradient: rec reads the WHOLE activated output of one
        plain convolution + BatchNorm layer, at full resolution, without dropout, and is its only reader (no skip connection, no
        pooled reader: those gradients meet in bn_act.hip's act_bwd) -- the second convolution of a DoubleConv reading the first
        (unet.py:12-17), the trunk's layers"""
        if not (self.actbwd_epilogue and self.train and self.dt == L.BF16):
            return None
        src = rec.src
        p = src.producer
        if not isinstance(p, Rec) or p.kind != "conv" or getattr(p, "is_head", False):
            return None
        if src.pool or getattr(src, "via_pool", False) or src.drop_p > 0 or src.planar or src.t is not p.y:
            return None
        if p.coff != 0 or p.ld != p.cout or src.coff != 0 or src.C != p.cout or rec.cin != p.cout or src.lh() != (p.H, p.W):
            return None
        readers = [r for r in self.recs + list(getattr(self, "head_recs", [])) if getattr(r, "src", None) is not None and
                   (r.src.producer is p or r.src.t is p.y)]
        if len({id(r) for r in readers}) != 1 or readers[0] is not rec:
            return None
        return p

    def _route(self, rec, dsrc):
        """hand d(src) of `rec` to whoever produced src"""
        src = rec.src
        prod = src.producer
        if isinstance(prod, tuple) and prod[0] == "cat":
            half = rec.cin // 2
            up = rec.cat_upper
            up.grad_out = (dsrc, rec.cin, half)
            # the skip half belongs to the conv that wrote channels [0:half) of the cat buffer
            skip = [r for r in self.recs if r.kind == "conv" and r.y is src.t and r.coff == 0][0]
            skip.grad_same = (dsrc, rec.cin, 0)
        elif src.pool or getattr(src, "via_pool", False):
            prod.grad_pool = (dsrc, rec.cin, 0)
        else:
            prod.grad_same = (dsrc, rec.cin, 0)

    def _build_backward(self):
        ops = self.bwd_ops
        self._heads_backward(ops)
        # ---- trunk, decoder, encoder in reverse
        body = [r for r in self.recs if not getattr(r, "is_head", False)]
        for rec in reversed(body):
            if rec.kind == "conv":
                dY = self._bn_backward(ops, rec, rec.grad_same, rec.grad_pool, defer=True)
                dsrc = self._conv_backward(ops, rec, dY)
                if dsrc is not None:
                    self._route(rec, dsrc)
            else:
                self._convT_backward(ops, rec)

    def _heads_backward(self, ops):
        B, h, w = self.B, self.h, self.w
        nh = len(self.heads)
        if self.hf is not None:
            return self._heads_backward_fused(ops)
        self.dlogits = [self.new((B, hc, h, w), torch.float32) for hc in self.heads]
        nchan = sum(self.heads)
        self.chan_scale = self.new((nchan,), torch.float32, 0.0)
        one = self.new((max(self.heads),), torch.float32, 1.0)
        zero = self.new((max(self.heads),), torch.float32, 0.0)
        dfeat = self.new((B, h, w, 128 * nh))
        # ---- heads' 1x1 convs (weight gradients one by one, the eight data gradients as one launch)
        head_dgrads, head_wgrads = [], []
        # K-splits of the heads' batched 1x1 weight gradient: ONE round of ~256 workgroups over all heads, instead of 256 splits
        # per head = 8 rounds of workgroups that each lived for 4 chunks and wrote a 64 KB slab.  Shared out by the measured
        # cost of a 128-pixel chunk: ~4 us for the feature tile alone, ~9 us with 128 rows of dL beside it (the rows of the
        # channel-planar dL are 36 KB apart: 128 concurrent 512-byte streams per workgroup, poor DRAM page locality)
        units = [-(-(-(-hc // 32)) // 4) for hc in self.heads]
        cost = [4.0 + 5.2 * (hc / u) / 128.0 for hc, u in zip(self.heads, units)]
        tot = sum(u * c for u, c in zip(units, cost))
        head_splits = [max(1, int(256 * c / tot)) for c in cost] if self.batched_heads else [None] * nh
        for r2 in self.head2:
            i, hc = r2.idx, r2.cout
            cs = self.chan_scale[self.head_off[i]:self.head_off[i] + hc]
            dl = Src(self.dlogits[i], L.F32, h, w, 0, 0, hc, coef=(cs, zero, one), planar=True)
            got = self.emit_wgrad(ops, dl, r2.src, hc, 128, [(0, 0)], 1, r2.cname + ".weight", "wgrad " + r2.cname,
                                  rowsum_to=r2.cname + ".bias", collect=head_wgrads, nsplit=head_splits[i])
            if got != "rowsum":
                lib = self.lib
                psw = self.new((lib.abc_plane_sum_work(hc),), torch.float32)
                a = (self.dlogits[i].data_ptr(), B, hc, h * w, cs.data_ptr(), psw.data_ptr(), self.G(r2.cname + ".bias"))
                ops.append((lambda _r, st, a=a: lib.abc_plane_sum(*a, st), None, "dbias " + r2.cname, (r2.cname + ".bias",),
                            {"kernel": "plane_sum", "flops": 0, "bytes": float(B * hc * h * w * 4)}))
            wd = self.packed(1, hc, 128)
            self.emit_pack(r2.cname + ".weight", wd, 1, hc, 128, 1, 128, hc)
            self.emit_conv(ops, dl, wd, None, dfeat, self.dt, h, w, 128 * nh, 128 * i, 128, [(0, 0)], what="dgrad " + r2.cname,
                           collect=head_dgrads)
        self.emit_wgrad_heads_batch(ops, head_wgrads, "wgrad out_modules.*.conv2")
        self.emit_heads_batch(ops, head_dgrads, 1, "dgrad out_modules.*.conv2")
        # ---- heads' BN + conv1: per-head BN backward, ONE data-gradient conv over the 8x128 concatenated channels
        taps = taps_square(3)
        dyh = self.new((B, h, w, 128 * nh))
        wd_all = self.packed(9, 128 * nh, 128)
        merged = None
        if self.dt == L.BF16 and self.batched_heads and nh <= 8:
            merged = self._heads_act_bwd_merged(ops, dfeat)
        one_wgrad = merged is not None and self._heads_conv1_wgrad_merged(ops, merged, dyh, taps)
        for i, rec in enumerate(self.head_recs):
            drop = (self.drop_p, self.drop_seed) if self.drop_p > 0 else None
            if one_wgrad:
                pass
            elif merged is not None:
                ok = self.emit_wgrad(ops, merged[i], rec.src, 128, 128, taps, 1, rec.cname + ".weight", "wgrad " + rec.cname,
                                     dual=(rec.y, rec.ld, rec.coff, dyh.data_ptr() + 128 * i * dyh.element_size(), 128 * nh))
                if not ok:
                    raise RuntimeError("fused BN-backward apply was refused for " + rec.cname)
            elif self.dt == L.BF16:
                # bf16: the weight-gradient kernel applies the BN-backward correction on load and writes dY into this
                # head's channel slice of dyh (no separate apply pass)
                gsrc, _apply = self._bn_backward(ops, rec, (dfeat, 128 * nh, 128 * i), None, drop=drop, defer=True)
                ok = self.emit_wgrad(ops, gsrc, rec.src, 128, 128, taps, 1, rec.cname + ".weight", "wgrad " + rec.cname,
                                     dual=(rec.y, rec.ld, rec.coff, dyh.data_ptr() + 128 * i * dyh.element_size(), 128 * nh))
                if not ok:
                    raise RuntimeError("fused BN-backward apply was refused for " + rec.cname)
            else:
                dY = self._bn_backward_into(ops, rec, (dfeat, 128 * nh, 128 * i), dyh, 128 * nh, 128 * i, drop)
                self.emit_wgrad(ops, dY, rec.src, 128, 128, taps, 1, rec.cname + ".weight", "wgrad " + rec.cname)
            self.emit_pack(rec.cname + ".weight", wd_all, 1, 128, 128, 3, 128, 128, red_total=128 * nh, red_off=128 * i)
        dtrunk = self.new((B, h, w, 128))
        dy_all = Src(dyh, self.dt, h, w, 128 * nh, 0, 128 * nh)
        self._heads_conv1_dgrad(ops, dy_all, wd_all, dtrunk, taps)

    def _heads_conv1_dgrad(self, ops, dy_all, wd_all, dtrunk, taps):
        """the eight heads' conv1 data gradients as ONE 8 x 128 -> 128 convolution (unet.py:66,116-118 backward); the trunk's last layer
        has no reader but the heads, so its act_bwd pass rides in this launch's epilogue where the library serves it"""
        h, w = self.h, self.w
        p = self.trunk.producer
        what = "dgrad heads.conv1"
        if (self.actbwd_epilogue and self.train and self.dt == L.BF16 and isinstance(p, Rec) and p.kind == "conv" and self.trunk.t is p.y and
                not self.trunk.pool and self.trunk.drop_p == 0 and p.coff == 0 and p.ld == p.cout == 128 and (p.H, p.W) == (h, w) and
                all(getattr(r, "is_head", False) for r in self.recs + list(self.head_recs)
                    if getattr(r, "src", None) is not None and (r.src.producer is p or r.src.t is p.y))):
            got = self.emit_conv(ops, dy_all, wd_all, None, dtrunk, self.dt, h, w, 128, 0, 128, taps_mirror(taps),
                                 what=what + " + act_bwd " + p.bname, actbwd=p)
            if got is not None:
                p.fused_g = (dtrunk, got[0], got[1])
                p.g = dtrunk
                self.heads_dgrad_is_g = True
                p.grad_same = (dtrunk, 128, 0)
                return
        self.emit_conv(ops, dy_all, wd_all, None, dtrunk, self.dt, h, w, 128, 0, 128, taps_mirror(taps), what=what)
        p.grad_same = (dtrunk, 128, 0)

    def _heads_backward_fused(self, ops):
        """backward plan behind the fused heads pass: conv2's weight / bias gradients from the blocked d(logits), the eight
        BatchNorm finalisations from the pass's partial sums (times the head's loss factor), then conv1 as in the unfused plan"""
        B, h, w = self.B, self.h, self.w
        nh = len(self.heads)
        Ct = 128 * nh
        lib, d = self.lib, self.hf
        npx = B * h * w
        writes = tuple(n for i in range(nh) for n in ("out_modules.%d.conv2.weight" % i, "out_modules.%d.conv2.bias" % i))
        ops.append((lambda _r, st: lib.abc_heads_fused_wgrad(C.byref(d), st), None, "wgrad out_modules.*.conv2", writes,
                    {"kernel": "heads_fused_wgrad", "flops": 2.0 * npx * sum(self.heads) * 128,
                     "bytes": float(npx * Ct * 2 + self.hf_dl.numel() * 2 + self.hf_work.numel() * 4)}))
        arr = (L.BnBwdDesc * nh)()
        merged, bwrites = [], []
        ca_all, cb_all, cc_all = (self.new((Ct,), torch.float32) for _ in range(3))
        for i, rec in enumerate(self.head_recs):
            k1, k2, gs = (self.new((128,), torch.float32) for _ in range(3))
            ca, cb, cc = (t[128 * i:128 * (i + 1)] for t in (ca_all, cb_all, cc_all))
            f = arr[i]
            f.partial, f.nblk, f.C, f.count = self.hf_bnpart.data_ptr() + 4 * 128 * i, self.hf_chunks, 128, float(npx)
            f.gamma, f.invstd = self.P(rec.bname + ".weight"), rec.invstd.data_ptr()
            f.dgamma, f.dbeta = self.G(rec.bname + ".weight"), self.G(rec.bname + ".bias")
            f.k1, f.k2, f.gscale = k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
            f.mean, f.ca, f.cb, f.cc = rec.mean.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
            f.in_scale = self.chan_scale.data_ptr() + 4 * self.head_off[i]
            bwrites += [rec.bname + ".weight", rec.bname + ".bias"]
            merged.append(Src(self.hf_g, self.dt, h, w, Ct, 128 * i, 128, coef=(ca_all, cc_all, cb_all)))
        self.keep.append(arr)
        ops.append((lambda _r, st, a=arr: lib.abc_bn_finalize_bwd_batch(a, nh, Ct, st), None, "bn_bwd out_modules.*.bn", tuple(bwrites),
                    {"kernel": "bn_bwd", "flops": 0, "bytes": 0}))
        taps = taps_square(3)
        dyh = self.new((B, h, w, Ct))
        wd_all = self.packed(9, Ct, 128)
        if not self._heads_conv1_wgrad_merged(ops, merged, dyh, taps):
            raise RuntimeError("fused heads: the merged conv1 weight gradient was refused")
        for i, rec in enumerate(self.head_recs):
            self.emit_pack(rec.cname + ".weight", wd_all, 1, 128, 128, 3, 128, 128, red_total=Ct, red_off=128 * i)
        dtrunk = self.new((B, h, w, 128))
        self.dyh, self.dtrunk = dyh, dtrunk
        dy_all = Src(dyh, self.dt, h, w, Ct, 0, Ct)
        self._heads_conv1_dgrad(ops, dy_all, wd_all, dtrunk, taps)

    def _heads_conv1_wgrad_merged(self, ops, merged, dyh, taps):
        """The eight heads' conv1 weight gradients (unet.py:66, 8 x [128,128,3,3]) as ONE weight gradient with 8 x 128
        a-channels: the eight share their Q operand (the trunk activation), their P operands sit side by side in g / hfeat,
        so one launch of 8 x 2 tiles x 16 splits reads Q once per XCD, writes an eighth of the split-K slabs (16 instead
        of 128 slabs per head) and runs 72 instead of 9 patches per workgroup.  The slab reduction scatters the eight
        row blocks to the eight parameters' gradients (one batched launch).  False when the library does not serve the
        fused BatchNorm-backward load for this descriptor (the caller then emits one launch per head)."""
        nh = len(self.heads)
        Ct = 128 * nh
        r0 = self.head_recs[0]
        q = r0.src
        g0 = merged[0]
        p = Src(g0.t, self.dt, g0.H, g0.W, Ct, 0, Ct, coef=g0.coef)
        d = L.WgradDesc()
        p.fill(d.p)
        q.fill(d.q)
        d.dtype_p, d.dtype_q, d.dtype_c = p.dt, q.dt, self.dt
        gh, gw = p.lh()
        d.B, d.Hg, d.Wg, d.Hq, d.Wq = self.B, gh, gw, gh, gw
        d.cp_off, d.cq_off, d.Ca, d.Cb, d.stride = 0, q.coff, Ct, 128, 1
        L.set_taps(d, taps)
        d.p2, d.ld_p2, d.cp2_off, d.p_dual, d.p_out, d.ld_pout = self.hfeat.data_ptr(), Ct, 0, 1, dyh.data_ptr(), Ct
        if not self.lib.abc_wgrad_fuses_apply(C.byref(d)):
            return False
        ca_pad, cb_pad = L.i32(), L.i32()
        L.check(self.lib.abc_wgrad_pads(C.byref(d), C.byref(ca_pad), C.byref(cb_pad)), "wgrad_pads")
        ca_pad, cb_pad = ca_pad.value, cb_pad.value
        per_split = self.lib.abc_wgrad_blocks(C.byref(d))
        npatch = self.B * (-(-gh // 8)) * (-(-gw // 16))
        nsplit = max(1, min(max(1, npatch // 2), 256 // per_split))
        d.nsplit = nsplit
        need = nsplit * len(taps) * ca_pad * cb_pad
        slabs = self.new((need,), torch.float32)
        d.partial = slabs.data_ptr()
        at_, bt_ = L.i32(), L.i32()
        L.check(self.lib.abc_wgrad_tile(C.byref(d), C.byref(at_), C.byref(bt_)), "wgrad_tile")
        esz = self._esz(self.dt)
        npx = self.B * gh * gw
        meta = {"kernel": "wgrad<%s,%s,%s,%dx%d,S1>" % (self._dn(p.dt), self._dn(q.dt), self._dn(self.dt), at_.value, bt_.value),
                "flops": 2.0 * npx * Ct * 128 * len(taps),
                # g + y_raw read, dY written, Q read once, the slabs written
                "bytes": float(3 * npx * Ct * esz + npx * 128 * self._esz(q.dt) + need * 4)}
        self._emit(ops, self.lib.abc_wgrad, d, "wgrad out_modules.*.conv1", meta=meta)
        rarr = (L.WgradReduceDesc * nh)()
        writes = []
        for i, rec in enumerate(self.head_recs):
            r = rarr[i]
            r.partial = slabs.data_ptr() + 4 * 128 * i * cb_pad      # this head's 128 rows of every slab
            r.nsplit, r.ntaps, r.Ca, r.Cb, r.Ca_pad, r.Cb_pad = nsplit, len(taps), 128, 128, ca_pad, cb_pad
            r.dw, r.accumulate = self.G(rec.cname + ".weight"), 0
            writes.append(rec.cname + ".weight")
        self.keep.append(rarr)
        lib = self.lib
        ops.append((lambda _r, st, a=rarr: lib.abc_wgrad_reduce_batch(a, nh, st), None, "wgrad out_modules.*.conv1 reduce", tuple(writes),
                    {"kernel": "wgrad_reduce_batch", "flops": 0, "bytes": float(need * 4 + Ct * 128 * len(taps) * 4)}))
        return True

    def _heads_act_bwd_merged(self, ops, dfeat):
        """BN -> LeakyReLU -> Dropout backward of ALL heads as one pass over the 8 x 128 channels of hfeat / dfeat (their
        coefficient and statistics arrays sit side by side), one batched finalisation; returns per head the Src of g with
        the deferred-apply coefficients (as _bn_backward(defer=True))"""
        nh = len(self.heads)
        Ct = 128 * nh
        r0 = self.head_recs[0]
        B, H, W = self.B, r0.H, r0.W
        g = self.new((B, H, W, Ct))
        d = L.ActBwdDesc()
        d.y_raw, d.ld_y = self.hfeat.data_ptr(), Ct
        d.dA_same, d.ld_same, d.csame_off = dfeat.data_ptr(), Ct, 0
        d.g, d.ld_g = g.data_ptr(), Ct
        sc, sh, sl = self.hcoef
        d.scale, d.shift, d.slope = sc.data_ptr(), sh.data_ptr(), sl.data_ptr()
        d.mean, d.invstd = self.hmean.data_ptr(), self.hinvstd.data_ptr()
        d.dtype, d.B, d.H, d.W, d.C, d.cy_off = self.dt, B, H, W, Ct, 0
        if self.drop_p > 0:
            d.drop_p, d.drop_seed, d.drop_ld, d.drop_salt = self.drop_p, self.drop_seed, Ct, self.drop_salt.data_ptr()
        nblk = self.lib.abc_act_bwd_blocks(C.byref(d))
        part = self.new((nblk, 2, Ct), torch.float32)
        d.partial = part.data_ptr()
        self._emit(ops, self.lib.abc_act_bwd, d, "act_bwd out_modules.*.bn",
                   meta={"kernel": "act_bwd", "flops": 0, "bytes": float(B * H * W * Ct * self._esz(self.dt) * 3)})
        arr = (L.BnBwdDesc * nh)()
        out, writes = [], []
        # the deferred-apply coefficients are indexed by the ABSOLUTE channel of g (abc_act_src): one array of 8 x 128 per
        # quantity, every head's finalisation writing its own slice
        ca_all, cb_all, cc_all = (self.new((Ct,), torch.float32) for _ in range(3))
        for i, rec in enumerate(self.head_recs):
            k1, k2, gs = (self.new((128,), torch.float32) for _ in range(3))
            ca, cb, cc = (t[128 * i:128 * (i + 1)] for t in (ca_all, cb_all, cc_all))
            f = arr[i]
            f.partial, f.nblk, f.C, f.count = part.data_ptr() + 4 * 128 * i, nblk, 128, float(B * H * W)
            f.gamma, f.invstd = self.P(rec.bname + ".weight"), rec.invstd.data_ptr()
            f.dgamma, f.dbeta = self.G(rec.bname + ".weight"), self.G(rec.bname + ".bias")
            f.k1, f.k2, f.gscale = k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
            f.mean, f.ca, f.cb, f.cc = rec.mean.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
            writes += [rec.bname + ".weight", rec.bname + ".bias"]
            out.append(Src(g, self.dt, H, W, Ct, 128 * i, 128, coef=(ca_all, cc_all, cb_all)))
        self.keep.append(arr)
        lib = self.lib
        ops.append((lambda _r, st, a=arr: lib.abc_bn_finalize_bwd_batch(a, nh, Ct, st), None, "bn_bwd out_modules.*.bn", tuple(writes),
                    {"kernel": "bn_bwd", "flops": 0, "bytes": 0}))
        return out

    def _convT_backward(self, ops, rec):
        """ConvTranspose2d(k3,s2) backward: bias (column sums of dOut), weight (stride-2 wgrad), data (stride-2 gather)"""
        B = self.B
        dcat, ld, coff = rec.grad_out
        hs, ws = rec.H, rec.W
        dOut = Src(dcat, self.dt, hs, ws, ld, coff, rec.cout)
        self.emit_colsum(ops, dcat, self.dt, B * hs * ws, ld, coff, rec.cout, None, rec.cname + ".bias", "dbias " + rec.cname)
        self.emit_wgrad(ops, rec.src, dOut, rec.cin, rec.cout, rec.taps_bwd, 2, rec.cname + ".weight", "wgrad " + rec.cname)
        lh, lw = rec.src.lh()
        rows_pad = -(-rec.cin // 32) * 32
        wd = self.packed(9, rec.cout, rows_pad)
        self.emit_pack(rec.cname + ".weight", wd, 3, rec.cout, rec.cin, 3, rows_pad, rec.cout)
        dsrc = self.new((B, lh, lw, rec.cin))
        rec.dsrc = dsrc
        self.emit_conv(ops, dOut, wd, None, dsrc, self.dt, lh, lw, rec.cin, 0, rec.cin, rec.taps_bwd, stride=2,
                       what="dgrad " + rec.cname)
        rec.src.producer.grad_same = (dsrc, rec.cin, 0)

    def _bn_backward_into(self, ops, rec, same, gbuf, ld_g, g_off, drop):
        """as _bn_backward, but G/dY live in a channel slice of a shared buffer (the heads)"""
        C_ = rec.cout
        esz = gbuf.element_size()
        gptr = gbuf.data_ptr() + g_off * esz
        d = L.ActBwdDesc()
        d.y_raw, d.ld_y = rec.y.data_ptr(), rec.ld
        d.dA_same, d.ld_same, d.csame_off = same[0].data_ptr(), same[1], same[2]
        d.g, d.ld_g = gptr, ld_g
        d.scale, d.shift, d.slope = rec.scale.data_ptr(), rec.shift.data_ptr(), rec.slopes.data_ptr()
        d.mean, d.invstd = rec.mean.data_ptr(), rec.invstd.data_ptr()
        d.dtype, d.B, d.H, d.W, d.C, d.cy_off = self.dt, self.B, rec.H, rec.W, C_, rec.coff
        if drop is not None:
            d.drop_p, d.drop_seed, d.drop_ld, d.drop_salt = drop[0], drop[1], rec.ld, self.drop_salt.data_ptr()
        nblk = self.lib.abc_act_bwd_blocks(C.byref(d))
        part = self.new((nblk, 2, C_), torch.float32)
        d.partial = part.data_ptr()
        nsrc = (1 if d.dA_same else 0) + (0.25 if d.dA_pool else 0)
        self._emit(ops, self.lib.abc_act_bwd, d, "act_bwd " + rec.bname,
                   meta={"kernel": "act_bwd", "flops": 0, "bytes": float(self.B * rec.H * rec.W * C_ * self._esz(self.dt) * (2 + nsrc))})
        k1, k2, gs = (self.new((C_,), torch.float32) for _ in range(3))
        f = L.BnBwdDesc()
        f.partial, f.nblk, f.C, f.count = part.data_ptr(), nblk, C_, float(self.B * rec.H * rec.W)
        f.gamma, f.invstd = self.P(rec.bname + ".weight"), rec.invstd.data_ptr()
        f.dgamma, f.dbeta = self.G(rec.bname + ".weight"), self.G(rec.bname + ".bias")
        f.k1, f.k2, f.gscale = k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
        self._emit_bn_bwd(ops, f, "bn_bwd " + rec.bname, (rec.bname + ".weight", rec.bname + ".bias"))
        a = L.BnApplyDesc()
        a.g, a.ld_g, a.y_raw, a.ld_y, a.cy_off = gptr, ld_g, rec.y.data_ptr(), rec.ld, rec.coff
        a.mean, a.invstd, a.k1, a.k2, a.gscale = rec.mean.data_ptr(), rec.invstd.data_ptr(), k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
        a.dtype, a.C, a.npix = self.dt, C_, self.B * rec.H * rec.W
        self._emit(ops, self.lib.abc_bn_apply_bwd, a, "bn_apply " + rec.bname,
                   meta={"kernel": "bn_apply", "flops": 0, "bytes": float(self.B * rec.H * rec.W * C_ * self._esz(self.dt) * 3)})
        return Src(gbuf, self.dt, rec.H, rec.W, ld_g, g_off, C_)

    # ------------------------------------------------------------------ unet2 (CBAM + residual, unet2.py)
    def f32buf(self, *shape, fill=0.0):
        return self.new(tuple(shape), torch.float32, fill)

    def block2(self, prefix, xin: Src, cout, k, dst=None):
        """unet2.DoubleConv (unet2.py:49-74): conv-BN-ReLU-conv-BN-CBAM, + residual, ReLU.  The block output is
        MATERIALISED (one element-wise pass), so its consumers load it with the identity transform."""
        lib = self.lib
        H, W = xin.lh()
        B, cin = self.B, xin.C
        p = prefix + ".double_conv"
        t1, c1 = self.act_buf(H, W, cout)
        rec1, a1 = self.conv_bn(p + ".0", p + ".1", xin, cout, k, (t1, c1, H, W, cout, 0), 0.0)
        t2, c2 = self.act_buf(H, W, cout)
        rec2, _ = self.conv_bn(p + ".3", p + ".4", a1, cout, k, (t2, c2, H, W, cout, 0), 1.0, stats_rows=4, force_stats=True)
        mid = cout // 16
        m = p + ".5.channel_attention.shared_MLP"
        blk = Rec(kind="blk2", prefix=prefix, rec1=rec1, rec2=rec2, xin=xin, cin=cin, cout=cout, k=k, H=H, W=W, mid=mid)
        blk.ca, blk.avgz, blk.maxz = self.f32buf(B, cout), self.f32buf(B, cout), self.f32buf(B, cout)
        # arg-max of AdaptiveMaxPool2d(1) (unet2.py:10,20): the extreme raw value per (image, channel) and the first pixel holding it
        blk.ext, blk.first = self.f32buf(B, cout), self.new((B, cout), torch.int32, 0x7FFFFFFF)
        blk.hid_a, blk.hid_m = self.f32buf(B, mid), self.f32buf(B, mid)
        ch = L.CbamChannelDesc()
        ch.partial, ch.tiles_per_img, ch.B, ch.C, ch.mid, ch.HW = rec2.stats.data_ptr(), rec2.nblk // B, B, cout, mid, float(H * W)
        ch.scale, ch.shift = rec2.scale.data_ptr(), rec2.shift.data_ptr()
        ch.w1, ch.b1, ch.w2, ch.b2 = self.P(m + ".0.weight"), self.P(m + ".0.bias"), self.P(m + ".2.weight"), self.P(m + ".2.bias")
        ch.ca, ch.avgz, ch.maxz = blk.ca.data_ptr(), blk.avgz.data_ptr(), blk.maxz.data_ptr()
        ch.hid_avg, ch.hid_max = blk.hid_a.data_ptr(), blk.hid_m.data_ptr()
        ch.ext, ch.first = blk.ext.data_ptr(), blk.first.data_ptr()
        self._emit(self.fwd_ops, lib.abc_cbam_channel_fwd, ch, "cbam_channel " + prefix)
        blk.st, blk.amax, blk.sa = self.f32buf(B, H, W, 2), self.new((B, H, W), torch.int32), self.f32buf(B, H, W)
        # residual branch (unet2.py:62-65,72)
        if cin != cout:
            tr = self.new((B, H, W, cout))
            rows_pad = -(-cout // 32) * 32
            wr = self.packed(1, cin, rows_pad)
            self.emit_pack(prefix + ".res_conv.weight", wr, 0, cout, cin, 1, rows_pad, cin)
            self.emit_conv(self.fwd_ops, xin, wr, self.P(prefix + ".res_conv.bias"), tr, self.dt, H, W, cout, 0, cout, [(0, 0)],
                           what="fwd %s.res_conv" % prefix)
            res = (tr, cout, 0, 0)
        else:
            res = (xin.t, xin.ld, xin.coff, 1 if xin.pool else 0)
        if dst is None:
            to = self.new((B, H, W, cout))
            dst = (to, H, W, cout, 0)
        to, _, _, ld_o, coff_o = dst
        blk.out, blk.ld_out, blk.coff_out, blk.res = to, ld_o, coff_o, res

        def pix():
            d = L.CbamPixDesc()
            d.y, d.ld_y, d.cy_off = t2.data_ptr(), cout, 0
            d.scale, d.shift, d.mean, d.invstd = rec2.scale.data_ptr(), rec2.shift.data_ptr(), rec2.mean.data_ptr(), rec2.invstd.data_ptr()
            d.ca, d.maxz, d.sa, d.st, d.amax = blk.ca.data_ptr(), blk.maxz.data_ptr(), blk.sa.data_ptr(), blk.st.data_ptr(), blk.amax.data_ptr()
            d.ext, d.first = blk.ext.data_ptr(), blk.first.data_ptr()
            d.res, d.ld_res, d.cres_off, d.res_pool = res[0].data_ptr(), res[1], res[2], res[3]
            d.out, d.ld_out, d.cout_off = to.data_ptr(), ld_o, coff_o
            d.dtype, d.B, d.H, d.W, d.C = self.dt, B, H, W, cout
            return d

        blk.pix = pix
        esz = self._esz(self.dt)
        npx = B * H * W
        self._emit(self.fwd_ops, lib.abc_cbam_spatial_stats, pix(), "cbam_spatial_stats " + prefix,
                   meta={"kernel": "cbam_spatial_stats", "flops": 0, "bytes": float(npx * cout * esz)})
        c7 = L.CbamConv7Desc()
        sp = p + ".5.spatial_attention.conv2d"
        c7.st, c7.w7, c7.b7, c7.sa = blk.st.data_ptr(), self.P(sp + ".weight"), self.P(sp + ".bias"), blk.sa.data_ptr()
        c7.B, c7.H, c7.W = B, H, W
        self._emit(self.fwd_ops, lib.abc_cbam_conv7_fwd, c7, "cbam_conv7 " + prefix)
        self._emit(self.fwd_ops, lib.abc_cbam_apply_fwd, pix(), "cbam_apply " + prefix,
                   meta={"kernel": "cbam_apply", "flops": 0, "bytes": float(npx * cout * esz * 3)})
        self.units2.append(("blk", blk))
        return Src(to, self.dt, H, W, ld_o, coff_o, cout, coef=None, producer=blk)

    def _build2(self):
        B, H, W = self.B, self.H, self.W
        S = [(H >> i, W >> i) for i in range(6)]
        self.units2 = []
        img_src = self._image_src()
        x = self.block2("inc1", img_src, 32, 5)
        x1 = self.block2("inc2", x, 32, 5)
        x2 = self.block2("down1.maxpool_conv.1", self.pooled(x1), 32, 3)
        x = self.block2("down2.maxpool_conv.1", self.pooled(x2), 64, 3)
        cat3, cc3 = self.act_buf(S[2][0], S[2][1], 128)
        cat2, cc2 = self.act_buf(S[3][0], S[3][1], 256)
        cat1, cc1 = self.act_buf(S[4][0], S[4][1], 512)
        x3 = self.block2("inc3", x, 64, 3, dst=(cat3, S[2][0], S[2][1], 128, 0))
        x4 = self.block2("down3.maxpool_conv.1", self.pooled(x3), 128, 3, dst=(cat2, S[3][0], S[3][1], 256, 0))
        x5 = self.block2("down4.maxpool_conv.1", self.pooled(x4), 256, 3, dst=(cat1, S[4][0], S[4][1], 512, 0))
        x6 = self.block2("down5.maxpool_conv.1", self.pooled(x5), 512, 3)
        u, _ = self.up("up1", x6, cat1, None, S[4][0], S[4][1], 512, 256, skip_producer=x5.producer)
        u, _ = self.up("up2", u, cat2, None, S[3][0], S[3][1], 256, 128, skip_producer=x4.producer)
        u, _ = self.up("up3", u, cat3, None, S[2][0], S[2][1], 128, 128, skip_producer=x3.producer)
        u = self.block2("dconv1", u, 128, 3)
        trunk = self.block2("dconv2", u, 128, 3)
        self.trunk = trunk
        self._build_heads(trunk)
        if self.train:
            self._build_backward2()
            self._flush_reduce(self.bwd_ops)
        self._finish_build()

    def _build_backward2(self):
        ops = self.bwd_ops
        lib = self.lib
        B = self.B
        self._heads_backward(ops)  # sets trunk.producer.grad_same
        for kind, u in reversed(self.units2):
            if kind == "convT":
                self._convT_backward(ops, u)
                continue
            blk = u
            H, W, Cc = blk.H, blk.W, blk.cout
            npx = B * H * W
            esz = self._esz(self.dt)
            g = self.new((B, H, W, Cc))
            dz = self.new((B, H, W, Cc))
            du, dst = self.f32buf(B, H, W), self.f32buf(B, H, W, 2)
            d_avgz, d_maxz = self.f32buf(B, Cc), self.f32buf(B, Cc)
            # (kept for the in-situ parity tests: every backward kernel's output against torch autograd on these tensors)
            blk.bw = {"g": g, "dz": dz, "du": du, "dst": dst, "d_avgz": d_avgz, "d_maxz": d_maxz}

            def pix(blk=blk, g=g, dz=dz, du=du, dst=dst, d_avgz=d_avgz, d_maxz=d_maxz):
                d = blk.pix()
                d.g, d.ld_g, d.dz, d.ld_dz = g.data_ptr(), Cc, dz.data_ptr(), Cc
                d.du, d.dst, d.d_avgz, d.d_maxz = du.data_ptr(), dst.data_ptr(), d_avgz.data_ptr(), d_maxz.data_ptr()
                return d

            d1 = pix()
            if blk.grad_same is not None:
                d1.d_same, d1.ld_same, d1.csame_off = blk.grad_same[0].data_ptr(), blk.grad_same[1], blk.grad_same[2]
            if blk.grad_pool is not None:
                d1.d_pool, d1.ld_pool, d1.cpool_off = blk.grad_pool[0].data_ptr(), blk.grad_pool[1], blk.grad_pool[2]
            self._emit(ops, lib.abc_cbam_bwd1, d1, "cbam_bwd1 " + blk.prefix,
                       meta={"kernel": "cbam_bwd1", "flops": 0, "bytes": float(npx * Cc * esz * 4)})
            p = blk.prefix + ".double_conv"
            sp = p + ".5.spatial_attention.conv2d"
            c7 = L.CbamConv7Desc()
            c7.st, c7.w7, c7.b7, c7.du, c7.dst = blk.st.data_ptr(), self.P(sp + ".weight"), self.P(sp + ".bias"), du.data_ptr(), dst.data_ptr()
            c7.B, c7.H, c7.W = B, H, W
            nb7 = lib.abc_cbam_conv7_blocks(C.byref(c7))
            part7 = self.f32buf(nb7, 99)
            c7.dw_partial, c7.dw7, c7.db7 = part7.data_ptr(), self.G(sp + ".weight"), self.G(sp + ".bias")
            # (the reduction of part7 rides in the first launch of this block's channel-attention backward below)
            self._emit(ops, lib.abc_cbam_conv7_bwd_partial, c7, "cbam_conv7_bwd " + blk.prefix)
            d2 = pix()
            nb2 = lib.abc_cbam_bwd2_blocks(C.byref(d2))
            part2 = self.f32buf(B, nb2, Cc)
            d2.partial = part2.data_ptr()
            self._emit(ops, lib.abc_cbam_bwd2, d2, "cbam_bwd2 " + blk.prefix,
                       meta={"kernel": "cbam_bwd2", "flops": 0, "bytes": float(npx * Cc * esz * 3)})
            m = p + ".5.channel_attention.shared_MLP"
            ch = L.CbamChannelDesc()
            ch.partial, ch.tiles_per_img, ch.B, ch.C, ch.mid, ch.HW = part2.data_ptr(), nb2, B, Cc, blk.mid, float(H * W)
            ch.w1, ch.b1, ch.w2, ch.b2 = self.P(m + ".0.weight"), self.P(m + ".0.bias"), self.P(m + ".2.weight"), self.P(m + ".2.bias")
            ch.ca, ch.avgz, ch.maxz = blk.ca.data_ptr(), blk.avgz.data_ptr(), blk.maxz.data_ptr()
            ch.hid_avg, ch.hid_max = blk.hid_a.data_ptr(), blk.hid_m.data_ptr()
            ch.dw1, ch.db1, ch.dw2, ch.db2 = self.G(m + ".0.weight"), self.G(m + ".0.bias"), self.G(m + ".2.weight"), self.G(m + ".2.bias")
            ch.d_avgz, ch.d_maxz = d_avgz.data_ptr(), d_maxz.data_ptr()
            ch.work = self.f32buf(B * (Cc + 2 * blk.mid)).data_ptr()
            self.keep += [ch, c7]
            ops.append((lambda _r, st, a=(ch, c7): lib.abc_cbam_channel_bwd_c7(C.byref(a[0]), C.byref(a[1]), st), None,
                        "cbam_channel_bwd " + blk.prefix + " + conv7 reduce",
                        (m + ".0.weight", m + ".0.bias", m + ".2.weight", m + ".2.bias", sp + ".weight", sp + ".bias"),
                        {"kernel": "cbam_channel_bwd", "flops": 0, "bytes": 0}))
            d3 = pix()
            nb3 = lib.abc_cbam_bwd3_blocks(C.byref(d3))
            part3 = self.f32buf(nb3, 2, Cc)
            d3.partial = part3.data_ptr()
            self._emit(ops, lib.abc_cbam_bwd3, d3, "cbam_bwd3 " + blk.prefix,
                       meta={"kernel": "cbam_bwd3", "flops": 0, "bytes": float(npx * Cc * esz * 3)})
            # BN2 backward on d_z, then the second conv
            rec2, rec1 = blk.rec2, blk.rec1
            # (the BN-backward apply fused into the weight gradient's load, as unet does: neutral in round 1 -- bn_apply -0.27 ms,
            #  dual weight gradients +0.33 ms -- +0.9 % since the weight-gradient kernel's prefetch got cheaper: 1231 -> 1242 img/s)
            defer2 = True
            dY2 = self._bn_finish(ops, rec2, part3, nb3, dz, defer=defer2)
            dA1 = self._conv_backward(ops, rec2, dY2)
            rec1.grad_same = (dA1, rec1.cout, 0)
            dY1 = self._bn_backward(ops, rec1, rec1.grad_same, None, defer=defer2)
            xin = blk.xin
            has_prod = xin.producer is not None
            # (identity residual, unet2.py:72: d(x) = d(conv path) + d(out) -- where the 32 -> 32 kernel serves it the data gradient is
            #  added into the tensor that holds d(out); nothing reads that tensor afterwards but the consumers of d(x))
            d_x = self._conv_backward(ops, rec1, dY1, want_dgrad=has_prod, into=g if (blk.cin == blk.cout and has_prod) else None)
            gsrc = Src(g, self.dt, H, W, Cc, 0, Cc)
            if blk.cin != blk.cout:
                rname = blk.prefix + ".res_conv"
                nv = 8 if self.dt == L.BF16 else 4
                if blk.cin == 1 and xin.dt == L.F32 and xin.coef is None and not xin.pool and xin.ld == 1 and xin.coff == 0 and \
                        Cc % nv == 0 and 256 % (Cc // nv) == 0:
                    # the one-channel image: the 1x1 weight gradient is a second, pixel-weighted row of the bias gradient's column sums
                    self.emit_colsum_w1(ops, g, self.dt, npx, Cc, 0, Cc, xin.t, rname + ".bias", rname + ".weight", "wgrad + dbias " + rname)
                else:
                    self.emit_wgrad(ops, gsrc, xin, blk.cout, blk.cin, [(0, 0)], 1, rname + ".weight", "wgrad " + rname)
                    self.emit_colsum(ops, g, self.dt, npx, Cc, 0, Cc, None, rname + ".bias", "dbias " + rname)
                if has_prod:
                    rows_pad = -(-blk.cin // 32) * 32
                    wd = self.packed(1, blk.cout, rows_pad)
                    self.emit_pack(rname + ".weight", wd, 1, blk.cout, blk.cin, 1, rows_pad, blk.cout)
                    self.emit_conv(ops, gsrc, wd, None, d_x, self.dt, H, W, blk.cin, 0, blk.cin, [(0, 0)], what="dgrad " + rname,
                                   accumulate=True)
            elif has_prod and not rec1.dsrc_accumulated:
                a = (d_x.data_ptr(), blk.cin, 0, g.data_ptr(), Cc, 0, Cc, npx, self.dt)
                ops.append((lambda _r, st, a=a: lib.abc_add_into(*a, st), None, "d_x += g " + blk.prefix, (),
                            {"kernel": "add_into", "flops": 0, "bytes": float(npx * Cc * esz * 3)}))
            if has_prod:
                self._route2(blk, d_x)

    def _route2(self, blk, d_x):
        src = blk.xin
        if getattr(src, "cat", None) is not None:
            skip_prod, up_rec = src.cat
            half = blk.cin // 2
            up_rec.grad_out = (d_x, blk.cin, half)
            skip_prod.grad_same = (d_x, blk.cin, 0)
        elif src.pool or getattr(src, "via_pool", False):
            src.producer.grad_pool = (d_x, blk.cin, 0)
        else:
            src.producer.grad_same = (d_x, blk.cin, 0)

    def _bn_finish(self, ops, rec, part, nblk, gbuf, defer=False, keep_g=False):
        """bn_finalize_bwd + bn_apply for a BN whose G and partials were produced elsewhere (CBAM bwd3; a data gradient with the
        act_bwd pass in its epilogue).
        defer=True: as _bn_backward(defer=True) -- the apply pass is left to the weight-gradient kernel's load where it can;
        keep_g: the apply pass writes a buffer of its own (g stays intact for the in-situ parity tests)"""
        C_ = rec.cout
        k1, k2, gs = (self.new((C_,), torch.float32) for _ in range(3))
        f = L.BnBwdDesc()
        f.partial, f.nblk, f.C, f.count = part.data_ptr(), nblk, C_, float(self.B * rec.H * rec.W)
        f.gamma, f.invstd = self.P(rec.bname + ".weight"), rec.invstd.data_ptr()
        f.dgamma, f.dbeta = self.G(rec.bname + ".weight"), self.G(rec.bname + ".bias")
        f.k1, f.k2, f.gscale = k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
        if defer:
            ca, cb, cc = (self.new((C_,), torch.float32) for _ in range(3))
            f.mean, f.ca, f.cb, f.cc = rec.mean.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr()
        self._emit_bn_bwd(ops, f, "bn_bwd " + rec.bname, (rec.bname + ".weight", rec.bname + ".bias"))
        a = L.BnApplyDesc()
        a.g, a.ld_g, a.y_raw, a.ld_y, a.cy_off = gbuf.data_ptr(), C_, rec.y.data_ptr(), rec.ld, rec.coff
        a.mean, a.invstd, a.k1, a.k2, a.gscale = rec.mean.data_ptr(), rec.invstd.data_ptr(), k1.data_ptr(), k2.data_ptr(), gs.data_ptr()
        a.dtype, a.C, a.npix = self.dt, C_, self.B * rec.H * rec.W

        def emit_apply():
            out = gbuf
            if keep_g:
                out = self.new((self.B, rec.H, rec.W, C_))
                a.out, a.ld_out = out.data_ptr(), C_
            self._emit(ops, self.lib.abc_bn_apply_bwd, a, "bn_apply " + rec.bname,
                       meta={"kernel": "bn_apply", "flops": 0, "bytes": float(self.B * rec.H * rec.W * C_ * self._esz(self.dt) * 3)})
            rec.dY = out
            return Src(out, self.dt, rec.H, rec.W, C_, 0, C_)

        if defer:
            return Src(gbuf, self.dt, rec.H, rec.W, C_, 0, C_, coef=(ca, cc, cb)), emit_apply
        return emit_apply()

    # ------------------------------------------------------------------ execution
    def _run(self, ops, stream):
        """launch ops in order on `stream` (the current stream's handle).  ONE stream: a second stream for work that is off the
        data-gradient chain was measured twice and lost both times inside the hipGraph -- the slab reductions (round 1:
        1733 vs 1823 img/s) and, with the BatchNorm-backward apply as its own pass so that nothing chains them in front of the
        data gradient, the weight gradients of the 12 x 12 .. 48 x 48 levels and the transposed convolutions (round 3,
        profiles/tools/ab_side.py history: 6.68 ms -> 6.93 ms with 16 launches forked, 6.99 ms with 46): every fork / join
        edge of the graph costs more than the 20-70 workgroup kernels it lets overlap."""
        if self.lib.abc_get_reserved_cus() != self.reserved_cus:
            raise L.AbcNetHipError("the library's reserved-CU count changed from %d to %d since this plan was built: its grids and "
                                   "statistics buffers were sized for the old value (abc_set_reserved_cus is process-wide; use "
                                   "abcnet_amd.engine.set_reserved_cus, which refuses while plans exist)" % (self.reserved_cus, self.lib.abc_get_reserved_cus()))
        for fn, ref, what, _w, m in ops:
            rc = fn(ref, stream)
            if rc != 0:
                L.check(rc, what)

    def run_pack(self, stream):
        self._run(self.pack_ops, stream)

    def run_forward(self, stream):
        self._run(self.fwd_ops, stream)

    def run_backward(self, stream):
        self._run(self.bwd_ops, stream)
