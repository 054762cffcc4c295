"""Inference harness: the heat-map half of /root/reference/src/img2smiles2.py:42-79 on the HIP kernels.

    model.eval(); preds = model(imgs)                       (img2smiles2.py:56-59)
    3x3 / circular 3-tap local-max masks, |rho|             (img2smiles2.py:61-79)

One step = eval-mode forward (running-statistics BatchNorm folded into the convolution weights / biases at refresh()
time, activations in the convolutions' epilogues, no dropout) + the peak-NMS kernel, on a batch already resident in HBM, replayed from one hipGraph.  The weights do not change between
steps, so re-packing them and deriving the eval-mode BatchNorm coefficients happens in `refresh()`, not in the step;
call it again after `load_state_dict`.  The SMILES assembly that follows in the reference (img2smiles2.py:104-344, RDKit) is out of
scope: the step ends with the four mask / |rho| maps the decoder reads.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class InferenceRunner:
    def __init__(self, model, batch, height, width, use_graph=True, device=None, extract=False, cap_atoms=512, cap_bonds=16384,
                 fold_bn=None, fp8=False, fp8_margin=1.0, guards=False, heads_epilogue=False, nms_in_heads=True, decode=False):
        """fp8: the e4m3 form of the BatchNorm-folded graph (unet.py, bf16 model): the 128-channel 3x3 convolutions at the output
        resolution on the block-scaled MFMA over e4m3 activations and weights (Engine(fp8=True)); the per-tensor activation scales
        are calibrated on the FIRST batch loaded (calibrate(); again on demand) by running the bf16 folded graph on it.
        fold_bn: run the eval graph with every BatchNorm folded into the convolution in front of it and the activation in
        that convolution's epilogue (weights re-packed times gamma / sqrt(running_var + eps) by refresh()); default: on for
        unet.py, off for unet2.py (whose CBAM reads the un-activated BatchNorm output)
        nms_in_heads (default): |rho| and the omega-bin mask of img2smiles2.py:73-79 are second outputs of the heads' 1x1 kernel
        (computed from the very f32 values it stores: bit-identical to the NMS kernel reading the maps back), and the NMS kernel
        does the two 3x3 centre masks only -- 0.5 GB less read per batch of 64; False: the round-3 plan
        decode (with nms_in_heads): store only what the decoder of img2smiles2.py:104-191 reads -- |rho| instead of the raw rho map
        (:73) and, for the 360 bond-type planes, their six-way arg max per omega bin as a uint8 map (:71,112; .btype_idx);
        .logits[5] and .logits[6] are then None, the candidate lists (extract=True) are unchanged bit for bit.  1.7 GB less written
        per batch of 64 at 512 x 512"""
        if not torch.cuda.is_available():
            raise L.AbcNetHipError("InferenceRunner needs an MI355X; abcnet_amd has no CPU fallback")
        self.model = model
        dev = torch.device(device or next(model.parameters()).device)
        if dev.type != "cuda":
            raise L.AbcNetHipError("InferenceRunner: the model must live on a GPU (got %s); abcnet_amd has no CPU fallback" % dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        model.eval()
        with torch.cuda.device(dev):
            x0 = torch.zeros((batch, model.n_channels, height, width), device=dev)
            if fold_bn is None:
                fold_bn = model.VARIANT == "unet"
            self.fold_bn = bool(fold_bn)
            self.fp8 = bool(fp8)
            self.fp8_margin = float(fp8_margin)
            if self.fp8 and not (self.fold_bn and model.VARIANT == "unet" and model.compute_dtype == "bf16"):
                raise L.AbcNetHipError("fp8 inference is a form of the BatchNorm-folded bf16 graph of unet.py")
            # heads_epilogue (folded graph, opt-in: exact but slower than the default plan): the heads' 1x1
            # convolutions in the epilogue of the convolution that makes their features
            self.eng = eng = model._engine_for(x0, False, fold_bn=self.fold_bn, fp8=self.fp8, guards=guards, heads_epilogue=heads_epilogue,
                                               nms_heads=bool(nms_in_heads) and not heads_epilogue,
                                               decode=bool(decode) and bool(nms_in_heads) and not heads_epilogue)
            # (bf16 graph with its feature tensor materialised: calibration only)
            self._ref = model._engine_for(x0, False, fold_bn=True, heads_epilogue=False) if self.fp8 else None
        lg = eng.logits
        self.atom_mask, self.bond_mask = torch.empty_like(lg[0]), torch.empty_like(lg[4])
        self.nms_in_heads = eng.nms_rho is not None
        self.decode = bool(getattr(eng, "decode", False)) and eng.btype_idx is not None
        self.btype_idx = eng.btype_idx
        if self.nms_in_heads:
            self.rho_abs, self.omega_mask = eng.nms_rho, eng.nms_omega      # (written by the forward plan itself)
        else:
            self.rho_abs, self.omega_mask = torch.empty_like(lg[6]), torch.empty_like(lg[7])
        d = L.NmsDesc()
        d.atom, d.bond, d.omega = lg[0].data_ptr(), lg[4].data_ptr(), lg[7].data_ptr()
        d.rho = None if lg[6] is None else lg[6].data_ptr()      # (decode: not stored, and not read -- n_omega = 0)
        d.B, d.h, d.w, d.n_omega = eng.B, eng.h, eng.w, (0 if self.nms_in_heads else lg[7].shape[1])
        d.atom_mask, d.bond_mask = self.atom_mask.data_ptr(), self.bond_mask.data_ptr()
        d.rho_abs, d.omega_mask = self.rho_abs.data_ptr(), self.omega_mask.data_ptr()
        self._nms = d
        # img2smiles2.py:113-191: compact candidate lists for the CPU graph-assembly stage, inside the same graph
        self.extractor = None
        if extract:
            from .ops import PeakExtractor
            self.extractor = PeakExtractor(lg, self.atom_mask, self.bond_mask, cap_atoms=cap_atoms, cap_bonds=cap_bonds,
                                           btype_idx=self.btype_idx if self.decode else None, rho_abs=self.rho_abs if self.decode else None)
        self.use_graph = use_graph
        self._graph = None
        self.steps = 0
        self.refresh()

    def refresh(self):
        """re-pack the (changed) weights and re-derive the eval-mode BatchNorm coefficients (both functions of the
        parameters alone; call again after load_state_dict)"""
        with torch.cuda.device(self.dev):
            self.eng.run_pack(torch.cuda.current_stream().cuda_stream)

    def load_batch(self, imgs):
        self.eng.img.copy_(imgs.reshape(self.eng.img.shape), non_blocking=True)
        if self.fp8 and not self.eng.fp8_calibrated:
            self.calibrate()

    def calibrate(self):
        """fp8: set the per-tensor e4m3 scales from the batch in the image buffer (the bf16 folded graph runs once on it), then
        re-pack the weights (their scales fold the input scale in).  No host sync; call again when the data distribution moves."""
        if not self.fp8:
            return
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream().cuda_stream
            ref = self._ref
            ref.img.copy_(self.eng.img)
            ref.run_pack(st)
            ref.run_forward(st)
            self.eng.calibrate_fp8(ref, st, self.fp8_margin)
            self.eng.run_pack(st)
            self._graph = None

    def _run(self, st):
        self.eng.run_forward(st)
        L.check(self.eng.lib.abc_nms_peaks(C.byref(self._nms), st), "nms_peaks")
        if self.extractor is not None:
            self.extractor.run(st)

    def candidates(self):
        """the per-image atom / bond candidate lists of the last step (host sync; needs extract=True)"""
        if self.extractor is None:
            raise L.AbcNetHipError("InferenceRunner was built without extract=True")
        return self.extractor.lists()

    def step(self):
        """forward + NMS on the batch in the static image buffer; results in .logits / .atom_mask / ..."""
        with torch.cuda.device(self.dev):
            self._step()

    def _step(self):
        if self.use_graph and self._graph is None and self.steps >= 1:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._run(torch.cuda.current_stream().cuda_stream)
            self._graph = g
        if self._graph is not None:
            self._graph.replay()
        else:
            self._run(torch.cuda.current_stream().cuda_stream)
        self.steps += 1

    @property
    def logits(self):
        return self.eng.logits

    def profile(self, iters=3):
        """eager steps with a HIP event pair around every launch on the launch stream (as Trainer.profile)"""
        eng = self.eng
        torch.cuda.set_device(self.dev)
        stream = torch.cuda.current_stream(self.dev)
        st = stream.cuda_stream
        acc = {}
        for _ in range(iters):
            marks = []
            for fn, ref, what, _w, meta in eng.fwd_ops:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                rc = fn(ref, st)
                e1.record(stream)
                if rc != 0:
                    L.check(rc, what)
                marks.append((meta["kernel"], meta["flops"], meta["bytes"], e0, e1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            L.check(eng.lib.abc_nms_peaks(C.byref(self._nms), st), "nms_peaks")
            e1.record(stream)
            marks.append(("nms", 0.0, float(eng.B * eng.h * eng.w * (2 if self.nms_in_heads else 122) * 4 * 2), e0, e1))
            torch.cuda.synchronize()
            for k, fl, by, a, b in marks:
                r = acc.setdefault(k, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
                r["calls"] += 1
                r["ms"] += a.elapsed_time(b)
                r["flops"] += fl
                r["bytes"] += by
        for r in acc.values():
            for f in ("calls", "ms", "flops", "bytes"):
                r[f] /= iters
        return acc
