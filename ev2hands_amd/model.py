"""Drop-in host interface of the Ev2Hands per-frame inference path on MI355X.

`TEHNetWrapper` and `TEHNet` keep the reference's surface
(/root/reference/src/Ev2Hands/model/model.py:10-64, model/TEHNet.py:115-197): same constructor
arguments, `forward(xyz, mano_hands)` positional signature, output dict, `state_dict()` keys
(342 entries, strict-loadable), `module.` prefix stripping, `train()/eval()`, `.net/.hands/.rot`.
The module tree below exists only to hold parameters under the reference's names; all arithmetic
runs in libev2hands_hip.so through `ev2h_forward` (no torch ops, no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib, synth
from .mano import ManoHand, ManoOutput, create_mano_layers
from .pack import PackedWeights


# ------------------------------------------------------------------------------------ parameter containers
class _MsgParams(nn.Module):
    """names: conv_blocks.{i}.{j}, bn_blocks.{i}.{j} (pointnet2_utils.py:205-222)"""

    def __init__(self, fan_in, mlps):
        super().__init__()
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for mlp in mlps:
            convs, bns, last = nn.ModuleList(), nn.ModuleList(), fan_in
            for o in mlp:
                convs.append(nn.Conv2d(last, o, 1))
                bns.append(nn.BatchNorm2d(o))
                last = o
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)


class _StackParams(nn.Module):
    """names: mlp_convs.{k}, mlp_bns.{k} (pointnet2_utils.py:161-174, 265-274)"""

    def __init__(self, fan_in, mlp, dims):
        super().__init__()
        conv, bn = (nn.Conv2d, nn.BatchNorm2d) if dims == 2 else (nn.Conv1d, nn.BatchNorm1d)
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = fan_in
        for o in mlp:
            self.mlp_convs.append(conv(last, o, 1))
            self.mlp_bns.append(bn(o))
            last = o


class _RegressorParams(nn.Module):
    """names: sa1.*, sa2.*, mano_regressor.{0,2,4} (TEHNet.py:30-55)"""

    def __init__(self, n_inp_features=4, n_pose_params=6, n_shape_params=10):
        super().__init__()
        self.sa1 = _MsgParams(n_inp_features + 3, synth.MANO_SA1_MLPS)
        self.sa2 = _StackParams(512 + 3, synth.MANO_SA2_MLP, 2)
        self.n_pose_params = n_pose_params
        self.n_mano_params = n_pose_params + n_shape_params
        self.mano_regressor = nn.Sequential(nn.Linear(512, 1024), nn.ReLU(), nn.BatchNorm1d(1024), nn.Dropout(0.3),
                                            nn.Linear(1024, 3 + self.n_mano_params + 3))


def _query_conv():
    return nn.Sequential(nn.Conv1d(256, 256, 3, 1, 1), nn.ReLU(), nn.BatchNorm1d(256), nn.Dropout(0.1),
                         nn.Conv1d(256, 256, 3, 1, 1), nn.BatchNorm1d(256))


class TEHNet(nn.Module):
    """TEHNet.py:115-197.  Input channel count follows env ERPC at construction (TEHNet.py:122)."""

    def __init__(self, n_pose_params, num_classes=4):
        super().__init__()
        # n_pose_params: any number of MANO PCA coefficients (TEHNet.py:114-125; the reference's settings use 6, the layer has 45).
        # num_classes: the reference's own forward only runs with 4 -- the attention context has num_classes channels
        # (TEHNet.py:13-27) and MANORegressor is built for n_inp_features = 4 of them (TEHNet.py:31,144-145), so another value
        # raises inside its first regressor convolution
        if num_classes != 4:
            raise ValueError("num_classes must be 4: the reference's MANORegressor takes the attention's 4 class channels (TEHNet.py:31,144-145)")
        if not 1 <= int(n_pose_params) <= 45:
            raise ValueError("n_pose_params must be 1 .. 45 (MANO has 45 pose PCA components)")
        self.in_channels = 3 + 1 + int(os.getenv("ERPC", 0))
        self.n_pose_params = n_pose_params
        self.sa1 = _MsgParams(self.in_channels + 3, synth.SA1_MLPS)
        self.sa2 = _MsgParams(320 + 3, synth.SA2_MLPS)
        self.sa3 = _StackParams(512 + 3, synth.SA3_MLP, 2)
        self.fp3 = _StackParams(1536, synth.FP3_MLP, 1)
        self.fp2 = _StackParams(576, synth.FP2_MLP, 1)
        self.fp1 = _StackParams(128, synth.FP1_MLP, 1)
        self.classifier = nn.Sequential(nn.Conv1d(256, 256, 1), nn.ReLU(), nn.BatchNorm1d(256), nn.Dropout(0.3),
                                        nn.Conv1d(256, num_classes, 1))
        self.left_mano_regressor = _RegressorParams(n_pose_params=n_pose_params)
        self.right_mano_regressor = _RegressorParams(n_pose_params=n_pose_params)
        self.mhlnes = int(os.getenv("MHLNES", 0))
        # arithmetic of the matrix contractions: "f16x2" (default: fp32-class two-plane fp16 split with exact
        # per-window range scaling, any checkpoint / input magnitude), "bf16x3" (fp32-class three-plane bf16 split), "f32"
        # (exact fp32 MFMA), "bf16" (reduced precision)
        # "auto" (the DEFAULT since round 5): the first forward (and the first after the weights change) runs verify_precision on ITS
        # OWN batch and keeps "f16x2" only if every output agrees with "bf16x3" to AUTO_TOLERANCE and the segmentation argmax is
        # identical; else "bf16x3".  A user who drops in a checkpoint therefore gets the check without asking for it; benchmarks
        # and tests name their mode (EV2H_PRECISION / precision=).
        # "f16" [r6]: the reduced-precision mode (BASELINE config 3): ONE fp16 plane per operand under f16x2's range records -- bf16's
        # cost, 8 x finer.  f16_families (EV2H_F16_FAMILIES): _lib.FAM_* mask of the kernel families that run reduced in that mode
        # (0 / 15 = all; the others run f16x2).
        self.precision = os.getenv("EV2H_PRECISION", "auto")
        self.f16_families = int(os.getenv("EV2H_F16_FAMILIES", "0")) & 15
        self._auto = None             # (pack key of the weights, chosen mode, report) of the last "auto" decision
        # exact power-of-two equalisation of the hidden channels when the checkpoint is packed (ev2h_pack_weights, csrc/pack.hip: equalize_channels): the
        # fp32 function is unchanged bit for bit, the 16-bit planes see well-conditioned operands whatever the BatchNorm scales are
        self.equalize = True
        self.left_query_conv = _query_conv()
        self.right_query_conv = _query_conv()
        self._packed = None
        self._packed_key = None
        self._packed_spare = None     # (key, image) of the other candidate while "auto" decides
        self._keep_spare = False
        self._key_tensors = None
        self._key_gen = 0
        self._faces_cache = {}
        self._ws = None
        self.fps_init = None          # optional override: list of four [B] int64 tensors for the next forward

    # -- weight packing (re-done whenever parameters/buffers change or move) ---------------------
    def _key_list(self):
        # (walking the module tree costs ~2 ms per call -- more than a B = 1 forward; the tensor list is cached and dropped
        # whenever nn.Module machinery may have replaced tensor objects: _apply (.to / .cuda / .float) and load_state_dict)
        ts = self._key_tensors
        self._key_calls = getattr(self, "_key_calls", 0) + 1
        if ts is not None and self._key_calls % 64 == 0:
            # tensors can also be replaced behind nn.Module's back (net.sa1.conv_blocks[0][0].weight = nn.Parameter(..), a
            # submodule's load_state_dict(assign=True), parametrize / prune): re-walk the tree now and then and compare identities
            fresh = list(self.parameters()) + list(self.buffers())
            if len(fresh) != len(ts) or any(a is not b for a, b in zip(fresh, ts)):
                ts = None
        if ts is None:
            ts = self._key_tensors = list(self.parameters()) + list(self.buffers())
            self._key_gen += 1
        return ts

    def _pack_key(self):
        """changes whenever a parameter / buffer is modified in place, replaced or moved"""
        ts = self._key_list()
        return (str(ts[0].device), self._key_gen, [t.data_ptr() for t in ts], [t._version for t in ts])

    def _version_key(self):
        """the cheap part of _pack_key (in-place modifications, module-level replacement): what a captured graph re-checks per replay"""
        ts = self._key_list()
        return (self._key_gen, [t._version for t in ts], [t.data_ptr() for t in ts])     # data_ptr: `p.data = new_tensor` keeps _version

    def _apply(self, fn, *args, **kwargs):
        self._key_tensors = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._key_tensors = None
        return super().load_state_dict(*args, **kwargs)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Parameter, nn.Module)):
            self.__dict__["_key_tensors"] = None
        super().__setattr__(name, value)

    # What "auto" accepts.  Two CORRECT fp32-class evaluations of a trained network differ by ~1.5e-5 (exact-fp32 MFMA vs bf16x3 vs
    # f16x2 on the optimiser-made checkpoints: 1.0e-5 .. 1.7e-5 pairwise, profiles/r5_trained_precision_report.txt; the exact-fp32 mode
    # itself sits 1.7e-5 from the reference's CPU sums on one fixture) -- round 4's 1e-5 was calibrated on hash-random weights
    # (1e-6) and would have sent every trained checkpoint to bf16x3 for no gain in accuracy.  A range failure of the two-plane split
    # shows as 6e-4 .. 0.2 (un-equalised checkpoints, tests/test_gpu_guard.py): half the 1e-4 parity bar separates the two.
    AUTO_TOLERANCE = 5e-5

    def effective_precision(self) -> str:
        """the arithmetic mode the next forward runs in ("auto" resolved; before its first decision: "f16x2")"""
        if self.precision != "auto":
            return self.precision
        return self._auto[1] if self._auto is not None else "f16x2"

    def packed(self, device) -> PackedWeights:
        prec = self.effective_precision()
        key = (str(device), prec, self.equalize, self.f16_families, self._pack_key())
        if self._packed is None or self._packed_key != key:
            # one spare image: while "auto" decides (f16x2 vs bf16x3 on the first batch) both candidates stay packed, so that the
            # forward after the decision does not pack the winner a second time; the loser is dropped by _auto_decide
            spare = getattr(self, "_packed_spare", None)
            if spare is not None and spare[0] == key:
                self._packed_spare = (self._packed_key, self._packed) if self._keep_spare else None
                self._packed, self._packed_key = spare[1], key
                return self._packed
            if self._keep_spare and self._packed is not None:
                self._packed_spare = (self._packed_key, self._packed)
            else:
                self._packed_spare = None
                self._packed = None               # (free the old device image first)
            self._packed = PackedWeights(self.state_dict(), device, self.in_channels, prec, equalize=self.equalize, f16_families=self.f16_families)
            self._packed_key = key
        return self._packed

    def workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != torch.device(device):
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    # -- forward ----------------------------------------------------------------------------------
    def draw_fps_init(self, B: int, N: int):
        """The reference seeds each farthest-point sampling with torch.randint on the global CPU
        RNG (pointnet2_utils.py:75); same draws, same order: enc.sa1, enc.sa2, left.sa1, right.sa1."""
        return [torch.randint(0, hi, (B,), dtype=torch.long) for hi in (N, synth.SA1_NPOINT, N, N)]

    @staticmethod
    def _inits_to_device(inits, device) -> torch.Tensor:
        """[4, B] int64 on the device.  Host tensors -- the reference's own case: torch.randint on the CPU generator,
        pointnet2_utils.py:75 -- go through PINNED memory: an "asynchronous" copy out of pageable memory blocks the host until the
        runtime has staged it, which put a host synchronisation in front of every forward and took the whole gain of forwards in
        flight (16 x 8192, two in flight: 7 490 windows/s with a list of host tensors against 10 470 with a device tensor,
        tools/debug/inflight_graph.py).  The caching host allocator keeps the pinned block alive until the copy has run."""
        if torch.is_tensor(inits):
            st = inits.to(torch.long)
        else:
            st = torch.stack([t.to(torch.long) for t in inits])
        if st.device.type == "cpu":
            pin = torch.empty(st.shape, dtype=torch.long, pin_memory=True)
            pin.copy_(st)
            return pin.to(device, non_blocking=True)
        return st.to(device, non_blocking=True).contiguous()

    def _check_input(self, xyz):
        if self.training:
            raise NotImplementedError("ev2hands_amd implements the inference forward (net.eval()) only")
        if xyz.dim() != 3 or xyz.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input [B, {self.in_channels}, N], got {tuple(xyz.shape)}")
        if not xyz.is_cuda:
            raise RuntimeError("ev2hands_amd runs on the GPU only (there is no CPU fallback); move the input to cuda")
        if xyz.dtype != torch.float32:
            raise RuntimeError("input must be float32")
        if self.mhlnes and not xyz.is_contiguous():
            raise RuntimeError("MHLNES=1 writes channel 2 in place and needs a contiguous input")

    def _enqueue(self, x, init_dev, mano_hands, rows=None, ws=None):
        """One ev2h_forward on the current stream: device work only (no host synchronisation, no host->device copy), so that it
        can run under stream capture.  x [B,C,N] contiguous float32, init_dev [4,B] int64 on the device.
        rows: optional float32 [B, >= dist.packed_width(N)] matrix (row stride = its second dimension) that receives every window's
        predictions side by side -- ev2h_outputs' window strides; the returned tensors are then views of it.  ws: workspace to use
        instead of the net's shared one (captured graphs own theirs)."""
        device = x.device
        B, Cin, N = x.shape
        L = _lib.lib()
        pw = self.packed(device)
        # hand models from ev2hands_amd.create_mano_layers run inside ev2h_forward; any other object with the reference adapter's
        # interface (model/utils.py:14-31: .shapedirs, .faces, __call__(global_orient, hand_pose, betas, transl) -> .vertices,
        # .joints) is called with the regressed parameters exactly as TEHNet.py:92-105 does
        native = {s: isinstance(mano_hands[s], ManoHand) for s in ("left", "right")}
        consts = {s: (C.byref(mano_hands[s].consts()) if native[s] else None) for s in ("left", "right")}

        f32 = dict(device=device, dtype=torch.float32)
        out = _lib.Outputs()
        NP, NVF = 3 + self.n_pose_params + 10 + 3, synth.MANO_NV * 3
        if rows is None:
            logits = torch.empty(B, 4, N, **f32)
            params = [torch.empty(B, NP, **f32) for _ in range(2)]
            verts = [torch.empty(B, synth.MANO_NV, 3, **f32) for _ in range(2)]
            joints = [torch.empty(B, 21, 3, **f32) for _ in range(2)]
        else:
            W = rows.shape[1] if rows.dim() == 2 else 0
            per_hand = NP + NVF + 63
            if (rows.dim() != 2 or rows.shape[0] != B or W < 4 * N + 2 * per_hand or rows.dtype != torch.float32 or rows.device != device
                    or rows.stride() != (W, 1)):
                raise RuntimeError(f"rows must be a float32 [B={B}, >= {4 * N + 2 * per_hand}] row-major matrix on {device}")
            logits = rows[:, :4 * N].view(B, 4, N)
            params, verts, joints = [], [], []
            for h in range(2):
                o = 4 * N + h * per_hand
                params.append(rows[:, o:o + NP])
                verts.append(rows[:, o + NP:o + NP + NVF].view(B, synth.MANO_NV, 3))
                joints.append(rows[:, o + NP + NVF:o + per_hand].view(B, 21, 3))
            out.logits_stride = out.params_stride = out.vertices_stride = out.joints_stride = W
        out.class_logits = logits.data_ptr()
        for h in range(2):
            out.params[h] = params[h].data_ptr()
            out.vertices[h] = verts[h].data_ptr()
            out.joints[h] = joints[h].data_ptr()
        nbytes = L.ev2h_workspace_bytes(B, N)
        if ws is None:
            ws = self.workspace(nbytes, device)
        elif ws.numel() < nbytes or ws.device != device:
            raise RuntimeError("workspace too small or on another device")
        with torch.cuda.device(device):
            # [r6] the side stream of THIS caller stream is chosen by measurement the first time (ev2h_bind_stream: a stream that shares
            # the caller's hardware queue would silently serialise the forward's overlaps); afterwards the call returns at once
            _lib.check(L.ev2h_bind_stream(_lib.stream_handle(), None), "ev2h_bind_stream")
            _lib.check(L.ev2h_forward(C.byref(pw.struct), consts["left"], consts["right"], x.data_ptr(), B, Cin, N, self.mhlnes,
                                      init_dev.data_ptr(), C.byref(out), ws.data_ptr(), nbytes, _lib.stream_handle()),
                       "ev2h_forward")
        res = {"class_logits": logits}
        npose = self.n_pose_params
        for h, side in enumerate(("left", "right")):
            p = params[h]
            d = {"global_orient": p[:, :3], "hand_pose": p[:, 3:3 + npose], "betas": p[:, 3 + npose:-3], "transl": p[:, -3:]}
            if native[side]:
                d = {"vertices": verts[h], "j3d": joints[h], **d}
            else:
                hd = mano_hands[side].shapedirs.device                       # TEHNet.py:92-105
                d = {k: v.to(hd) for k, v in d.items()}
                o = mano_hands[side](**d)
                d = {"vertices": o.vertices, "j3d": o.joints, **d}
                if rows is not None:                                         # a foreign hand model's results join the row as well
                    verts[h].copy_(o.vertices.to(device))
                    joints[h].copy_(o.joints.to(device))
            res[side] = d
        self._last_shape = (B, N)
        self._last_ws = ws
        return res

    # -- the f16x2 guard: what the DATA does to the split arithmetic ---------------------------------
    def range_report(self) -> dict:
        """After a forward in "f16x2": for every contraction operand that is materialised in the workspace (ev2h_range_report),
        {"nonzero", "below_2^-17", "below_2^-28"}: int64 [B] counts per window of the values that sit more than ~2^17 / ~2^28 below
        the window's maximum -- the ones the two-plane fp16 split resolves less finely than fp32 -- plus "worst_fraction" =
        max over the windows of below_2^-17 / nonzero."""
        B, N = self._last_shape
        L = _lib.lib()
        names = (C.c_char_p * 64)()
        n = L.ev2h_range_report_entries(names, 64)
        ws = self._last_ws
        counts = torch.zeros(n, B, 3, dtype=torch.int32, device=ws.device)
        with torch.cuda.device(ws.device):
            _lib.check(L.ev2h_range_report(ws.data_ptr(), B, N, counts.data_ptr(), _lib.stream_handle()), "ev2h_range_report")
        c = counts.cpu().to(torch.int64)
        out = {}
        for i in range(n):
            nz, lo, hi = c[i, :, 0], c[i, :, 1], c[i, :, 2]
            out[names[i].decode()] = {"nonzero": nz, "below_2^-17": lo, "below_2^-28": hi,
                                      "worst_fraction": float((lo.double() / nz.clamp(min=1).double()).max())}
        return out

    def verify_precision(self, xyz, mano_hands, fps_init=None, reference: str = "bf16x3") -> dict:
        """Run THIS batch in "f16x2" and in `reference` ("bf16x3": three exact bf16 planes, 8 exponent bits, no range assumption;
        or "f32") with the same FPS start indices and compare: {"max_rel": worst per-tensor relative difference
        (max|a - b| / max|b|) over class_logits and both hands' parameters / vertices / joints, "per_output": ..., "argmax_agreement":
        fraction of points with the same segmentation class, "range": range_report() of the f16x2 run, "weights": packed-weight
        spread, "ok": max_rel <= AUTO_TOLERANCE and identical argmax}.  Costs two forwards and two packs; the network's precision
        setting is restored."""
        self._check_input(xyz)
        B, _, N = xyz.shape
        inits = fps_init if fps_init is not None else (self.fps_init if self.fps_init is not None else self.draw_fps_init(B, N))
        keep = self.precision
        outs = {}
        try:
            for prec in ("f16x2", reference):
                self.precision = prec
                self.fps_init = inits
                x = xyz.clone() if self.mhlnes else xyz
                with torch.no_grad():
                    o = self.forward(x, mano_hands)
                outs[prec] = {"class_logits": o["class_logits"].clone(),
                              **{f"{s}.{k}": o[s][k].clone() for s in ("left", "right") for k in ("global_orient", "hand_pose", "betas", "transl", "vertices", "j3d")}}
                if prec == "f16x2":
                    rng = self.range_report()
                    wsp = self._packed.weight_spread()
        finally:
            self.precision = keep
            self.fps_init = None
        a, b = outs["f16x2"], outs[reference]
        per = {k: float((a[k] - b[k]).abs().max() / b[k].abs().max().clamp(min=1e-30)) for k in a}
        agree = float((a["class_logits"].argmax(1) == b["class_logits"].argmax(1)).float().mean())
        nzw = sum(v[0] for v in wsp.values())
        rep = {"max_rel": max(per.values()), "per_output": per, "argmax_agreement": agree, "reference": reference, "range": rng,
               "range_worst": max(rng.items(), key=lambda kv: kv[1]["worst_fraction"])[0] if rng else None,
               "weights": {"nonzero": nzw, "below_2^-17": sum(v[1] for v in wsp.values())}}
        rep["ok"] = bool(rep["max_rel"] <= self.AUTO_TOLERANCE and agree == 1.0)
        return rep

    def _auto_decide(self, xyz, mano_hands, inits):
        """The "auto" decision (once per set of weights): f16x2 against bf16x3 on THIS batch.  Costs two extra forwards and two packs
        (both images stay packed until the decision is taken; the loser is dropped).
        Under torch.distributed the decision is COLLECTIVE: every rank verifies on its own shard and the group takes the minimum
        (one small all-reduce), so that all ranks run the same arithmetic -- a lone rank on bf16x3 would break sharded == unsharded
        and stall every all-gather at 0.6 x the others' rate.  Every rank therefore has to reach its first forward
        (and the first after a weight change); name a mode (precision= / EV2H_PRECISION) where ranks may diverge in that."""
        key = (str(xyz.device), self.equalize, self._pack_key())
        if self._auto is not None and self._auto[0] == key:
            return
        self._auto = None
        self.precision = "f16x2"
        self._keep_spare = True
        try:
            rep = self.verify_precision(xyz, mano_hands, fps_init=inits)
        finally:
            self.precision = "auto"
            self._keep_spare = False
        ok = bool(rep["ok"])
        rep["ok_this_rank"] = ok
        try:
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
                from .dist import _group_moves_device_memory
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=xyz.device if _group_moves_device_memory(None) else "cpu")
                tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
                ok = bool(int(t.item()))
                rep["ranks"] = tdist.get_world_size()
        except ImportError:
            pass
        rep["ok"] = ok
        self._auto = (key, "f16x2" if ok else "bf16x3", rep)
        # drop the image that lost (the winner is `_packed` or the spare; packed() swaps it in without packing again)
        want = (str(xyz.device), self._auto[1], self.equalize, self.f16_families, self._pack_key())
        if self._packed_key != want and self._packed_spare is not None and self._packed_spare[0] == want:
            self._packed, self._packed_key = self._packed_spare[1], want
        self._packed_spare = None

    @property
    def auto_report(self):
        """verify_precision's report behind the current "auto" choice (None before the first forward)"""
        return None if self._auto is None else self._auto[2]

    def forward(self, xyz, mano_hands, rows=None, ws=None):
        """TEHNet.py:168-197.  rows (extension, optional): float32 [B, >= 4N + 2 * 2419] matrix that receives each window's
        predictions as one row ([4N logits | left 22 params, 778x3 vertices, 21x3 joints | right ...], ev2hands_amd/dist.py);
        the returned tensors are then views of it -- the multi-GPU path passes its slice of the all-gather buffer.
        ws (extension, optional): a workspace of its own for this call (uint8, >= ev2h_workspace_bytes(B, N)) -- forwards that
        are in flight at the same time on different streams must not share one (ev2hands_amd/inflight.py)."""
        self._check_input(xyz)
        device = xyz.device
        B, Cin, N = xyz.shape
        x = xyz if self.mhlnes else xyz.contiguous()
        inits = self.fps_init if self.fps_init is not None else self.draw_fps_init(B, N)
        self.fps_init = None
        if self.precision == "auto":
            self._auto_decide(xyz, mano_hands, inits)
        init_dev = self._inits_to_device(inits, device)
        res = self._enqueue(x, init_dev, mano_hands, rows=rows, ws=ws)
        for side in ("left", "right"):
            res[side]["faces"] = self._tiled_faces(mano_hands[side], B)          # eval only (TEHNet.py:109-110)
        return res

    def _tiled_faces(self, hand, B: int):
        """np.tile(faces, (B, 1, 1)) as TEHNet.py:110 returns it, built once per (hand model, B): at B = 256 the two tilings are
        19 MB of host copies per forward."""
        key = (id(hand), id(hand.faces), B)
        hit = self._faces_cache.get(key)
        if hit is None:
            if len(self._faces_cache) > 8:
                self._faces_cache.clear()
            tiled = np.tile(hand.faces, (B, 1, 1))
            tiled.flags.writeable = False        # shared between forwards (the reference returns a fresh array each time): editing it in place must raise
            hit = self._faces_cache[key] = (tiled, hand, hand.faces)
        return hit[0]

    def capture(self, xyz, mano_hands, fps_init=None) -> "CapturedForward":
        """Capture one forward for inputs of xyz's shape into a hipGraph (torch.cuda.CUDAGraph: stream capture of the kernel
        sequence ev2h_forward enqueues, including its fork onto the library's side stream).  Replays cost one graph launch
        instead of ~90 kernel launches -- what matters at the reference's operating point, one small batch at a time
        (demo.py:24-33).  The returned object owns static input / FPS-init / output tensors; `replay(xyz, fps_init)` copies new
        inputs in and returns the (static) output dict."""
        self._check_input(xyz)
        if not all(isinstance(mano_hands[s], ManoHand) for s in ("left", "right")):
            raise TypeError("capture needs hand models from ev2hands_amd.create_mano_layers (foreign hand models run host code)")
        return CapturedForward(self, xyz, mano_hands, fps_init)

    def debug_buffer(self, name: str, dtype=torch.float32) -> torch.Tensor:
        """Copy of a named workspace buffer of the last forward (parity tests)."""
        B, N = self._last_shape
        L = _lib.lib()
        cnt = C.c_size_t(0)
        ws = self._last_ws
        et = C.c_int(0)
        p = L.ev2h_workspace_buffer_ex(ws.data_ptr(), B, N, name.encode(), C.byref(cnt), C.byref(et))
        if not p:
            raise KeyError(name)
        off = p - ws.data_ptr()
        if et.value in (1, 2):        # the last forward stored this buffer as 16-bit values (BF16 / F16 mode: l0): widened here, `cnt` counts VALUES
            if dtype != torch.float32:
                raise TypeError(f"workspace buffer {name!r} holds {'bf16' if et.value == 1 else 'fp16'} values in this mode: ask for float32")
            if et.value == 1:
                return ws[off:off + cnt.value * 2].view(torch.bfloat16).float()
            # F16: fp16 values times the window's power of two (workspace "p1scale" row 5): undone here, exactly
            v = ws[off:off + cnt.value * 2].view(torch.float16).float().view(B, -1)
            return (v / self.debug_buffer("p1scale").view(6, B)[5].view(B, 1)).view(-1)
        return ws[off:off + cnt.value * 4].view(dtype).clone()


class CapturedForward:
    """hipGraph of TEHNet.forward for one input shape (TEHNet.capture)."""

    def __init__(self, net: TEHNet, xyz, mano_hands, fps_init=None):
        device = xyz.device
        B, _, N = xyz.shape
        self.net, self.hands, self.B, self.N = net, mano_hands, B, N
        self.x = xyz.detach().clone().contiguous()
        inits = fps_init if fps_init is not None else net.draw_fps_init(B, N)
        self.init = torch.stack([t.to(torch.long) for t in inits]).to(device).contiguous()
        # The graph bakes in device addresses: the workspace and the packed weights.  Both are owned HERE -- the net's shared
        # workspace is re-allocated by a later, larger eager forward, and the packed weights are dropped when the parameters or
        # the precision change (a replay would then read and write freed memory).
        self._ws = torch.empty(_lib.lib().ev2h_workspace_bytes(B, N), dtype=torch.uint8, device=device)
        if net.precision == "auto":
            net._auto_decide(self.x, mano_hands, [t.cpu() for t in self.init])
        self._pw = net.packed(device)
        self._key = (net.effective_precision(), net.equalize, net._version_key())
        for s in ("left", "right"):
            mano_hands[s].consts()
        self._mano_keep = {s: (mano_hands[s]._keep, mano_hands[s].shapedirs._version) for s in ("left", "right")}    # MANO constant tensors
        with torch.no_grad():
            net._enqueue(self.x, self.init, mano_hands, ws=self._ws)   # warm-up outside the capture: kernel attributes, occupancy queries
            torch.cuda.synchronize(device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = net._enqueue(self.x, self.init, mano_hands, ws=self._ws)
        self.faces = {s: np.tile(mano_hands[s].faces, (B, 1, 1)) for s in ("left", "right")}
        for s in ("left", "right"):
            self.faces[s].flags.writeable = False             # shared by every replay
            self.out[s]["faces"] = self.faces[s]
        if net.mhlnes:
            self.x.copy_(xyz)                                 # MHLNES=1 overwrote channel 2 during warm-up and capture

    def replay(self, xyz=None, fps_init=None) -> dict:
        """Run the captured forward (optionally on new inputs of the captured shape).  The returned tensors are the graph's static
        outputs: they are overwritten by the next replay."""
        if (self.net.effective_precision(), self.net.equalize, self.net._version_key()) != self._key:
            raise RuntimeError("the network's parameters, device or precision changed since this forward was captured: the graph "
                               "holds the old packed weights -- capture again")
        if any(self.hands[s].shapedirs._version != self._mano_keep[s][1] for s in ("left", "right")):
            raise RuntimeError("a hand model's shapedirs changed since this forward was captured -- capture again")
        if xyz is not None:
            if tuple(xyz.shape) != tuple(self.x.shape):
                raise RuntimeError(f"captured for input {tuple(self.x.shape)}, got {tuple(xyz.shape)}")
            self.x.copy_(xyz, non_blocking=True)
        if fps_init is not None:                              # (host tensors go through pinned memory: TEHNet._inits_to_device)
            self.init.copy_(TEHNet._inits_to_device(fps_init, self.init.device), non_blocking=True)
        elif xyz is not None:
            self.init.copy_(TEHNet._inits_to_device(self.net.draw_fps_init(self.B, self.N), self.init.device), non_blocking=True)     # reference RNG order
        self.graph.replay()
        if self.net.mhlnes and xyz is not None:
            xyz[:, 2].copy_(self.x[:, 2])                     # the in-place overwrite of TEHNet.py:176-177 reaches the caller's tensor
        return self.out


# ------------------------------------------------------------------------------------ wrapper
def _rotation_x_180() -> torch.Tensor:
    """trimesh.transformations.rotation_matrix(radians(180), [1,0,0]) (model.py:59), closed form."""
    c, s = math.cos(math.pi), math.sin(math.pi)
    return torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, c, -s, 0.0], [0.0, s, c, 0.0], [0.0, 0.0, 0.0, 1.0]],
                        dtype=torch.float64)


class TEHNetWrapper:
    """model.py:10-64.  `mano_assets` / `mano_path` say where the MANO constants come from (the
    reference reads settings.MANO_PATH, which cannot be imported without pyrender)."""

    def __init__(self, device, mano_path="../data/models", mano_assets=None, precision=None, n_pose_params=None):
        # n_pose_params: the reference reads settings.MANO_CMPS (= 6, model.py:52-55); another value (1 .. 45) is an extension of
        # the constructor only -- checkpoint head width, MANO layer and the `rows=` layout follow it
        n_pose = synth.MANO_CMPS if n_pose_params is None else int(n_pose_params)
        self.net = TEHNet(n_pose_params=n_pose).to(device)
        if precision is not None:           # otherwise EV2H_PRECISION, default "auto" (f16x2 after a self-check on the first batch)
            self.net.precision = precision
        self.net.eval()
        self.training = False
        if torch.device(device).type == "cuda":
            # create the forward's side stream now: streams created later (torch's pool, RCCL) must not push it onto the caller's
            # hardware queue (ev2h_init)
            with torch.cuda.device(device):
                _lib.check(_lib.lib().ev2h_init(), "ev2h_init")
        self.hands = create_mano_layers(mano_path, device, n_pose, assets=mano_assets)
        self.rot = _rotation_x_180().to(device).float()

    def state_dict(self):
        return self.net.state_dict()

    def load_state_dict(self, params, *args, **kwargs):
        stripped = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in params.items()}
        # The input width is fixed at construction by the environment (TEHNet.py:122: 4 channels, 5 with ERPC=1); a checkpoint
        # trained with the other setting fails strict loading with a size mismatch buried in a long message -- say what it is.
        w = stripped.get("sa1.conv_blocks.0.0.weight")
        if w is not None and w.dim() == 4 and int(w.shape[1]) - 3 != self.net.in_channels:
            c = int(w.shape[1]) - 3
            raise RuntimeError(f"checkpoint expects {c} input channels (trained with ERPC={int(c == 5)}) but this model was built for "
                               f"{self.net.in_channels} (ERPC={os.getenv('ERPC', '0')} when it was constructed): set ERPC={int(c == 5)} "
                               f"before creating TEHNetWrapper (the reference reads it at construction, TEHNet.py:122)")
        return self.net.load_state_dict(stripped, *args, **kwargs)

    def parameters(self):
        return self.net.parameters()

    def train(self):
        self.training = True
        return self.net.train()

    def eval(self):
        self.training = False
        return self.net.eval()

    def P3dtoP2d(self, j3d, scale, translation):
        B, N = j3d.shape[:2]
        h = torch.cat([j3d, torch.ones(B, N, 1, device=j3d.device)], 2) @ self.rot.detach()
        translation = translation.unsqueeze(1)
        scale = scale.unsqueeze(1)
        j2d = torch.zeros(B, N, 2, device=j3d.device)
        j2d[:, :, 0] = translation[:, :, 0] + scale[:, :, 0] * h[:, :, 0]
        j2d[:, :, 1] = translation[:, :, 1] + scale[:, :, 1] * h[:, :, 1]
        return j2d

    def __call__(self, inp):
        return self.net(inp, self.hands)

    def capture(self, inp, fps_init=None):
        """hipGraph of `self(inp)` for inputs of inp's shape: `g = net.capture(inp); out = g.replay(new_inp)` (TEHNet.capture)."""
        return self.net.capture(inp, self.hands, fps_init)
