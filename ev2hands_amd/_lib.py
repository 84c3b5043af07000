"""ctypes binding of libev2hands_hip.so (include/ev2hands_hip.h).

The product path has no CPU fallback: if the shared library is missing or fails to load,
`lib()` raises.  Build it with `python -m ev2hands_amd.build` (or __graft_entry__.build()).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EV2H_LIB_PATH: another BUILD of the same library (tools/asan_host.sh: host code under AddressSanitizer / UBSan); never a fallback --
# a missing file still raises
LIB_PATH = os.environ.get("EV2H_LIB_PATH") or os.path.join(_HERE, "libev2hands_hip.so")

vp = C.c_void_p
ci = C.c_int


class GemmDesc(C.Structure):
    _fields_ = [("X", vp), ("ldx", ci), ("W", vp), ("ldw", ci), ("Y", vp), ("ldy", ci),
                ("M", ci), ("N", ci), ("K", ci),
                ("bias", vp), ("bias_group_rows", ci), ("ldbias", ci), ("relu", ci),
                ("post_scale", vp), ("post_shift", vp),
                ("taps", ci), ("rows_per_seq", ci), ("rowmax_rows", ci), ("precision", ci), ("Ws", vp), ("ws_tile_rows", ci), ("w_unscale", C.c_float),
                ("x_amax", vp), ("x_amax2", vp), ("x_group_rows", ci), ("y_amax", vp), ("y_group_rows", ci), ("y_scale", vp),
                ("y_bound_w", C.c_float), ("y_bound_b", C.c_float), ("skinny", ci)]


class SaDesc(C.Structure):
    _fields_ = [("P1", vp), ("ldp", ci), ("pts4", vp), ("ctr4", vp), ("gidx", vp),
                ("W1x", vp), ("W2", vp), ("b2", vp), ("W3", vp), ("b3", vp),
                ("out", vp), ("ldo", ci),
                ("B", ci), ("Npts", ci), ("S", ci), ("K", ci),
                ("C1", ci), ("C2", ci), ("C3", ci), ("precision", ci), ("W2s", vp), ("W3s", vp), ("cnt", vp), ("w2_unscale", C.c_float), ("w3_unscale", C.c_float), ("cnt_ld", ci),
                ("p1_scale", vp), ("p1_amax", vp), ("w1x_norm", C.c_float), ("dmax", C.c_float), ("w2_norm", C.c_float), ("b2_max", C.c_float),
                ("out_amax", vp), ("feat", vp), ("ldf", ci), ("W1f", vp), ("ldw1f", ci), ("b1", vp), ("nfeat", ci),
                ("feat_amax", vp), ("w1f_norm", C.c_float), ("b1_max", C.c_float), ("w1f_unscale", C.c_float), ("w1x_unscale", C.c_float),
                ("xyz_out", vp), ("xyz_ld", ci), ("S_total", ci), ("s_off", ci)]


class FpDesc(C.Structure):
    _fields_ = [("T", vp), ("ldt", ci), ("nn_idx", vp), ("nn_w", vp), ("b2", vp), ("b3", vp), ("W2s", vp), ("W3s", vp),
                ("w2_unscale", C.c_float), ("w3_unscale", C.c_float), ("out", vp), ("ldo", ci),
                ("B", ci), ("N", ci), ("S", ci), ("C1", ci), ("C2", ci), ("C3", ci), ("precision", ci),
                ("t_scale", vp), ("t_amax", vp), ("w2_norm", C.c_float), ("b2_max", C.c_float), ("out_amax", vp),
                ("out_cols", ci), ("no_relu_out", ci), ("out_cm", vp), ("out_cm_stride", C.c_size_t)]


class SaBranch(C.Structure):
    _fields_ = [("W1x", vp), ("W2", vp), ("b2", vp), ("W3", vp), ("b3", vp),
                ("C1", ci), ("C2", ci), ("C3", ci), ("K", ci), ("radius", C.c_double), ("W2s", vp), ("W3s", vp),
                ("w2_unscale", C.c_float), ("w3_unscale", C.c_float), ("w1x_norm", C.c_float), ("w2_norm", C.c_float), ("b2_max", C.c_float),
                ("w1f_unscale", C.c_float), ("w1x_unscale", C.c_float), ("w3_norm", C.c_float), ("b3_max", C.c_float)]


class SaModule(C.Structure):
    _fields_ = [("W1f", vp), ("b1", vp), ("kf", ci), ("npoint", ci), ("nbranch", ci), ("br", SaBranch * 3),
                ("w1f_unscale", C.c_float), ("W1fs", vp), ("w1f_norm", C.c_float), ("b1_max", C.c_float)]


class Dense(C.Structure):
    _fields_ = [("W", vp), ("b", vp), ("post_scale", vp), ("post_shift", vp), ("O", ci), ("K", ci), ("ldw", ci), ("Ws", vp), ("ws_tile_rows", ci), ("w_unscale", C.c_float)]


class Weights(C.Structure):
    _fields_ = [("sa1", SaModule), ("sa2", SaModule), ("mano_sa1", SaModule * 2),
                ("sa3", Dense * 3),
                ("fp3_skip", Dense), ("fp3_bcast", Dense), ("fp3_1", Dense),
                ("fp2", Dense * 2), ("fp1", Dense * 3), ("fp1m", SaModule),
                ("cls0", Dense), ("cls4", Dense), ("clsm", SaBranch),
                ("qconv0", Dense), ("qconv4", Dense * 2), ("qconv4T", vp * 2),
                ("mano_sa2", (Dense * 2) * 2),
                ("head0", Dense * 2), ("head4", Dense * 2), ("precision", ci), ("l0_unscale", vp), ("flags", ci), ("f16_families", ci)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", vp), ("dtype", ci), ("ndim", ci), ("shape", C.c_int64 * 4)]


DT_F32, DT_F64, DT_I64 = 0, 1, 2
PACK_EQUALIZE, PACK_HOST_ONLY, PACK_UNEQUALIZED_OK = 1, 2, 4
FAM_SA, FAM_ROWS, FAM_QCONV, FAM_DENSE, FAM_ALL = 1, 2, 4, 8, 15      # EV2H_FAM_*: kernel families of the "f16" mode (ev2h_weights.f16_families)


def pack_f16_families(mask: int) -> int:
    """EV2H_PACK_F16_FAMILIES(mask)"""
    return (int(mask) & 15) << 8
W_EQUALIZED, W_UNEQUALIZED_OK = 1, 2
ABI_VERSION = 8     # 8: EV2H_PREC_F16 (one fp16 plane with f16x2's range machinery); 3: F16X2 range records; 4: ev2h_fp_mlp, ev2h_weights.fp1m; 5: window strides of the outputs; 6: ev2h_pack_weights, ev2h_weights.flags; 7: ev2h_sa_desc.xyz_out

PREC = {"f32": 0, "bf16": 1, "f16x2": 2, "bf16x3": 3, "f16": 4}


class ManoConsts(C.Structure):
    _fields_ = [("hands_mean", vp), ("comps", vp), ("blend_T", vp), ("v_template", vp),
                ("J_template", vp), ("J_shape", vp), ("weights", vp),
                ("tips", C.c_int32 * 5), ("ncomps", C.c_int32)]


class Outputs(C.Structure):
    _fields_ = [("class_logits", vp), ("params", vp * 2), ("vertices", vp * 2), ("joints", vp * 2),
                ("logits_stride", C.c_size_t), ("params_stride", C.c_size_t), ("vertices_stride", C.c_size_t), ("joints_stride", C.c_size_t)]


EXPORTS = [
    "ev2h_abi_version", "ev2h_source_hash", "ev2h_build_defs", "ev2h_last_error", "ev2h_init", "ev2h_set_side_stream", "ev2h_side_stream_probe", "ev2h_streams_concurrent", "ev2h_bind_stream", "ev2h_shader_clock_probe", "ev2h_struct_sizes",
    "ev2h_prep_points", "ev2h_fps", "ev2h_fps_multi", "ev2h_ball_query", "ev2h_three_nn_interp",
    "ev2h_gemm", "ev2h_transpose_logits", "ev2h_sa_mlp_max", "ev2h_fp_mlp", "ev2h_tile_geometry",
    "ev2h_attn_sim", "ev2h_attn_sim_folded", "ev2h_attn_sim_folded_scratch", "ev2h_attn_context", "ev2h_mano", "ev2h_mano_rotations",
    "ev2h_workspace_bytes", "ev2h_forward", "ev2h_workspace_buffer", "ev2h_workspace_buffer_ex", "ev2h_profile_set", "ev2h_range_report_entries", "ev2h_range_report",
    "ev2h_pack_weights", "ev2h_packed_free", "ev2h_packed_weights", "ev2h_packed_bytes", "ev2h_packed_tensor_count", "ev2h_packed_tensor",
    "ev2h_packed_equalization_count", "ev2h_packed_equalization", "ev2h_packed_weight_spread_count", "ev2h_packed_weight_spread", "ev2h_pack_sa_image_bytes", "ev2h_pack_sa_images",
    "ev2h_pack_gemm_image_bytes", "ev2h_pack_gemm_image", "ev2h_plane_unscale",
    "ev2h_event_window_build", "ev2h_event_window_timesort", "ev2h_event_window_sample", "ev2h_joint_metrics", "ev2h_mesh_collisions", "ev2h_mesh_collisions_ws", "ev2h_mesh_collisions_scratch_bytes", "ev2h_collision_penalty",
]

_lib = None


class Ev2hError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported before the library: both depend on libamdhip64 and the process must end up with ONE HIP
    # runtime (torch's bundled copy).  Loading ours first pulls /opt/rocm's copy and leaves it without a device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise Ev2hError(f"{LIB_PATH} is missing: build the HIP library first "
                        f"(python -m ev2hands_amd.build). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    # [r6] the library must have been built from THESE sources: the sha256 of csrc/ + include/ is compiled in (ev2hands_amd/build.py).
    # `*.so` is git-ignored yet travels to the GPU box, so a stale binary next to newer sources is otherwise possible and silent.
    # EV2H_LIB_PATH (another build of the library, e.g. tools/asan_host.sh) opts out.
    if not os.environ.get("EV2H_LIB_PATH"):
        from . import build as _build
        try:
            L.ev2h_source_hash.restype = C.c_char_p
            have = L.ev2h_source_hash().decode()
        except AttributeError:
            have = "(no stamp: built before round 6)"
        want = _build.source_hash()
        if have != want:
            raise Ev2hError(f"{LIB_PATH} was built from other sources (stamp {have[:16]}, sources {want[:16]}): rebuild it "
                            f"(python -m ev2hands_amd.build), or set EV2H_LIB_PATH to use another build on purpose.")
    L.ev2h_last_error.restype = C.c_char_p
    L.ev2h_workspace_bytes.restype = C.c_size_t
    L.ev2h_workspace_bytes.argtypes = [ci, ci]
    L.ev2h_workspace_buffer.restype = vp
    L.ev2h_workspace_buffer.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(C.c_size_t)]
    L.ev2h_workspace_buffer_ex.restype = vp
    L.ev2h_workspace_buffer_ex.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(ci)]
    L.ev2h_struct_sizes.restype = None
    L.ev2h_struct_sizes.argtypes = [C.c_size_t * 8]
    L.ev2h_set_side_stream.argtypes = [ci]
    L.ev2h_side_stream_probe.argtypes = [vp, ci, C.POINTER(C.c_float)]
    L.ev2h_shader_clock_probe.argtypes = [vp, ci, vp]
    L.ev2h_streams_concurrent.argtypes = [vp, vp, ci, C.POINTER(C.c_float)]
    L.ev2h_bind_stream.argtypes = [vp, C.POINTER(ci)]
    L.ev2h_prep_points.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp]
    L.ev2h_fps.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
    L.ev2h_fps_multi.argtypes = [vp, ci, ci, ci, C.POINTER(ci), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]
    L.ev2h_ball_query.argtypes = [vp, vp, ci, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(ci), C.POINTER(vp), vp, vp]
    L.ev2h_three_nn_interp.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, vp, ci, vp, vp, vp, vp]
    L.ev2h_gemm.argtypes = [C.POINTER(GemmDesc), vp]
    L.ev2h_transpose_logits.argtypes = [vp, ci, ci, vp, C.c_size_t, vp]
    L.ev2h_sa_mlp_max.argtypes = [C.POINTER(SaDesc), vp]
    L.ev2h_fp_mlp.argtypes = [C.POINTER(FpDesc), vp]
    L.ev2h_tile_geometry.argtypes = [ci, ci, ci, ci, C.c_int * 10]
    L.ev2h_attn_sim.argtypes = [vp, vp, ci, C.c_size_t, ci, ci, vp, vp]
    L.ev2h_attn_context.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp, vp]
    L.ev2h_attn_sim_folded_scratch.restype = C.c_size_t
    L.ev2h_attn_sim_folded_scratch.argtypes = [ci, ci]
    L.ev2h_attn_sim_folded.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    L.ev2h_mano.argtypes = [C.POINTER(ManoConsts), vp, ci, ci, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.ev2h_mano_rotations.argtypes = [C.POINTER(ManoConsts), vp, ci, ci, vp, vp]
    L.ev2h_forward.argtypes = [C.POINTER(Weights), C.POINTER(ManoConsts), C.POINTER(ManoConsts), vp, ci, ci, ci, ci, vp,
                               C.POINTER(Outputs), vp, C.c_size_t, vp]
    L.ev2h_event_window_build.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp]
    L.ev2h_event_window_timesort.argtypes = [vp, vp, ci, vp, ci, ci, vp, ci, vp, vp, vp]
    L.ev2h_event_window_sample.argtypes = [vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, vp, vp]
    L.ev2h_joint_metrics.argtypes = [vp, vp, vp, ci, ci, ci, C.c_double, vp, vp, vp, vp, vp, vp]
    L.ev2h_mesh_collisions.argtypes = [vp, vp, vp, vp, ci, ci, ci, C.c_float, ci, vp, vp, ci, vp]
    L.ev2h_mesh_collisions_ws.argtypes = [vp, vp, vp, vp, ci, ci, ci, C.c_float, ci, vp, vp, ci, vp, C.c_size_t, vp]
    L.ev2h_mesh_collisions_scratch_bytes.restype = C.c_size_t
    L.ev2h_mesh_collisions_scratch_bytes.argtypes = [ci, ci]
    L.ev2h_collision_penalty.argtypes = [vp, vp, vp, vp, ci, ci, ci, C.c_float, C.c_double, vp, vp, ci, vp, vp]
    L.ev2h_range_report_entries.argtypes = [C.POINTER(C.c_char_p), ci]
    L.ev2h_range_report.argtypes = [vp, ci, ci, vp, vp]
    L.ev2h_profile_set.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(vp), ci]
    L.ev2h_pack_weights.argtypes = [C.POINTER(TensorDesc), ci, ci, ci, ci, C.POINTER(vp)]
    L.ev2h_packed_free.restype = None
    L.ev2h_packed_free.argtypes = [vp]
    L.ev2h_packed_weights.restype = vp
    L.ev2h_packed_weights.argtypes = [vp]
    L.ev2h_packed_bytes.restype = C.c_size_t
    L.ev2h_packed_bytes.argtypes = [vp]
    L.ev2h_packed_tensor_count.argtypes = [vp]
    L.ev2h_packed_tensor.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(vp), C.POINTER(vp)]
    L.ev2h_packed_equalization_count.argtypes = [vp]
    L.ev2h_packed_equalization.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_double)), C.POINTER(ci)]
    L.ev2h_packed_weight_spread_count.argtypes = [vp]
    L.ev2h_packed_weight_spread.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.c_uint64 * 3]
    L.ev2h_pack_sa_image_bytes.argtypes = [ci, ci, ci, ci, C.c_size_t * 2]
    L.ev2h_pack_sa_images.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ev2h_pack_gemm_image_bytes.restype = C.c_size_t
    L.ev2h_pack_gemm_image_bytes.argtypes = [ci, ci, ci, ci]
    L.ev2h_pack_gemm_image.argtypes = [vp, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    L.ev2h_plane_unscale.restype = C.c_float
    L.ev2h_plane_unscale.argtypes = [vp, C.c_size_t, ci]
    sizes = (C.c_size_t * 8)()
    L.ev2h_struct_sizes(sizes)
    mine = [C.sizeof(t) for t in (GemmDesc, SaDesc, SaModule, Weights, ManoConsts, Outputs, FpDesc, TensorDesc)]
    if list(sizes) != mine:
        raise Ev2hError(f"struct layout mismatch between ev2hands_hip.h and _lib.py: {list(sizes)} vs {mine}")
    if L.ev2h_abi_version() != ABI_VERSION:
        raise Ev2hError(f"ABI version mismatch: library {L.ev2h_abi_version()}, binding {ABI_VERSION}")
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ev2h_last_error().decode(errors="replace")
        raise Ev2hError(f"{what} failed (rc={rc}): {msg}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor, 0 for None."""
    return 0 if t is None else t.data_ptr()


def stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def side_stream_probe(spin_us: int = 50) -> float:
    """ev2h_side_stream_probe on the current device and stream: (time of one spin kernel on each of the caller's stream and the
    library's side stream at once) / (time of one alone).  ~1.0: the two streams run concurrently; ~2.0: they share a hardware
    queue and the forward's overlaps are lost (create the wrapper / call ev2h_init() before other code creates streams,
    INTEGRATION.md section 3).  Raises if the side stream is switched off."""
    r = C.c_float(0.0)
    check(lib().ev2h_side_stream_probe(stream_handle(), int(spin_us), C.byref(r)), "ev2h_side_stream_probe")
    return float(r.value)


def streams_concurrent(a: int, b: int, spin_us: int = 40) -> float:
    """ev2h_streams_concurrent on two raw stream handles of the current device: ~1.0-1.3 concurrent, ~2 = one hardware queue"""
    r = C.c_float(0.0)
    check(lib().ev2h_streams_concurrent(a, b, int(spin_us), C.byref(r)), "ev2h_streams_concurrent")
    return float(r.value)


def bind_stream(handle: int | None = None) -> tuple:
    """ev2h_bind_stream for a raw stream handle (default: the current stream): (candidates tried, ratio of the chosen pair, earlier-bound
    streams still sharing a queue with it); (0, 0.0, 0) when there was nothing to do (bound before, capturing, single-stream mode)."""
    info = (ci * 3)()
    check(lib().ev2h_bind_stream(stream_handle() if handle is None else handle, info), "ev2h_bind_stream")
    return int(info[0]), info[1] / 1000.0, int(info[2])


def concurrent_streams(device, n: int, pool: int = 8) -> list:
    """n torch streams of `device` that the device runs CONCURRENTLY with each other, as far as the hardware queues allow: torch hands
    out streams of a pool round robin and HIP maps them onto a few hardware queues, so two fresh streams may well share one and then
    run in order.  Greedy: keep a candidate if it is concurrent with every stream kept so far; if the pool is exhausted first, the
    rest are taken as they come (correct, only serialised)."""
    import torch
    device = torch.device(device)
    with torch.cuda.device(device):
        cands = [torch.cuda.Stream(device) for _ in range(max(pool, n))]
        kept = []
        for c in cands:
            if len(kept) == n:
                break
            if all(streams_concurrent(k.cuda_stream, c.cuda_stream) < 1.6 for k in kept):
                kept.append(c)
        for c in cands:
            if len(kept) == n:
                break
            if all(c is not k for k in kept):
                kept.append(c)
    return kept


class ShaderClockSampler:
    """Shader clock observed WHILE a workload runs (ev2h_shader_clock_probe): `sample()` enqueues a one-wave probe on a stream of
    its own (it runs beside whatever the other streams are doing); `mhz()` synchronises that stream and returns the clocks of all
    samples taken so far.  bench.py: `value_sustained`."""

    def __init__(self, device, max_samples: int = 64, spin_us: int = 200):
        import torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.buf = torch.zeros(max_samples, 2, dtype=torch.int64, device=self.device)
        self.n, self.spin_us = 0, int(spin_us)

    def sample(self) -> None:
        import torch
        if self.n >= self.buf.shape[0]:
            return
        with torch.cuda.device(self.device):
            check(lib().ev2h_shader_clock_probe(self.stream.cuda_stream, self.spin_us, self.buf[self.n].data_ptr()), "ev2h_shader_clock_probe")
        self.n += 1

    def mhz(self) -> list:
        self.stream.synchronize()
        v = self.buf[:self.n].cpu().double()
        return [float(100.0 * c / r) for c, r in v.tolist() if r > 0]
