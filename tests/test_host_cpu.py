"""CPU: the C-ABI library loads and exports every symbol of include/ev2hands_hip.h, struct layouts agree,
the host-side interface mirrors the reference's (state_dict schema, module. prefix, error behaviour),
weight packing produces the documented layouts, and the product path refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from ev2hands_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from ev2hands_amd import build
    build.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "ev2hands_hip.h")).read()
    declared = set(re.findall(r"\b(ev2h_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(built, name), f"{name} declared in ev2hands_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)


def test_struct_layouts_and_abi_version(built):
    sizes = (C.c_size_t * 8)()
    built.ev2h_struct_sizes(sizes)
    mine = [C.sizeof(t) for t in (_lib.GemmDesc, _lib.SaDesc, _lib.SaModule, _lib.Weights, _lib.ManoConsts, _lib.Outputs, _lib.FpDesc, _lib.TensorDesc)]
    assert list(sizes) == mine
    assert built.ev2h_abi_version() == 8 == _lib.ABI_VERSION


def test_workspace_size_grows_linearly(built):
    a = built.ev2h_workspace_bytes(1, 2048)
    b = built.ev2h_workspace_bytes(2, 2048)
    c = built.ev2h_workspace_bytes(4, 2048)
    assert 0 < a < b < c and abs((c - b) - 2 * (b - a)) < 64 * 1024
    assert built.ev2h_workspace_bytes(0, 2048) == 0


def test_bad_arguments_return_error_codes_not_crashes(built):
    d = _lib.GemmDesc()
    assert built.ev2h_gemm(C.byref(d), None) != 0
    assert b"bad argument" in built.ev2h_last_error()
    s = _lib.SaDesc()
    assert built.ev2h_sa_mlp_max(C.byref(s), None) != 0
    f = _lib.FpDesc()
    assert built.ev2h_fp_mlp(C.byref(f), None) != 0
    # a chain shape the fused row kernel does not have: a clean error before anything is launched
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p).value
    f.T, f.ldt, f.b2, f.b3, f.W2s, f.W3s, f.out, f.ldo = p, 64, p, p, p, p, p, 64
    f.B, f.N, f.C1, f.C2, f.C3, f.precision = 1, 32, 64, 64, 64, _lib.PREC["f16x2"]
    assert built.ev2h_fp_mlp(C.byref(f), None) != 0 and b"unsupported chain" in built.ev2h_last_error()
    f.ldt, f.C1, f.C2, f.C3, f.precision = 256, 256, 256, 32, _lib.PREC["f32"]
    assert built.ev2h_fp_mlp(C.byref(f), None) != 0 and b"16-bit" in built.ev2h_last_error()
    assert built.ev2h_attn_sim_folded(None, None, 512, 1, 128, None, None, None, None, None, None, None) != 0
    assert built.ev2h_attn_sim_folded_scratch(2, 2048) == 2 * 8 * 12 * 512 and built.ev2h_attn_sim_folded_scratch(1, 130) == 12 * 512


def test_checkpoint_schema():
    for C_ in (4, 5):
        sch = synth.checkpoint_schema(C_)
        assert len(sch) == 342
        n = sum(int(np.prod(s)) for s, k in sch.values() if k != "bn_count" and not k.startswith("bn_mean") and k != "bn_var")
        assert n == {4: 4494676, 5: 4494836}[C_]            # SURVEY.md 8b parameter counts
    assert sch["sa1.conv_blocks.0.0.weight"][0] == (32, 8, 1, 1)
    assert sch["left_query_conv.4.weight"][0] == (256, 256, 3)


@pytest.mark.parametrize("C_", [4, 5])
def test_wrapper_state_dict_interface(C_):
    from ev2hands_amd.model import TEHNetWrapper
    os.environ["ERPC"] = "1" if C_ == 5 else "0"
    assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
    net = TEHNetWrapper("cpu", mano_assets=assets)
    sd = synth.synth_state_dict(C_, 0)
    assert list(net.state_dict().keys()) == list(sd.keys())
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    net.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)     # model.py:14-23
    assert torch.equal(net.state_dict()["fp1.mlp_convs.2.weight"], sd["fp1.mlp_convs.2.weight"])
    bad = dict(sd)
    bad["sa2.conv_blocks.0.0.weight"] = torch.zeros(128, 300, 1, 1)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    assert net.training is False and net.net.training is False
    net.train()
    assert net.training and net.net.training
    net.eval()
    assert set(net.hands) == {"left", "right"} and net.hands["left"].faces.shape == (1538, 3)
    assert tuple(net.rot.shape) == (4, 4) and abs(float(net.rot[1, 1]) + 1) < 1e-6
    j2d = net.P3dtoP2d(torch.ones(2, 21, 3), torch.ones(2, 2) * 2, torch.ones(2, 2) * 3)
    assert tuple(j2d.shape) == (2, 21, 2) and abs(float(j2d[0, 0, 0]) - 5.0) < 1e-6 and abs(float(j2d[0, 0, 1]) - 1.0) < 1e-6


def test_no_cpu_fallback():
    from ev2hands_amd.model import TEHNetWrapper
    os.environ["ERPC"] = "0"
    assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
    net = TEHNetWrapper("cpu", mano_assets=assets)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 4, 256))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 5, 256))          # wrong channel count is reported before anything runs


def test_lib_path_override_is_not_a_fallback():
    """EV2H_LIB_PATH selects another BUILD of the library (tools/asan_host.sh); a path that does not exist must raise, not fall back to
    the in-tree build or to anything on the CPU"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from ev2hands_amd import _lib\ntry:\n    _lib.lib()\nexcept _lib.Ev2hError as e:\n    print('RAISED', e)\n"
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, EV2H_LIB_PATH="/nonexistent/libev2hands_hip.so"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert "RAISED" in r.stdout and "/nonexistent/libev2hands_hip.so is missing" in r.stdout, r.stdout


def test_packed_weight_layouts():
    from ev2hands_amd.pack import PackedWeights
    sd = synth.synth_state_dict(5, 3)
    pw = PackedWeights(sd, "cpu", 5, equalize=False)      # (the equalisation multiplies rows / columns by powers of two: test_range_host.py)
    t = pw.tensors
    assert tuple(t["sa1.W1f"].shape) == (160, 8) and tuple(t["sa2.W1f"].shape) == (256, 320)
    assert tuple(t["sa2.1.W2"].shape) == (224, 128) and float(t["sa2.1.W2"][196:].abs().max()) == 0.0
    assert tuple(t["sa2.1.W3"].shape) == (256, 200) and float(t["sa2.1.W3"][:, 196:].abs().max()) == 0.0
    assert tuple(t["sa3.0.W"].shape) == (256, 520) and float(t["sa3.0.W"][:, 515:].abs().max()) == 0.0
    assert tuple(t["qconv0.W"].shape) == (512, 768) and tuple(t["fp3.bcast.W"].shape) == (256, 1024)
    # BN folding: first sa1 conv, branch 0 -- features part and xyz part
    W = sd["sa1.conv_blocks.0.0.weight"][:, :, 0, 0].double()
    a = sd["sa1.bn_blocks.0.0.weight"].double() / torch.sqrt(sd["sa1.bn_blocks.0.0.running_var"].double() + 1e-5)
    assert torch.allclose(t["sa1.W1f"][:32, :5].double(), (W * a[:, None])[:, :5], atol=1e-7)
    assert torch.allclose(t["sa1.0.W1x"][:, :3].double(), (W * a[:, None])[:, 5:8], atol=1e-7)
    # group-all column permutation [xyz | feat] -> [feat | xyz | pad]
    W0 = sd["sa3.mlp_convs.0.weight"][:, :, 0, 0].double()
    a0 = sd["sa3.mlp_bns.0.weight"].double() / torch.sqrt(sd["sa3.mlp_bns.0.running_var"].double() + 1e-5)
    assert torch.allclose(t["sa3.0.W"][:, 512:515].double(), (W0 * a0[:, None])[:, :3], atol=1e-7)
    assert torch.allclose(t["sa3.0.W"][:, :512].double(), (W0 * a0[:, None])[:, 3:], atol=1e-7)
    # k=3 conv is tap-major
    Wq = sd["right_query_conv.0.weight"]
    assert torch.equal(t["qconv0.W"][256:, 256:512], Wq[:, :, 1])
    assert pw.struct.sa1.nbranch == 3 and pw.struct.mano_sa1[1].br[1].C2 == 196 and pw.struct.qconv0.K == 256


def test_mano_pkl_reader_without_chumpy(tmp_path):
    """The chumpy-free unpickler on a file shaped like MANO_RIGHT.pkl (arrays wrapped in chumpy.Ch objects)."""
    import pickle
    import sys
    import types
    from ev2hands_amd import mano
    a = synth.synth_mano_assets("right", 1)
    # build a fake 'chumpy' module only to WRITE the fixture; it is removed before reading
    ch = types.ModuleType("chumpy")
    chch = types.ModuleType("chumpy.ch")

    Ch = type("Ch", (), {"__init__": lambda self, x: setattr(self, "x", x), "__getstate__": lambda self: {"x": self.x}})
    Ch.__module__ = "chumpy.ch"
    Ch.__qualname__ = "Ch"
    chch.Ch = Ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = ch, chch
    try:
        import scipy.sparse as sp
        d = {"v_template": a["v_template"], "shapedirs": Ch(a["shapedirs"]), "posedirs": a["posedirs"],
             "J_regressor": sp.csc_matrix(a["J_regressor"]), "weights": a["weights"],
             "hands_components": a["hands_components"], "hands_mean": a["hands_mean"], "f": a["faces"].astype(np.uint32),
             "kintree_table": np.array([[4294967295] + a["parents"][1:], list(range(16))], dtype=np.int64)}
        p = tmp_path / "MANO_RIGHT.pkl"
        with open(p, "wb") as f:
            pickle.dump(d, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    got = mano.load_mano_pkl(str(p), "right")
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "hands_components", "hands_mean"):
        assert np.allclose(got[k], a[k]), k
    assert got["parents"] == a["parents"] and np.array_equal(got["faces"], a["faces"])


def test_tile_geometry_comes_from_the_kernels(built):
    """One source of truth for the weight tile images: the library exports the kernels' compile-time geometry and
    its packer (csrc/pack.hip) asserts its own numbers against it on every pack, as does the numpy restatement tests/ref_pack.py;
    a drift raises at pack time, not in a GPU parity test.  No GPU needed."""
    import ctypes as C
    from ev2hands_amd import pack
    out = (C.c_int * 10)()
    assert built.ev2h_tile_geometry(128, 196, 256, 2, out) == 0
    assert list(out) == [7, 208, 144, 848, 7 * 32 * 144, 32 * 848, 144, 32, 4, 1]     # 4 leftover channels share MFMAs (f16x2 only); W2 k slots in D-register order
    assert built.ev2h_tile_geometry(128, 196, 256, 3, out) == 0 and out[8] == 0
    assert built.ev2h_tile_geometry(100, 100, 100, 2, out) != 0 and b"unsupported chain" in built.ev2h_last_error()
    rng = np.random.default_rng(0)
    for (c1, c2, c3) in ((32, 32, 64), (64, 96, 128), (128, 196, 256), (256, 256, 32)):
        for ns in (1, 2, 3):
            i2, i3, u2, u3 = pack.sa_bf16_images(rng.normal(size=(c2, c1)), rng.normal(size=(c3, c2)), ns)     # asserts inside
            g = pack.kernel_geometry(c1, c2, c3, ns)
            assert i2.size == (c1 // 32) * g["TB2"] and i3.size == (c3 // 32) * g["TB3"]
    # a drifted packer is caught (shown on the restatement, whose geometry function can be patched; the C packer runs the same check)
    import ref_pack
    real = ref_pack.sa_bf16_geometry
    ref_pack.sa_bf16_geometry = lambda C2: (real(C2)[0], real(C2)[1] + 16)
    try:
        with pytest.raises(Exception, match="tile geometry"):
            ref_pack.sa_bf16_images(rng.normal(size=(196, 128)), rng.normal(size=(256, 196)), 2)
    finally:
        ref_pack.sa_bf16_geometry = real


# ---- real-asset hardening: the day MANO_{LEFT,RIGHT}.pkl and best_model_state_dict.pth appear -----------------
class _Py2Pickler(__import__("pickle")._Pickler):
    """Writes what Python 2's cPickle wrote at protocol 2 -- the format of the licensed MANO files: `str` and raw array buffers as
    (SHORT_)BINSTRING (bytes that Python 3 has to decode with encoding='latin1'), classes under their 2017 module paths
    (numpy.core.multiarray, scipy.sparse.csc, chumpy.ch).  Pure-Python pickler with the str / bytes / global writers replaced."""
    import pickle as _p
    import struct as _s
    dispatch = dict(_p._Pickler.dispatch)

    def _binstring(self, b):
        p, s = self._p, self._s
        self.write((p.SHORT_BINSTRING + bytes([len(b)]) if len(b) < 256 else p.BINSTRING + s.pack("<i", len(b))) + b)

    def save_str(self, obj):
        self._binstring(obj.encode("latin1"))
        self.memoize(obj)

    def save_bytes(self, obj):
        self._binstring(obj)
        self.memoize(obj)

    def save_global(self, obj, name=None):
        name = name or getattr(obj, "__qualname__", obj.__name__)
        module = getattr(obj, "__module__", None) or self._p.whichmodule(obj, name)
        module = {"numpy._core.multiarray": "numpy.core.multiarray", "numpy._core.numeric": "numpy.core.numeric", "numpy": "numpy",
                  "copyreg": "copy_reg"}.get(module, module)
        self.write(self._p.GLOBAL + module.encode() + b"\n" + name.encode() + b"\n")
        self.memoize(obj)

    dispatch[str] = save_str
    dispatch[bytes] = save_bytes


def _fake_class(module, name):
    """A picklable new-style class living under a foreign module path, state = its __dict__ (what chumpy.Ch and scipy's matrices do)."""
    import sys
    import types
    mod = sys.modules.get(module) or types.ModuleType(module)
    cls = type(name, (), {"__module__": module})
    setattr(mod, name, cls)
    return mod, cls


def test_mano_pkl_reader_on_a_python2_style_file(tmp_path):
    """The reader on a file with the REAL files' format features: Python-2 protocol-2 byte strings, chumpy objects pickled as
    NEWOBJ + a state dict that carries chumpy's bookkeeping next to 'x', `scipy.sparse.csc.csc_matrix` under its 2017 module path
    with the 2017 attribute set, uint32 faces, the 4294967295 root parent.  Neither chumpy nor the pickled scipy layout is needed."""
    import io
    import sys
    from ev2hands_amd import mano
    a = synth.synth_mano_assets("left", 2)
    saved = {k: sys.modules.get(k) for k in ("chumpy", "chumpy.ch", "scipy.sparse.csc")}
    made = {}
    try:
        for m in ("chumpy", "chumpy.ch"):
            made[m], Ch = _fake_class(m, "Ch")
            sys.modules[m] = made[m]
        made["scipy.sparse.csc"], Csc = _fake_class("scipy.sparse.csc", "csc_matrix")
        sys.modules["scipy.sparse.csc"] = made["scipy.sparse.csc"]

        def ch(x):
            o = Ch()
            o.__dict__.update({"x": np.asarray(x), "_dirty_vars": set(), "_itr": None, "_depends_on_deps": {}, "_status": "new", "_cache": {"drs": {}}})
            return o

        jr = a["J_regressor"]
        cols, rows = np.nonzero(jr.T)                                         # column-major order of the non-zeros
        m = Csc()
        m.__dict__.update({"_shape": jr.shape, "maxprint": 50, "data": jr[rows, cols].copy(), "indices": rows.astype(np.int32),
                           "indptr": np.concatenate([[0], np.cumsum((jr != 0).sum(0))]).astype(np.int32), "_has_sorted_indices": True})
        d = {"v_template": ch(a["v_template"]), "shapedirs": ch(a["shapedirs"]), "posedirs": ch(a["posedirs"]), "J_regressor": m,
             "weights": ch(a["weights"]), "hands_components": a["hands_components"], "hands_mean": a["hands_mean"],
             "hands_coeffs": np.zeros((3, 45)), "bs_style": "lbs", "bs_type": "lrotmin", "J": ch(np.zeros((16, 3))),
             "f": a["faces"].astype(np.uint32), "kintree_table": np.array([[4294967295] + a["parents"][1:], list(range(16))], dtype=np.int64)}
        buf = io.BytesIO()
        _Py2Pickler(buf, protocol=2).dump(d)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    raw = buf.getvalue()
    assert b"cchumpy.ch\nCh\n" in raw and b"cscipy.sparse.csc\ncsc_matrix\n" in raw and b"cnumpy.core.multiarray\n_reconstruct\n" in raw
    assert b"U\nv_template" in raw                                            # SHORT_BINSTRING key: a Python-2 str
    p = tmp_path / "MANO_LEFT.pkl"
    p.write_bytes(raw)
    assert "chumpy" not in sys.modules
    got = mano.load_mano_pkl(str(p), "left")
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "hands_components", "hands_mean"):
        assert np.array_equal(got[k], a[k]), k
    assert got["parents"] == a["parents"] and np.array_equal(got["faces"], a["faces"]) and got["faces"].dtype == np.int64


def test_checkpoint_with_the_other_input_width_is_refused_clearly():
    """A C=4 checkpoint (ERPC=0) loaded into a model built with ERPC=1, and the other way round: a clear error naming ERPC, not a
    size mismatch buried in strict loading; the validation tool refuses the same mismatch."""
    from ev2hands_amd.model import TEHNetWrapper
    assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
    for env, ck_c in (("1", 4), ("0", 5)):
        os.environ["ERPC"] = env
        net = TEHNetWrapper("cpu", mano_assets=assets)
        with pytest.raises(RuntimeError, match="ERPC"):
            net.load_state_dict({"module." + k: v for k, v in synth.synth_state_dict(ck_c, 1).items()}, strict=True)
        net.load_state_dict(synth.synth_state_dict(9 - ck_c, 1), strict=True)          # the matching width loads
    os.environ["ERPC"] = "0"


def test_only_test_infrastructure_imports_the_oracle():
    """oracle/ is the checker: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it -- never the product
    package, and no tool either (fuzzers / diagnostics that need it live under tests/)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    offenders = []
    for d, _, files in os.walk(root):
        rel = os.path.relpath(d, root)
        if rel.split(os.sep)[0] in ("tests", "oracle", ".git", "gpurun_out", "__pycache__", ".pytest_cache"):
            continue
        for f in files:
            if f.endswith(".py") and pat.search(open(os.path.join(d, f), errors="replace").read()):
                offenders.append(os.path.join(rel, f))
    assert sorted(offenders) == ["./__graft_entry__.py", "./bench.py"], offenders
    # ... and inside those two only where the contract allows it
    bench = open(os.path.join(root, "bench.py")).read()
    assert bench.count("from oracle import") == 1 and bench.split("from oracle import")[0].rsplit("\ndef ", 1)[1].startswith("cpu_baseline(")
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    assert entry.count("from oracle import") == 1 and entry.split("from oracle import")[0].rsplit("\ndef ", 1)[1].startswith("smoke(")


def test_surface_like_synthetic_hand_assets():
    """synth_mano_surface_assets: same shapes / invariants as the parity assets (rows of J_regressor and of the skinning weights
    non-negative and summing to 1, 3 joints per vertex), but the template is a surface: neighbouring faces share edges, and a posed
    pair of hands collides in tens of triangle pairs, not tens of thousands (checked with the collision oracle)."""
    from oracle import collision_oracle as CO, mano_oracle
    a = {s: synth.synth_mano_surface_assets(s, 0) for s in ("left", "right")}
    b = synth.synth_mano_assets("left", 0)
    for k, v in b.items():
        if hasattr(v, "shape"):
            assert a["left"][k].shape == v.shape, k
    for s in ("left", "right"):
        jr, w = a[s]["J_regressor"], a[s]["weights"]
        assert (jr >= 0).all() and np.allclose(jr.sum(1), 1) and (w >= 0).all() and np.allclose(w.sum(1), 1) and ((w > 0).sum(1) <= 3).all()
        f = a[s]["faces"]
        assert f.min() == 0 and f.max() == 777
        edges = np.sort(np.concatenate([f[:1533, [0, 1]], f[:1533, [1, 2]], f[:1533, [2, 0]]]), 1)
        _, cnt = np.unique(edges, axis=0, return_counts=True)
        assert (cnt <= 2).all() and (cnt == 2).mean() > 0.98          # a manifold with one open rim (the wrist)
    hands = mano_oracle.make_hands(a["left"], a["right"])
    g = torch.Generator().manual_seed(0)
    v = {}
    for s in ("left", "right"):
        o = hands[s](global_orient=torch.randn(2, 3, generator=g) * 0.2, hand_pose=torch.randn(2, 6, generator=g) * 0.3,
                     betas=torch.randn(2, 10, generator=g) * 0.3, transl=torch.randn(2, 3, generator=g) * 0.05)
        v[s] = o.vertices.numpy()
    for i in range(2):
        vv, ff = CO.build_triangles(v["left"][i], v["right"][i], a["left"]["faces"], a["right"]["faces"], scale=1000.0)
        assert CO.collision_pairs(vv, ff, 8).shape[0] < 1000


def test_bench_launch_site_matching_is_independent_of_template_arguments():
    """bench.py picks the dispatches of its two launch sites out of a rocprofv3 counter table: the set-abstraction kernel by its
    widths, the k = 3 query convolution as the dense-layer launch with the largest grid -- in every arithmetic mode, whatever
    trailing template arguments the kernels have grown (a hard-coded `<2, true>` matched nothing at N = 8192)."""
    import bench
    for gemm in ("gemm_nt_bf16_occ_kernel<2, true, false>", "gemm_nt_bf16_occ_kernel<1, true, false, 7>", "gemm_nt_kernel"):
        rows = [("void (anonymous namespace)::sa_mlp_max_bf16_kernel<128, 196, 256, 2, false, 0>(SaArgs)", 1048576, 1, 5.0),
                ("void (anonymous namespace)::sa_mlp_max_bf16_kernel<128, 128, 256, 2, false, 0>(SaArgs)", 1048576, 2, 3.0),
                (gemm, 8388608, 3, 7.0), (gemm.replace("true", "false"), 262144, 4, 1.0), (gemm, 8388608, 5, 9.0)]
        assert bench.site_dispatch_values("sa2.1", rows) == [5.0]
        assert bench.site_dispatch_values("qconv0", rows) == [7.0, 9.0]
    assert bench.site_dispatch_values("qconv0", [("fps_kernel<8>", 256, 1, 1.0)]) == []
    assert {s["tag"] for s in bench.launch_sites(8192)} == {"sa2.1", "qconv0"} and bench.launch_sites(8192)[0]["tag"] == "qconv0"
    assert bench.launch_sites(2048)[0]["tag"] == "sa2.1"


def test_every_environment_switch_of_the_library_is_listed_in_the_gpu_switch_test():
    """grep of csrc/: the set of getenv() names is exactly the tested set (a new switch must come with its test)."""
    import re
    from test_gpu_forward import AB_SWITCHES
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ev2hands_amd", "csrc")
    found = set()
    for fn in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))):
        found |= set(re.findall(r'getenv\("(EV2H_[A-Z0-9_]+)"\)', open(os.path.join(csrc, fn)).read()))
    found -= {"EV2H_PACK_HOST_ONLY"}          # not an A/B path: pack without a device (CPU layout tests)
    assert found == {n for n, _, _ in AB_SWITCHES}, found ^ {n for n, _, _ in AB_SWITCHES}
    assert len(found) <= 8          # keep the A/B surface small


def test_default_arithmetic_mode_is_auto(monkeypatch):
    """A user who names no mode gets "auto" (f16x2 after a self-check against bf16x3 on the first batch); EV2H_PRECISION and
    precision= still choose explicitly."""
    from ev2hands_amd.model import TEHNetWrapper
    monkeypatch.delenv("EV2H_PRECISION", raising=False)
    assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
    assert TEHNetWrapper("cpu", mano_assets=assets).net.precision == "auto"
    assert TEHNetWrapper("cpu", mano_assets=assets, precision="bf16x3").net.precision == "bf16x3"
    monkeypatch.setenv("EV2H_PRECISION", "f32")
    assert TEHNetWrapper("cpu", mano_assets=assets).net.precision == "f32"
    assert TEHNetWrapper("cpu", mano_assets=assets).net.AUTO_TOLERANCE == 5e-5


def test_a_library_built_from_other_sources_is_refused(tmp_path, built):
    """[r6] The binary is git-ignored but travels (gpurun snapshot, copies between machines): ev2hands_amd/build.py stamps the
    sha256 of csrc/ + include/ into it and _lib.lib() recomputes the hash of the sources next to it.  A copy of the package whose
    sources differ by ONE byte from what its .so was built from must refuse to load (and say how to fix it); the same copy with
    EV2H_LIB_PATH naming the binary explicitly loads (the documented opt-out); needs_build() is the same comparison, so
    __graft_entry__.build() either matches provably or rebuilds."""
    import shutil
    import subprocess
    import sys
    from ev2hands_amd import build
    assert build.built_hash() == build.source_hash() and not build.needs_build()
    assert built.ev2h_source_hash().decode() == build.source_hash()
    root = tmp_path / "copy"
    shutil.copytree(os.path.join(ROOT, "ev2hands_amd"), root / "ev2hands_amd", ignore=shutil.ignore_patterns(".obj", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    probe = ("import sys; sys.path.insert(0, %r)\nfrom ev2hands_amd import _lib, build\nprint('needs_build', build.needs_build())\n"
             "_lib.lib(); print('loaded')\n" % str(root))
    env = {k: v for k, v in os.environ.items() if k != "EV2H_LIB_PATH"}
    ok = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert ok.returncode == 0 and "needs_build False" in ok.stdout and "loaded" in ok.stdout, ok.stdout + ok.stderr
    with open(root / "ev2hands_amd" / "csrc" / "mano.hip", "a") as f:
        f.write("\n")                                              # one byte
    bad = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert bad.returncode != 0 and "needs_build True" in bad.stdout and "built from other sources" in bad.stderr and "loaded" not in bad.stdout, bad.stdout + bad.stderr
    env["EV2H_LIB_PATH"] = str(root / "ev2hands_amd" / "libev2hands_hip.so")
    opt = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert opt.returncode == 0 and "loaded" in opt.stdout, opt.stdout + opt.stderr
