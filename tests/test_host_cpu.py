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
    sizes = (C.c_size_t * 7)()
    built.ev2h_struct_sizes(sizes)
    mine = [C.sizeof(t) for t in (_lib.GemmDesc, _lib.SaDesc, _lib.SaModule, _lib.Weights, _lib.ManoConsts, _lib.Outputs, _lib.FpDesc)]
    assert list(sizes) == mine
    assert built.ev2h_abi_version() == 4


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


def test_packed_weight_layouts():
    from ev2hands_amd.pack import PackedWeights
    sd = synth.synth_state_dict(5, 3)
    pw = PackedWeights(sd, "cpu", 5)
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
