"""ev2h_pack_weights (csrc/pack.hip) against the independent numpy restatement tests/ref_pack.py: every packed array, every plane
image and every scalar of ev2h_weights byte for byte, for plain and stress checkpoints, both input widths, all four arithmetic
modes, with and without the channel equalisation; plus the C ABI's error behaviour (strict schema, as
/root/reference/src/Ev2Hands/demo.py:83-84 loads with strict=True) and the F16X2 accuracy-contract flags."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_pack  # noqa: E402
import stress_checkpoints as sc  # noqa: E402
from ev2hands_amd import _lib, pack, synth  # noqa: E402


def _scalars(s, path=""):
    """every non-pointer field of a ctypes struct, recursively: path -> value"""
    out = {}
    for name, tp in s._fields_:
        v = getattr(s, name)
        p = f"{path}.{name}" if path else name
        if isinstance(v, C.Structure):
            out.update(_scalars(v, p))
        elif isinstance(v, C.Array):
            for i, e in enumerate(v):
                if isinstance(e, C.Structure):
                    out.update(_scalars(e, f"{p}[{i}]"))
                elif isinstance(e, C.Array):
                    for j, ee in enumerate(e):
                        out.update(_scalars(ee, f"{p}[{i}][{j}]"))
                elif tp._type_ is not C.c_void_p:
                    out[f"{p}[{i}]"] = e
                else:
                    out[f"{p}[{i}]"] = bool(e)
        elif tp is C.c_void_p:
            out[p] = bool(v)              # NULL-ness must agree; addresses cannot
        else:
            out[p] = v
    return out


def _compare(sd, C_in, precision, equalize):
    ref = ref_pack.PackedWeights(sd, "cpu", C_in, precision, equalize=equalize)
    got = pack.PackedWeights(sd, "cpu", C_in, precision, equalize=equalize)
    rt, gt = ref.tensors, got.tensors
    assert sorted(rt) == sorted(gt)
    for k in rt:
        a, b = rt[k].numpy(), gt[k].numpy()
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        assert a.tobytes() == b.tobytes(), f"{k}: {int((a.view(np.uint8).reshape(-1) != b.view(np.uint8).reshape(-1)).sum())} bytes differ"
    rs, gs = _scalars(ref.struct), _scalars(got.struct)
    gs.pop("flags")
    rs.pop("flags", None)
    assert rs == gs, {k: (rs[k], gs[k]) for k in rs if rs[k] != gs.get(k)}
    assert sorted(ref.equalization) == sorted(got.equalization)
    for k, e in ref.equalization.items():
        assert np.array_equal(e, got.equalization[k]), k
    assert got.struct.flags == (_lib.W_EQUALIZED if equalize else _lib.W_UNEQUALIZED_OK)


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16x2", "bf16x3", "f16"])
@pytest.mark.parametrize("C_in,seed", [(4, 0), (5, 3)])
def test_c_packer_equals_the_numpy_restatement(precision, C_in, seed):
    _compare(synth.synth_state_dict(C_in, seed), C_in, precision, True)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_c_packer_without_equalisation(precision):
    _compare(synth.synth_state_dict(5, 2), 5, precision, False)


@pytest.mark.parametrize("kind", ["channels8", "channels16_dead", "hidden1e4", "hidden1e-4", "heavy", "module_prefix"])
def test_c_packer_on_stress_checkpoints(kind):
    sd = synth.synth_state_dict(4, 7)
    if kind == "channels8":
        sd = sc.rescale_channels(sd, 8.0, 9)
    elif kind == "channels16_dead":
        sd = sc.rescale_channels(sd, 16.0, 11, dead_fraction=0.1, include_l0=True)
    elif kind == "hidden1e4":
        sd = sc.rescale_hidden(sd, 1e4)
    elif kind == "hidden1e-4":
        sd = sc.rescale_hidden(sd, 1e-4)
    elif kind == "heavy":
        sd = sc.heavy_tailed(sd, 1.5, 5)
    if kind == "module_prefix":          # nn.DataParallel checkpoints (model.py:16-21): the C packer strips the prefix itself
        got = pack.PackedWeights({"module." + k: v for k, v in sd.items()}, "cpu", 4, "f16x2")
        ref = pack.PackedWeights(sd, "cpu", 4, "f16x2")
        assert all(torch.equal(v, got.tensors[k]) for k, v in ref.tensors.items())
        return
    _compare(sd, 4, "f16x2", True)
    _compare(sd, 4, "bf16x3", True)
    if kind in ("channels8", "hidden1e-4"):
        _compare(sd, 4, "f16", True)


def test_f16_family_mask_packs_each_family_in_its_own_plane_mode():
    """ev2h_pack_weights(EV2H_PREC_F16, EV2H_PACK_F16_FAMILIES(mask)): the families inside the mask get ONE fp16 plane (the images
    the all-families pack holds), the others the two-plane images the F16X2 pack holds -- byte for byte -- and the fp32 arrays
    are the same in all three."""
    sd = synth.synth_state_dict(4, 5)
    full = pack.PackedWeights(sd, "cpu", 4, "f16")
    two = pack.PackedWeights(sd, "cpu", 4, "f16x2")
    assert full.struct.f16_families == _lib.FAM_ALL and two.struct.f16_families == 0
    fam_of = lambda name: (_lib.FAM_SA if name.startswith(("sa1.", "sa2.")) or ".sa1." in name else
                           _lib.FAM_ROWS if name.startswith(("fp1m.", "clsm.")) else
                           _lib.FAM_QCONV if name.startswith("qconv0.") else _lib.FAM_DENSE)      # noqa: E731
    for mask in (_lib.FAM_SA, _lib.FAM_ROWS | _lib.FAM_QCONV, _lib.FAM_DENSE):
        mixed = pack.PackedWeights(sd, "cpu", 4, "f16", f16_families=mask)
        assert mixed.struct.f16_families == mask and mixed.struct.precision == _lib.PREC["f16"]
        assert sorted(mixed.tensors) == sorted(full.tensors) == sorted(two.tensors)
        nimg = 0
        for name, t in mixed.tensors.items():
            if t.dtype == torch.uint8:                   # a plane image
                src = full if fam_of(name) & mask else two
                assert torch.equal(t, src.tensors[name]), (mask, name)
                nimg += 1
            else:
                assert torch.equal(t, full.tensors[name]) and torch.equal(t, two.tensors[name]), name
        assert nimg > 20


def test_strict_schema_errors():
    sd = synth.synth_state_dict(4, 0)
    missing = {k: v for k, v in sd.items() if k != "fp2.mlp_bns.1.running_var"}
    with pytest.raises(_lib.Ev2hError, match='missing key "fp2.mlp_bns.1.running_var"'):
        pack.PackedWeights(missing, "cpu", 4, "f32")
    extra = dict(sd, **{"sa9.weight": torch.zeros(3)})
    with pytest.raises(_lib.Ev2hError, match='unexpected key "sa9.weight"'):
        pack.PackedWeights(extra, "cpu", 4, "f32")
    wrong = dict(sd, **{"classifier.4.weight": torch.zeros(5, 256, 1)})
    with pytest.raises(_lib.Ev2hError, match='size mismatch for "classifier.4.weight"'):
        pack.PackedWeights(wrong, "cpu", 4, "f32")
    with pytest.raises(_lib.Ev2hError, match="size mismatch"):          # a C = 5 checkpoint packed for 4 input channels
        pack.PackedWeights(synth.synth_state_dict(5, 0), "cpu", 4, "f32")
    big = dict(sd)
    big["fp1.mlp_convs.1.weight"] = sd["fp1.mlp_convs.1.weight"] * 1e9
    with pytest.raises(_lib.Ev2hError, match="65504"):                   # one matrix spans more than fp16 can hold after its scale
        w = big["fp1.mlp_convs.1.weight"].clone()
        w[0, 0, 0] = float("inf")
        pack.PackedWeights(dict(big, **{"fp1.mlp_convs.1.weight": w}), "cpu", 4, "f16x2", equalize=False)
    pack.PackedWeights(big, "cpu", 4, "f16x2")                            # large but finite weights are rescaled, not refused
    # float64 checkpoints are accepted as they are
    pack.PackedWeights({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, "cpu", 4, "f32")


def test_device_pack_needs_a_device_and_says_so():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    descs, keep = pack.tensor_descs(synth.synth_state_dict(4, 0))
    h = C.c_void_p()
    rc = L.ev2h_pack_weights(descs, len(descs), 4, _lib.PREC["f32"], _lib.PACK_EQUALIZE, C.byref(h))
    assert rc != 0 and not h.value and b"upload" in L.ev2h_last_error()


def test_single_image_exports_equal_the_restatement():
    rng = np.random.default_rng(5)
    for (c1, c2, c3) in [(32, 32, 64), (64, 96, 128), (128, 196, 256), (256, 256, 32)]:
        for ns in (1, 2, 3):
            W2, W3 = rng.normal(size=(c2, c1)) * 1e-3, rng.normal(size=(c3, c2)) * 40.0
            a, b = ref_pack.sa_bf16_images(W2, W3, ns), pack.sa_bf16_images(W2, W3, ns)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2:] == b[2:]
    for (n, k, rows) in [(160, 8, 128), (256, 320, 128), (22, 1024, 128), (512, 768, 256)]:
        for ns in (1, 2, 3):
            W = rng.normal(size=(n, k)) * 10.0 ** rng.integers(-5, 3)
            a, b = ref_pack.gemm_bf16_w_image(W, ns, rows), pack.gemm_bf16_w_image(W, ns, rows)
            assert a[0].tobytes() == b[0].tobytes() and a[1] == b[1]
            assert ref_pack.plane_unscale(W, ns) == pack.plane_unscale(W, ns)


def test_weight_spread_report():
    """The weight side of the f16x2 accuracy contract (ev2h_packed_weight_spread): equalised checkpoints keep all but a few weights
    within 2^17 of their matrix maximum; the same rescaled checkpoint without equalisation does not."""
    sd = sc.rescale_channels(synth.synth_state_dict(4, 3), 16.0, 1)
    tot = lambda pw: tuple(sum(v[i] for v in pw.weight_spread().values()) for i in range(3))      # noqa: E731
    nz, lo, hi = tot(pack.PackedWeights(sd, "cpu", 4, "f16x2", equalize=True))
    assert nz > 4_000_000 and lo < 1e-4 * nz and hi == 0
    nz, lo, hi = tot(pack.PackedWeights(sd, "cpu", 4, "f16x2", equalize=False))
    assert lo > 0.5 * nz and hi > 0.3 * nz
    assert pack.PackedWeights(sd, "cpu", 4, "bf16x3").weight_spread() == {}                      # f16x2 only


def test_integration_md_binding_stub():
    """The ctypes stub INTEGRATION.md section 2 shows for ev2h_pack_weights, executed as written (plus EV2H_PACK_HOST_ONLY: no GPU
    here) on a DataParallel-style checkpoint."""
    L = C.CDLL(_lib.LIB_PATH)
    L.ev2h_last_error.restype = C.c_char_p

    class TensorDesc(C.Structure):
        _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]
    sd = {"module." + k: v for k, v in synth.synth_state_dict(5, 1).items()}
    keep = [(k.encode(), v.numpy()) for k, v in sd.items() if v.dtype == torch.float32]
    descs = (TensorDesc * len(keep))()
    for d, (k, a) in zip(descs, keep):
        d.name, d.data, d.dtype, d.ndim = k, a.ctypes.data, 0, a.ndim
        d.shape[:a.ndim] = a.shape
    handle = C.c_void_p()
    EQUALIZE, HOST_ONLY, F16X2 = 1, 2, 2
    assert L.ev2h_pack_weights(descs, len(descs), 5, F16X2, EQUALIZE | HOST_ONLY, C.byref(handle)) == 0, L.ev2h_last_error()
    L.ev2h_packed_weights.restype = C.c_void_p
    w = _lib.Weights.from_address(L.ev2h_packed_weights(handle))
    assert w.precision == 2 and w.flags == _lib.W_EQUALIZED and w.sa1.nbranch == 3 and w.qconv0.O == 512
    L.ev2h_packed_free.argtypes = [C.c_void_p]
    L.ev2h_packed_free(handle)
