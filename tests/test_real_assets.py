"""Validation on real assets (tools/validate_real_assets.py).

* CPU: a dry run of the fixture-making step with the oracle and synthetic assets written in the REAL file formats
  ({'state_dict': {'module.…': …}} checkpoint, chumpy-free MANO pkl reader).
* GPU: `check` of that dry-run fixture through libev2hands_hip.so -- and, when EV2H_REAL_FIXTURE / EV2H_MANO_PATH / EV2H_CKPT
  point at a fixture made by the reference itself and at the licensed files, of the real thing (skipped otherwise:
  the files cannot be redistributed).  INTEGRATION.md section 5 has the two commands."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "validate_real_assets.py")


def _make_dry_run(tmp_path):
    out = tmp_path / "dry.npz"
    assets = tmp_path / "assets"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "real_assets_dryrun.py"), str(assets), str(out)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return str(out), str(assets)


def test_make_fixture_dry_run(tmp_path):
    fixture, assets = _make_dry_run(tmp_path)
    g = np.load(fixture)
    assert int(g["channels"]) == 5 and int(g["ncases"]) == 3
    assert g["0.xyz"].shape == (2, 5, 2048) and g["0.fps_init"].shape == (4, 2) and g["0.left.vertices"].shape == (2, 778, 3)
    assert len(str(g["sha256.ckpt"])) == 64
    assert os.path.exists(os.path.join(assets, "mano", "MANO_LEFT.pkl")) and os.path.exists(os.path.join(assets, "best_model_state_dict.pth"))
    # no asset content in the fixture: it is far smaller than the checkpoint
    assert os.path.getsize(fixture) < os.path.getsize(os.path.join(assets, "best_model_state_dict.pth")) / 4


@pytest.mark.gpu
def test_check_dry_run_fixture_on_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import validate_real_assets as V
    fixture, assets = _make_dry_run(tmp_path)
    rep = V.check_fixture(fixture, assets, os.path.join(assets, "best_model_state_dict.pth"), verbose=False)
    for prec in ("f32", "f16x2", "bf16x3"):
        assert rep[prec]["max_rel"] < 1e-4 and rep[prec]["argmax_agreement"] == 1.0 and rep[prec]["mpjpe_mm"] < 1e-3, (prec, rep[prec])
    assert rep["bf16"]["mpjpe_mm"] < 5.0
    # a different checkpoint file is refused (digest mismatch), not silently compared
    other = tmp_path / "other.pth"
    torch.save({"state_dict": {}}, other)
    with pytest.raises(RuntimeError, match="SHA-256"):
        V.check_fixture(fixture, assets, str(other), verbose=False)


@pytest.mark.gpu
def test_real_assets_if_present():
    fixture, mano, ckpt = (os.environ.get(k) for k in ("EV2H_REAL_FIXTURE", "EV2H_MANO_PATH", "EV2H_CKPT"))
    if not (fixture and mano and ckpt and all(os.path.exists(p) for p in (fixture, mano, ckpt))):
        pytest.skip("licensed MANO files / pretrained checkpoint / reference-made fixture not provided (INTEGRATION.md section 5)")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import validate_real_assets as V
    rep = V.check_fixture(fixture, mano, ckpt)
    for prec in ("f32", "f16x2", "bf16x3"):
        assert rep[prec]["max_rel"] < 1e-4, (prec, rep[prec])
        assert rep[prec]["argmax_agreement"] > 0.9999, (prec, rep[prec])
