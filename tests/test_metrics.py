"""Joint metrics.  CPU: oracle vs the fixture produced by the reference's own functions
(oracle/make_golden_metrics.py).  GPU: ev2h_joint_metrics vs the fixture: PCK curves, candidate choice and rounded AUCs exact,
float64 MPJPE / root distance to 1e-9 mm (the reference's reduction order over 42 doubles is not specified)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

FIX = os.path.join(os.path.dirname(__file__), "golden", "metrics_0.npz")


@pytest.mark.parametrize("steps", [100, 20])
def test_oracle_matches_reference_fixture(steps):
    g = np.load(FIX)
    t = f"s{steps}"
    pred, gts = torch.from_numpy(g[t + ".pred"]), torch.from_numpy(g[t + ".gts"])
    for b in range(pred.shape[0]):
        m = MO.evaluate_joints(pred[b] * 1000, gts[b] * 1000, steps)
        assert m["best"] == int(g[t + ".best"][b])
        assert np.array_equal(m["absolute_pck3d"], g[t + ".abs"][b])
        assert np.array_equal(m["relative_pck3d"], g[t + ".rel"][b])
        assert np.array_equal(m["right_root_relative_pck3d"], g[t + ".rrr"][b])
        assert m["joint_loss"] == g[t + ".mpjpe"][b] and m["root_distance"][0] == g[t + ".rootd"][b]
        assert [MO.auc(m[k]) for k in ("absolute_pck3d", "relative_pck3d", "right_root_relative_pck3d")] == list(g[t + ".auc"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [100, 20])
def test_gpu_metrics_match_reference_fixture(steps):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.metrics import evaluate_joints_real_batch
    g = np.load(FIX)
    t = f"s{steps}"
    pred, gts = torch.from_numpy(g[t + ".pred"]).cuda(), torch.from_numpy(g[t + ".gts"])
    res = evaluate_joints_real_batch(pred[:, 0].contiguous(), pred[:, 1].contiguous(), gts, steps)
    for b, r in enumerate(res):
        assert r["gt_index"] == int(g[t + ".best"][b])
        assert np.array_equal(r["absolute_pck3d"], g[t + ".abs"][b])
        assert np.array_equal(r["relative_pck3d"], g[t + ".rel"][b])
        assert np.array_equal(r["right_root_relative_pck3d"], g[t + ".rrr"][b])
        assert [r["absolute_auc"], r["relative_auc"], r["right_root_relative_auc"]] == list(g[t + ".auc"][b])
        assert abs(r["joint_loss"] - g[t + ".mpjpe"][b]) < 1e-9
        assert abs(r["root_distance"][0] - g[t + ".rootd"][b]) < 1e-9
