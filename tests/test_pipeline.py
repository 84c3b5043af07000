"""The whole evaluation flow of the reference (evaluate_ev2hands_r.py:91-160) composed from the GPU pieces: raw event windows ->
[B, 5, N] tensors (8f-1) -> TEHNet forward (C = 5, ERPC = 1) -> per-frame joint metrics (8f-3) and the two-hand non-collision
score (8f-4).  Each piece has its own parity tests; this one checks that they fit together (devices, dtypes, shapes, ranges) and
that the composition equals the composition of the oracles."""
import os

import numpy as np
import pytest
import torch

from ev2hands_amd import synth

pytestmark = pytest.mark.gpu


def test_events_to_scores_end_to_end():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ev2hands_amd.collision import compute_non_collision_score
    from ev2hands_amd.events import EventWindowBuilder
    from ev2hands_amd.metrics import evaluate_joints_real_batch
    from ev2hands_amd.model import TEHNetWrapper
    from oracle import collision_oracle, event_window_oracle as EW, mano_oracle, metrics_oracle, tehnet_oracle

    B, N, seed = 3, 2048, 17
    windows = []
    for k in range(B):
        s = EW.synth_event_stream(2500 + 300 * k, 90 + k).astype(np.float64)
        s[:, 2] *= 1e-3                                  # us -> ms, as the reference's stream reader does
        windows.append(s)
    rs = np.random.RandomState(seed)
    table_counts = [int(np.unique(w[:, 0].astype(np.int64) + 346 * w[:, 1].astype(np.int64)).size) for w in windows]
    idx = np.stack([rs.randint(0, m, N) for m in table_counts])

    # ---- GPU flow
    data = EventWindowBuilder("cuda:0")(windows, idx)
    assert data.shape == (B, 5, N) and data.dtype == torch.float32 and data.is_cuda
    os.environ["ERPC"] = "1"
    assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
    sd = synth.synth_state_dict(5, seed)
    net = TEHNetWrapper("cuda:0", mano_assets=assets, precision="f16x2")
    net.load_state_dict(sd, strict=True)
    net.eval()
    inits = synth.fps_inits(B, N, seed)
    net.net.fps_init = inits
    with torch.no_grad():
        out = net(data)
    gts = torch.from_numpy(synth.hash_normal("gt", (B, 2, 2, 21, 3), seed) * 0.05)
    scores = evaluate_joints_real_batch(out["left"]["j3d"], out["right"]["j3d"], gts, 20)
    ncs, _ = compute_non_collision_score(out["left"]["vertices"], net.hands["left"].faces, out["right"]["vertices"], net.hands["right"].faces)
    # (the synthetic MANO assets have random faces: a triangle soup with far more intersecting pairs than triangles, so the score
    # is hugely negative -- the formula, not the range, is what is checked below)
    assert len(scores) == B and len(ncs) == B and all(np.isfinite(s) and s <= 100.0 for s in ncs)

    # ---- the same flow through the oracles
    ref_data = torch.stack([EW.build_window(w, i)[0] for w, i in zip(windows, idx)])
    assert torch.equal(data.cpu(), ref_data)
    hands = mano_oracle.make_hands(assets["left"], assets["right"])
    with torch.no_grad():
        ref = tehnet_oracle.tehnet_forward(sd, ref_data.clone(), hands, fps_init=inits)
    rel = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(out["left"]["j3d"], ref["left"]["j3d"]) < 1e-4 and rel(out["right"]["vertices"], ref["right"]["vertices"]) < 1e-4
    assert torch.equal(out["class_logits"].argmax(1).cpu(), ref["class_logits"].argmax(1))
    for b in range(B):
        pred_mm = torch.stack([ref["left"]["j3d"][b], ref["right"]["j3d"][b]]).double() * 1000
        want = metrics_oracle.evaluate_joints(pred_mm, gts[b].double() * 1000, 20)
        assert abs(scores[b]["joint_loss"] - want["joint_loss"]) < 1e-3 * max(1.0, abs(want["joint_loss"]))
        # the collision count is discrete: compare on the GPU's own vertices
        sc, _n = collision_oracle.non_collision_score(out["left"]["vertices"][b].cpu().numpy(), out["right"]["vertices"][b].cpu().numpy(),
                                                      np.asarray(net.hands["left"].faces), np.asarray(net.hands["right"].faces), 8)
        assert ncs[b] == sc                                   # (max_collisions = 8 per triangle, evaluate_ev2hands_r.py:131)
