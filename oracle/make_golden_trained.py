"""Checkpoints that came out of an OPTIMISER, with the reference as the trainer (survey container only).

Every other checkpoint in tests/ is hash-random (`synth_state_dict`) or a constructed transform of one.  Here the reference's own
network (/root/reference/src/Ev2Hands/model/TEHNet.py:115-197, imported as in make_golden.py) is put in TRAIN mode
(BatchNorm batch statistics + running-stat updates `pointnet2_utils.py:253-256,312-314`, Dropout active) and stepped with the
reference's optimiser settings (Adam, lr 1e-3, weight decay 0: train.py:22-23,53; loop shape train.py:78-92) on synthetic event
clouds with a learnable target, so that weight scales, BN running statistics, dead / saturated units and the logit margins EMERGE
instead of being constructed.  Then `eval()`, and the reference's forward on the result is recorded exactly as make_golden.py does.

Training signal (the datasets and the MANO assets are absent; the point is the optimiser's footprint, not the task):
  * per-point class = which of the window's two event blobs a point belongs to (1 = the blob with the smaller x "left",
    2 = "right"), 3 ("noise") for far points late in the window, 0 (background) for far points early in it
    -> cross entropy on `class_logits` (losses.py uses CE on the same tensor);
  * per hand the 22 regressed parameters -> smooth functions of that blob's centre; MSE on them and on the 21 joints the
    (differentiable) MANO restatement makes of them (the reference's loss also mixes parameter and joint terms).

Schedule: 480 steps of 4 windows x 2048 points, C = 4 (steps 0-319: cross entropy x 1, joint term x 100 -- the joint error of a
random start is huge; steps 320-479: cross entropy x 3, joint term x 10, so that the segmentation head gets its share of the
gradient: CE 2.4 -> 1.3, accuracy 0.30 -> 0.58), checkpointed every 20 steps in /tmp so that a run can be resumed; then 48 steps
for the C = 5 variant.  About 35 minutes on 8 CPU threads.

Outputs:
  tests/golden/trained_weights_c4.npz   fp16 deltas on synth_state_dict(4, 100) + BN statistics (tests/trained_ckpt.py)
  tests/golden/trained_weights_c5.npz   the C = 5 checkpoint's differing entries (enc.sa1 first convolutions re-trained)
  tests/golden/trained_*.npz            reference-run fixtures on those checkpoints
  profiles/r5_trained_checkpoint_report.txt   what the optimiser did (spreads, dead units, BN statistics, loss curve)

[r6] EV2H_TRAIN_RUN=b: a SECOND, independent run (tests/trained_ckpt.py: RUNS) -- initial weights synth_state_dict(4, 101), torch seed
2026, other clouds (data seeds 20 000 + step), 1 500 steps (900 with cross entropy x 2 / joint term x 30, then 600 with x 4 / x 5),
C = 4 only; outputs tests/golden/trained2_weights_c4.npz, trained2_{E,U}_c4_n2048.npz, profiles/r6_trained2_checkpoint_report.txt.
About two hours on 6 CPU threads.

Run:  python oracle/make_golden_trained.py [train|fixtures|all]      (needs /root/reference; never runs on the GPU box)
"""
from __future__ import annotations

import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ev2hands_amd import synth  # noqa: E402
from oracle import make_golden, mano_oracle  # noqa: E402
import trained_ckpt  # noqa: E402

RUN = os.getenv("EV2H_TRAIN_RUN", "a")
STEPS = int(os.getenv("EV2H_TRAIN_STEPS", "480" if RUN == "a" else "1500"))
PHASE2 = 320 if RUN == "a" else 900      # from this step on the joint term is weighted less and the cross entropy more
LOSS_W = {"a": ((1.0, 100.0), (3.0, 10.0)), "b": ((2.0, 30.0), (4.0, 5.0))}[RUN]          # (w_ce, w_j3d) of the two phases
TORCH_SEED, DATA_SEED0 = {"a": (2024, 10_000), "b": (2026, 20_000)}[RUN]
STEPS_C5 = int(os.getenv("EV2H_TRAIN_STEPS_C5", "48"))
BATCH, POINTS = 4, 2048
STATE = "/tmp/ev2h_trained_state.pt" if RUN == "a" else "/tmp/ev2h_trained_state_b.pt"
MANO_SEED = 0
CASES = [
    # name,                 kind, C, N,   B, seed
    ("trained_E_c4_n2048", "E", 4, 2048, 2, 41),
    ("trained_E_c5_n2048", "E", 5, 2048, 2, 42),
    ("trained_U_c4_n2048", "U", 4, 2048, 2, 43),      # inputs the network never saw: other activation ranges
    ("trained_E_c4_n8192", "E", 4, 8192, 1, 44),
] if RUN == "a" else [
    ("trained2_E_c4_n2048", "E", 4, 2048, 2, 45),
    ("trained2_U_c4_n2048", "U", 4, 2048, 2, 46),
]


def blob_centres(b: int, seed: int) -> np.ndarray:
    """The two blob centres synth_cloud_events draws for window b, in the cloud's normalised [-1, 1] coordinates,
    sorted by x (row 0 = 'left')."""
    W, H = synth.OUTPUT_WIDTH, synth.OUTPUT_HEIGHT
    c = synth.hash_uniform(f"cloudE/{b}/ctr", (2, 2), seed) * np.array([W * 0.6, H * 0.6]) + np.array([W * 0.2, H * 0.2])
    c = 2 * c / np.array([W, H]) - 1
    return c[np.argsort(c[:, 0])]


def make_batch(C: int, step: int):
    seed = DATA_SEED0 + step
    xyz = synth.synth_cloud("E", BATCH, C, POINTS, seed)
    labels = torch.zeros(BATCH, POINTS, dtype=torch.long)
    targets = {"left": torch.zeros(BATCH, 22), "right": torch.zeros(BATCH, 22)}
    sig = np.array([2 * 30.0 / synth.OUTPUT_WIDTH, 2 * 30.0 / synth.OUTPUT_HEIGHT])          # the blobs' sigma, normalised
    for b in range(BATCH):
        ctr = blob_centres(b, seed)
        p = xyz[b, :2].numpy().T                                                                # [N, 2]
        d = np.stack([np.sqrt((((p - ctr[h]) / sig) ** 2).sum(1)) for h in range(2)], 1)        # in sigmas
        near = d.argmin(1)
        far = d.min(1) > 2.5
        lab = np.where(far, np.where(xyz[b, 2].numpy() > 0, 3, 0), near + 1)
        labels[b] = torch.from_numpy(lab)
        for h, side in enumerate(("left", "right")):
            cx, cy = ctr[h]
            k6, k10 = np.arange(1, 7), np.arange(1, 11)
            targets[side][b] = torch.from_numpy(np.concatenate([
                [0.8 * cx, 0.8 * cy, 0.5 * cx * cy],                         # global_orient
                0.6 * np.sin(k6 * cx + cy),                                 # hand_pose (6 PCA coefficients)
                0.5 * np.cos(k10 * cy - 0.3 * cx),                          # betas
                [0.2 * cx, 0.2 * cy, 0.5 + 0.1 * cx],                       # transl
            ]).astype(np.float32))
    return xyz, labels, targets


def loss_fn(out, labels, targets, hands, w_ce=1.0, w_j3d=100.0):
    ce = F.cross_entropy(out["class_logits"], labels)
    loss = w_ce * ce
    parts = {"ce": float(ce.detach())}
    for side in ("left", "right"):
        prm = torch.cat([out[side][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1)
        t = targets[side]
        lp = F.mse_loss(prm, t)
        with torch.no_grad():
            tj = hands[side](t[:, 0:3], t[:, 3:9], t[:, 9:19], t[:, 19:22]).joints
        lj = F.mse_loss(out[side]["j3d"], tj) * w_j3d
        loss = loss + lp + lj
        parts[side + ".prm"] = float(lp.detach())
        parts[side + ".j3d"] = float(lj.detach())
    return loss, parts


def train_c4(te, hands, log):
    os.environ["ERPC"], os.environ["MHLNES"] = "0", "0"
    torch.manual_seed(TORCH_SEED)
    net = te.TEHNet(n_pose_params=synth.MANO_CMPS)
    init = synth.synth_state_dict(4, trained_ckpt.RUNS[RUN]["init_seed"])
    net.load_state_dict(init, strict=True)
    opt = torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=0.0)          # train.py:22-23,53
    start = 0
    if os.path.exists(STATE):
        st = torch.load(STATE)
        net.load_state_dict(st["net"])
        opt.load_state_dict(st["opt"])
        torch.set_rng_state(st["rng"])
        start = st["step"]
        log(f"resumed at step {start}")
    net.train()
    t0 = time.time()
    for step in range(start, STEPS):
        xyz, labels, targets = make_batch(4, step)
        out = net(xyz, hands)                                                     # train.py:83
        # two phases: the joint term (huge at a random start) first, then the segmentation head gets its share of the gradient
        loss, parts = loss_fn(out, labels, targets, hands, *(LOSS_W[0] if step < PHASE2 else LOSS_W[1]))
        opt.zero_grad()                                                           # train.py:90-92
        loss.backward()
        opt.step()
        if step % 10 == 0 or step == STEPS - 1:
            acc = float((out["class_logits"].argmax(1) == labels).float().mean())
            log(f"step {step:4d} loss {float(loss.detach()):8.4f} " + " ".join(f"{k} {v:.4f}" for k, v in parts.items())
                + f" acc {acc:.3f} ({time.time() - t0:.0f} s)")
        if (step + 1) % 20 == 0:
            torch.save({"net": net.state_dict(), "opt": opt.state_dict(), "rng": torch.get_rng_state(), "step": step + 1}, STATE)
    net.eval()
    return OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items()), init


C5_KEYS = [f"sa1.conv_blocks.{i}.0." for i in range(3)] + [f"sa1.bn_blocks.{i}.0." for i in range(3)]


def train_c5(te, hands, sd4, log):
    """C = 5 (ERPC) from the trained C = 4 network: enc.sa1's first convolutions get a column for the second event-count channel
    ([x, y, t, pol | dx, dy, dz] -> [x, y, t, pos, neg | dx, dy, dz], pointnet2_utils.py:248) and are re-trained with their
    BatchNorms while everything downstream stays as trained (frozen, eval-mode BN)."""
    os.environ["ERPC"] = "1"
    net = te.TEHNet(n_pose_params=synth.MANO_CMPS)
    init5 = synth.synth_state_dict(5, trained_ckpt.INIT_SEED)
    sd5 = OrderedDict()
    for k, v in init5.items():
        if k.startswith("sa1.conv_blocks.") and k.endswith(".0.weight"):
            w4 = sd4[k]
            sd5[k] = torch.cat([w4[:, :4], v[:, 4:5], w4[:, 4:]], 1).contiguous()
        else:
            sd5[k] = sd4[k].clone()
    net.load_state_dict(sd5, strict=True)
    net.eval()
    live = []
    for name, p in net.named_parameters():
        p.requires_grad_(any(name.startswith(k) for k in C5_KEYS))
        if p.requires_grad:
            live.append(p)
    for i in range(3):
        net.sa1.bn_blocks[i][0].train()
    opt = torch.optim.Adam(live, lr=0.001, weight_decay=0.0)
    torch.manual_seed(2025)
    for step in range(STEPS_C5):
        xyz, labels, targets = make_batch(5, 50_000 + step)
        out = net(xyz, hands)
        loss, parts = loss_fn(out, labels, targets, hands)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 8 == 0 or step == STEPS_C5 - 1:
            log(f"c5 step {step:3d} loss {float(loss.detach()):8.4f} ce {parts['ce']:.4f}")
    net.eval()
    out = OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items())
    return out, {k: v.numpy() for k, v in out.items() if any(k.startswith(p) for p in C5_KEYS)}


def report(sd, init, log):
    """What the optimiser did, in the terms the f16x2 contract cares about."""
    log("\n== checkpoint report (trained vs its hash-random start) ==")
    tot = 0
    for k, v in sd.items():
        if v.dtype != torch.float32 or v.dim() < 2:
            continue
        w, w0 = v.flatten(1), init[k].flatten(1)
        rn, rn0 = w.norm(dim=1), w0.norm(dim=1)
        tot += w.numel()
        log(f"{k:58s} |dW|/|W0| {float((w - w0).norm() / w0.norm()):6.3f}  row-norm spread 2^{float(torch.log2(rn.max() / rn.min().clamp_min(1e-30))):5.2f}"
            f" (init 2^{float(torch.log2(rn0.max() / rn0.min())):4.2f})  max|w| {float(w.abs().max()):.3f}")
    for k, v in sd.items():
        if k.endswith("running_var"):
            base = k[:-len("running_var")]
            g, var, mu = sd[base + "weight"], v, sd[base + "running_mean"]
            s = g / torch.sqrt(var + 1e-5)
            log(f"{base:58s} var [{float(var.min()):.2e}, {float(var.max()):.2e}]  |mean| max {float(mu.abs().max()):.3f}  "
                f"gamma [{float(g.min()):.3f}, {float(g.max()):.3f}]  fold scale spread 2^{float(torch.log2(s.abs().max() / s.abs().min().clamp_min(1e-30))):.2f}"
                f"  tracked {int(sd[base + 'num_batches_tracked'])}")
    log(f"{tot} weights in conv / linear tensors")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.set_num_threads(int(os.getenv("EV2H_TRAIN_THREADS", "8")))
    pn, te = make_golden.load_reference()
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", MANO_SEED), synth.synth_mano_assets("right", MANO_SEED))
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    if what in ("train", "all"):
        sd4, init = train_c4(te, hands, log)
        arrays = trained_ckpt.encode(sd4, init)
        np.savez_compressed(trained_ckpt.weights_path(4, RUN), **arrays)
        sd4q = trained_ckpt.decode(arrays, init)                  # the checkpoint every consumer reconstructs
        drift = max(float((sd4q[k].double() - sd4[k].double()).abs().max()) for k in sd4 if sd4[k].dtype == torch.float32)
        log(f"wrote {trained_ckpt.weights_path(4, RUN)} ({os.path.getsize(trained_ckpt.weights_path(4, RUN)) / 2**20:.2f} MiB); "
            f"fp16-delta rounding moved a weight by at most {drift:.2e}")
        if RUN == "a":
            _sd5, changed = train_c5(te, hands, sd4q, log)
            np.savez_compressed(trained_ckpt.weights_path(5), **changed)
            log(f"wrote {trained_ckpt.weights_path(5)} ({os.path.getsize(trained_ckpt.weights_path(5)) / 1024:.1f} KiB)")
        report(sd4q, init, log)
        with open(os.path.join(ROOT, "profiles", "r5_trained_checkpoint_report.txt" if RUN == "a" else "r6_trained2_checkpoint_report.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    if what in ("fixtures", "all"):
        only = sys.argv[2:]
        for name, kind, C, N, B, seed in CASES:
            if only and name not in only:
                continue
            sd = trained_ckpt.trained_state_dict(C, RUN)
            make_golden.run_case(pn, te, name, kind, C, N, B, seed, sd_override=sd,
                                 extra={"ckpt": np.array("trained"), "mano_seed": np.array(seed)})
    os.environ["ERPC"], os.environ["MHLNES"] = "0", "0"


if __name__ == "__main__":
    main()
