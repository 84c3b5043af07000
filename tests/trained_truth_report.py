"""How far is each arithmetic from the EXACT value of the network function, on the checkpoints that came out of training?

oracle/tehnet_oracle.tehnet_forward_f64 evaluates the reference's function in float64 on the float32 run's own discrete selections.
Against that yardstick: the reference-style float32 CPU evaluation (the parity oracle), and the library's four arithmetic modes,
per window (relative error of each window's tensor, max|d| / max|truth|), for the trained and the hash-random checkpoint.
    python tests/trained_truth_report.py [B]      ->  profiles/r5_trained_truth_report.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402
from oracle import mano_oracle, tehnet_oracle  # noqa: E402   (measurement tool: the oracle is the checker here)
import trained_ckpt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N, C = 2048, 4
os.environ["ERPC"] = "0"
torch.set_num_threads(min(32, os.cpu_count() or 8))


def per_window(a, b):
    a, b = a.double().cpu().flatten(1), b.double().cpu().flatten(1)
    return (a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-300)


def keys(o, tr=None):
    prm = lambda s: torch.cat([o[s][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1)      # noqa: E731
    return {"logits": o["class_logits"], "params": torch.cat([prm("left"), prm("right")], 1),
            "vertices": torch.cat([o["left"]["vertices"], o["right"]["vertices"]], 1), "joints": torch.cat([o["left"]["j3d"], o["right"]["j3d"]], 1)}


for tag, sd, kind, seed in (("trained", trained_ckpt.trained_state_dict(C), "E", 91), ("trained", trained_ckpt.trained_state_dict(C), "U", 92),
                            ("hash-random", synth.synth_state_dict(C, 93), "E", 93)):
    assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
    xyz, inits = synth.synth_cloud(kind, B, C, N, seed), synth.fps_inits(B, N, seed)
    tr = {}
    with torch.no_grad():
        r32 = tehnet_oracle.tehnet_forward(sd, xyz.clone(), mano_oracle.make_hands(assets["left"], assets["right"]), fps_init=inits, trace=tr)
        r64 = tehnet_oracle.tehnet_forward_f64(sd, xyz.clone(), mano_oracle.make_hands(assets["left"], assets["right"], dtype=torch.float64), tr)
    truth = keys(r64)
    rows = {"reference-style fp32 on the CPU (the parity oracle)": keys(r32)}
    for prec in ("f32", "bf16x3", "f16x2", "bf16"):
        net = TEHNetWrapper("cuda:0", mano_assets=assets, precision=prec)
        net.load_state_dict(sd, strict=True)
        net.eval()
        net.net.fps_init = inits
        with torch.no_grad():
            o = net(xyz.cuda())
        torch.cuda.synchronize()
        rows[f"library, {prec}"] = keys(o)
    print(f"## {tag} checkpoint, {kind}-clouds, {B} windows of {N} points: relative error per window against the float64 value of the function -- median / max over the windows")
    for name, got in rows.items():
        cells = []
        for k in truth:
            e = per_window(got[k], truth[k])
            cells.append(f"{k} {float(e.median()):.1e} / {float(e.max()):.1e}")
        agree = float((got["logits"].argmax(1).cpu() == truth["logits"].argmax(1)).float().mean())
        print(f"  {name:52s} " + "   ".join(cells) + f"   argmax = truth at {agree * 100:.3f} %")
