"""Arbitrate one case of FUZZ_TRAINED=1 tests/fuzz_modes.py with the float64 yardstick: which mode is far from the exact value?
usage: python tests/replay_trained_case.py <case> [fuzz seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stress_checkpoints as sc  # noqa: E402
import trained_ckpt  # noqa: E402
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402
from oracle import mano_oracle, tehnet_oracle  # noqa: E402

want = int(sys.argv[1])
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for case in range(want + 1):               # replay the fuzzer's draws
    C = int(rng.choice([4, 5])); kind = str(rng.choice(["E", "E", "U", "L"])); B = int(rng.integers(1, 6))
    N = int(rng.choice([128, 129, 200, 333, 512, 777, 1024, 1500, 2048, 2049, 3000, int(rng.integers(130, 4000)), int(rng.integers(130, 4000)), 5000, 8192, 8200, 12345]))
    if N > 4096:
        B = min(B, 2)
    if kind == "L":
        N = min(N, 4096)
    mh = int(rng.random() < 0.2); seed = int(rng.integers(0, 10 ** 6)); variant = str(rng.choice(["plain", "channels", "dead", "heavy", "hidden"]))
    p1, p2 = float(rng.random()), float(rng.random())
sd = trained_ckpt.trained_state_dict(C)
if variant == "channels":
    sd = sc.rescale_channels(sd, [3, 8, 14][int(p1 * 3)], seed, include_l0=p2 < 0.5)
elif variant == "dead":
    sd = sc.rescale_channels(sd, 4, seed, dead_fraction=0.15)
elif variant == "hidden":
    sd = sc.rescale_hidden(sd, [1e-4, 1e3, 1e6][int(p1 * 3)])
print(f"case {want}: C={C} {kind} B={B} N={N} mhlnes={mh} variant={variant} p1={p1:.3f} seed={seed}")
os.environ["ERPC"] = "1" if C == 5 else "0"
os.environ["MHLNES"] = str(mh)
assets = {s: synth.synth_mano_assets(s, seed % 7) for s in ("left", "right")}
xyz = synth.synth_cloud(kind, B, C, N, seed)
inits = synth.fps_inits(B, N, seed)
tr = {}
with torch.no_grad():
    x32 = xyz.clone()
    r32 = tehnet_oracle.tehnet_forward(sd, x32, mano_oracle.make_hands(assets["left"], assets["right"]), fps_init=inits, trace=tr, mhlnes=bool(mh))
    r64 = tehnet_oracle.tehnet_forward_f64(sd, x32, mano_oracle.make_hands(assets["left"], assets["right"], dtype=torch.float64), tr)
KEYS = ("global_orient", "hand_pose", "betas", "transl")
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
def row(o):
    return {"logits": o["class_logits"], **{f"{s}.{k}": o[s][k] for s in ("left", "right") for k in KEYS}}
truth = row(r64)
print("  CPU fp32 oracle vs f64:", {k: f"{rel(v, truth[k]):.1e}" for k, v in row(r32).items()})
for prec in ("f32", "bf16x3", "f16x2"):
    net = TEHNetWrapper("cuda:0", mano_assets=assets, precision=prec)
    net.load_state_dict(sd, strict=True); net.eval()
    net.net.fps_init = inits
    with torch.no_grad():
        o = net(xyz.clone().cuda())
    print(f"  {prec:7s} vs f64:", {k: f"{rel(v, truth[k]):.1e}" for k, v in row(o).items()})
