"""Which kernel families carry the error of the one-plane fp16 mode on the TRAINED checkpoints, and what each costs:
"f16" with every EV2H_FAM_* mask (families outside the mask run f16x2) against the exact-fp32 mode, B = 32 for accuracy and
B = 256 back-to-back forwards for the rate.    python tools/debug/f16_family_sweep.py [masks...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402
import trained_ckpt  # noqa: E402

MASKS = [int(v) for v in sys.argv[1:]] or [15, 1, 2, 4, 8, 14, 13, 11, 7, 3, 5, 9]
NAMES = {1: "sa", 2: "rows", 4: "qconv", 8: "dense"}
B, N = 32, 2048


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rootrel(j):
    j = j.double().view(j.shape[0], 2, 21, 3)
    return (j - j[:, :, :1]).view(j.shape[0], 42, 3)


def run(net, xyz, inits):
    net.net.fps_init = inits
    with torch.no_grad():
        o = net(xyz)
    return {"logits": o["class_logits"].clone(),
            "params": torch.cat([o[s][k] for s in ("left", "right") for k in ("global_orient", "hand_pose", "betas", "transl")], 1).clone(),
            "j3d": torch.cat([o["left"]["j3d"], o["right"]["j3d"]], 1).clone()}


cases = []
for C, kind, seed in ((4, "E", 51), (4, "U", 52), (5, "E", 53)):
    os.environ["ERPC"] = "1" if C == 5 else "0"
    assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
    sd = trained_ckpt.trained_state_dict(C)
    xyz = synth.synth_cloud(kind, B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    net = TEHNetWrapper("cuda:0", mano_assets=assets, precision="f32")
    net.load_state_dict(sd, strict=True)
    net.eval()
    cases.append((C, kind, net, xyz, inits, run(net, xyz, inits)))

# rate: B = 256, hash-random weights, C = 4
os.environ["ERPC"] = "0"
rnet = TEHNetWrapper("cuda:0", mano_assets={s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}, precision="f16")
rnet.load_state_dict(synth.synth_state_dict(4, 0), strict=True)
rnet.eval()
rx = synth.synth_cloud("E", 256, 4, N, 0).cuda()
rinits = synth.fps_inits(256, N, 0)


def rate(prec, mask):
    rnet.net.precision, rnet.net.f16_families = prec, mask
    for _ in range(3):
        rnet.net.fps_init = rinits
        with torch.no_grad():
            rnet(rx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 20
    for _ in range(k):
        rnet.net.fps_init = rinits
        with torch.no_grad():
            rnet(rx)
    torch.cuda.synchronize()
    return 256 * k / (time.perf_counter() - t0)


rows = [("f16x2", 0), ("bf16", 0)] + [("f16", m) for m in MASKS]
for prec, mask in rows:
    tag = prec if prec != "f16" else "f16[" + "+".join(NAMES[b] for b in (1, 2, 4, 8) if mask & b) + "]"
    parts = []
    for C, kind, net, xyz, inits, r in cases:
        net.net.precision, net.net.f16_families = prec, mask
        o = run(net, xyz, inits)
        agree = float((o["logits"].argmax(1) == r["logits"].argmax(1)).float().mean())
        mp = float((o["j3d"] - r["j3d"]).double().norm(dim=-1).mean()) * 1e3
        mpr = float((rootrel(o["j3d"]) - rootrel(r["j3d"])).norm(dim=-1).mean()) * 1e3
        parts.append(f"{kind}{C}: logits {rel(o['logits'], r['logits']):.1e} params {rel(o['params'], r['params']):.1e} MPJPE {mp:8.3f} rr {mpr:7.4f} mm argmax {agree * 100:8.4f} %")
        net.net.precision = "f32"
    print(f"{tag:28s} {rate(prec, mask):8.0f} win/s | " + " | ".join(parts), flush=True)
