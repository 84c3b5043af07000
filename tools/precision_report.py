"""Arithmetic modes of the matrix-pipe kernels against the exact fp32 MFMA mode at the bench size (B=256, N=2048):
per-tensor relative error (||d||inf / ||ref||inf), segmentation argmax agreement and the minimum top-2 logit margin.
    python tools/precision_report.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C, N = 4, 2048


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


for kind, seed in (("E", 11), ("U", 12)):
    os.environ["ERPC"] = "0"
    assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
    sd = synth.synth_state_dict(C, seed)
    xyz = synth.synth_cloud(kind, B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    outs = {}
    for prec in ("f32", "bf16x3", "f16x2", "bf16"):
        os.environ["EV2H_PRECISION"] = prec
        net = TEHNetWrapper("cuda:0", mano_assets=assets)
        net.load_state_dict(sd, strict=True)
        net.eval()
        net.net.fps_init = inits
        with torch.no_grad():
            o = net(xyz)
        outs[prec] = {"logits": o["class_logits"].clone(), "verts": torch.cat([o["left"]["vertices"], o["right"]["vertices"]], 1).clone(),
                      "j3d": torch.cat([o["left"]["j3d"], o["right"]["j3d"]], 1).clone(),
                      "pose": torch.cat([o["left"]["hand_pose"], o["right"]["hand_pose"]], 1).clone()}
    r = outs["f32"]
    top2 = r["logits"].topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    print(f"{kind}-clouds B={B} N={N}: {margin.numel()} points, min top-2 logit margin {float(margin.min()):.3e}, "
          f"logit scale {float(r['logits'].abs().max()):.3f}")
    for prec in ("bf16x3", "f16x2", "bf16"):
        o = outs[prec]
        agree = (o["logits"].argmax(1) == r["logits"].argmax(1))
        print(f"  {prec:7s} vs f32: logits {rel(o['logits'], r['logits']):.2e}  vertices {rel(o['verts'], r['verts']):.2e}  "
              f"joints {rel(o['j3d'], r['j3d']):.2e}  pose {rel(o['pose'], r['pose']):.2e}  argmax differs at {int((~agree).sum())} of {agree.numel()} points")
