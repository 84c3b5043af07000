"""Where does plain bf16 lose the trained checkpoint?  Every materialised stage of the forward in bf16 against the same stage in
exact fp32 (same library, same selections), trained checkpoint vs the hash-random one:  python tools/debug/bf16_trained_layers.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402
import trained_ckpt  # noqa: E402

B, C, N, seed = 2, 4, 2048, 41
BUFS = [("l1cat", 576, 320), ("l2buf", 520, 512), ("l3", 1024, 1024), ("fp3o", 256, 256), ("l1new", 128, 128), ("l0", 256, 256), ("logits_pm", 4, 4),
        ("sim", 256, 256), ("hf8", 8, 4), ("m1bufL", 520, 512), ("m2L", 512, 512), ("fc1L", 1024, 1024), ("m1bufR", 520, 512), ("m2R", 512, 512), ("fc1R", 1024, 1024)]


def run(sd, prec, xyz, inits):
    os.environ["ERPC"] = "0"
    net = TEHNetWrapper("cuda:0", mano_assets={s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}, precision=prec)
    net.load_state_dict(sd, strict=True)
    net.eval()
    net.net.fps_init = inits
    with torch.no_grad():
        out = net(xyz)
    torch.cuda.synchronize()
    eq = net.net.packed(xyz.device).equalization
    res = {}
    for name, ld, ncol in BUFS:
        t = net.net.debug_buffer(name).view(-1, ld)[:, :ncol].clone()
        res[name] = t
    for s in ("left", "right"):
        res[s + ".params"] = torch.cat([out[s][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1).clone()
        res[s + ".j3d"] = out[s]["j3d"].clone()
    return res, eq


xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
inits = synth.fps_inits(B, N, seed)
for tag, sd in (("trained", trained_ckpt.trained_state_dict(C)), ("hash-random", synth.synth_state_dict(C, seed))):
    ref, _ = run(sd, "f32", xyz, inits)
    for prec in ("bf16", "f16x2"):
        got, _ = run(sd, prec, xyz, inits)
        print(f"## {tag} checkpoint, {prec} vs f32 (max |diff| / max |ref| per stage; equalised channels compared as stored where both modes equalise alike)")
        for k in ref:
            a, b = got[k].double(), ref[k].double()
            if a.shape != b.shape:
                continue
            print(f"   {k:14s} {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):9.2e}   (max |ref| {float(b.abs().max()):9.3g})")
