"""Per-stage error of a reduced mode against exact fp32 on a (rescaled) hash-random checkpoint:
python tools/debug/f16_stage_errors.py [alpha] [mode]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402
import stress_checkpoints as sc  # noqa: E402

alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
mode = sys.argv[2] if len(sys.argv) > 2 else "f16"
B, C, N, seed = 2, 4, 1024, 11
BUFS = [("l1cat", 576, 320), ("l2buf", 520, 512), ("l3", 1024, 1024), ("fp3o", 256, 256), ("l1new", 128, 128), ("l0", 256, 256), ("logits_pm", 4, 4),
        ("sim", 256, 256), ("hf8", 8, 4), ("m1bufL", 520, 512), ("m2L", 512, 512), ("fc1L", 1024, 1024), ("m1bufR", 520, 512), ("m2R", 512, 512), ("fc1R", 1024, 1024)]
sd = sc.rescale_hidden(synth.synth_state_dict(C, seed), alpha)
xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
inits = synth.fps_inits(B, N, seed)


def run(prec):
    os.environ["ERPC"] = "0"
    net = TEHNetWrapper("cuda:0", mano_assets={s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}, precision=prec)
    net.load_state_dict(sd, strict=True)
    net.eval()
    net.net.fps_init = inits
    with torch.no_grad():
        out = net(xyz)
    torch.cuda.synchronize()
    res = {name: net.net.debug_buffer(name).view(-1, ld)[:, :ncol].clone() for name, ld, ncol in BUFS}
    res["logits"] = out["class_logits"].clone()
    return res


ref, got = run("f32"), run(mode)
print(f"## hidden x {alpha:g}, {mode} vs f32")
for k in ref:
    a, b = got[k].double(), ref[k].double()
    print(f"   {k:10s} {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):9.2e}   (max |ref| {float(b.abs().max()):9.3g}, finite {bool(torch.isfinite(a).all())})")
