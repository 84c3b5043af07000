"""Largest activation magnitudes of the path on the bench workload (the f16x2 mode needs |x| < 65504): python tools/activation_range.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
B, C, N = 64, 4, 2048
os.environ["ERPC"] = "0"
assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
net = TEHNetWrapper("cuda:0", mano_assets=assets, precision="f32")
net.load_state_dict(synth.synth_state_dict(C, 0), strict=True); net.eval()
for kind in ("E", "U"):
    xyz = synth.synth_cloud(kind, B, C, N, 1000).cuda()
    net.net.fps_init = synth.fps_inits(B, N, 7)
    with torch.no_grad(): net(xyz)
    torch.cuda.synchronize()
    out = []
    for name in ("P1a", "l1cat", "P1b", "l2buf", "sa3h1", "sa3h2", "l3", "fp3o", "fp2h", "l1new", "fp1T", "l0", "clsh", "q1",
                 "P1mL", "m1bufL", "msa2hL", "m2L", "fc1L"):
        t = net.net.debug_buffer(name)
        out.append(f"{name} {float(t.abs().max()):.3g}")
    print(kind, "max |activation|:", ", ".join(out))
