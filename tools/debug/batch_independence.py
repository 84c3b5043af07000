"""Find the first workspace tensor of window 0 that differs between a batched and a single-window f16x2 forward."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper

C, N, seed = 5, 512, 12
os.environ["ERPC"] = "1"
sd = synth.synth_state_dict(C, seed)
assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
net = TEHNetWrapper("cuda:0", mano_assets=assets, precision="f16x2")
net.load_state_dict(sd, strict=True); net.eval()
xyz = synth.synth_cloud("E", 2, C, N, seed)
xyz[1, 3:] *= float(os.environ.get("DBG_MAG", "1e4"))
inits = synth.fps_inits(2, N, seed)
names = [("feat8", N * 8), ("P1a", N * 160), ("l1cat", 512 * 576), ("P1b", 512 * 256), ("l2buf", 128 * 520), ("sa3h1", 128 * 256), ("sa3h2", 128 * 512),
         ("l3", 1024), ("fp3bias", 256), ("fp3h", 128 * 256), ("fp3o", 128 * 256), ("fp2h", 512 * 256), ("l1new", 512 * 128), ("fp1T", 512 * 128),
         ("l0", N * 256), ("clsh", N * 256), ("logits_pm", N * 4), ("q1", N * 512),
         ("sim", 2 * 4 * 256), ("hf8", N * 8), ("P1mL", N * 256), ("m1bufL", 128 * 520), ("msa2hL", 128 * 256), ("m2L", 512), ("fc1L", 1024)]
recs = ["feat", "l1a", "l1b", "l2", "sa3h1", "sa3h2", "l3", "fp3h", "fp3o", "fp2h", "l1new", "fp1t", "l0", "clsh", "q1", "hfL", "hfR",
        "m1L", "msa2hL", "m2L", "fc1L", "p1a", "p1b", "p1mL"]
with torch.no_grad():
    net.net.fps_init = inits
    net(xyz.cuda()); torch.cuda.synchronize()
    full = {n: net.net.debug_buffer(n)[:c].clone() for n, c in names}
    fr = {r: net.net.debug_buffer("rng." + r).view(torch.float32)[:2].clone() for r in recs}
    fs = net.net.debug_buffer("p1scale").view(5, 2).clone()
    net.net.fps_init = [t[:1] for t in inits]
    net(xyz[:1].cuda()); torch.cuda.synchronize()
    one = {n: net.net.debug_buffer(n)[:c].clone() for n, c in names}
    orr = {r: net.net.debug_buffer("rng." + r).view(torch.float32)[:1].clone() for r in recs}
    os_ = net.net.debug_buffer("p1scale").view(5, 1).clone()
for n, c in names:
    a, b = full[n], one[n]
    d = float((a - b).abs().max())
    print(f"{n:10s} max|diff| {d:.3e}  max|val| {float(b.abs().max()):.3e}  equal {torch.equal(a, b)}")
for r in recs:
    print(f"rng.{r:8s} batched {fr[r].tolist()}  single {orr[r].tolist()}")
print("p1scale batched", fs.tolist(), "single", os_.tolist())
pw = net.net.packed(torch.device("cuda:0"))
W = pw.tensors["left_mano_regressor.sa2.0.W"].double(); bb = pw.tensors["left_mano_regressor.sa2.0.b"].double()
for tag, d in (("batched", full), ("single", one)):
    X = d["m1bufL"].view(128, 520).double()
    ref = torch.relu(X @ W.T + bb)
    got = d["msa2hL"].view(128, 256).double()
    e = (got - ref).abs()
    print(tag, "msa2hL vs fp64 GEMM of m1bufL: max err", float(e.max()), "rows with err>1e-4:", (e.amax(1) > 1e-4).nonzero().flatten().tolist()[:20],
          "cols:", (e.amax(0) > 1e-4).nonzero().flatten().tolist()[:20])
print("W shape", tuple(W.shape), "m1bufL xyz cols window0 max", float(full["m1bufL"].view(128, 520)[:, 512:].abs().max()))
