import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
net = TEHNetWrapper("cuda:0", mano_assets=assets); net.load_state_dict(synth.synth_state_dict(4, 0), strict=True); net.eval()
outs = {}
for B in (1, 8):
    x = synth.synth_cloud("E", B, 4, 2048, 77).cuda(); inits = synth.fps_inits(B, 2048, 77)
    def once():
        net.net.fps_init = inits
        with torch.no_grad(): return net(x)
    for _ in range(5): o = once()
    ts = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); once(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    g = net.capture(x, inits)
    for _ in range(3): g.replay()
    tg = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); tg.append((time.perf_counter() - t0) * 1e3)
    print(f"B={B}: eager median {sorted(ts)[20]:.3f} ms, hipgraph median {sorted(tg)[20]:.3f} ms")
    outs[B] = torch.cat([o["class_logits"].flatten(), o["left"]["vertices"].flatten(), o["right"]["j3d"].flatten()]).cpu()
torch.save(outs, sys.argv[1])
