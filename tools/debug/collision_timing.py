"""Time the two-hand collision kernels on posed synthetic hands (surface-like or soup assets): count-only search (cap 8), pair-list
search (cap 16), penalty.   python tools/debug/collision_timing.py [surface|soup] [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import synth, collision as col
from ev2hands_amd.model import TEHNetWrapper
kind = sys.argv[1] if len(sys.argv) > 1 else "surface"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mk = synth.synth_mano_surface_assets if kind == "surface" else synth.synth_mano_assets
assets = {s: mk(s, 0) for s in ("left", "right")}
net = TEHNetWrapper("cuda:0", mano_assets=assets); net.load_state_dict(synth.synth_state_dict(4, 0), strict=True); net.eval()
xyz = synth.synth_cloud("E", B, 4, 2048, 1000).cuda()
net.net.fps_init = synth.fps_inits(B, 2048, 7)
with torch.no_grad(): out = net(xyz)
fl, fr = col.device_faces(net.hands["left"].faces, "cuda:0"), col.device_faces(net.hands["right"].faces, "cuda:0")
vl, vr = out["left"]["vertices"], out["right"]["vertices"]
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
c8, _ = col.mesh_collisions(vl, vr, fl, fr, max_per_triangle=8)
c16, p16 = col.mesh_collisions(vl, vr, fl, fr, max_pairs=2 * 1538 * 16, scale=1.0, max_per_triangle=16)
print(kind, "B", B, "pairs per window (cap 8): mean %.0f max %d; (cap 16): mean %.0f max %d" % (c8.float().mean(), c8.max(), c16.float().mean(), c16.max()))
print("count-only search cap 8        : %.3f ms" % t(lambda: col.mesh_collisions(vl, vr, fl, fr, max_per_triangle=8)))
print("pair-list search cap 16 (metres): %.3f ms" % t(lambda: col.mesh_collisions(vl, vr, fl, fr, max_pairs=2 * 1538 * 16, scale=1.0, max_per_triangle=16)))
cl = col.CollisionLoss("cuda:0")
print("CollisionLoss.per_window (search + penalty): %.3f ms" % t(lambda: cl.per_window(out, faces=(fl, fr))))
