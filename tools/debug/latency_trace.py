import os, sys, torch
sys.path.insert(0, os.getcwd())
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
B, C, N = int(os.environ.get("LB", "1")), 4, 2048
assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
net = TEHNetWrapper("cuda:0", mano_assets=assets); net.load_state_dict(synth.synth_state_dict(C, 0), strict=True); net.eval()
xyz = synth.synth_cloud("E", B, C, N, 1000).cuda()
with torch.no_grad():
    for i in range(12):
        net.net.fps_init = synth.fps_inits(B, N, 7)
        net(xyz)
torch.cuda.synchronize()
