import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
os.environ["ERPC"]="0"
for kind in ("E","U"):
    B,C,N=16,4,2048
    assets = {s: synth.synth_mano_assets(s, 0) for s in ("left","right")}
    net = TEHNetWrapper("cuda:0", mano_assets=assets); net.load_state_dict(synth.synth_state_dict(C,0)); net.eval()
    xyz = synth.synth_cloud(kind,B,C,N,3).cuda()
    with torch.no_grad(): net(xyz)
    for name, S, Ks in (("cnt1",512,(32,64,128)),("cnt2",128,(64,128)),("cntmL",128,(64,128))):
        c = net.net.debug_buffer(name, torch.int32).view(B,S,len(Ks)).cpu().numpy()
        for i,K in enumerate(Ks):
            strips = np.ceil(c[:,:,i]/32)
            print(kind, name, "K=%d"%K, "mean cnt %.1f"%c[:,:,i].mean(), "saturated %.2f"%(c[:,:,i]>=K).mean(), "mean strips %.2f of %d"%(strips.mean(), K//32))
