import os, sys, torch
sys.path.insert(0, "/root/repo")
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
from oracle import mano_oracle, tehnet_oracle
sys.path.insert(0, "/root/repo/tools")
from fuzz_modes import run, rel, KEYS
C, kind, B, N, seed = 4, "L", 3, 2628, 946906
os.environ["ERPC"]="0"; os.environ["MHLNES"]="0"
for alpha in (1e-4, 1e3, 1e6):
    sd = synth.rescale_hidden(synth.synth_state_dict(C, seed), alpha)
    assets = {s: synth.synth_mano_assets(s, seed % 7) for s in ("left", "right")}
    net = TEHNetWrapper("cuda:0", mano_assets=assets); net.load_state_dict(sd, strict=True); net.eval()
    xyz = synth.synth_cloud(kind, B, C, N, seed).cuda(); inits = synth.fps_inits(B, N, seed)
    hands = mano_oracle.make_hands(assets["left"], assets["right"])
    with torch.no_grad():
        o = tehnet_oracle.tehnet_forward(sd, xyz.cpu().clone(), hands, fps_init=inits)
    truth = {"class_logits": o["class_logits"], **{f"{s_}.{k}": o[s_][k] for s_ in ("left", "right") for k in KEYS}}
    res = {}
    for prec in ("f32", "f16x2", "bf16x3"):
        got, _ = run(net, xyz, inits, prec); res[prec] = got
        e = {k: rel(got[k], truth[k]) for k in got}; k = max(e, key=e.get)
        print(f"alpha {alpha:g} {prec:7s} vs CPU oracle: worst {k} {e[k]:.2e};  right.j3d {e['right.j3d']:.2e} right.hand_pose {e['right.hand_pose']:.2e} right.global_orient {e['right.global_orient']:.2e}")
    e = {k: rel(res['f16x2'][k], res['f32'][k]) for k in res['f32']}; k = max(e, key=e.get)
    print(f"   f16x2 vs f32 mode: worst {k} {e[k]:.2e}")
