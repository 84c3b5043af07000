import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
import trained_ckpt, stress_checkpoints as sc
from oracle import mano_oracle, tehnet_oracle
C, N, B, seed = 4, 1024, 2, 71
os.environ["ERPC"] = "0"
assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
xyz = synth.synth_cloud("E", B, C, N, seed); inits = synth.fps_inits(B, N, seed)
hands = mano_oracle.make_hands(assets["left"], assets["right"])
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for name, sd in (("trained", trained_ckpt.trained_state_dict(C)), ("trained x hidden 1e-3", sc.rescale_hidden(trained_ckpt.trained_state_dict(C), 1e-3)),
                 ("trained x hidden 1e5", sc.rescale_hidden(trained_ckpt.trained_state_dict(C), 1e5))):
    with torch.no_grad():
        ref = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits)
    ref64 = None
    outs = {}
    for prec in ("f32", "bf16x3", "f16x2"):
        net = TEHNetWrapper("cuda:0", mano_assets=assets, precision=prec); net.load_state_dict(sd, strict=True); net.eval()
        net.net.fps_init = inits
        with torch.no_grad():
            o = net(xyz.cuda())
        outs[prec] = {"logits": o["class_logits"].cpu(), "L": torch.cat([o["left"][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1).cpu(),
                      "R": torch.cat([o["right"][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1).cpu()}
    r = {"logits": ref["class_logits"], "L": torch.cat([ref["left"][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1),
         "R": torch.cat([ref["right"][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1)}
    print("##", name)
    for prec in outs:
        print(f"  {prec:7s} vs CPU fp32 oracle: " + "  ".join(f"{k} {rel(outs[prec][k], r[k]):.2e}" for k in r))
    if ref64 is not None:
        r64 = {"logits": ref64["class_logits"], "L": torch.cat([ref64["left"][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1),
               "R": torch.cat([ref64["right"][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1)}
        print(f"  CPU fp32 oracle vs CPU fp64 evaluation: " + "  ".join(f"{k} {rel(r[k], r64[k]):.2e}" for k in r))
        for prec in outs:
            print(f"  {prec:7s} vs CPU fp64 evaluation: " + "  ".join(f"{k} {rel(outs[prec][k], r64[k]):.2e}" for k in r))
