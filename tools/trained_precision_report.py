"""The arithmetic modes on the TRAINED checkpoints (tests/golden/trained_weights_c{4,5}.npz, oracle/make_golden_trained.py) at a
batch the CPU reference would need minutes for: every mode against the exact-fp32 MFMA mode of this library (which the GPU tests
hold to the reference-run fixtures), plus what the f16x2 guard reports -- TEHNet.verify_precision (f16x2 vs bf16x3 on the batch),
the range report of the materialised operands and the packed-weight spread.
    python tools/trained_precision_report.py [B]      ->  profiles/r6_trained_precision_report.txt
[r6] with "f16" (one fp16 plane under f16x2's range records) next to "bf16", and the root-relative MPJPE of
evaluate_ev2hands_r.py:43-54 (each hand's joints minus its root joint) next to the absolute one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ev2hands_amd import synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402
import trained_ckpt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 2048


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


CASES = [(4, "E", 51, "a"), (4, "U", 52, "a"), (5, "E", 53, "a")] + ([(4, "E", 54, "b"), (4, "U", 55, "b")] if trained_ckpt.available("b") else [])
for C, kind, seed, run in CASES:
    os.environ["ERPC"] = "1" if C == 5 else "0"
    assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
    sd = trained_ckpt.trained_state_dict(C, run)
    xyz = synth.synth_cloud(kind, B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    outs = {}
    for prec in ("f32", "bf16x3", "f16x2", "f16", "bf16"):
        net = TEHNetWrapper("cuda:0", mano_assets=assets, precision=prec)
        net.load_state_dict(sd, strict=True)
        net.eval()
        net.net.fps_init = inits
        with torch.no_grad():
            o = net(xyz)
        outs[prec] = {"logits": o["class_logits"].clone(),
                      "params": torch.cat([o[s][k] for s in ("left", "right") for k in ("global_orient", "hand_pose", "betas", "transl")], 1).clone(),
                      "verts": torch.cat([o["left"]["vertices"], o["right"]["vertices"]], 1).clone(),
                      "j3d": torch.cat([o["left"]["j3d"], o["right"]["j3d"]], 1).clone()}
        if prec == "f16x2":
            rep = net.net.verify_precision(xyz, net.hands, fps_init=inits)
            eq = net.net.packed(xyz.device).equalization
    r = outs["f32"]
    top2 = r["logits"].topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    scale = float(r["logits"].abs().max())
    print(f"## trained checkpoint (run {run}) C={C}, {kind}-clouds, B={B}, N={N}: logit scale {scale:.3f}, top-2 margin < 1e-4 scale at "
          f"{float((margin < 1e-4 * scale).float().mean()) * 100:.3f} % of the points, class histogram {torch.bincount(r['logits'].argmax(1).flatten(), minlength=4).tolist()}")
    def rootrel(j):          # [B, 42, 3]: left joints 0..20, right 21..41, each minus its own root (evaluate_ev2hands_r.py:43-54)
        j = j.double().view(j.shape[0], 2, 21, 3)
        return (j - j[:, :, :1]).view(j.shape[0], 42, 3)
    for prec in ("bf16x3", "f16x2", "f16", "bf16"):
        o = outs[prec]
        agree = (o["logits"].argmax(1) == r["logits"].argmax(1))
        mp = float((o["j3d"] - r["j3d"]).double().norm(dim=-1).mean()) * 1e3
        mpr = float((rootrel(o["j3d"]) - rootrel(r["j3d"])).norm(dim=-1).mean()) * 1e3
        mpw = float((o["j3d"] - r["j3d"]).double().norm(dim=-1).mean(1).max()) * 1e3
        print(f"  {prec:7s} vs f32: logits {rel(o['logits'], r['logits']):.2e}  params {rel(o['params'], r['params']):.2e}  vertices {rel(o['verts'], r['verts']):.2e}  "
              f"joints {rel(o['j3d'], r['j3d']):.2e}  MPJPE {mp:.4f} mm (worst window {mpw:.4f})  root-relative MPJPE {mpr:.4f} mm  "
              f"argmax agreement {float(agree.float().mean()) * 100:.4f} % ({int((~agree).sum())} of {agree.numel()} points differ)")
    worst = sorted(rep["range"].items(), key=lambda kv: -kv[1]["worst_fraction"])[:4]
    print(f"  verify_precision (f16x2 vs bf16x3 on this batch): max_rel {rep['max_rel']:.2e}, argmax agreement {rep['argmax_agreement']:.6f}, ok(<= {net.net.AUTO_TOLERANCE:g}) = {rep['ok']}")
    print("  range report, operands with the largest share of values more than 2^17 below their window's maximum: "
          + ", ".join(f"{k} {v['worst_fraction'] * 100:.3f} %" for k, v in worst))
    print(f"  packed weights: {rep['weights']['below_2^-17']} of {rep['weights']['nonzero']} non-zero plane-image weights sit more than 2^17 below their matrix's scale")
    fac = torch.cat([torch.as_tensor(v).flatten().double() for v in eq.values()]) if eq else torch.ones(1)
    print(f"  channel equalisation (data-free, power-of-two): {len(eq)} tensors, factors 2^{float(torch.log2(fac.min())):.0f} .. 2^{float(torch.log2(fac.max())):.0f}")
