"""Validate the MI355X path on the REAL assets (licensed MANO_{LEFT,RIGHT}.pkl + best_model_state_dict.pth), which this
repository cannot ship and its CI has never seen.

Two steps, on two machines:

  1. where the reference runs (its conda env with manopth; CPU is enough):

       python tools/validate_real_assets.py make --reference /path/to/Ev2Hands/src/Ev2Hands \
              --mano /path/to/Ev2Hands/src/data/models --ckpt /path/to/best_model_state_dict.pth --out real_assets.npz

     imports the reference's own `model.TEHNetWrapper` (model/model.py:10-64, manopth MANO layer and all), loads the checkpoint
     strict=True as demo.py:83-84 does, runs it on seeded synthetic event clouds and stores inputs, the FPS start indices the
     reference drew (pointnet2_utils.py:75) and every output.  No asset byte goes into the file (only SHA-256 digests, to
     recognise a mismatch later).

  2. on the MI355X box (assets needed again, the reference is not):

       python tools/validate_real_assets.py check --fixture real_assets.npz --mano /path/to/models --ckpt /path/to/best_model_state_dict.pth
       # or:  EV2H_REAL_FIXTURE=real_assets.npz EV2H_MANO_PATH=/path/to/models EV2H_CKPT=/path/to/ckpt.pth python -m pytest tests/test_real_assets.py -m gpu

     loads the pkl files with the chumpy-free reader (ev2hands_amd/mano.py), the checkpoint with the drop-in wrapper, replays the
     recorded inputs through libev2hands_hip.so and reports max relative error per output, segmentation argmax agreement and the
     root-relative MPJPE (mm, evaluate_ev2hands_r.py:43-54) against the reference -- in every arithmetic mode.

`python tests/real_assets_dryrun.py DIR OUT` replaces the reference by this repository's CPU oracle and a synthetic checkpoint /
MANO-shaped assets written in the real file formats: a dry run of the whole procedure (tests/test_real_assets.py uses it; the oracle
is test infrastructure, so that part lives under tests/).
"""
from __future__ import annotations

import argparse
import hashlib
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ev2hands_amd import synth  # noqa: E402

CASES = [("E", 2, 2048, 0), ("E", 1, 2048, 1), ("U", 2, 2048, 2)]       # (cloud kind, B, N, seed)
KEYS = ("vertices", "j3d", "global_orient", "hand_pose", "betas", "transl")


def sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def load_checkpoint(path: str) -> dict:
    ck = torch.load(path, map_location="cpu")
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck              # demo.py:83
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def channels_of(sd: dict) -> int:
    return int(sd["sa1.conv_blocks.0.0.weight"].shape[1]) - 3


# ------------------------------------------------------------------------------------------------ make
def run_reference(ref_dir: str, mano_dir: str, sd: dict, C: int):
    """The reference's own wrapper on the CPU.  Returns a function (xyz) -> (outputs, [4 fps init vectors])."""
    os.environ["ERPC"] = "1" if C == 5 else "0"
    sys.path.insert(0, os.path.abspath(ref_dir))
    sys.path.insert(0, os.path.abspath(os.path.join(ref_dir, "..")))             # settings.py lives in src/
    import settings                                                                # noqa: F401  (reference module)
    settings.MANO_PATH = mano_dir
    from model import TEHNetWrapper                                               # the reference's class
    net = TEHNetWrapper(device=torch.device("cpu"))
    net.load_state_dict(sd, strict=True)
    net.eval()

    def run(xyz):
        drawn = []
        o_randint = torch.randint

        def spy(*a, **k):
            v = o_randint(*a, **k)
            drawn.append(v.clone())
            return v

        torch.randint = spy
        try:
            with torch.no_grad():
                out = net(xyz.clone())
        finally:
            torch.randint = o_randint
        assert len(drawn) == 4, f"expected 4 FPS start draws, saw {len(drawn)}"
        return out, drawn
    return run


def make_fixture(run, C: int, ckpt: str, mano_dir: str, out_path: str, source: str) -> int:
    """Record `run(xyz) -> (outputs, [4 FPS start vectors])` on the seeded cases.  `run` is the reference's own wrapper (cmd_make);
    the test suite's dry run passes the CPU oracle instead (tests/real_assets_dryrun.py -- the oracle is test infrastructure and is
    not imported by this tool)."""
    torch.manual_seed(1234)
    out = {"channels": np.array(C), "ncases": np.array(len(CASES)), "source": np.array(source),
           "sha256.ckpt": np.array(sha256(ckpt))}
    for side in ("left", "right"):
        out[f"sha256.mano_{side}"] = np.array(sha256(os.path.join(mano_dir, "mano", f"MANO_{side.upper()}.pkl")))
    for i, (kind, B, N, seed) in enumerate(CASES):
        xyz = synth.synth_cloud(kind, B, C, N, seed)
        res, drawn = run(xyz)
        out[f"{i}.xyz"] = xyz.numpy()
        out[f"{i}.fps_init"] = torch.stack(drawn).numpy().astype(np.int64)
        out[f"{i}.class_logits"] = res["class_logits"].numpy()
        for side in ("left", "right"):
            for k in KEYS:
                out[f"{i}.{side}.{k}"] = res[side][k].detach().numpy()
        print(f"case {i}: {kind}-cloud B={B} N={N}: logits scale {float(res['class_logits'].abs().max()):.3g}, "
              f"class histogram {torch.bincount(res['class_logits'].argmax(1).flatten(), minlength=4).tolist()}")
    np.savez_compressed(out_path, **out)
    print(f"wrote {out_path} ({os.path.getsize(out_path) // 1024} KiB) from the {source}")
    return 0


def cmd_make(a) -> int:
    sd = load_checkpoint(a.ckpt)
    C = channels_of(sd)
    return make_fixture(run_reference(a.reference, a.mano, sd, C), C, a.ckpt, a.mano, a.out, "reference")


# ------------------------------------------------------------------------------------------------ check
def rel(x, y) -> float:
    x, y = torch.as_tensor(x).double().cpu(), torch.as_tensor(y).double().cpu()
    return float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))


def check_fixture(fixture: str, mano_dir: str, ckpt: str, precisions=("f32", "f16x2", "bf16x3", "bf16"), device="cuda:0", verbose=True):
    """Returns {precision: {"max_rel": .., "argmax_agreement": .., "mpjpe_mm": ..}}; raises if an asset digest differs."""
    from ev2hands_amd.model import TEHNetWrapper
    g = np.load(fixture)
    for name, path in (("ckpt", ckpt), ("mano_left", os.path.join(mano_dir, "mano", "MANO_LEFT.pkl")),
                       ("mano_right", os.path.join(mano_dir, "mano", "MANO_RIGHT.pkl"))):
        if sha256(path) != str(g[f"sha256.{name}"]):
            raise RuntimeError(f"{path} is not the file the fixture was made with (SHA-256 differs)")
    C = int(g["channels"])
    os.environ["ERPC"] = "1" if C == 5 else "0"
    sd = load_checkpoint(ckpt)
    if channels_of(sd) != C:
        raise RuntimeError(f"the checkpoint has {channels_of(sd)} input channels, the fixture was made with {C} (ERPC mismatch)")
    report = {}
    for prec in precisions:
        net = TEHNetWrapper(device, mano_path=mano_dir, precision=prec)
        net.load_state_dict(sd, strict=True)
        net.eval()
        worst, agree, npts, mp_sum, mp_n = 0.0, 0, 0, 0.0, 0
        for i in range(int(g["ncases"])):
            xyz = torch.from_numpy(g[f"{i}.xyz"]).to(device)
            net.net.fps_init = [torch.from_numpy(v) for v in g[f"{i}.fps_init"]]
            with torch.no_grad():
                out = net(xyz)
            torch.cuda.synchronize()
            errs = {"class_logits": rel(out["class_logits"], g[f"{i}.class_logits"])}
            for side in ("left", "right"):
                for k in KEYS:
                    errs[f"{side}.{k}"] = rel(out[side][k], g[f"{i}.{side}.{k}"])
            worst = max(worst, max(errs.values()))
            am = out["class_logits"].argmax(1).cpu()
            want = torch.from_numpy(g[f"{i}.class_logits"]).argmax(1)
            agree += int((am == want).sum())
            npts += am.numel()
            j = torch.cat([out["left"]["j3d"], out["right"]["j3d"]], 1).cpu().double()
            r = torch.cat([torch.from_numpy(g[f"{i}.left.j3d"]), torch.from_numpy(g[f"{i}.right.j3d"])], 1).double()
            mp_sum += float(((j - j[:, :1]) - (r - r[:, :1])).norm(dim=-1).sum() * 1000)
            mp_n += j.shape[0] * j.shape[1]
            if verbose:
                print(f"  [{prec}] case {i}: worst output {max(errs, key=errs.get)} rel err {max(errs.values()):.2e}")
        report[prec] = {"max_rel": worst, "argmax_agreement": agree / npts, "mpjpe_mm": mp_sum / mp_n}
        if verbose:
            print(f"[{prec}] max relative error {worst:.2e}, argmax agreement {100.0 * agree / npts:.4f} %, "
                  f"root-relative MPJPE vs reference {mp_sum / mp_n:.5f} mm")
    return report


def cmd_check(a) -> int:
    rep = check_fixture(a.fixture, a.mano, a.ckpt)
    ok = all(rep[p]["max_rel"] < 1e-4 and rep[p]["argmax_agreement"] > 0.9999 for p in rep if p != "bf16")
    print("PARITY", "OK" if ok else "FAILED", "(1e-4 relative on every output, argmax agreement; bf16 is reported only)")
    return 0 if ok else 1


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("make")
    m.add_argument("--reference", help="path to Ev2Hands/src/Ev2Hands of the reference checkout")
    m.add_argument("--mano", help="directory that holds mano/MANO_{LEFT,RIGHT}.pkl (settings.MANO_PATH)")
    m.add_argument("--ckpt", help="best_model_state_dict.pth")
    m.add_argument("--out", required=True)
    c = sub.add_parser("check")
    c.add_argument("--fixture", required=True)
    c.add_argument("--mano", required=True)
    c.add_argument("--ckpt", required=True)
    a = ap.parse_args()
    if a.cmd == "make":
        if not (a.reference and a.mano and a.ckpt):
            ap.error("make needs --reference, --mano and --ckpt (a dry run with synthetic assets: python tests/real_assets_dryrun.py DIR OUT)")
        return cmd_make(a)
    return cmd_check(a)


if __name__ == "__main__":
    sys.exit(main())
