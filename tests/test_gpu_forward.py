"""GPU: the whole path through the drop-in interface (TEHNetWrapper -> ev2h_forward) against the
oracle on the same seeded inputs, and against the committed fixtures captured from the reference.
Tolerance (BASELINE.json north_star): every float output <= 1e-4 relative (||d||inf / ||ref||inf per
tensor), class_logits argmax bit-exact, FPS / ball-query / 3-NN selections identical."""
import glob
import os

import numpy as np
import pytest
import torch

from ev2hands_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if not os.path.basename(p).startswith(("events_", "metrics_", "rodrigues_", "trained_", "trained2_")))
# reference-run fixtures on checkpoints that came out of the reference's own training loop (oracle/make_golden_trained.py)
TRAINED = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "trained_*.npz")) if "weights" not in os.path.basename(p))
# [r6] ... and of a SECOND, independent training run (other start, seed, clouds, schedule and loss weighting; tests/trained_ckpt.py: RUNS)
TRAINED += sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "trained2_*.npz")) if "weights" not in os.path.basename(p))
TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def nn_mismatches(query, known, got, want):
    """3-NN parity up to ties.  query [B,N,3], known [B,S,3] coordinates, got / want [B,N,3] neighbour indices.  The reference
    takes the first three of `dists.sort(dim=-1)` (pointnet2_utils.py:297-298); torch.sort is not stable, so among candidates at
    EXACTLY the same distance (duplicated sample points, lattice-like coordinates) its choice is implementation-defined (the
    AVX-512 sort of one host and the stable sort of another differ).  A query counts as a mismatch unless the chosen
    neighbours have the same coordinates or, failing that, exactly the same fp32 distances (as the reference computes them) in
    the same order."""
    from oracle.tehnet_oracle import pairwise_sqdist          # the reference's own distance arithmetic (checker only)
    B = query.shape[0]
    bi = torch.arange(B).view(B, 1, 1)
    same_pts = (known[bi, got] == known[bi, want]).all(-1)
    d = pairwise_sqdist(query.float().contiguous(), known.float().contiguous())       # [B,N,S]
    same_dist = torch.gather(d, 2, got) == torch.gather(d, 2, want)
    return int((~(same_pts | same_dist)).any(-1).sum())


def make_net(C, seed, device="cuda:0", precision="f32", sd=None, n_pose=None):
    from ev2hands_amd.model import TEHNetWrapper
    os.environ["ERPC"] = "1" if C == 5 else "0"
    os.environ["EV2H_PRECISION"] = precision
    assets = {s: synth.synth_mano_assets(s, seed) for s in ("left", "right")}
    sd = (synth.synth_state_dict(C, seed) if n_pose is None else synth.synth_state_dict(C, seed, n_pose)) if sd is None else sd
    net = TEHNetWrapper(device, mano_assets=assets, n_pose_params=n_pose)
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net, sd, assets


def run_oracle(sd, assets, xyz, inits, n_pose=None):
    from oracle import mano_oracle, tehnet_oracle
    n_pose = synth.MANO_CMPS if n_pose is None else n_pose
    hands = mano_oracle.make_hands(assets["left"], assets["right"], ncomps=n_pose)
    trace = {}
    with torch.no_grad():
        out = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits, trace=trace, n_pose=n_pose)
    return out, trace


def check_against(out, net, ref, trace, B, N, tol=None):
    errs = {"class_logits": rel(out["class_logits"], ref["class_logits"])}
    for side in ("left", "right"):
        for k in ("vertices", "j3d", "global_orient", "hand_pose", "betas", "transl"):
            errs[f"{side}.{k}"] = rel(out[side][k], ref[side][k])
        assert out[side]["faces"].shape == (B, 1538, 3)
    # selections
    sel = {
        "sa1.fps": ("fps1", (B, 512)), "sa2.fps": ("fps2", (B, 128)),
        "left_mano_regressor.sa1.fps": ("fpsmL", (B, 128)), "right_mano_regressor.sa1.fps": ("fpsmR", (B, 128)),
        "sa1.group0": ("gidx1_0", (B, 512, 32)), "sa1.group1": ("gidx1_1", (B, 512, 64)), "sa1.group2": ("gidx1_2", (B, 512, 128)),
        "sa2.group0": ("gidx2_0", (B, 128, 64)), "sa2.group1": ("gidx2_1", (B, 128, 128)),
        "left_mano_regressor.sa1.group0": ("gidxm0L", (B, 128, 64)), "left_mano_regressor.sa1.group1": ("gidxm1L", (B, 128, 128)),
        "right_mano_regressor.sa1.group0": ("gidxm0R", (B, 128, 64)), "right_mano_regressor.sa1.group1": ("gidxm1R", (B, 128, 128)),
    }
    bad = {}
    # 3-NN: compare the selected neighbours by coordinates -- with fewer than 512 unique points the
    # sampled set holds duplicates, equal distances tie, and torch.sort's order among ties is unspecified
    l1_xyz = trace["l1_xyz"].permute(0, 2, 1)
    l2_xyz = trace["l2_xyz"].permute(0, 2, 1)
    l0_xyz = net.net.debug_buffer("pts4").view(B, N, 4)[:, :, :3].cpu()
    for tname, bname, shape, qry, pts in (("fp2.nn_idx", "nn2_idx", (B, 512, 3), l1_xyz, l2_xyz), ("fp1.nn_idx", "nn1_idx", (B, N, 3), l0_xyz, l1_xyz)):
        got = net.net.debug_buffer(bname, torch.int32).view(shape).cpu().long()
        want = torch.as_tensor(np.asarray(trace[tname])).long()
        n = nn_mismatches(qry, pts, got, want)
        if n:
            bad[tname] = n
    for tname, (bname, shape) in sel.items():
        got = net.net.debug_buffer(bname, torch.int32).view(shape).cpu().long()
        want = torch.as_tensor(np.asarray(trace[tname])).long()
        n = int((got != want).sum())
        if n:
            bad[tname] = n
    # intermediate features (diagnostics + tolerance)
    # (the workspace holds the EQUALISED channels -- pack.py: equalize_channels multiplies channel c of a hidden tensor by the power
    # of two e_c -- so the buffers are divided by those factors before they are compared with the reference's tensors)
    feats = {
        "sa1_points": ("l1cat", (B, 512, 576), slice(0, 320), "sa1.out"),
        "l0_points": ("l0", (B, N, 256), slice(0, 256), "l0"),
    }
    eq = net.net.packed(out["class_logits"].device).equalization
    for tname, (bname, shape, cols, ename) in feats.items():
        if tname in trace:
            got = net.net.debug_buffer(bname).view(shape)[:, :, cols]
            if ename in eq:
                got = got / torch.from_numpy(eq[ename]).to(got)
            errs["buf." + tname] = rel(got.permute(0, 2, 1), trace[tname])
    same = bool((out["class_logits"].argmax(1).cpu() == torch.as_tensor(ref["class_logits"]).argmax(1)).all())
    print("errors:", {k: f"{v:.2e}" for k, v in errs.items()}, "selection mismatches:", bad, "argmax identical:", same)
    assert not bad, bad
    assert max(errs.values()) < (TOL if tol is None else tol), errs
    assert same
    return errs


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("kind,C,N,B,seed", [("U", 4, 2048, 2, 0), ("E", 5, 2048, 2, 1), ("E", 4, 2048, 3, 4), ("U", 5, 256, 2, 2),
                                              ("E", 4, 8192, 1, 5), ("E", 5, 16384, 1, 9), ("U", 5, 1000, 2, 6), ("E", 4, 128, 1, 7), ("U", 4, 640, 5, 8),
                                              # odd sizes: partial 32-row strips in the row-chain kernels, ragged 256-point partial sums of
                                              # the folded attention, an odd point count, one point past a power of two
                                              ("E", 4, 333, 3, 9), ("U", 5, 2049, 1, 10), ("E", 4, 130, 2, 11)])
def test_forward_matches_oracle(kind, C, N, B, seed, precision):
    """Both fp32-class arithmetic modes must meet the full parity bar (1e-4, argmax and selections exact)."""
    _need_gpu()
    net, sd, assets = make_net(C, seed, precision=precision)
    xyz = synth.synth_cloud(kind, B, C, N, seed)
    inits = synth.fps_inits(B, N, seed)
    ref, trace = run_oracle(sd, assets, xyz, inits)
    net.net.fps_init = inits
    keep = xyz.clone()
    xg = xyz.cuda()
    with torch.no_grad():
        out = net(xg)
    torch.cuda.synchronize()
    assert torch.equal(xg.cpu(), keep), "input mutated with MHLNES=0"
    check_against(out, net, ref, trace, B, N)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_forward_matches_reference_fixture(path, precision):
    """Against numbers produced by the reference itself (oracle/make_golden.py)."""
    run_reference_fixture(path, precision)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("path", TRAINED, ids=[os.path.basename(p)[:-4] for p in TRAINED])
def test_forward_matches_reference_fixture_on_trained_weights(path, precision):
    """The same bar -- 1e-4, identical argmax, identical selections -- on weights that came out of an OPTIMISER: the reference's
    own network trained by its own loop's settings (oracle/make_golden_trained.py; BN running statistics, weight scales and logit
    margins as training left them), outputs recorded from the reference's forward.  The headline arithmetic (f16x2) is held to
    this on checkpoints nobody constructed for it."""
    import trained_ckpt
    C = int(np.load(path)["meta"][1])
    errs = run_reference_fixture(path, precision, sd=trained_ckpt.trained_state_dict(C, trained_ckpt.run_of_fixture(path)))
    print(f"trained fixture {os.path.basename(path)} [{precision}]: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))


@pytest.mark.parametrize("path", TRAINED, ids=[os.path.basename(p)[:-4] for p in TRAINED])
def test_bf16_mode_on_trained_weights(path):
    """BASELINE config 3's plain-bf16 arithmetic on the trained checkpoints: not a parity mode -- reported as MPJPE against the
    reference's joints and argmax agreement with the reference's classes, with the bounds the synthetic checkpoints hold."""
    _need_gpu()
    import trained_ckpt
    g = np.load(path)
    B, C, N, seed = [int(v) for v in g["meta"]]
    net, _sd, _assets = make_net(C, seed, precision="bf16", sd=trained_ckpt.trained_state_dict(C, trained_ckpt.run_of_fixture(path)))
    net.net.fps_init = [torch.from_numpy(g["fps_init"][i].astype(np.int64)) for i in range(4)]
    with torch.no_grad():
        out = net(torch.from_numpy(g["xyz"]).cuda())
    torch.cuda.synchronize()
    agree = float((out["class_logits"].argmax(1).cpu().numpy() == g["argmax"]).mean())
    lerr = rel(out["class_logits"], g["class_logits"])
    rep = []
    for s in ("left", "right"):
        a, b = out[s]["j3d"].cpu().double(), torch.from_numpy(g[f"unpinned.{s}.j3d"]).double()
        prm = torch.cat([out[s][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1)
        rep.append((float((a - b).norm(dim=-1).mean()) * 1e3, float(((a - a[:, :1]) - (b - b[:, :1])).norm(dim=-1).mean()) * 1e3,
                    rel(prm, g[s + ".params"]), rel(net.net.debug_buffer("hf8").view(2, B, N, 8)[("left", "right").index(s), :, :, :4].permute(0, 2, 1), g[s + ".hand_features"])))
    print(f"bf16 on trained weights {os.path.basename(path)}: argmax agreement {agree:.4f}, logits rel {lerr:.2e}; per hand (MPJPE mm, root-relative "
          f"MPJPE mm, params rel, attention features rel): " + ", ".join(f"({a:.2f}, {b:.2f}, {c:.1e}, {d:.1e})" for a, b, c, d in rep))
    # NOT a parity mode.  On these checkpoints training left post-ReLU BatchNorms with collapsed running variances (dead units:
    # var ~ 1e-23, fold scale 1 / sqrt(eps) = 316, profiles/r5_trained_checkpoint_report.txt); a unit that is dead in fp32 but gets a
    # slightly positive pre-activation from bf16 rounding is amplified 316-fold, so the regressed parameters are NOT held to a
    # bound here -- the numbers are reported (README "arithmetic modes") and only the discrete output and the logits are bounded.
    assert agree > 0.97 and lerr < 5e-2
    assert all(np.isfinite(v) for r in rep for v in r)


# [r6] What the ONE-PLANE fp16 mode ("f16": BASELINE config 3's throughput arithmetic that is usable on a trained network) must hold on
# the optimiser-made checkpoints, against the outputs the REFERENCE recorded (tests/golden/trained_*.npz; two windows each, the worse
# hand), per fixture kind: (argmax agreement >=, logits rel <=, MPJPE mm <=, root-relative MPJPE mm <=, params rel <=).
# Measured (this test's printout, profiles/r6_gpu_tests.txt):   f16                                   plain bf16
#   trained_E_c4_n2048    argmax 1.0000  logits 2.5e-3  MPJPE  8.4 mm  root-relative 0.25 mm  params 1.3e-2 | 0.9963  2.5e-2   71.6   1.97  1.1e-1
#   trained_E_c4_n8192           0.9995         6.1e-4         0.11                  0.003           7.7e-4 | 0.9946  6.2e-3    0.65  0.02  4.8e-3
#   trained_E_c5_n2048           0.9985         4.1e-3        30.4                   0.39            2.2e-2 | 0.9888  2.7e-2  271     3.67  2.4e-1
#   trained_U_c4_n2048           1.0000         1.1e-3        63                     6.2             5.5e-3 | 0.9990  8.3e-3  763    38     5.6e-2
#   trained2_E_c4_n2048          0.9998         3.8e-4         2.1                   0.043           7.7e-3 | 0.9954  2.4e-3    1.7   0.12  2.6e-2
#   trained2_U_c4_n2048          0.9990         7.3e-4         9.6                   0.70            1.8e-3 | 0.9961  6.5e-3  289     9.9   3.1e-2
# (two windows per fixture: the absolute MPJPE moved 7 -> 30 mm on E_c5 between two builds of the same arithmetic whose 32-window
#  figures are 5.0 and 5.9 mm, profiles/r6_trained_precision_report.txt.)
# The bounds are 2.5-3 x the measured worst.  Root-relative MPJPE is the reference's own metric (evaluate_ev2hands_r.py:43-54); the
# absolute figure is dominated by the regressed TRANSLATION, which passes through the regression head's Linear -> ReLU -> BatchNorm
# with collapsed running variances (fold scale 316, DESIGN.md "trained checkpoints") -- any upstream rounding is amplified there, in
# every mode, in proportion to its size (f16x2: 0.001-0.04 mm).  E = event-like clouds (what the network was trained on); U = uniform
# clouds, far outside its training distribution (one class everywhere, poses at the edge of the regressor's range): the looser row.
F16_BOUNDS = {"E": (0.997, 1e-2, 80.0, 1.0, 5.5e-2), "U": (0.998, 4e-3, 170.0, 16.0, 1.5e-2)}


def _reduced_mode_metrics(path, precision):
    import trained_ckpt
    g = np.load(path)
    B, C, N, seed = [int(v) for v in g["meta"]]
    net, _sd, _assets = make_net(C, seed, precision=precision, sd=trained_ckpt.trained_state_dict(C, trained_ckpt.run_of_fixture(path)))
    net.net.fps_init = [torch.from_numpy(g["fps_init"][i].astype(np.int64)) for i in range(4)]
    with torch.no_grad():
        out = net(torch.from_numpy(g["xyz"]).cuda())
    torch.cuda.synchronize()
    agree = float((out["class_logits"].argmax(1).cpu().numpy() == g["argmax"]).mean())
    lerr = rel(out["class_logits"], g["class_logits"])
    mp, mpr, prm = [], [], []
    for s in ("left", "right"):
        a, b = out[s]["j3d"].cpu().double(), torch.from_numpy(g[f"unpinned.{s}.j3d"]).double()
        mp.append(float((a - b).norm(dim=-1).mean()) * 1e3)
        mpr.append(float(((a - a[:, :1]) - (b - b[:, :1])).norm(dim=-1).mean()) * 1e3)       # evaluate_ev2hands_r.py:43-54
        prm.append(rel(torch.cat([out[s][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1), g[s + ".params"]))
    for name, buf in (("sa1.fps", "fps1"), ("sa2.fps", "fps2"), ("sa1.group2", "gidx1_2"), ("sa2.group1", "gidx2_1")):      # selections stay fp32: exact in every mode
        assert np.array_equal(net.net.debug_buffer(buf, torch.int32).cpu().numpy().reshape(g[name].shape), g[name].astype(np.int32)), name
    return {"argmax": agree, "logits": lerr, "mpjpe_mm": max(mp), "rr_mpjpe_mm": max(mpr), "params": max(prm)}


@pytest.mark.parametrize("path", TRAINED, ids=[os.path.basename(p)[:-4] for p in TRAINED])
def test_f16_mode_on_trained_weights_is_bounded(path):
    """the reduced-precision mode gets an accuracy CONTRACT on weights that came out of an optimiser -- bounded, not
    just finite -- and plain bf16 stays next to it as the comparison (it is 4-10 x further out on every metric)."""
    _need_gpu()
    kind = os.path.basename(path).split("_")[1]
    m = _reduced_mode_metrics(path, "f16")
    b = _reduced_mode_metrics(path, "bf16")
    print(f"reduced modes on {os.path.basename(path)} vs the reference-run fixture:\n   f16 : " + ", ".join(f"{k} {v:.4g}" for k, v in m.items())
          + "\n   bf16: " + ", ".join(f"{k} {v:.4g}" for k, v in b.items()))
    agree, lerr, mp, mpr, prm = F16_BOUNDS[kind]
    assert m["argmax"] >= agree and m["logits"] <= lerr and m["mpjpe_mm"] <= mp and m["rr_mpjpe_mm"] <= mpr and m["params"] <= prm, m
    # 11 mantissa bits against 8.  (Not asserted fixture by fixture: the ABSOLUTE MPJPE -- the regressed translation behind a collapsed-variance
    # BatchNorm, two windows per fixture -- is a noisy sample: on trained2_E the two windows give 2.1 mm (f16) against 1.7 mm (bf16) where the same
    # checkpoint at 32 windows gives 4.4 against 68.5, profiles/r6_trained_precision_report.txt.)
    assert m["logits"] < 0.5 * b["logits"] and m["rr_mpjpe_mm"] < 0.5 * b["rr_mpjpe_mm"] and m["params"] < 0.5 * b["params"] and m["argmax"] >= b["argmax"], (m, b)


def test_f16_family_masks():
    """ev2h_weights.f16_families: the F16 mode with a partial family mask runs the other families as F16X2 (same range records,
    images packed per family).  Every mask runs; reducing only the k = 3 query convolution leaves the logits at fp32 class (the
    segmentation head does not read it) and is closer to the exact-fp32 mode than reducing everything."""
    _need_gpu()
    import trained_ckpt
    from ev2hands_amd import _lib
    C, seed, B, N = 4, 61, 4, 2048
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    net, _sd, _assets = make_net(C, seed, precision="f32", sd=trained_ckpt.trained_state_dict(C))

    def run(prec, mask=0):
        net.net.precision, net.net.f16_families = prec, mask
        net.net.fps_init = inits
        with torch.no_grad():
            o = net(xyz)
        return o["class_logits"].clone(), torch.cat([o[s][k] for s in ("left", "right") for k in ("global_orient", "hand_pose", "betas", "transl")], 1).clone()

    ref = run("f32")
    full = run("f16")
    assert net.net.packed(xyz.device).struct.f16_families == _lib.FAM_ALL
    same = run("f16", _lib.FAM_ALL)
    assert torch.equal(full[0], same[0]) and torch.equal(full[1], same[1])
    errs = {}
    for mask in (_lib.FAM_SA, _lib.FAM_ROWS, _lib.FAM_QCONV, _lib.FAM_DENSE, _lib.FAM_SA | _lib.FAM_QCONV):
        got = run("f16", mask)
        assert net.net.packed(xyz.device).struct.f16_families == mask
        errs[mask] = (rel(got[0], ref[0]), rel(got[1], ref[1]))
    efull = (rel(full[0], ref[0]), rel(full[1], ref[1]))
    print("f16 family masks, (logits, params) rel vs f32:", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in errs.items()}, "all:", tuple(f"{v:.1e}" for v in efull))
    assert errs[_lib.FAM_QCONV][0] < 2e-5                      # the logits never see the query convolution
    assert errs[_lib.FAM_QCONV][1] < efull[1]
    assert all(1e-7 < e[1] < 0.2 for e in errs.values())
    net.net.f16_families = 0


def run_reference_fixture(path, precision, sd=None):
    _need_gpu()
    g = np.load(path)
    B, C, N, seed = [int(v) for v in g["meta"]]
    mhlnes = bool(int(g["mhlnes"])) if "mhlnes" in g.files else False
    os.environ["MHLNES"] = "1" if mhlnes else "0"          # read at construction, like TEHNet.py:148
    n_pose = int(g["n_pose"]) if "n_pose" in g.files else None              # TEHNet(n_pose_params): fixtures *_pose<K>
    try:
        net, sd, assets = make_net(C, seed, precision=precision, sd=sd, n_pose=n_pose)
    finally:
        os.environ["MHLNES"] = "0"
    assert net.net.mhlnes == int(mhlnes)
    xyz = torch.from_numpy(g["xyz"])
    inits = [torch.from_numpy(g["fps_init"][i].astype(np.int64)) for i in range(4)]
    ties = "tie_eps" in g.files
    if ties:         # near-tie segmentation head: the fixture was made with the checkpoint of oracle/stress.py
        from oracle import mano_oracle, stress
        hands = mano_oracle.make_hands(assets["left"], assets["right"])
        net.load_state_dict(stress.near_tie_state_dict(sd, xyz, inits, hands, float(g["tie_eps"])), strict=True)
    net.net.fps_init = inits
    xg = xyz.cuda()
    with torch.no_grad():
        out = net(xg)
    torch.cuda.synchronize()
    if mhlnes:       # the reference overwrites channel 2 of the caller's tensor in place (TEHNet.py:176-177): so must the drop-in
        assert np.array_equal(xg.cpu().numpy(), g["xyz_after"]) and not np.array_equal(g["xyz_after"], g["xyz"])
    else:
        assert torch.equal(xg.cpu(), xyz)
    if ties:
        # 5.6 % of the points have a top-2 margin below 1e-5 of the logit scale, 0.7 % below 1e-6.  Two correct fp32 evaluations of
        # these 256-term sums differ by a few 1e-6 of the scale (tests/test_gpu_stress.py), so: every point whose margin in the
        # reference is at least 2e-5 of the scale must get the reference's class; inside that band agreement is reported
        lg = torch.from_numpy(g["class_logits"]).double()
        top = lg.topk(2, dim=1).values
        margin, scale = top[:, 0] - top[:, 1], float(lg.abs().max())
        got = out["class_logits"].argmax(1).cpu()
        want = torch.from_numpy(g["argmax"].astype(np.int64))
        safe = margin >= 2e-5 * scale
        inside = ~safe
        print(f"near-ties [{precision}]: {int((margin < 1e-5 * scale).sum())} points < 1e-5 scale, {int(inside.sum())} inside the 2e-5 band, "
              f"{int((got[inside] == want[inside]).sum())} of those agree; max |logit err| / scale = "
              f"{float((out['class_logits'].cpu().double() - lg).abs().max()) / scale:.2e}")
        assert float((margin < 1e-5 * scale).float().mean()) > 0.01
        assert torch.equal(got[safe], want[safe])
    else:
        assert np.array_equal(out["class_logits"].argmax(1).cpu().numpy(), g["argmax"])
    errs = {"class_logits": rel(out["class_logits"], g["class_logits"])}
    assert errs["class_logits"] < TOL, errs
    for h, side in enumerate(("left", "right")):
        prm = torch.cat([out[side][k] for k in ("global_orient", "hand_pose", "betas", "transl")], 1)
        errs[side + ".params"] = rel(prm, g[side + ".params"])
        errs[side + ".vertices"] = rel(out[side]["vertices"], g[f"unpinned.{side}.vertices"])
        errs[side + ".j3d"] = rel(out[side]["j3d"], g[f"unpinned.{side}.j3d"])
    assert max(errs.values()) < TOL, errs
    for name, buf in (("sa1.fps", "fps1"), ("sa2.fps", "fps2"), ("sa1.group2", "gidx1_2"), ("sa2.group1", "gidx2_1")):
        got = net.net.debug_buffer(buf, torch.int32).cpu().numpy().reshape(g[name].shape)
        assert np.array_equal(got, g[name].astype(np.int32)), name
    # 3-NN by neighbour coordinates (ties among duplicated sample points have no defined order in torch.sort)
    l1_xyz = torch.from_numpy(g["sa1.new_xyz"]).permute(0, 2, 1)
    got = net.net.debug_buffer("nn1_idx", torch.int32).view(B, N, 3).cpu().long()
    want = torch.from_numpy(g["fp1.nn_idx"].astype(np.int64))
    l0_xyz = torch.from_numpy(g["xyz_after"] if mhlnes else g["xyz"])[:, :3].permute(0, 2, 1)
    assert nn_mismatches(l0_xyz, l1_xyz, got, want) == 0
    assert rel(net.net.debug_buffer("nn1_w").view(B, N, 3), g["fp1.nn_w"]) < 1e-5
    hf = net.net.debug_buffer("hf8").view(2, B, N, 8)
    for h, side in enumerate(("left", "right")):
        errs[side + ".hand_features"] = rel(hf[h, :, :, :4].permute(0, 2, 1), g[side + ".hand_features"])
        assert errs[side + ".hand_features"] < TOL, errs
    return errs


def test_foreign_mano_hands_are_called_like_the_reference_does():
    """TEHNet.forward accepts ANY hand model with the adapter's interface (model/utils.py:14-31; TEHNet.py:92-105 only uses
    .shapedirs.device, __call__ and .faces).  With such objects -- here the CPU oracle's hands -- the regressed parameters are
    handed to them on their own device and their vertices / joints are returned, as the reference does."""
    _need_gpu()
    from oracle import mano_oracle
    B, C, N, seed = 2, 4, 512, 21
    net, sd, assets = make_net(C, seed)
    xyz = synth.synth_cloud("E", B, C, N, seed)
    inits = synth.fps_inits(B, N, seed)
    foreign = mano_oracle.make_hands(assets["left"], assets["right"])          # CPU tensors, not ManoHand objects
    with torch.no_grad():
        net.net.fps_init = inits
        native = net(xyz.cuda())
        native = {s: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in native[s].items()} for s in ("left", "right")}
        net.net.fps_init = inits
        out = net.net(xyz.cuda(), foreign)
        net.net.fps_init = inits
        mixed = net.net(xyz.cuda(), {"left": net.hands["left"], "right": foreign["right"]})
    for side in ("left", "right"):
        assert out[side]["vertices"].device.type == "cpu" and out[side]["global_orient"].device.type == "cpu"
        for k in ("global_orient", "hand_pose", "betas", "transl"):
            assert torch.equal(out[side][k], native[side][k].cpu())
        want = foreign[side](**{k: native[side][k].cpu() for k in ("global_orient", "hand_pose", "betas", "transl")})
        assert torch.equal(out[side]["vertices"], want.vertices) and torch.equal(out[side]["j3d"], want.joints)
        assert rel(out[side]["vertices"], native[side]["vertices"]) < 1e-5     # oracle MANO vs the HIP kernel
        assert out[side]["faces"].shape == (B, 1538, 3)
    assert mixed["left"]["vertices"].is_cuda and torch.equal(mixed["left"]["vertices"], native["left"]["vertices"])
    assert torch.equal(mixed["right"]["vertices"], out["right"]["vertices"])


def test_rng_draw_order_matches_reference():
    """Without an explicit fps_init the wrapper consumes torch's global CPU RNG like the reference."""
    _need_gpu()
    B, C, N, seed = 2, 4, 2048, 0
    net, sd, assets = make_net(C, seed)
    xyz = synth.synth_cloud("U", B, C, N, seed)
    torch.manual_seed(123)
    inits = [torch.randint(0, hi, (B,), dtype=torch.long) for hi in (N, 512, N, N)]
    net.net.fps_init = inits
    with torch.no_grad():
        a = net(xyz.cuda())
        torch.manual_seed(123)
        b = net(xyz.cuda())
    assert torch.equal(a["class_logits"], b["class_logits"])
    assert torch.equal(a["left"]["vertices"], b["left"]["vertices"])


def test_sharded_equals_unsharded():
    """Windows are independent: a batch slice with the matching slice of FPS inits gives bit-identical rows."""
    _need_gpu()
    B, C, N, seed = 4, 4, 2048, 3
    net, sd, assets = make_net(C, seed)
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    net.net.fps_init = inits
    with torch.no_grad():
        full = net(xyz)
        full = {"class_logits": full["class_logits"].clone(), "v": full["right"]["vertices"].clone()}
        net.net.fps_init = [t[2:] for t in inits]
        part = net(xyz[2:].contiguous())
    assert torch.equal(full["class_logits"][2:], part["class_logits"])
    assert torch.equal(full["v"][2:], part["right"]["vertices"])


def test_state_dict_roundtrip_and_module_prefix():
    _need_gpu()
    net, sd, assets = make_net(4, 0)
    assert list(net.state_dict().keys()) == list(sd.keys()) and len(sd) == 342
    net.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)
    with pytest.raises(RuntimeError):
        bad = dict(sd)
        bad.pop("classifier.4.bias")
        net.load_state_dict(bad, strict=True)


def test_window_size_limits_match_the_reference_domain():
    """N < 128 is outside the reference's own domain: query_ball_point takes `[:, :, :nsample]` of an N-column tensor and then
    indexes it with an nsample-column mask (pointnet2_utils.py:103-106), an IndexError as soon as N < 128 = the largest nsample
    (checked here on the oracle, which restates those lines).  The library rejects such windows with an error instead."""
    _need_gpu()
    from ev2hands_amd import _lib
    net, sd, assets = make_net(4, 0)
    xyz = synth.synth_cloud("U", 1, 4, 64, 0)
    with pytest.raises(IndexError):
        run_oracle(sd, assets, xyz, synth.fps_inits(1, 64, 0))
    with pytest.raises(_lib.Ev2hError, match="bad argument"):
        net(xyz.cuda())
    with pytest.raises(_lib.Ev2hError, match="bad argument"):
        net(torch.zeros(1, 4, 32800, device="cuda"))             # > 32768: the documented upper limit of the selection kernels


def test_training_forward_is_refused():
    _need_gpu()
    net, sd, assets = make_net(4, 0)
    net.train()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 4, 256, device="cuda"))


def test_bf16_mode_reports_mpjpe_and_argmax_agreement():
    """BASELINE.json config 3: plain-bf16 MFMA operands (selection math stays fp32).  Not a parity mode:
    report root-relative MPJPE (mm, formula of evaluate_ev2hands_r.py:43-54) and argmax agreement vs fp32."""
    _need_gpu()
    B, C, N, seed = 4, 4, 2048, 2
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    outs = {}
    for prec in ("f32", "bf16"):
        net, sd, assets = make_net(C, seed, precision=prec)
        net.net.fps_init = inits
        with torch.no_grad():
            o = net(xyz)
        outs[prec] = {"logits": o["class_logits"].clone(), "j": torch.cat([o["left"]["j3d"], o["right"]["j3d"]], 1).clone()}
    os.environ["EV2H_PRECISION"] = "f32"
    a, b = outs["f32"]["j"], outs["bf16"]["j"]
    a = a - a[:, :1]
    b = b - b[:, :1]
    mpjpe_mm = float((a - b).norm(dim=-1).mean() * 1000)
    agree = float((outs["f32"]["logits"].argmax(1) == outs["bf16"]["logits"].argmax(1)).float().mean())
    print(f"bf16 vs fp32: root-relative MPJPE {mpjpe_mm:.4f} mm over 42 joints, argmax agreement {agree * 100:.3f} %")
    assert mpjpe_mm < 5.0 and agree > 0.97


def test_bf16_mode_against_the_oracle():
    """BASELINE.json config 3 against the CPU oracle (not only against the library's own fp32 mode): root-relative MPJPE in
    mm (evaluate_ev2hands_r.py:43-54), segmentation argmax agreement, and exact selections of the first stage (selection math
    stays fp32 in every mode; later selections depend on bf16-rounded features only through FPS/ball-query inputs, which are
    coordinates -- so ALL selections must still be identical)."""
    _need_gpu()
    B, C, N, seed = 3, 4, 2048, 14
    net, sd, assets = make_net(C, seed, precision="bf16")
    xyz = synth.synth_cloud("E", B, C, N, seed)
    inits = synth.fps_inits(B, N, seed)
    ref, trace = run_oracle(sd, assets, xyz, inits)
    net.net.fps_init = inits
    with torch.no_grad():
        out = net(xyz.cuda())
    torch.cuda.synchronize()
    a = torch.cat([out["left"]["j3d"], out["right"]["j3d"]], 1).cpu()
    b = torch.cat([ref["left"]["j3d"], ref["right"]["j3d"]], 1)
    mpjpe_mm = float(((a - a[:, :1]) - (b - b[:, :1])).norm(dim=-1).mean() * 1000)
    agree = float((out["class_logits"].argmax(1).cpu() == ref["class_logits"].argmax(1)).float().mean())
    err = rel(out["class_logits"], ref["class_logits"])
    print(f"bf16 vs oracle: root-relative MPJPE {mpjpe_mm:.4f} mm, argmax agreement {agree * 100:.3f} %, logits rel err {err:.2e}")
    assert mpjpe_mm < 5.0 and agree > 0.97 and err < 5e-2
    for tname, bname, shape in (("sa1.fps", "fps1", (B, 512)), ("sa2.fps", "fps2", (B, 128)), ("sa1.group2", "gidx1_2", (B, 512, 128)),
                                ("sa2.group1", "gidx2_1", (B, 128, 128)), ("left_mano_regressor.sa1.group1", "gidxm1L", (B, 128, 128))):
        got = net.net.debug_buffer(bname, torch.int32).view(shape).cpu().long()
        assert torch.equal(got, torch.as_tensor(np.asarray(trace[tname])).long()), tname


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3"])
def test_config5_as_eight_shards_of_16_windows_equals_the_128_window_batch(precision):
    """BASELINE.json config 5 reads "N=8192 ..., B=128, 8 GPUs" the way config 4 reads "B=2048 sharded across 8": a rank then
    holds SIXTEEN windows of 8192 points.  That shape takes other launch decisions than 128 windows per GPU (one ball-query
    centroid per wave, strips spread over a workgroup's waves, quarter-size GEMM tiles, fewer range-record atomics per address),
    none of which may change a number: the eight 16-window shards, run one after the other with their slices of the globally
    drawn FPS starts (ev2hands_amd/dist.py), must reproduce the 128-window forward bit for bit, selections included."""
    _need_gpu()
    from ev2hands_amd import dist as evdist
    B, C, N, seed, world = 128, 4, 8192, 21, 8
    net, sd, assets = make_net(C, seed, precision=precision)
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    keys = (("class_logits", None), ("left", "vertices"), ("right", "vertices"), ("left", "j3d"), ("right", "j3d"), ("left", "betas"), ("right", "hand_pose"))
    pick = lambda o: [o[a].clone() if b is None else o[a][b].clone() for a, b in keys]      # noqa: E731
    with torch.no_grad():
        net.net.fps_init = inits
        full = pick(net(xyz))
        gi = net.net.debug_buffer("gidx1_2", torch.int32).view(B, 512, 128).clone()
        for r in range(world):
            lo, hi = evdist.shard_range(B, r, world)
            assert hi - lo == 16
            net.net.fps_init = evdist.shard_fps_inits(inits, lo, hi)
            part = pick(net(xyz[lo:hi].contiguous()))
            for f, p_ in zip(full, part):
                assert torch.isfinite(p_).all()
                assert torch.equal(f[lo:hi], p_), (r,)
            assert torch.equal(gi[lo:hi], net.net.debug_buffer("gidx1_2", torch.int32).view(16, 512, 128))
    del net
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision,B,N", [("bf16", 256, 2048), ("f16x2", 128, 8192), ("f16x2", 16, 8192), ("bf16", 128, 8192), ("f32", 128, 8192)])
def test_baseline_config_sizes_properties(precision, B, N):
    """BASELINE.json config 3 (B=256, N=2048, bf16) and config 5 (N=8192 dense windows, B=128 per GPU) at full size, where the
    CPU oracle would need minutes per window: size-independent properties -- all outputs finite, and reversing the order of the
    windows (with their FPS inits) reverses every output bit-exactly; plus parity of window 0 run alone (B=1 is covered against
    the oracle / the reference fixture at these N by test_forward_matches_oracle / ..._reference_fixture)."""
    _need_gpu()
    C, seed = 4, 15
    net, sd, assets = make_net(C, seed, precision=precision)
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    keys = (("class_logits", None), ("left", "vertices"), ("right", "j3d"), ("left", "betas"))
    pick = lambda o: [o[a].clone() if b is None else o[a][b].clone() for a, b in keys]      # noqa: E731
    with torch.no_grad():
        net.net.fps_init = inits
        fwd = pick(net(xyz))
        net.net.fps_init = [t.flip(0) for t in inits]
        rev = pick(net(xyz.flip(0).contiguous()))
        net.net.fps_init = [t[:1] for t in inits]
        one = pick(net(xyz[:1].contiguous()))
    torch.cuda.synchronize()
    for f, r, o in zip(fwd, rev, one):
        assert torch.isfinite(f).all()
        assert torch.equal(f.flip(0), r)
        assert torch.equal(f[:1], o)
    del net
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_full_size_batch_properties(precision):
    """BASELINE.json config 2 size (B=64, N=2048, C=4), too slow for the CPU oracle: size-independent properties instead --
    permuting the windows of a batch permutes every output bit-exactly (windows are independent, weights shared), and the
    two fp32-class arithmetic modes agree with each other within the parity tolerance."""
    _need_gpu()
    B, C, N, seed = 64, 4, 2048, 9
    net, sd, assets = make_net(C, seed, precision=precision)
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    perm = torch.from_numpy(np.argsort(synth.hash_uniform("perm", (B,), seed)))
    net.net.fps_init = inits
    with torch.no_grad():
        a = net(xyz)
        a = {"l": a["class_logits"].clone(), "v": a["left"]["vertices"].clone(), "j": a["right"]["j3d"].clone(),
             "p": a["right"]["betas"].clone()}
        net.net.fps_init = [t[perm] for t in inits]
        b = net(xyz[perm.cuda()].contiguous())
    pc = perm.cuda()
    assert torch.equal(a["l"][pc], b["class_logits"]) and torch.equal(a["v"][pc], b["left"]["vertices"])
    assert torch.equal(a["j"][pc], b["right"]["j3d"]) and torch.equal(a["p"][pc], b["right"]["betas"])
    assert torch.isfinite(a["l"]).all() and torch.isfinite(a["v"]).all()
    if precision != "f32":
        net32, _, _ = make_net(C, seed, precision="f32")
        net32.net.fps_init = inits
        with torch.no_grad():
            r = net32(xyz)
        assert rel(a["l"], r["class_logits"]) < TOL and rel(a["v"], r["left"]["vertices"]) < TOL
        agree = float((a["l"].argmax(1) == r["class_logits"].argmax(1)).float().mean())
        print(f"{precision} vs f32 at B=64: logits rel {rel(a['l'], r['class_logits']):.2e}, argmax agreement {agree * 100:.4f} %")
        assert agree == 1.0


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
@pytest.mark.parametrize("B", [1, 8])
def test_hipgraph_replay_is_bit_identical_to_eager(B, precision):
    """ev2h_forward neither allocates nor synchronises, so the whole path (including the fork of the right-hand regressor onto the
    library's side stream) can be captured into a hipGraph (TEHNet.capture).  Replays must reproduce the eager forward bit for
    bit, on the captured inputs and on new inputs copied into the graph's static buffers."""
    _need_gpu()
    C, N, seed = 4, 2048, 17
    net, sd, assets = make_net(C, seed, precision=precision)
    xa = synth.synth_cloud("E", B, C, N, seed).cuda()
    xb = synth.synth_cloud("U", B, C, N, seed + 1).cuda()
    ia, ib = synth.fps_inits(B, N, seed), synth.fps_inits(B, N, seed + 1)

    def flat(o):
        return torch.cat([o["class_logits"].flatten()] + [o[s][k].flatten() for s in ("left", "right")
                                                           for k in ("vertices", "j3d", "global_orient", "hand_pose", "betas", "transl")]).clone()
    with torch.no_grad():
        net.net.fps_init = ia
        ea = flat(net(xa))
        net.net.fps_init = ib
        eb = flat(net(xb))
        g = net.capture(xa, fps_init=ia)
        ga = flat(g.replay())
        ga2 = flat(g.replay())
        gb = flat(g.replay(xb, fps_init=ib))
        ga3 = flat(g.replay(xa, fps_init=ia))
        net.net.fps_init = ib
        eb2 = flat(net(xb))                                   # eager still works after a capture
    torch.cuda.synchronize()
    assert torch.equal(ea, ga) and torch.equal(ea, ga2) and torch.equal(ea, ga3)
    assert torch.equal(eb, gb) and torch.equal(eb, eb2)
    assert g.out["left"]["faces"].shape == (B, 1538, 3)


def test_captured_forward_owns_what_its_graph_points_at():
    """A hipGraph bakes device addresses in.  The net's shared workspace is re-allocated by a later, larger eager forward, and the
    packed weights are dropped when the precision or the parameters change: a captured forward therefore owns its workspace and
    holds the packed weights it was captured with -- replays after such events stay bit-identical, or refuse to run."""
    _need_gpu()
    C, N, seed = 4, 1024, 19
    net, sd, assets = make_net(C, seed, precision="f16x2")
    x1 = synth.synth_cloud("E", 1, C, N, seed).cuda()
    x8 = synth.synth_cloud("U", 8, C, N, seed + 1).cuda()
    i1, i8 = synth.fps_inits(1, N, seed), synth.fps_inits(8, N, seed + 1)

    def flat(o):
        return torch.cat([o["class_logits"].flatten()] + [o[s][k].flatten() for s in ("left", "right")
                                                           for k in ("vertices", "j3d", "global_orient", "hand_pose", "betas", "transl")]).clone()
    with torch.no_grad():
        net.net.fps_init = i1
        e1 = flat(net(x1))
        g = net.capture(x1, fps_init=i1)
        assert torch.equal(flat(g.replay()), e1)
        net.net.fps_init = i8
        net(x8)                                               # larger batch: the net's own workspace is freed and re-allocated
        junk = torch.full((64 << 20,), 7.0, device="cuda")    # ... and whatever the allocator freed gets overwritten
        assert torch.equal(flat(g.replay()), e1)
        net.net.precision = "f32"                             # the eager path re-packs; the graph keeps the weights it captured
        net.net.fps_init = i1
        net(x1)
        with pytest.raises(RuntimeError, match="capture again"):
            g.replay()
        net.net.precision = "f16x2"                           # back to the captured arithmetic (re-packed: same key)
        net.net.fps_init = i1
        net(x1)
        assert torch.equal(flat(g.replay()), e1)
        net.load_state_dict(synth.synth_state_dict(C, seed + 5), strict=True)
        with pytest.raises(RuntimeError, match="capture again"):
            g.replay()
    torch.cuda.synchronize()
    del junk


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_forward_into_caller_owned_rows(precision):
    """ev2h_outputs' window strides: the forward writes every window's predictions as one row of a caller-owned matrix (the layout
    of the multi-GPU gather buffer, ev2hands_amd/dist.py) -- bit-identical to the dense outputs, nothing written outside the
    row's first packed_width columns, the returned tensors are views of the matrix."""
    _need_gpu()
    from ev2hands_amd import dist as evdist
    C, N, B, seed = 5, 777, 3, 20
    net, sd, assets = make_net(C, seed, precision=precision)
    x = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    W = evdist.packed_width(N)
    big = torch.full((B + 2, W + 5), -123.0, device="cuda")
    rows = big[1:B + 1]                                       # a slice of a larger buffer, row stride W + 5
    with torch.no_grad():
        net.net.fps_init = inits
        dense = net(x)
        net.net.fps_init = inits
        out = net.net(x, net.hands, rows=rows)
    torch.cuda.synchronize()
    assert out["class_logits"].data_ptr() == rows.data_ptr()
    assert torch.equal(out["class_logits"], dense["class_logits"])
    for side in ("left", "right"):
        for k in ("vertices", "j3d", "global_orient", "hand_pose", "betas", "transl"):
            assert torch.equal(out[side][k], dense[side][k]), (side, k)
    back = evdist.unpack_outputs(rows[:, :W], N)
    assert torch.equal(back["class_logits"], dense["class_logits"]) and torch.equal(back["right"]["j3d"], dense["right"]["j3d"])
    assert torch.equal(evdist.pack_outputs(dense), rows[:, :W])
    assert bool((big[0] == -123.0).all()) and bool((big[-1] == -123.0).all()) and bool((big[:, W:] == -123.0).all())
    with pytest.raises(RuntimeError, match="rows must be"):
        net.net(x, net.hands, rows=torch.empty(B, W - 1, device="cuda"))


def test_two_stream_fork_is_bit_identical(tmp_path):
    """EV2H_TWO_STREAMS=1 (opt-in: right-hand regressor and the MANO ball queries on a second stream) must not change a bit.
    The switch is read once per process, so the forked run happens in a child process."""
    _need_gpu()
    import subprocess
    import sys
    script = tmp_path / "run.py"
    script.write_text(
        "import os, sys, torch\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from ev2hands_amd import synth\n"
        "from ev2hands_amd.model import TEHNetWrapper\n"
        "os.environ['ERPC'] = '0'\n"
        "assets = {s: synth.synth_mano_assets(s, 3) for s in ('left', 'right')}\n"
        "net = TEHNetWrapper('cuda:0', mano_assets=assets, precision='f16x2')\n"
        "net.load_state_dict(synth.synth_state_dict(4, 3), strict=True); net.eval()\n"
        "xyz = synth.synth_cloud('E', 6, 4, 2048, 3).cuda()\n"
        "outs = []\n"
        "for _ in range(3):\n"
        "    net.net.fps_init = synth.fps_inits(6, 2048, 3)\n"
        "    with torch.no_grad():\n"
        "        o = net(xyz)\n"
        "    outs.append(torch.cat([o['class_logits'].flatten(), o['left']['vertices'].flatten(), o['right']['vertices'].flatten(),\n"
        "                           o['left']['j3d'].flatten(), o['right']['betas'].flatten()]).cpu())\n"
        "assert all(torch.equal(outs[0], x) for x in outs[1:])\n"
        "torch.save(outs[0], sys.argv[1])\n")
    res = {}
    for fork in ("0", "1"):
        env = dict(os.environ, EV2H_TWO_STREAMS=fork)      # "0" = single stream, "1" = forked (the default)
        out = tmp_path / f"out{fork}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[fork] = torch.load(out)
    assert torch.equal(res["0"], res["1"])


# Every A/B switch the library still reads from the environment (each `static const bool`, read once per process) selects an
# alternative code path that ALSO serves another arithmetic mode or window shape (the layer-1 tables: BF16X3 / F32; the un-fused
# fp1 / classifier / q1 forms: F32, BF16X3, windows that do not tile by 128; the resident vs streamed set abstraction: the MLP
# widths that do not fit LDS; the generic k = 3 loader: odd window sizes).  Round 5 retired the fourteen switches whose A/B was
# settled and whose path nothing else needs.  Each remaining one runs the whole forward in a child
# process and is compared with the default path -- bit-identical where the arithmetic is the same, <= 1e-5 where the switch
# re-associates sums (the un-fused forms), selections and argmax identical in every case.
AB_SWITCHES = [("EV2H_TWO_STREAMS", "0", True), ("EV2H_SA_STREAMED", "1", True), ("EV2H_FPS_CHUNKS", "100000", True),     # (3 windows: one launch by default -- the value forces the chunked sampling of 8+ windows)
               ("EV2H_FP1_UNFUSED", "1", False), ("EV2H_CLS_UNFUSED", "1", False), ("EV2H_ATTN_UNFUSED_ZSUM", "1", False),
               ("EV2H_L1_TABLE", "1", False), ("EV2H_L0_F32", "1", True)]       # (EV2H_L0_F32 acts in BF16 only: see test_bf16_l0_storage below)


_AB_SCRIPT = """
import os, sys, torch
sys.path.insert(0, {root!r})
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
os.environ['ERPC'] = '1'
assets = {{s: synth.synth_mano_assets(s, 3) for s in ('left', 'right')}}
res = {{}}
for prec in ('f16x2', 'bf16x3'):
    net = TEHNetWrapper('cuda:0', mano_assets=assets, precision=prec)
    net.load_state_dict(synth.synth_state_dict(5, 3), strict=True); net.eval()
    xyz = synth.synth_cloud('E', 3, 5, 1100, 3).cuda()
    net.net.fps_init = synth.fps_inits(3, 1100, 3)
    with torch.no_grad():
        o = net(xyz)
    torch.cuda.synchronize()
    res[prec] = {{'logits': o['class_logits'].cpu(), 'gidx': net.net.debug_buffer('gidxm1R', torch.int32).cpu(), 'nn': net.net.debug_buffer('nn1_idx', torch.int32).cpu(),
                 **{{f'{{s}}.{{k}}': o[s][k].cpu() for s in ('left', 'right') for k in ('vertices', 'j3d', 'global_orient', 'hand_pose', 'betas', 'transl')}}}}
torch.save(res, sys.argv[1])
"""


@pytest.fixture(scope="module")
def ab_default(tmp_path_factory):
    _need_gpu()
    import subprocess
    import sys
    d = tmp_path_factory.mktemp("ab")
    script = d / "run.py"
    script.write_text(_AB_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = {k: v for k, v in os.environ.items() if k not in {n for n, _, _ in AB_SWITCHES}}
    r = subprocess.run([sys.executable, str(script), str(d / "default.pt")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return script, env, torch.load(d / "default.pt")


@pytest.mark.parametrize("name,value,bit_identical", AB_SWITCHES, ids=[n for n, _, _ in AB_SWITCHES])
def test_ab_switch_paths_agree_with_the_default(ab_default, tmp_path, name, value, bit_identical):
    import subprocess
    import sys
    script, env, want = ab_default
    out = tmp_path / "alt.pt"
    r = subprocess.run([sys.executable, str(script), str(out)], env=dict(env, **{name: value}), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = torch.load(out)
    for prec in want:
        for k, w in want[prec].items():
            g = got[prec][k]
            if bit_identical or k in ("gidx", "nn"):
                assert torch.equal(g, w), (name, prec, k)
            else:
                assert rel(g, w) < 1e-5, (name, prec, k, rel(g, w))
        assert torch.equal(got[prec]["logits"].argmax(1), want[prec]["logits"].argmax(1)), (name, prec)


def test_bf16_table_switch_agrees_with_the_table_free_layer1(tmp_path):
    """EV2H_L1_TABLE=1 (layer-1 tables + gathers, as in the other modes) against the default BF16 path (layer 1 on the matrix pipe
    from the raw feature rows): same selections, outputs within bf16 rounding of each other, both inside the bf16 bar against the
    oracle (test_bf16_mode_against_the_oracle runs the default)."""
    _need_gpu()
    import subprocess
    import sys
    script = tmp_path / "run.py"
    script.write_text(_AB_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).replace("('f16x2', 'bf16x3')", "('bf16',)"))
    res = {}
    for tag, env in (("mfma", {}), ("table", {"EV2H_L1_TABLE": "1"})):
        out = tmp_path / f"{tag}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(out)["bf16"]
    for k in ("gidx", "nn"):
        assert torch.equal(res["mfma"][k], res["table"][k]), k
    worst = max(rel(res["mfma"][k], res["table"][k]) for k in res["mfma"] if k not in ("gidx", "nn"))
    agree = float((res["mfma"]["logits"].argmax(1) == res["table"]["logits"].argmax(1)).float().mean())
    print(f"bf16: table-free layer 1 vs layer-1 tables: worst relative difference {worst:.2e}, argmax agreement {agree:.4f}")
    assert worst < 3e-2 and agree > 0.97


@pytest.mark.parametrize("name,value,bit_identical", AB_SWITCHES, ids=[n for n, _, _ in AB_SWITCHES])
def test_ab_switch_paths_in_f16(tmp_path, name, value, bit_identical):
    """[r6] The same switches under `f16` (one fp16 plane): the alternative paths -- layer-1 tables whose producer chooses the chain's
    power of two (EV2H_L1_TABLE), streamed weights, one stream, chunked sampling at 3 windows, the un-fused row chains, l0 as fp32 --
    must give the same selections and outputs within the mode's own rounding (bit-identical where the arithmetic is the same)."""
    _need_gpu()
    import subprocess
    import sys
    script = tmp_path / "run.py"
    script.write_text(_AB_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).replace("('f16x2', 'bf16x3')", "('f16',)"))
    env0 = {k: v for k, v in os.environ.items() if k not in {n for n, _, _ in AB_SWITCHES}}
    res = {}
    for tag, env in (("default", {}), ("alt", {name: value})):
        out = tmp_path / f"{tag}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(env0, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(out)["f16"]
    for k in ("gidx", "nn"):
        assert torch.equal(res["default"][k], res["alt"][k]), k
    floats = [k for k in res["default"] if k not in ("gidx", "nn")]
    assert all(torch.isfinite(res["alt"][k]).all() for k in floats)
    worst = max(rel(res["default"][k], res["alt"][k]) for k in floats)
    agree = float((res["default"]["logits"].argmax(1) == res["alt"]["logits"].argmax(1)).float().mean())
    print(f"f16, {name}={value}: worst relative difference to the default path {worst:.2e}, argmax agreement {agree:.5f}")
    if bit_identical and name != "EV2H_L0_F32":
        assert worst == 0.0
    else:
        assert worst < 1e-2 and agree > 0.995


def test_query_convolution_without_q1_agrees_with_the_two_pass_form(tmp_path):
    """[r4, bf16x3: r5] bf16 / f16x2 / bf16x3 at window sizes that tile by 128 rows: the first query convolution forms the attention's key-weighted sums in its
    epilogue and never writes q1 (gemm_bf16.hip: zsum_epilogue).  EV2H_ATTN_UNFUSED_ZSUM=1 = q1 to memory + attn_zsum_kernel (the
    fp32 fma chain): same selections, f16x2 within 1e-5 of each other (different summation order), bf16 within its own rounding;
    both inside the parity bar against the oracle (test_forward_matches_oracle runs the default at N = 256 .. 2048)."""
    _need_gpu()
    import subprocess
    import sys
    script = tmp_path / "run.py"
    text = _AB_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    text = text.replace("('f16x2', 'bf16x3')", "('f16x2', 'bf16', 'bf16x3')").replace("1100", "1024")
    assert "1024" in text
    script.write_text(text)
    env0 = {k: v for k, v in os.environ.items() if k != "EV2H_ATTN_UNFUSED_ZSUM"}
    res = {}
    for tag, env in (("fused", {}), ("two_pass", {"EV2H_ATTN_UNFUSED_ZSUM": "1"})):
        out = tmp_path / f"{tag}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(env0, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(out)
    for prec, tol in (("f16x2", 1e-5), ("bf16", 3e-2), ("bf16x3", 1e-5)):
        a, b = res["fused"][prec], res["two_pass"][prec]
        for k in ("gidx", "nn"):
            assert torch.equal(a[k], b[k]), (prec, k)
        assert torch.equal(a["logits"], b["logits"]), prec             # the head does not depend on the attention
        worst = max(rel(a[k], b[k]) for k in a if k not in ("gidx", "nn"))
        print(f"{prec}: q1 never written vs two-pass: worst relative difference {worst:.2e}")
        assert worst < tol, (prec, worst)
        assert worst > 0.0 or prec == "bf16", "the switch changed nothing: is the fused path taken?"


@pytest.mark.parametrize("precision,B,N", [("f16x2", 3, 2048), ("bf16x3", 2, 1000)])
def test_inflight_forwards_are_bit_identical(precision, B, N):
    """ev2hands_amd/inflight.py: forwards issued round-robin on two (three) streams with their own workspaces -- different
    batches in flight at the same time, sharing the library's one side stream and its events -- must each reproduce the plain
    forward of their batch bit for bit (a shared workspace, or an event recorded by one forward and consumed by the other, would
    show as a mismatch on some repetition)."""
    _need_gpu()
    from ev2hands_amd.inflight import InflightForward
    C, seed = 4, 17
    net, sd, assets = make_net(C, seed, precision=precision)
    clouds = [synth.synth_cloud("E" if i % 2 else "U", B, C, N, seed + i).cuda() for i in range(4)]
    inits = [synth.fps_inits(B, N, seed + i) for i in range(4)]
    keys = (("class_logits", None), ("left", "vertices"), ("right", "j3d"), ("left", "betas"), ("right", "transl"))
    pick = lambda o: [o[a].clone() if b is None else o[a][b].clone() for a, b in keys]      # noqa: E731
    want = []
    with torch.no_grad():
        for x, ini in zip(clouds, inits):
            net.net.fps_init = ini
            want.append(pick(net(x)))
    torch.cuda.synchronize()
    for depth in (2, 3):
        pipe = InflightForward(net, depth=depth)
        for rep in range(3):
            tickets = []
            for x, ini in zip(clouds, inits):
                net.net.fps_init = ini
                tickets.append(pipe.submit(x))
            for i, t in enumerate(tickets):
                got = pick(t.result())
                for g_, w_ in zip(got, want[i]):
                    assert torch.equal(g_, w_), (depth, rep, i)
        pipe.drain()
    torch.cuda.synchronize()


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("n_pose,C", [(1, 4), (12, 5), (45, 4)])
def test_forward_with_other_pose_widths(n_pose, C, precision):
    """TEHNet(n_pose_params) takes any number of MANO PCA coefficients (TEHNet.py:114-125): the head is 3 + n_pose + 10 + 3 wide,
    the MANO layer takes n_pose components, the `rows=` layout follows.  Against the oracle at the full parity bar, and through
    the gather-buffer layout (ev2hands_amd/dist.py: packed_width / unpack_outputs with n_pose)."""
    _need_gpu()
    from ev2hands_amd import dist as evdist
    B, N, seed = 2, 512, 30 + n_pose
    net, sd, assets = make_net(C, seed, precision=precision, n_pose=n_pose)
    assert sd["left_mano_regressor.mano_regressor.4.weight"].shape == (16 + n_pose, 1024)
    xyz, inits = synth.synth_cloud("E", B, C, N, seed), synth.fps_inits(B, N, seed)
    ref, trace = run_oracle(sd, assets, xyz, inits, n_pose=n_pose)
    net.net.fps_init = inits
    with torch.no_grad():
        out = net(xyz.cuda())
    torch.cuda.synchronize()
    assert out["left"]["hand_pose"].shape == (B, n_pose) and out["right"]["betas"].shape == (B, 10)
    check_against(out, net, ref, trace, B, N)
    rows = torch.zeros(B, evdist.packed_width(N, n_pose), device="cuda")
    net.net.fps_init = inits
    with torch.no_grad():
        net.net(xyz.cuda(), net.hands, rows=rows)
    got = evdist.unpack_outputs(rows, N, n_pose)
    for side in ("left", "right"):
        for k in ("global_orient", "hand_pose", "betas", "transl", "vertices", "j3d"):
            assert torch.equal(got[side][k], out[side][k]), (side, k)
    # a hand model with another number of components than the checkpoint regresses is refused, by name
    from ev2hands_amd.mano import create_mano_layers
    wrong = create_mano_layers(None, "cuda:0", 6 if n_pose != 6 else 7, assets=assets)
    net.net.fps_init = inits
    with pytest.raises(Exception, match="pose coefficients"):
        net.net(xyz.cuda(), wrong)


def test_bf16_l0_storage_agrees_with_fp32_storage(tmp_path):
    """[r5] BF16 stores l0 -- written once by the fp1 chain, read by the segmentation head, the k = 3 query convolution and the
    attention context -- as bf16 (2.1 of the step's 6.0 GB of HBM traffic touch it).  Every BF16 reader rounds it to bf16 anyway, so
    against EV2H_L0_F32=1 (fp32 storage, the round-4 form): identical selections, IDENTICAL logits (the head sees the same bf16
    values either way), the regressed outputs within bf16 rounding of each other (only the attention context used to read the
    unrounded values)."""
    _need_gpu()
    import subprocess
    import sys
    script = tmp_path / "run.py"
    text = _AB_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    text = text.replace("('f16x2', 'bf16x3')", "('bf16',)").replace("1100", "1024")
    script.write_text(text)
    env0 = {k: v for k, v in os.environ.items() if k != "EV2H_L0_F32"}
    res = {}
    for tag, env in (("bf16_l0", {}), ("f32_l0", {"EV2H_L0_F32": "1"})):
        out = tmp_path / f"{tag}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(env0, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(out)["bf16"]
    a, b = res["bf16_l0"], res["f32_l0"]
    for k in ("gidx", "nn"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["logits"], b["logits"])
    worst = max(rel(a[k], b[k]) for k in a if k not in ("gidx", "nn"))
    print(f"bf16: l0 stored as bf16 vs fp32: worst relative difference {worst:.2e}")
    assert 0.0 < worst < 3e-2


def test_debug_buffer_knows_what_l0_holds_in_bf16_mode():
    """in BF16 mode the workspace's "l0" holds bf16 values (2 bytes each); ev2h_workspace_buffer_ex reports the element
    type and TEHNet.debug_buffer widens it -- read as float32 bytes (what the old accessor implied) it was garbage."""
    _need_gpu()
    C, seed, B, N = 4, 17, 2, 1024
    xyz = synth.synth_cloud("E", B, C, N, seed).cuda()
    inits = synth.fps_inits(B, N, seed)
    l0 = {}
    for prec in ("f32", "bf16", "f16x2", "f16"):
        net, _sd, _assets = make_net(C, seed, precision=prec)
        net.net.fps_init = inits
        with torch.no_grad():
            net(xyz)
        torch.cuda.synchronize()
        l0[prec] = net.net.debug_buffer("l0").view(B, N, 256).cpu()
        if prec == "bf16":
            with pytest.raises(TypeError):
                net.net.debug_buffer("l0", torch.int32)
    assert l0["bf16"].dtype == torch.float32 and torch.isfinite(l0["bf16"]).all()
    assert rel(l0["f16x2"], l0["f32"]) < 1e-5
    e = rel(l0["bf16"], l0["f32"])
    print(f"bf16-stored l0 against the exact-fp32 mode's: {e:.2e}")
    assert 1e-5 < e < 2e-2
    e16 = rel(l0["f16"], l0["f32"])          # fp16 values times the window's power of two (workspace "p1scale" row 5): undone by debug_buffer
    print(f"fp16-stored l0 (F16 mode) against the exact-fp32 mode's: {e16:.2e}")
    assert 1e-6 < e16 < 3e-3
