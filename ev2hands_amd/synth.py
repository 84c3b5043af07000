"""Deterministic synthetic checkpoints, event clouds and MANO-shaped assets.

Everything here is generated from a counter-based integer hash (splitmix64), so
the numbers do not depend on torch / numpy RNG streams and are reproduced
bit-for-bit in the survey container (where the golden fixtures are made) and on
the GPU box (where they are consumed).  SURVEY.md section 8c/8d defines the
distributions.

Reference facts this mirrors (shapes only, no code):
  * checkpoint schema, 342 entries          -> /root/reference/src/Ev2Hands/model/TEHNet.py:116-166,
                                               pointnet2_utils.py:161-275
  * event window -> [C, N] tensor           -> /root/reference/src/Ev2Hands/dataset/ev2hands_r.py:108-159, 21-35
  * MANO asset shapes                       -> /root/reference/src/Ev2Hands/model/utils.py:13-42 (manopth ManoLayer buffers)
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

OUTPUT_WIDTH = 346    # /root/reference/src/settings.py:21
OUTPUT_HEIGHT = 260   # /root/reference/src/settings.py:22
MANO_CMPS = 6         # /root/reference/src/settings.py:38


# --------------------------------------------------------------------------- hash RNG
def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _key(name: str, seed: int) -> np.uint64:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    k = np.array([(h << 32) ^ (seed & 0xFFFFFFFF) ^ 0xA5A5A5A5], dtype=np.uint64)
    return _splitmix64(_splitmix64(k))[0]


def hash_uniform(name: str, shape, seed: int = 0) -> np.ndarray:
    """float64 uniform [0, 1) of `shape`, a pure function of (name, seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(ctr * np.uint64(0x2545F4914F6CDD1D) + _key(name, seed))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return u.reshape(shape)


def hash_normal(name: str, shape, seed: int = 0) -> np.ndarray:
    u1 = hash_uniform(name + "#1", shape, seed)
    u2 = hash_uniform(name + "#2", shape, seed)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def hash_randint(name: str, lo: int, hi: int, shape, seed: int = 0) -> np.ndarray:
    return (lo + np.floor(hash_uniform(name, shape, seed) * (hi - lo))).astype(np.int64)


# --------------------------------------------------------------------------- checkpoint schema
SA1_MLPS = [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
SA1_RADII = [0.1, 0.2, 0.4]
SA1_NSAMPLE = [32, 64, 128]
SA1_NPOINT = 512
SA2_MLPS = [[128, 128, 256], [128, 196, 256]]
SA2_RADII = [0.4, 0.8]
SA2_NSAMPLE = [64, 128]
SA2_NPOINT = 128
SA3_MLP = [256, 512, 1024]
FP3_MLP = [256, 256]
FP2_MLP = [256, 128]
FP1_MLP = [128, 128, 256]
MANO_SA1_MLPS = [[128, 128, 256], [128, 196, 256]]
MANO_SA1_RADII = [0.4, 0.8]
MANO_SA1_NSAMPLE = [64, 128]
MANO_SA1_NPOINT = 128
MANO_SA2_MLP = [256, 512]
N_CLASSES = 4
N_MANO_OUT = 3 + MANO_CMPS + 10 + 3


def checkpoint_schema(in_channels: int = 4, n_pose: int = MANO_CMPS) -> "OrderedDict[str, tuple]":
    """name -> (shape, kind) for all 342 state_dict entries, in module order.

    kind in {conv_w, conv_b, bn_w, bn_b, bn_mean, bn_var, bn_count}.
    `in_channels` is C of the [B, C, N] input (4 with ERPC=0, 5 with ERPC=1).
    """
    sch: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(prefix, o, i, tail):
        sch[prefix + ".weight"] = ((o, i) + tail, "conv_w")
        sch[prefix + ".bias"] = ((o,), "conv_b")

    def bn(prefix, c):
        sch[prefix + ".weight"] = ((c,), "bn_w")
        sch[prefix + ".bias"] = ((c,), "bn_b")
        sch[prefix + ".running_mean"] = ((c,), "bn_mean")
        sch[prefix + ".running_var"] = ((c,), "bn_var")
        sch[prefix + ".num_batches_tracked"] = ((), "bn_count")

    def msg(prefix, fan_in, mlps):
        for i, mlp in enumerate(mlps):
            last = fan_in
            for j, o in enumerate(mlp):
                conv(f"{prefix}.conv_blocks.{i}.{j}", o, last, (1, 1))
                last = o
        for i, mlp in enumerate(mlps):
            for j, o in enumerate(mlp):
                bn(f"{prefix}.bn_blocks.{i}.{j}", o)

    def stack(prefix, fan_in, mlp, tail):
        last = fan_in
        for k, o in enumerate(mlp):
            conv(f"{prefix}.mlp_convs.{k}", o, last, tail)
            last = o
        for k, o in enumerate(mlp):
            bn(f"{prefix}.mlp_bns.{k}", o)

    msg("sa1", in_channels + 3, SA1_MLPS)
    msg("sa2", 320 + 3, SA2_MLPS)
    stack("sa3", 512 + 3, SA3_MLP, (1, 1))
    stack("fp3", 1536, FP3_MLP, (1,))
    stack("fp2", 576, FP2_MLP, (1,))
    stack("fp1", 128, FP1_MLP, (1,))
    conv("classifier.0", 256, 256, (1,))
    bn("classifier.2", 256)
    conv("classifier.4", N_CLASSES, 256, (1,))
    for side in ("left", "right"):
        p = f"{side}_mano_regressor"
        msg(p + ".sa1", 4 + 3, MANO_SA1_MLPS)
        stack(p + ".sa2", 512 + 3, MANO_SA2_MLP, (1, 1))
        conv(p + ".mano_regressor.0", 1024, 512, ())
        bn(p + ".mano_regressor.2", 1024)
        conv(p + ".mano_regressor.4", 3 + n_pose + 10 + 3, 1024, ())
    for side in ("left", "right"):
        p = f"{side}_query_conv"
        conv(p + ".0", 256, 256, (3,))
        bn(p + ".2", 256)
        conv(p + ".4", 256, 256, (3,))
        bn(p + ".5", 256)
    return sch


def synth_state_dict(in_channels: int = 4, seed: int = 0, n_pose: int = MANO_CMPS) -> "OrderedDict[str, torch.Tensor]":
    """Random-init checkpoint with non-trivial BN running stats:
    conv/linear ~ U(+-1/sqrt(fan_in)), gamma in [.5,1.5], beta, mean ~ N(0,.1), var in [.5,1.5]."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (shape, kind) in checkpoint_schema(in_channels, n_pose).items():
        if kind == "conv_w":
            fan_in = int(np.prod(shape[1:]))
            v = (hash_uniform(name, shape, seed) * 2 - 1) / np.sqrt(fan_in)
        elif kind == "conv_b":
            # fan-in of the matching weight is not known here; a fixed small range is enough
            v = (hash_uniform(name, shape, seed) * 2 - 1) * 0.05
        elif kind == "bn_w":
            v = 0.5 + hash_uniform(name, shape, seed)
        elif kind in ("bn_b", "bn_mean"):
            v = 0.1 * hash_normal(name, shape, seed)
        elif kind == "bn_var":
            v = 0.5 + hash_uniform(name, shape, seed)
        elif kind == "bn_count":
            sd[name] = torch.tensor(1000, dtype=torch.int64)
            continue
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return sd


# --------------------------------------------------------------------------- event clouds
def synth_cloud_uniform(B: int, C: int, N: int, seed: int = 0) -> torch.Tensor:
    """Distribution U: every channel iid uniform [-1, 1).  Returns float32 [B, C, N]."""
    u = hash_uniform("cloudU", (B, C, N), seed) * 2 - 1
    return torch.from_numpy(u.astype(np.float32))


def synth_cloud_events(B: int, C: int, N: int, seed: int = 0) -> torch.Tensor:
    """Distribution E (event-like): unique pixels from two Gaussian blobs + uniform noise on a
    346x260 sensor, per-pixel mean timestamp and polarity counts, resampled WITH replacement
    to N points, x/y/t normalised to [-1, 1].  C=5: [x, y, t, pos_cnt, neg_cnt]; C=4: [x, y, t, polarity]."""
    out = np.zeros((B, C, N), dtype=np.float32)
    W, H = OUTPUT_WIDTH, OUTPUT_HEIGHT
    for b in range(B):
        tag = f"cloudE/{b}"
        M = int(600 + np.floor(hash_uniform(tag + "/M", (), seed) * 3400))
        n_raw = 3 * M
        centres = hash_uniform(tag + "/ctr", (2, 2), seed) * np.array([W * 0.6, H * 0.6]) + np.array([W * 0.2, H * 0.2])
        which = (hash_uniform(tag + "/which", (n_raw,), seed) < 0.5).astype(np.int64)
        g = hash_normal(tag + "/g", (n_raw, 2), seed) * 30.0
        xy = centres[which] + g
        noise = hash_uniform(tag + "/noise", (n_raw,), seed) < (1.0 / 32.0)
        uxy = hash_uniform(tag + "/uxy", (n_raw, 2), seed) * np.array([W, H])
        xy = np.where(noise[:, None], uxy, xy)
        x = np.clip(np.floor(xy[:, 0]), 0, W - 1).astype(np.int64)
        y = np.clip(np.floor(xy[:, 1]), 0, H - 1).astype(np.int64)
        pix = np.unique(y * W + x)[:M]
        order = np.argsort(hash_uniform(tag + "/perm", (pix.size,), seed), kind="stable")
        pix = pix[order]
        yi, xi = pix // W, pix % W
        # per-pixel mean timestamp: ramp across the sensor + jitter
        t = (xi / W) * 0.7 + (yi / H) * 0.2 + 0.1 * hash_uniform(tag + "/t", (pix.size,), seed)
        pos = np.floor(hash_uniform(tag + "/p", (pix.size,), seed) * 8)
        neg = np.floor(hash_uniform(tag + "/n", (pix.size,), seed) * 8)
        both0 = (pos + neg) == 0
        pos = np.where(both0, 1.0, pos)
        sel = hash_randint(tag + "/sel", 0, pix.size, (N,), seed)
        ev = np.stack([xi[sel].astype(np.float32), yi[sel].astype(np.float32),
                       t[sel].astype(np.float32), pos[sel].astype(np.float32), neg[sel].astype(np.float32)], 0)
        ev[0] = 2 * (ev[0] / np.float32(W)) - 1
        ev[1] = 2 * (ev[1] / np.float32(H)) - 1
        tmin, tmax = ev[2].min(), ev[2].max()
        ev[2] = 2 * ((ev[2] - tmin) / (tmax - tmin)) - 1
        if C == 5:
            out[b] = ev
        elif C == 4:
            out[b, :3] = ev[:3]
            out[b, 3] = (ev[3] >= ev[4]).astype(np.float32)
        else:
            raise ValueError("C must be 4 or 5")
    return torch.from_numpy(out)


def synth_cloud_lattice(B: int, C: int, N: int, seed: int = 0) -> torch.Tensor:
    """Distribution L (stress case for the discrete selections): N distinct points of the cubic lattice 0.02 * Z^3 inside
    [-0.24, 0.24]^3.  Lattice vectors of squared length 25, 100, 400, 1600 (x 0.02^2) put many point pairs at EXACTLY the
    ball-query radii 0.1, 0.2, 0.4, 0.8 -- where the fp32 rounding of the matmul-form distance decides `d > r*r`
    (pointnet2_utils.py:100-102) -- and give every query point several equidistant 3-NN candidates (:296-298) and FPS ties
    (:83).  Feature channels as in distribution U."""
    if N > 25 ** 3:
        raise ValueError("lattice cloud holds at most 15625 distinct points")
    out = np.zeros((B, C, N), dtype=np.float32)
    for b in range(B):
        order = np.argsort(hash_uniform(f"cloudL/{b}", (25 ** 3,), seed), kind="stable")[:N]
        k = np.stack([order // 625, (order // 25) % 25, order % 25], 0).astype(np.float64) - 12.0
        out[b, :3] = (k * 0.02).astype(np.float32)
        out[b, 3:] = (hash_uniform(f"cloudL/{b}/f", (C - 3, N), seed) * 2 - 1).astype(np.float32)
    return torch.from_numpy(out)


def synth_cloud(kind: str, B: int, C: int, N: int, seed: int = 0) -> torch.Tensor:
    if kind == "U":
        return synth_cloud_uniform(B, C, N, seed)
    if kind == "E":
        return synth_cloud_events(B, C, N, seed)
    if kind == "L":
        return synth_cloud_lattice(B, C, N, seed)
    raise ValueError(kind)


def fps_inits(B: int, N: int, seed: int = 0) -> list:
    """The four FPS start-index vectors of one forward, drawn the way the reference's
    forward consumes the global CPU RNG (pointnet2_utils.py:75): randint(0,N) enc.sa1,
    randint(0,512) enc.sa2, randint(0,N) left.sa1, randint(0,N) right.sa1."""
    g = torch.Generator().manual_seed(1000 + seed)
    return [torch.randint(0, hi, (B,), dtype=torch.long, generator=g) for hi in (N, SA1_NPOINT, N, N)]


# --------------------------------------------------------------------------- MANO-shaped assets
MANO_NV = 778
MANO_NJ = 16
MANO_NF = 1538
MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
MANO_TIPS = {"right": [745, 317, 444, 556, 673], "left": [745, 317, 445, 556, 673]}
MANO_JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def synth_mano_assets(side: str, seed: int = 0) -> dict:
    """A seeded, MANO-shaped (not MANO) asset: the licensed MANO_{LEFT,RIGHT}.pkl files are
    absent.  float64 numpy arrays with manopth's buffer shapes; J_regressor and
    skinning-weight rows are non-negative and sum to 1."""
    tag = f"mano/{side}"
    s = 1.0 if side == "right" else -1.0
    box = np.array([0.10, 0.09, 0.03])
    v = (hash_uniform(tag + "/v", (MANO_NV, 3), seed) - 0.5) * box
    v[:, 0] = v[:, 0] * s + 0.09 * s
    shapedirs = hash_normal(tag + "/sd", (MANO_NV, 3, 10), seed) * 3e-3
    posedirs = hash_normal(tag + "/pd", (MANO_NV, 3, 135), seed) * 1e-3
    jr = hash_uniform(tag + "/jr", (MANO_NJ, MANO_NV), seed)
    jr = np.where(jr > 0.97, jr, 0.0)
    jr[:, 0] += 1e-3
    jr /= jr.sum(1, keepdims=True)
    w = hash_uniform(tag + "/w", (MANO_NV, MANO_NJ), seed)
    thresh = np.sort(w, axis=1)[:, -3][:, None]
    w = np.where(w >= thresh, w, 0.0)
    w /= w.sum(1, keepdims=True)
    comps = hash_normal(tag + "/comps", (45, 45), seed) * 0.3
    mean = hash_normal(tag + "/mean", (45,), seed) * 0.2
    faces = hash_randint(tag + "/f", 0, MANO_NV, (MANO_NF, 3), seed)
    return {
        "side": side,
        "v_template": v, "shapedirs": shapedirs, "posedirs": posedirs,
        "J_regressor": jr, "weights": w,
        "hands_components": comps, "hands_mean": mean,
        "faces": faces.astype(np.int64),
        "parents": list(MANO_PARENTS),
    }


def synth_mano_surface_assets(side: str, seed: int = 0) -> dict:
    """A MANO-shaped asset whose template IS a surface: a mitten (half ellipsoid, open at the wrist like MANO's mesh) of 37 rings x 21
    segments + a tip vertex = 778 vertices, 1512 + 21 triangles + 5 repeated ones = 1538 faces, smooth low-frequency blend
    shapes, 16 joints inside the volume with a J_regressor of local vertex averages and smooth (distance-based, 3 joints per
    vertex) skinning weights.  synth_mano_assets() above draws vertices and faces independently -- triangle soup in which every
    mesh intersects itself ~24 000 times, the worst case for the collision search; this one deforms like a hand mesh does (few
    or no self-intersections), which is what the two-hand collision term of BASELINE config 5 meets in practice.  Same array
    shapes and keys; used by `bench.py --collision` (both geometries are reported), not by the parity fixtures."""
    tag = f"mano_surface/{side}"
    s = 1.0 if side == "right" else -1.0
    R, S = 37, 21
    t = (np.arange(R) / R)[:, None]                                   # 0 (wrist) .. 0.973
    phi = (2 * np.pi * np.arange(S) / S)[None, :]
    prof = np.sqrt(1.0 - 0.95 * t ** 2)
    vx, vy, vz = 0.040 * prof * np.cos(phi), 0.09 * t + 0 * phi, 0.012 * prof * np.sin(phi)
    v = np.concatenate([np.stack([vx, vy, vz], -1).reshape(-1, 3), np.array([[0.0, 0.0915, 0.0]])], 0)      # 777 + tip
    assert v.shape[0] == MANO_NV
    faces = []
    for i in range(R - 1):
        for j in range(S):
            a, b, c, d = i * S + j, i * S + (j + 1) % S, (i + 1) * S + (j + 1) % S, (i + 1) * S + j
            faces += [(a, b, c), (a, c, d)]
    tip = R * S
    faces += [((R - 1) * S + j, (R - 1) * S + (j + 1) % S, tip) for j in range(S)]
    faces += faces[:MANO_NF - len(faces)]                             # 5 repeats (share all vertices with their originals: never a pair)
    faces = np.asarray(faces, dtype=np.int64)
    assert faces.shape == (MANO_NF, 3)
    # smooth fields over the surface: a few low-frequency modes of (t, phi) per direction
    tt = np.concatenate([np.repeat(t[:, 0], S), [1.0]])
    pp = np.concatenate([np.tile(phi[0], R), [0.0]])

    def smooth(name, k, amp):
        c = hash_normal(tag + "/" + name, (k, 3, 4), seed)
        ph = hash_uniform(tag + "/" + name + "/ph", (k, 3, 4), seed) * 2 * np.pi
        basis = np.stack([np.sin(np.pi * tt), np.sin(2 * np.pi * tt), np.cos(pp) * np.sin(np.pi * tt), np.sin(pp) * np.sin(np.pi * tt)], 0)   # [4, V]
        f = np.einsum("kcm,kcmv->vck", c, np.cos(ph)[..., None] * basis[None, None])
        return amp * f / 2.0
    shapedirs = smooth("sd", 10, 1.0e-3)
    posedirs = smooth("pd", 135, 2.0e-4)
    # joints: the root at the wrist, five chains of three along the mitten
    targets = [np.array([0.0, 0.005, 0.0])]
    for f_, x0 in enumerate(np.linspace(-0.026, 0.026, 5)):
        for y0 in (0.030, 0.052, 0.072):
            targets.append(np.array([x0 * (1.0 - 0.5 * (y0 / 0.09) ** 2), y0, 0.0]))
    targets = np.stack(targets)                                       # [16, 3] in MANO's joint order (root, 5 x 3)
    d2 = ((v[None] - targets[:, None]) ** 2).sum(-1)                   # [16, V]
    jr = np.exp(-d2 / 0.012 ** 2) + 1e-12
    jr /= jr.sum(1, keepdims=True)
    J = jr @ v
    w = np.exp(-(((v[:, None] - J[None]) ** 2).sum(-1)) / 0.018 ** 2) + 1e-12
    thresh = np.sort(w, axis=1)[:, -3][:, None]
    w = np.where(w >= thresh, w, 0.0)
    w /= w.sum(1, keepdims=True)
    v = v.copy()
    v[:, 0] = v[:, 0] * s + 0.09 * s
    base = synth_mano_assets(side, seed)
    return {"side": side, "v_template": v, "shapedirs": shapedirs, "posedirs": posedirs, "J_regressor": jr, "weights": w,
            "hands_components": base["hands_components"], "hands_mean": base["hands_mean"], "faces": faces, "parents": list(MANO_PARENTS)}
