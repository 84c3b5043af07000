"""Randomised check of the event-window builder (ev2h_event_window_build / _sample) against its oracle, which is pinned to the
reference's own ERPCParser.__getitem__: ragged batches of windows with 1 ... 32768 events, uniform / clustered / single-row / edge
pixels, tied and huge timestamps, polarity values other than {0, 1}.  Bit-exact tables and normalised tensors (NaN positions
included: a window whose pixels share one mean time normalises to 0/0 in the reference as well).
usage: python tests/fuzz_events.py [nbatches] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ev2hands_amd.events import EventWindowBuilder  # noqa: E402
from oracle import event_window_oracle as EW  # noqa: E402

W, H = 346, 260


def window(rng, E):
    kind = rng.choice(["uniform", "cluster", "row", "edge", "one"])
    if kind == "uniform":
        x, y = rng.integers(0, W, E), rng.integers(0, H, E)
    elif kind == "cluster":
        cx, cy = rng.integers(0, W), rng.integers(0, H)
        x = np.clip(cx + rng.normal(0, 3, E).astype(int), 0, W - 1)
        y = np.clip(cy + rng.normal(0, 3, E).astype(int), 0, H - 1)
    elif kind == "row":
        x, y = rng.integers(0, W, E), np.full(E, rng.integers(0, H))
    elif kind == "edge":
        x = rng.choice([0, W - 1], E)
        y = rng.choice([0, H - 1], E)
    else:
        x, y = np.full(E, rng.integers(0, W)), np.full(E, rng.integers(0, H))
    base = float(rng.choice([0.0, 1e3, 1e9]))
    dt = rng.choice([0.0, 1e-3, 0.37, 5.0]) if rng.random() < 0.3 else rng.random() * 0.1
    t = base + np.cumsum(np.where(rng.random(E) < 0.3, 0.0, rng.random(E) * dt + 0.0))      # non-decreasing, many ties
    p = rng.choice([0, 1, 1, 0, -1, 2], E) if rng.random() < 0.3 else rng.integers(0, 2, E)
    return np.stack([x, y, t, p], 1).astype(np.float64), kind


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(nb):
        B = int(rng.integers(1, 7))
        sizes = [int(rng.choice([1, 2, 3, 17, 300, 2500, 2500, 9000, 32768])) for _ in range(B)]
        wins, kinds = zip(*[window(rng, E) for E in sizes])
        n = int(rng.choice([128, 512, 2048]))
        bld = EventWindowBuilder("cuda:0", n_events=n)
        table, counts = bld.accumulate(list(wins))
        msgs = []
        Ms = []
        for w, raw in enumerate(wins):
            xi, yi, t_avg, p_evn, n_evn = EW.accumulate_pixels(raw)
            M = xi.shape[0]
            Ms.append(M)
            if int(counts[w]) != M:
                msgs.append(f"window {w}: count {int(counts[w])} != {M}")
                continue
            got = table[w, :M, :5].cpu().numpy()
            ref = np.stack([xi, yi, t_avg, p_evn, n_evn], 1).astype(np.float32)
            if not np.array_equal(got, ref):
                msgs.append(f"window {w} ({kinds[w]}, E={sizes[w]}): table differs in {int((got != ref).sum())} entries")
        if not msgs:
            idx = np.stack([rng.integers(0, M, n) for M in Ms])
            out = bld.sample(table, counts, idx).cpu().numpy()
            for w, raw in enumerate(wins):
                with np.errstate(all="ignore"):
                    ref, _, _ = EW.build_window(raw, idx[w], n_events=n)
                if not np.array_equal(out[w], ref.numpy(), equal_nan=True):
                    msgs.append(f"window {w} ({kinds[w]}, E={sizes[w]}, M={Ms[w]}): tensor differs")
        print(f"batch {it:3d}: sizes {sizes} kinds {list(kinds)} n={n}  {'OK' if not msgs else 'FAIL ' + '; '.join(msgs)}", flush=True)
        bad += bool(msgs)
    print(f"{nb} batches, {bad} with violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
