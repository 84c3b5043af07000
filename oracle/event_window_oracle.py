"""ORACLE -- test infrastructure, not product code.

NumPy restatement of the reference's event-window -> [5, N] tensor builder (the step immediately before the
hot path):
  /root/reference/src/Ev2Hands/dataset/evaluation_stream.py:177-231  (ERPCParser.__getitem__, evaluation)
  /root/reference/src/Ev2Hands/dataset/ev2hands_r.py:108-159          (Ev2HandRDataset, same arithmetic)
  pc_normalize: evaluation_stream.py:13-28 / ev2hands_r.py:21-35

PINNED: oracle/make_golden_events.py imports the real evaluation_stream.py (with `dv`, `settings`, `camera`
stubbed, none of which the window builder touches), runs ERPCParser.__getitem__ on synthetic event streams and
asserts bit-equality with this restatement before writing tests/golden/events_*.npz.
"""
from __future__ import annotations

import numpy as np
import torch

OUTPUT_WIDTH, OUTPUT_HEIGHT = 346, 260      # /root/reference/src/settings.py:21-22


def accumulate_pixels(events: np.ndarray, width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT):
    """evaluation_stream.py:187-208.  events [E,4] float64 rows (x, y, t_ms, polarity) in stream order.
    Returns (xi, yi, t_avg, p_evn, n_evn) of the pixels that received at least one event, in row-major pixel order;
    the per-pixel timestamp sum is accumulated in event order by np.add.at (each step adds in float64 and rounds the
    running float32 sum)."""
    ev = np.array(events, dtype=np.float64, copy=True)
    ev[:, 2] -= ev[0, 2]
    grid = np.zeros((height, width, 3), dtype=np.float32)
    cnt = np.zeros((height, width), dtype=np.float32)
    x, y, t, p = ev.T
    x, y = x.astype(np.int32), y.astype(np.int32)
    np.add.at(grid, (y, x, 0), t)
    np.add.at(grid, (y, x, 1), p == 1)
    np.add.at(grid, (y, x, 2), p != 1)
    np.add.at(cnt, (y, x), 1)
    yi, xi = np.nonzero(cnt)
    t_avg = grid[yi, xi, 0] / cnt[yi, xi]
    return xi, yi, t_avg, grid[yi, xi, 1], grid[yi, xi, 2]


def normalize_points(pc: torch.Tensor, width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT) -> torch.Tensor:
    """pc_normalize (evaluation_stream.py:13-28) on a float32 [n,3] tensor (x, y, t), in place like the reference."""
    pc[:, 0] /= width
    pc[:, 1] /= height
    pc[:, :2] = 2 * pc[:, :2] - 1
    ts = pc[:, 2:]
    t_max = ts.max(0).values
    t_min = ts.min(0).values
    pc[:, 2:] = (2 * ((ts - t_min) / (t_max - t_min))) - 1
    return pc


def build_window(events: np.ndarray, sample_idx: np.ndarray | None = None, n_events: int = 2048,
                 width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT):
    """evaluation_stream.py:187-225: [E,4] raw events -> float32 [5, n_events] = (x, y, t, pos_cnt, neg_cnt).
    `sample_idx` = the resampling-with-replacement indices; None draws them with np.random.choice like the reference.
    Returns (tensor [5,n], unique-pixel table [M,5] float64, sample_idx)."""
    xi, yi, t_avg, p_evn, n_evn = accumulate_pixels(events, width, height)
    table = np.hstack([xi[..., None], yi[..., None], t_avg[..., None], p_evn[..., None], n_evn[..., None]])
    if sample_idx is None:
        sample_idx = np.random.choice(table.shape[0], n_events)
    ev = torch.tensor(table[sample_idx], dtype=torch.float32)
    ev[:, :3] = normalize_points(ev[:, :3], width, height)
    return ev.permute(1, 0).contiguous(), table, sample_idx


def synth_event_stream(n: int, seed: int, width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT) -> np.ndarray:
    """Seeded synthetic raw event stream, int64 [n,4] rows (x, y, t_us, polarity) like the reference's pickled
    streams (evaluation_stream.py:36-42), time-ordered, two moving blobs + uniform noise."""
    from ev2hands_amd.synth import hash_normal, hash_uniform
    tag = f"evstream/{seed}"
    t = np.cumsum(hash_uniform(tag + "/dt", (n,), seed) * 1.6) + 1_000_000.0        # microseconds, ~1.25 events/us
    which = hash_uniform(tag + "/w", (n,), seed) < 0.5
    cx = np.where(which, 110.0, 230.0) + 25.0 * np.sin(t * 2e-4)
    cy = np.where(which, 120.0, 140.0) + 20.0 * np.cos(t * 2e-4)
    g = hash_normal(tag + "/g", (n, 2), seed) * 18.0
    noise = hash_uniform(tag + "/n", (n,), seed) < 0.03
    ux = hash_uniform(tag + "/ux", (n,), seed) * width
    uy = hash_uniform(tag + "/uy", (n,), seed) * height
    x = np.clip(np.where(noise, ux, cx + g[:, 0]), 0, width - 1e-3)
    y = np.clip(np.where(noise, uy, cy + g[:, 1]), 0, height - 1e-3)
    p = (hash_uniform(tag + "/p", (n,), seed) < 0.55)
    return np.stack([np.floor(x), np.floor(y), np.floor(t), p], 1).astype(np.int64)


# ------------------------------------------------------------------------------------------------ Ev2Hands-S variant
def build_window_s(rows: np.ndarray, sample_idx: np.ndarray | None = None, sampling: bool = True, n_events: int = 2048,
                   width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT):
    """The synthetic-dataset builder, /root/reference/src/Ev2Hands/dataset/erpc.py:169-249 (Ev2HandSDataset.__getitem__,
    augment off).  rows [n,6] float64 = (x, y, t, p, annotation_index, event_label) of n raw events.  Differences from the
    evaluation builder above: timestamps are accumulated as they are and the per-pixel mean is multiplied by 1e-6 (:191);
    the unique pixels are re-ordered by that mean time with np.argsort (:207) and the first one's time is subtracted (:211);
    the per-EVENT labels are indexed with those per-PIXEL sort indices (:209, kept as it is); sampling=False keeps all pixels
    and pads with resampled ones (:220-227).
    Returns (events float32 [5, N'], labels int64 [N'], table float32 [M,5] in time order, time-ordered labels [M],
    sample_idx) -- N' = n_events, or M when sampling is off and M == n_events.
    np.argsort's default sort is not stable: among pixels with exactly equal mean time the reference's order depends on the
    numpy build; this restatement (and the GPU kernel) take them in pixel order (stable)."""
    x, y, t, p, _, lab = np.asarray(rows).T
    grid = np.zeros((height, width, 3), dtype=np.float32)
    cnt = np.zeros((height, width), dtype=np.float32)
    x, y = x.astype(np.int32), y.astype(np.int32)
    np.add.at(grid, (y, x, 0), t)
    np.add.at(grid, (y, x, 1), p == 1)
    np.add.at(grid, (y, x, 2), p != 1)
    np.add.at(cnt, (y, x), 1)
    yi, xi = np.nonzero(cnt)
    t_avg = (grid[yi, xi, 0] / cnt[yi, xi]) * 1e-6
    ev = np.hstack([xi[:, None], yi[:, None], t_avg[:, None], grid[yi, xi, 1][:, None], grid[yi, xi, 2][:, None]]).astype(np.float32)
    lab = lab.astype(np.int32)
    order = np.argsort(ev[:, 2], kind="stable")
    ev = ev[order]
    lab = lab[order]                       # erpc.py:209: per-event labels indexed by per-pixel sort positions, as the reference does
    ev[:, 2] -= ev[0, 2]
    table, table_lab = ev.copy(), lab.copy()
    M = ev.shape[0]
    if sampling:
        if sample_idx is None:
            sample_idx = np.random.choice(M, n_events)
        ev, lab = ev[sample_idx], lab[sample_idx]
    elif M < n_events:
        if sample_idx is None:
            sample_idx = np.random.choice(M, n_events - M)
        ev, lab = np.concatenate([ev, ev[sample_idx]], 0), np.concatenate([lab, lab[sample_idx]], 0)
    evt = torch.tensor(ev, dtype=torch.float32)
    evt[:, :3] = normalize_points(evt[:, :3], width, height)
    return evt.permute(1, 0).contiguous(), torch.tensor(lab, dtype=torch.long), table, table_lab, sample_idx


def synth_s_rows(n: int, seed: int, start: int = 0) -> np.ndarray:
    """n consecutive rows of a synthetic Ev2Hands-S event table: float64 (x, y, t_ns, p, annotation_index, label).
    Timestamps are strictly increasing (1-3 us apart, nanosecond jitter) and counted from the start of the recording: unique raw times and small
    magnitudes keep the float32 per-pixel mean times free of exact ties, whose order the reference leaves to np.argsort (real
    tables have several events per microsecond and t ~ 1e9 ns, where such ties abound)."""
    from ev2hands_amd.synth import hash_uniform
    s = synth_event_stream(start + n, seed)
    t_us = np.cumsum(1.0 + np.floor(hash_uniform(f"evS/{seed}/dt", (start + n,), seed) * 3.0))
    t_ns = t_us * 1000.0 + np.floor(hash_uniform(f"evS/{seed}/ns", (start + n,), seed) * 1000.0)      # nanosecond jitter
    lab = np.floor(hash_uniform(f"evS/{seed}/lab", (start + n,), seed) * 4)
    rows = np.stack([s[:, 0], s[:, 1], t_ns, s[:, 3], np.zeros(start + n), lab], 1).astype(np.float64)
    return rows[start:]
