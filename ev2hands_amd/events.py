"""Host side of the event-window builder: ragged raw event windows -> the hot path's [B, 5, N] input.

Mirrors what the reference's dataset classes do per item on the CPU
(/root/reference/src/Ev2Hands/dataset/evaluation_stream.py:177-231, dataset/ev2hands_r.py:108-159; the synthetic-dataset
variant dataset/erpc.py:169-249 is EventWindowBuilderS), batched on the GPU
through ev2h_event_window_build / ev2h_event_window_timesort / ev2h_event_window_sample.  The resampling indices are drawn on the host with
np.random.choice(M, N) per window, like the reference, which needs the unique-pixel counts M back from the device (one
small copy); pass `sample_idx` to avoid that synchronisation.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

OUTPUT_WIDTH, OUTPUT_HEIGHT = 346, 260       # /root/reference/src/settings.py:21-22


class EventWindowBuilder:
    def __init__(self, device, n_events: int = 2048, width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT, cap: int = 32768):
        self.device = torch.device(device)
        self.n, self.w, self.h, self.cap = n_events, width, height, cap
        self.raw_time = 0            # evaluation builders subtract the window's first timestamp before accumulating

    def accumulate(self, windows):
        """windows: list of [E_i, >=4] float64 arrays (x, y, t, polarity, ...).  Returns (table [B,cap,8] f32, counts [B] i32), on device."""
        B = len(windows)
        offs = np.zeros(B + 1, dtype=np.int32)
        offs[1:] = np.cumsum([w.shape[0] for w in windows])
        ev = torch.from_numpy(np.ascontiguousarray(np.concatenate(windows, 0), dtype=np.float64)).to(self.device)
        off = torch.from_numpy(offs).to(self.device)
        table = torch.empty(B, self.cap, 8, device=self.device, dtype=torch.float32)
        counts = torch.empty(B, device=self.device, dtype=torch.int32)
        L = _lib.lib()
        _lib.check(L.ev2h_event_window_build(ev.data_ptr(), ev.shape[1], off.data_ptr(), B, self.w, self.h, self.cap, self.raw_time,
                                             counts.data_ptr(), table.data_ptr(), _lib.stream_handle()), "ev2h_event_window_build")
        self._last = (ev, off)
        return table, counts

    def sample(self, table, counts, sample_idx=None, labels=None):
        """-> float32 [B, 5, N] (and int64 [B, N] labels when the per-pixel `labels` [B, cap] int32 are given).  sample_idx [B, N]
        (any integer type); None draws np.random.choice(M_b, N) per window in batch order from numpy's global RNG
        (evaluation_stream.py:209)."""
        B = table.shape[0]
        if sample_idx is None:
            ms = counts.cpu().numpy()
            if (ms <= 0).any():
                raise RuntimeError("an event window is empty or exceeds 32768 events")
            sample_idx = np.stack([np.random.choice(int(m), self.n) for m in ms])
        idx = torch.as_tensor(np.asarray(sample_idx), dtype=torch.int32).to(self.device).contiguous()
        n = idx.shape[1]
        out = torch.empty(B, 5, n, device=self.device, dtype=torch.float32)
        lab = torch.empty(B, n, device=self.device, dtype=torch.int64) if labels is not None else None
        L = _lib.lib()
        _lib.check(L.ev2h_event_window_sample(table.data_ptr(), counts.data_ptr(), self.cap, idx.data_ptr(), B, n, self.w,
                                              self.h, out.data_ptr(), _lib.ptr(labels), _lib.ptr(lab), _lib.stream_handle()),
                   "ev2h_event_window_sample")
        return out if labels is None else (out, lab)

    def __call__(self, windows, sample_idx=None):
        table, counts = self.accumulate(windows)
        return self.sample(table, counts, sample_idx)


class EventWindowBuilderS(EventWindowBuilder):
    """The synthetic-dataset (Ev2Hands-S) item builder, /root/reference/src/Ev2Hands/dataset/erpc.py:169-249 with augment off:
    windows are [n, 6] float64 tables (x, y, t_ns, p, annotation_index, event_label).  Timestamps are accumulated as they are,
    the per-pixel means are scaled by 1e-6, the unique pixels are ordered by mean time (first one's time subtracted) and the
    labels are gathered the way erpc.py:209 does.  `sampling=False` keeps all M pixels and pads with N - M resampled ones
    (:220-227).  Returns {'events': [B,5,N] float32, 'class_logits': [B,N] int64} like the dataset item."""

    def __init__(self, device, n_events: int = 2048, width: int = OUTPUT_WIDTH, height: int = OUTPUT_HEIGHT, cap: int = 4096):
        super().__init__(device, n_events, width, height, cap)
        self.raw_time = 1

    def __call__(self, windows, sampling: bool = True, sample_idx=None):
        B = len(windows)
        table, counts = self.accumulate(windows)
        ev, off = self._last
        sorted_t = torch.empty_like(table)
        labels = torch.zeros(B, self.cap, device=self.device, dtype=torch.int32)
        L = _lib.lib()
        _lib.check(L.ev2h_event_window_timesort(table.data_ptr(), counts.data_ptr(), self.cap, ev.data_ptr(), ev.shape[1], 5,
                                                off.data_ptr(), B, sorted_t.data_ptr(), labels.data_ptr(), _lib.stream_handle()),
                   "ev2h_event_window_timesort")
        ms = counts.cpu().numpy()
        if (ms <= 0).any() or (ms > self.cap).any():
            raise RuntimeError("an event window is empty or has more unique pixels than `cap`")
        if sampling:
            idx = sample_idx if sample_idx is not None else np.stack([np.random.choice(int(m), self.n) for m in ms])
        else:
            # erpc.py:220-227: keep all M pixels and append n_events - M resampled ones (n raw events give M <= n pixels)
            if (ms > self.n).any():
                raise RuntimeError("sampling=False needs at most n_events unique pixels per window")
            rows = []
            for b, m in enumerate(int(v) for v in ms):
                if m == self.n:
                    extra = np.zeros(0, dtype=np.int64)
                elif sample_idx is not None:
                    extra = np.asarray(sample_idx[b], dtype=np.int64)
                else:
                    extra = np.random.choice(m, self.n - m)
                rows.append(np.concatenate([np.arange(m, dtype=np.int64), extra]))
            idx = np.stack(rows)
        events, lab = self.sample(sorted_t, counts, idx, labels)
        self.table, self.table_labels = sorted_t, labels
        return {"events": events, "class_logits": lab}
