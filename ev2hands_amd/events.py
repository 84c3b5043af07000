"""Host side of the event-window builder: ragged raw event windows -> the hot path's [B, 5, N] input.

Mirrors what the reference's dataset classes do per item on the CPU
(/root/reference/src/Ev2Hands/dataset/evaluation_stream.py:177-231, dataset/ev2hands_r.py:108-159), batched on the GPU
through ev2h_event_window_build / ev2h_event_window_sample.  The resampling indices are drawn on the host with
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

    def accumulate(self, windows):
        """windows: list of [E_i, 4] float64 arrays (x, y, t_ms, polarity).  Returns (table [B,cap,8] f32, counts [B] i32), on device."""
        B = len(windows)
        offs = np.zeros(B + 1, dtype=np.int32)
        offs[1:] = np.cumsum([w.shape[0] for w in windows])
        ev = torch.from_numpy(np.ascontiguousarray(np.concatenate(windows, 0), dtype=np.float64)).to(self.device)
        off = torch.from_numpy(offs).to(self.device)
        table = torch.empty(B, self.cap, 8, device=self.device, dtype=torch.float32)
        counts = torch.empty(B, device=self.device, dtype=torch.int32)
        L = _lib.lib()
        _lib.check(L.ev2h_event_window_build(ev.data_ptr(), off.data_ptr(), B, self.w, self.h, self.cap, counts.data_ptr(),
                                             table.data_ptr(), _lib.stream_handle()), "ev2h_event_window_build")
        return table, counts

    def sample(self, table, counts, sample_idx=None):
        """-> float32 [B, 5, N].  sample_idx [B, N] (any integer type); None draws np.random.choice(M_b, N) per window in
        batch order from numpy's global RNG (evaluation_stream.py:209)."""
        B = table.shape[0]
        if sample_idx is None:
            ms = counts.cpu().numpy()
            if (ms <= 0).any():
                raise RuntimeError("an event window is empty or exceeds 32768 events")
            sample_idx = np.stack([np.random.choice(int(m), self.n) for m in ms])
        idx = torch.as_tensor(np.asarray(sample_idx), dtype=torch.int32).to(self.device).contiguous()
        out = torch.empty(B, 5, self.n, device=self.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.ev2h_event_window_sample(table.data_ptr(), counts.data_ptr(), self.cap, idx.data_ptr(), B, self.n, self.w,
                                              self.h, out.data_ptr(), _lib.stream_handle()), "ev2h_event_window_sample")
        return out

    def __call__(self, windows, sample_idx=None):
        table, counts = self.accumulate(windows)
        return self.sample(table, counts, sample_idx)
