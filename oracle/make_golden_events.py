"""Generate tests/golden/events_*.npz from the REAL reference window builder (survey container only).

Imports /root/reference/src/Ev2Hands/dataset/evaluation_stream.py with its unavailable imports (`dv`, `settings`,
`camera`: none is used by the window arithmetic) replaced by stubs, builds an ERPCParser over a synthetic event stream
without running its file-reading constructor, calls the reference's own __getitem__ and asserts that
oracle/event_window_oracle.py reproduces its output bit for bit before writing the fixture.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import event_window_oracle as EW  # noqa: E402

REF = "/root/reference/src/Ev2Hands/dataset/evaluation_stream.py"


def load_reference():
    dv = types.ModuleType("dv")
    dv.AedatFile = object
    settings = types.ModuleType("settings")
    settings.OUTPUT_HEIGHT, settings.OUTPUT_WIDTH = EW.OUTPUT_HEIGHT, EW.OUTPUT_WIDTH
    camera = types.ModuleType("camera")
    camera.undistort = camera.opencv_camera_view_to_screen_space_transform = lambda *a, **k: None
    sys.modules.update({"dv": dv, "settings": settings, "camera": camera})
    spec = importlib.util.spec_from_file_location("ref_evaluation_stream", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    # numpy version skew: evaluation_stream.py:207 indexes with a LIST `n_evn[[..., None]]`, which numpy < 1.23 treated
    # as the tuple `n_evn[..., None]` (the reference pins numpy 1.21) and numpy 2 rejects.  Give the module a numpy proxy
    # whose zeros() returns an ndarray subclass that restores the old interpretation; nothing else changes.
    class Legacy(np.ndarray):
        def __getitem__(self, k):
            if isinstance(k, list) and any(e is Ellipsis or e is None for e in k):
                k = tuple(k)
            return super().__getitem__(k)

    class NumpyProxy:
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def zeros(*a, **k):
            return np.zeros(*a, **k).view(Legacy)

    m.np = NumpyProxy()
    return m


def main():
    ref = load_reference()
    for case, (n_stream, seed, nwin) in enumerate([(12000, 1, 3), (40000, 2, 4)]):
        stream = EW.synth_event_stream(n_stream, seed)
        p = ref.ERPCParser.__new__(ref.ERPCParser)            # skip the pickle / aedat reading constructor
        p.events = np.concatenate([stream, np.zeros((n_stream, 1), dtype=np.int64)], 1)      # 5th column = frame index
        p.joints = np.zeros((1, 2, 21, 3))
        p.camera = {}
        p.e_id, p.n_events = 0, 0
        out = {}
        for w in range(nwin):
            e0 = p.e_id
            np.random.seed(100 + 10 * case + w)
            d = p[0]                                          # reference ERPCParser.__getitem__
            # the raw events the reference consumed for this window (evaluation_stream.py:124-146)
            q = ref.ERPCParser.__new__(ref.ERPCParser)
            q.events, q.e_id, q.n_events = p.events, e0, 0
            raw, _ = q.get_events_by_time()
            np.random.seed(100 + 10 * case + w)
            mine, table, idx = EW.build_window(raw)
            assert torch.equal(mine, d["data"]), f"oracle != reference (case {case} window {w})"
            out[f"raw{w}"] = raw
            out[f"idx{w}"] = idx.astype(np.int32)
            out[f"data{w}"] = d["data"].numpy()
            out[f"table{w}"] = table
            print(f"case {case} window {w}: {raw.shape[0]} events, {table.shape[0]} unique pixels, next e_id {p.e_id}")
        out["nwin"] = np.array(nwin)
        path = os.path.join(ROOT, "tests", "golden", f"events_{case}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
