"""Generate tests/golden/events_s_*.npz from the REAL reference builder of the synthetic dataset (survey container only).

Imports /root/reference/src/Ev2Hands/dataset/erpc.py as part of a synthetic `refdataset` package (its relative import of
augmentations.py resolves; `h5py` and `settings` are stubbed -- the window arithmetic touches neither), builds an
Ev2HandSDataset without running its file-opening constructor, calls the reference's own __getitem__ (augment off, sampling on
and off) on synthetic event tables and asserts that oracle/event_window_oracle.py: build_window_s reproduces its output bit for
bit before writing the fixture.  Tables whose per-pixel mean times tie exactly are rejected (np.argsort's order among ties is
not defined by the reference).
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

REF_DIR = "/root/reference/src/Ev2Hands/dataset"


def load_reference():
    settings = types.ModuleType("settings")
    settings.OUTPUT_HEIGHT, settings.OUTPUT_WIDTH, settings.LNES_WINDOW_MS = EW.OUTPUT_HEIGHT, EW.OUTPUT_WIDTH, 5
    sys.modules["settings"] = settings
    sys.modules["h5py"] = types.ModuleType("h5py")
    pkg = types.ModuleType("refdataset")
    pkg.__path__ = [REF_DIR]
    sys.modules["refdataset"] = pkg

    def load(name):
        spec = importlib.util.spec_from_file_location(f"refdataset.{name}", f"{REF_DIR}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        return m

    load("augmentations")
    return load("erpc")


def main():
    ref = load_reference()
    hand = {"global_orient": np.zeros(3), "hand_pose": np.zeros(6), "shape": np.zeros(10), "trans": np.zeros(3)}
    for case, (n_table, seed, starts, sampling) in enumerate([(9000, 11, (0, 1500, 5000), True), (9000, 12, (100, 3000), False)]):
        rows = EW.synth_s_rows(n_table, seed)
        ds = ref.Ev2HandSDataset.__new__(ref.Ev2HandSDataset)       # skip the h5 / pickle reading constructor
        ds.dataset, ds.annotations = rows, {0: {"left": dict(hand), "right": dict(hand)}}
        ds.augment, ds.sampling, ds.demo, ds.nSamples = False, sampling, False, n_table
        out = {"sampling": np.array(int(sampling)), "nwin": np.array(len(starts))}
        for w, st in enumerate(starts):
            np.random.seed(300 + 10 * case + w)
            d = ds[st]                                              # reference Ev2HandSDataset.__getitem__
            np.random.seed(300 + 10 * case + w)
            ev, lab, table, table_lab, idx = EW.build_window_s(rows[st:st + 2048], sampling=sampling)
            assert len(np.unique(table[:, 2])) == table.shape[0], "tied mean times: the reference's order is undefined"
            assert torch.equal(ev, d["events"]) and torch.equal(lab, d["class_logits"]), f"oracle != reference (case {case} window {w})"
            out[f"rows{w}"] = rows[st:st + 2048]
            out[f"idx{w}"] = np.asarray(idx, dtype=np.int32)
            out[f"events{w}"] = d["events"].numpy()
            out[f"labels{w}"] = d["class_logits"].numpy()
            out[f"table{w}"] = table
            out[f"table_lab{w}"] = table_lab
            print(f"case {case} window {w}: start {st}, {table.shape[0]} unique pixels, sampling {sampling}, output {tuple(d['events'].shape)}")
        path = os.path.join(ROOT, "tests", "golden", f"events_s_{case}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
