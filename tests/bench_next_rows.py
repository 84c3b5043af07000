"""Timing of the next-row kernels (event-window builder, mesh collisions) beside their CPU oracles.  Lives under tests/
because it imports oracle/ (test infrastructure):  python tests/bench_next_rows.py [events|collision|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from kbench import timeit  # noqa: E402


def bench_collision(B=256):
    import time
    import numpy as np
    from ev2hands_amd.collision import mesh_collisions
    from oracle import collision_oracle as CO
    v, f = CO.icosphere(3)                       # 642 vertices, 1280 faces per "hand"
    rng = np.random.default_rng(0)
    vl = np.stack([(v * 0.04).astype(np.float32)] * B)
    vr = np.stack([(v * 0.04 + np.array([0.03 + 0.04 * rng.random(), 0.01 * rng.normal(), 0.01 * rng.normal()])).astype(np.float32) for _ in range(B)])
    a, b = torch.from_numpy(vl).cuda(), torch.from_numpy(vr).cuda()
    ms = timeit(lambda: mesh_collisions(a, b, f, f), iters=5)
    t0 = time.time()
    verts, faces = CO.build_triangles(vl[0], vr[0], f, f)
    n = CO.collision_pairs(verts, faces).shape[0]
    cpu = time.time() - t0
    print(f"mesh collisions B={B}, 2x{f.shape[0]} triangles: {ms:8.3f} ms on the GPU = {B / ms * 1e3:9.0f} windows/s; NumPy oracle {cpu * 1e3:7.1f} ms "
          f"per window ({n} pairs in window 0)")


def bench_events(B=256, n_ev=2500):
    import time
    import numpy as np
    from ev2hands_amd.events import EventWindowBuilder
    from oracle import event_window_oracle as EW
    wins = []
    for k in range(8):
        s_ = EW.synth_event_stream(n_ev, 50 + k).astype(np.float64)
        s_[:, 2] *= 1e-3
        wins.append(s_)
    wins = (wins * (B // 8))[:B]
    bld = EventWindowBuilder("cuda:0")
    table, counts = bld.accumulate(wins)
    idx = np.stack([np.random.RandomState(i).randint(0, int(c), 2048) for i, c in enumerate(counts.cpu().numpy())])
    # device-only timing: inputs already resident
    import ctypes as C
    from ev2hands_amd import _lib
    offs = np.zeros(B + 1, dtype=np.int32); offs[1:] = np.cumsum([w.shape[0] for w in wins])
    ev = torch.from_numpy(np.concatenate(wins, 0)).cuda(); off = torch.from_numpy(offs).cuda()
    idt = torch.from_numpy(idx.astype(np.int32)).cuda()
    out = torch.empty(B, 5, 2048, device="cuda")
    L = _lib.lib()
    def fn():
        _lib.check(L.ev2h_event_window_build(ev.data_ptr(), 4, off.data_ptr(), B, 346, 260, bld.cap, 0, counts.data_ptr(), table.data_ptr(), _lib.stream_handle()), "b")
        _lib.check(L.ev2h_event_window_sample(table.data_ptr(), counts.data_ptr(), bld.cap, idt.data_ptr(), B, 2048, 346, 260, out.data_ptr(), None, None, _lib.stream_handle()), "s")
    ms = timeit(fn, iters=10)
    t0 = time.time()
    for w, i in zip(wins[:32], idx[:32]):
        EW.build_window(w, i)
    cpu = 32 / (time.time() - t0)
    print(f"event-window builder: B={B} windows x {n_ev} events -> [B,5,2048]: {ms:.3f} ms  ({B / ms * 1e3:.0f} windows/s on the GPU, "
          f"oracle (numpy, 1 thread) {cpu:.0f} windows/s)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("events", "all"):
        bench_events()
    if what in ("collision", "all"):
        bench_collision()
