"""Experiment: K forwards in flight as hipGraph REPLAYS (one captured graph per slot, each with its own static buffers and
workspace) against K eager forwards in flight (ev2hands_amd/inflight.py).   python tools/debug/inflight_graph.py [B [N [prec]]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import _lib, synth  # noqa: E402
from ev2hands_amd.inflight import InflightForward  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x2"
C = 4
os.environ["ERPC"] = "0"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
net = TEHNetWrapper(dev, mano_assets=assets, precision=prec)
net.load_state_dict(synth.synth_state_dict(C, 0), strict=True)
net.eval()
xyz = synth.synth_cloud("E", B, C, N, seed=1000).to(dev)
inits = synth.fps_inits(B, N, 7)
inits_dev = torch.stack(inits).to(dev)                   # (a list of host tensors would cost a blocking H2D copy per forward)
STEPS = 300


def eager(k, inits_dev=inits_dev):
    pipe = InflightForward(net, depth=k)
    for _ in range(10):
        net.net.fps_init = inits_dev
        pipe.submit(xyz)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        net.net.fps_init = inits_dev
        pipe.submit(xyz)
    pipe.drain(); torch.cuda.synchronize()
    return B * STEPS / (time.perf_counter() - t0)


def graphs(k):
    streams = _lib.concurrent_streams(dev, k) if k > 1 else [torch.cuda.current_stream()]
    gs = []
    for s in streams:
        with torch.cuda.stream(s):
            _lib.bind_stream()
            gs.append(net.capture(xyz, fps_init=inits))
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % k]):
                gs[i % k].graph.replay()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
    run(10)
    t0 = time.perf_counter()
    run(STEPS)
    return B * STEPS / (time.perf_counter() - t0)


for k in ([int(sys.argv[4])] if len(sys.argv) > 4 else (1, 2, 3)):
    if os.environ.get("HOST_INITS"):          # the drop-in default: FPS starts drawn on the host (a list of CPU tensors) for every forward
        print(f"{B} x {N} {prec}: {k} in flight, host-side FPS starts: eager {eager(k, inits):9.1f} windows/s", flush=True)
        continue
    print(f"{B} x {N} {prec}: {k} in flight: eager {eager(k):9.1f} windows/s   hipGraph replays {graphs(k):9.1f} windows/s", flush=True)
