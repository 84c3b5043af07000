"""Two full batches in flight: step i on stream i % 2 with its own workspace and output buffers, so that the latency-bound head of
step i+1 (sampling, ball queries, tables) can run under the MFMA-bound tail of step i.   python tools/debug/inflight.py [B [N]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import _lib, synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, C = (int(sys.argv[2]) if len(sys.argv) > 2 else 2048), 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ["ERPC"] = "0"
assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
net = TEHNetWrapper(dev, mano_assets=assets)
net.load_state_dict(synth.synth_state_dict(C, 0), strict=True)
net.eval()
xyz = synth.synth_cloud("E", B, C, N, seed=1000).to(dev)
inits = torch.stack(synth.fps_inits(B, N, 7)).to(dev)
L = _lib.lib()
nbytes = L.ev2h_workspace_bytes(B, N)


def run(nstreams, steps=100):
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    wss = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(max(nstreams, 1))]

    def loop(n):
        for i in range(n):
            k = i % len(streams)
            with torch.cuda.stream(streams[k]), torch.no_grad():
                net.net._enqueue(xyz, inits, net.hands, ws=wss[k])
    loop(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


for n in (1, 2, 1, 2, 3):
    print(f"streams in flight {n}: {run(n):9.1f} windows/s")
