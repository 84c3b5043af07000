"""Where the world-size-1 cost of the multi-GPU step goes (EV2H_BENCH_FORCE_DIST=1 is ~5 % slower than the plain step): times, on one
GPU with a 1-rank RCCL group, (a) the plain forward, (b) the forward writing into the gather buffer's rows, (c) b + the in-place
all-gather, (d) the all-gather alone, (e) c with the all-gather on a non-default stream of ours.   python tools/debug/dist_overhead.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ev2hands_amd import dist as evdist, synth  # noqa: E402
from ev2hands_amd.model import TEHNetWrapper  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, N, C = 256, 2048, 4
if os.environ.get("DIST_FIRST") == "1":
    dist.init_process_group("nccl", device_id=dev)
os.environ["ERPC"] = "0"
assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
net = TEHNetWrapper(dev, mano_assets=assets)
net.load_state_dict(synth.synth_state_dict(C, 0), strict=True)
net.eval()
xyz = synth.synth_cloud("E", B, C, N, seed=1000).to(dev)
inits = synth.fps_inits(B, N, 7)
gbuf = None


def fwd(rows=None):
    net.net.fps_init = inits
    with torch.no_grad():
        return net.net(xyz, net.hands, rows=rows)


def timed(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def host_only(fn, n=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return t


res = {}
res["0 plain forward BEFORE init_process_group"] = timed(lambda: fwd())
res["0' again"] = timed(lambda: fwd())
if dist.is_initialized():
    pass
elif os.environ.get("DIST_EAGER", "1") == "1":
    dist.init_process_group("nccl", device_id=dev)          # eager communicator creation
else:
    dist.init_process_group("nccl")                         # lazy: the communicator is created by the first collective
res["0'' plain forward after init (no collective yet)"] = timed(lambda: fwd())
gbuf = evdist.GatherBuffer(N, B, dev)
res["a plain forward"] = timed(lambda: fwd())
res["b forward into rows"] = timed(lambda: fwd(gbuf.rows()))
res["c rows + in-place all_gather"] = timed(lambda: (fwd(gbuf.rows()), gbuf.gather()))
res["d all_gather alone"] = timed(lambda: gbuf.gather())
res["host: forward enqueue"] = host_only(lambda: fwd(gbuf.rows()))
res["host: gather call"] = host_only(lambda: gbuf.gather())
res["a' plain forward again"] = timed(lambda: fwd())
if os.environ.get("DIST_BARRIER") == "1":
    def loop_with_barrier(n=100):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fwd()
        dist.barrier(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res["z 100 forwards bracketed by dist.barrier()"] = loop_with_barrier()
    res["z' again"] = loop_with_barrier()
for k, v in res.items():
    print(f"{k:50s} {v:8.3f} ms")
dist.destroy_process_group()
