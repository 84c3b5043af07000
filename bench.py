"""bench.py -- event-windows/s of the Ev2Hands per-frame inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W           (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path (TEHNet.forward + MANO, both hands) over one batch of synthetic
event windows already resident in HBM: B=256 windows per GPU, N=2048 points, C=4, fp32 (the
configuration BASELINE.json's metric is quoted on); with N>1 GPUs every rank runs its own shard of the
global batch and the step ends with the RCCL all-gather of the predictions (weak scaling).
Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel (the fused set-abstraction MLP of
mano.sa1, r=0.8, K=128), timed with HIP events inside the timed region; `cpu_baseline` is the oracle
(oracle/tehnet_oracle.py, PyTorch-CPU) on the host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ev2hands_amd import _lib, dist as evdist, synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same table, dense bf16 matrix peak
# plane products executed per algorithmic multiply-add in each arithmetic mode
PRODUCTS = {"f32": 1, "bf16x3": 6, "f16x2": 3, "bf16": 1}
DTYPE = {"f32": "f32 (v_mfma_f32_32x32x2_f32)",
         "bf16x3": "f32 (each f32 operand split exactly into 3 bf16 planes, 6 plane products per MAC on the bf16 MFMA, f32 accumulate)",
         "f16x2": "f32 (each f32 operand split into 2 fp16 planes, 3 plane products per MAC on the f16 MFMA, f32 accumulate)",
         "bf16": "bf16 (f32 accumulate)"}
# algorithmic work of the profiled kernel per window: layers 2+3 of enc.sa2 branch 1 (same MLP and group shape as mano.sa1 branch 1;
# launched before the two-hand stream fork, so its HIP-event duration is free of overlap)
# (16384 rows x (128*196 + 196*256) MAC; layer 1 is not in this kernel) -- DESIGN.md "Measurement"
PROFILED_TAG = "sa2.1"
PROFILED_MAC_PER_WINDOW = 128 * 128 * (128 * 196 + 196 * 256)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="event windows per GPU per step")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--cloud", default="E", choices=["U", "E"])
    ap.add_argument("--precision", default=os.environ.get("EV2H_PRECISION", "f16x2"), choices=["f32", "bf16x3", "f16x2", "bf16"],
                    help="arithmetic of the MFMA contractions: f16x2 = fp32-class 2-plane fp16 split (default), bf16x3 = fp32-class "
                         "3-plane bf16 split (full fp32 range) -- both pass the 1e-4 / exact-argmax parity bar -- "
                         "f32 = v_mfma_f32_32x32x2_f32, bf16 = reduced precision (BASELINE.json config 3)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra exact-f32-MFMA timing leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


class HipEvents:
    """hipEvent_t pairs created through libamdhip64 directly (torch.cuda.Event only sees torch's own records)."""

    def __init__(self, n):
        self.hip = C.CDLL("libamdhip64.so")
        self.n = n
        self.start = (C.c_void_p * n)()
        self.stop = (C.c_void_p * n)()
        for arr in (self.start, self.stop):
            for i in range(n):
                ev = C.c_void_p()
                assert self.hip.hipEventCreate(C.byref(ev)) == 0
                arr[i] = ev

    def elapsed_ms(self, used):
        out = []
        for i in range(min(used, self.n)):
            ms = C.c_float()
            rc = self.hip.hipEventElapsedTime(C.byref(ms), C.c_void_p(self.start[i]), C.c_void_p(self.stop[i]))
            if rc == 0:
                out.append(ms.value)
        return out


def pmc_traffic(precision="f32"):
    """HBM bytes per launch of the profiled kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_hbm_traffic_*.json: (2*FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_{precision}_*.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        k = ks[[n for n in ks if "128, 196, 256" in n][0]]
        return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch_corrected"], "source": os.path.basename(files[-1])}
    except Exception:
        return None


def cpu_baseline(sd, assets, C_, N, cloud, seconds):
    """Oracle (port of the reference's CPU path) on the host cores, bounded sample.  PyTorch-CPU scales
    badly past a few dozen threads on these small ops (256 threads measured 100x slower than 16), so
    the baseline uses at most 16 threads and says so in `cores`."""
    from oracle import mano_oracle, tehnet_oracle
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    hands = mano_oracle.make_hands(assets["left"], assets["right"])
    b = 4
    xyz = synth.synth_cloud(cloud, b, C_, N, 99)
    inits = synth.fps_inits(b, N, 99)
    with torch.no_grad():
        t0 = time.time()
        tehnet_oracle.tehnet_forward(sd, xyz[:1].clone(), hands, fps_init=[t[:1] for t in inits])   # warm-up + sizing
        t1 = (time.time() - t0) * b
        reps = max(1, min(16, int(seconds / max(t1, 1e-3))))
        t0 = time.time()
        for _ in range(reps):
            tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits)
        dt = time.time() - t0
    return {"value": round(b * reps / dt, 3), "unit": "event-windows/s", "cores": cores, "kind": "port",
            "sample": f"{reps} forwards of B={b} N={N} C={C_} {cloud}-clouds, torch {torch.__version__} CPU, {cores} threads"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EV2H_BENCH_FORCE_DIST=1 runs the RCCL code path (init, all-gather, barrier, all-reduce) with a single rank, e.g.
    # under `torchrun --nproc-per-node 1`, to check it on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("EV2H_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    from ev2hands_amd.model import TEHNetWrapper

    B, N, Cc = a.batch, a.points, a.channels
    os.environ["ERPC"] = "1" if Cc == 5 else "0"
    os.environ["EV2H_PRECISION"] = a.precision
    assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
    sd = synth.synth_state_dict(Cc, 0)
    net = TEHNetWrapper(dev, mano_assets=assets)
    net.load_state_dict(sd, strict=True)
    net.eval()

    # this rank's shard of the global synthetic batch; FPS inits drawn for the global batch, then sliced
    gB = B * world
    lo, hi = evdist.shard_range(gB, rank, world)
    full = synth.synth_cloud(a.cloud, B, Cc, N, seed=1000 + rank)          # per-rank seed == distinct windows
    xyz = full.to(dev)
    g_inits = synth.fps_inits(gB, N, 7)
    inits = evdist.shard_fps_inits(g_inits, lo, hi)

    L = _lib.lib()
    nprof = max(a.steps, 1)
    ev = HipEvents(nprof)

    def step():
        net.net.fps_init = inits
        with torch.no_grad():
            out = net(xyz)
        if use_dist:
            out = evdist.all_gather_outputs(out, N)
        return out

    def sync():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps):
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        sync()
        dt_ = time.perf_counter() - t0
        tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
        if use_dist:
            import torch.distributed as dist
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    for _ in range(a.warmup):
        step()
    sync()
    L.ev2h_profile_set(PROFILED_TAG.encode(), ev.start, ev.stop, nprof)
    dt = timed(a.steps)
    L.ev2h_profile_set(None, None, None, 0)

    # transparency legs: the same workload in the other arithmetic modes (not part of `value`)
    legs = {}
    if not a.no_f32_leg:
        for prec in ("f32", "bf16x3", "f16x2", "bf16"):
            if prec == a.precision:
                continue
            net.net.precision = prec
            for _ in range(max(1, a.warmup)):
                step()
            k = max(2, a.steps // 2)
            dtf = timed(k)
            legs[prec] = {"value": round(gB * k / dtf, 2), "ms_per_step": round(dtf / k * 1e3, 3), "steps": k, "dtype": DTYPE[prec]}
        net.net.precision = a.precision
    f32_leg = legs.pop("f32", None)

    if rank == 0:
        kms = ev.elapsed_ms(a.steps)
        kavg = sum(kms) / max(len(kms), 1)
        nprod = PRODUCTS[a.precision]
        peak = PEAK_F32_MFMA_TFLOPS if a.precision == "f32" else PEAK_BF16_MFMA_TFLOPS
        alg_flops = 2.0 * PROFILED_MAC_PER_WINDOW * B            # fp32 multiply-adds of the layer, x2
        flops = alg_flops * nprod                                # MFMA flops the arithmetic mode needs for them
        ach = flops / (kavg * 1e-3) / 1e12 if kavg > 0 else 0.0
        res = {
            "metric": "event-windows/sec at B=256 N=2048",
            "value": round(gB * a.steps / dt, 2),
            "unit": "event-windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[a.precision], "data": "synthetic",
            "config": {"workload": f"TEHNet.forward+MANO both hands, B={B}/GPU N={N} C={Cc} fp32, {a.cloud}-clouds, "
                                   f"random-init 342-key checkpoint, synthetic MANO-shaped assets",
                       "global_batch": gB, "points": N, "channels": Cc, "precision": a.precision,
                       "parallelism": f"batch-shard x{world}" + (" + RCCL all-gather of predictions" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": f"sa_mlp_max<128,196,256> ({PROFILED_TAG}, K=128, {B} windows/launch, {a.precision})",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": pmc_traffic(a.precision),
                         "kernel_ms": round(kavg, 4), "flop_per_launch": flops,
                         "products_per_mac": nprod, "fp32_equivalent_tflops": round(ach / nprod, 2)},
        }
        if f32_leg:
            res["f32_mfma_leg"] = f32_leg
        if legs:
            res["other_modes"] = legs
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, assets, Cc, N, a.cloud, a.cpu_seconds)
        print(json.dumps(res), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
