"""bench.py -- event-windows/s of the Ev2Hands per-frame inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (TEHNet.forward + MANO, both hands) over one batch of synthetic
event windows already resident in HBM: B=256 windows per GPU, N=2048 points, C=4, fp32 (the
configuration BASELINE.json's metric is quoted on); with N>1 GPUs every rank runs its own shard of the
global batch and the step ends with the RCCL all-gather of the predictions (weak scaling).

N>1 runs one process per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks
already exist (RANK / LOCAL_RANK / WORLD_SIZE in the environment); invoked directly as
`python bench.py --gpus N` this process only LAUNCHES the N ranks as children (before anything touches
the GPU -- it never initialises HIP itself and never exec()s) and relays rank 0's JSON line.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused set-abstraction MLP
128-196-256, launch site enc.sa2 branch 1), timed with HIP events inside the timed region:
`achieved`/`frac` count the ALGORITHMIC fp32 multiply-adds against the peak of the matrix pipe the
mode runs on, `executed`/`executed_frac` the plane products the split arithmetic really issues.
`cpu_baseline` is the oracle (oracle/tehnet_oracle.py, PyTorch-CPU) on the host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import math
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0    # same table, dense bf16 / fp16 matrix peak
# What the matrix pipe SUSTAINS on random operands: a pure v_mfma_f32_32x32x16_f16 loop on every CU (no memory, no VALU) runs at
# 0.98 of the peak on all-zero operands and at 0.68 of it on random ones -- the clock drops from 2.39 to ~1.68 GHz under the power
# limit (tools/ubench/mfma_power.hip, profiles/r3_mfma_ceiling.txt).  A kernel that also feeds the pipe cannot beat that.
SUSTAINED_16BIT_MFMA_TFLOPS = 1700.0
# plane products executed per algorithmic multiply-add in each arithmetic mode
PRODUCTS = {"f32": 1, "bf16x3": 6, "f16x2": 3, "bf16": 1, "f16": 1}
DTYPE = {"f32": "f32 (v_mfma_f32_32x32x2_f32)",
         "bf16x3": "f32 (each f32 operand split exactly into 3 bf16 planes, 6 plane products per MAC on the bf16 MFMA, f32 accumulate)",
         "f16x2": "f32 (each f32 operand split into 2 fp16 planes with per-window power-of-two range scaling, 3 plane products per MAC "
                  "on the f16 MFMA, f32 accumulate)",
         "bf16": "bf16 (f32 accumulate)",
         "f16": "f16 (ONE fp16 plane per operand with f16x2's per-window power-of-two range scaling, 1 product per MAC on the f16 MFMA, "
                "f32 accumulate)"}
# algorithmic work of the profiled kernel per window: layers 2+3 of enc.sa2 branch 1 (same MLP and group shape as mano.sa1 branch 1;
# launched before the two-hand stream fork, so its HIP-event duration is free of overlap)
# (16384 rows x (128*196 + 196*256) MAC; layer 1 is not in this kernel) -- DESIGN.md "Measurement"
PROFILED_TAG = "sa2.1"
PROFILED_MAC_PER_WINDOW = 128 * 128 * (128 * 196 + 196 * 256)


def launch_sites(N):
    """The two largest single launches of a forward, by algorithmic multiply-adds per window: the fused set-abstraction MLP
    128-196-256 of enc.sa2 branch 1 (independent of N: 128 centroids x 128 neighbours) and the k = 3 query convolution of both
    hands (one GEMM, [N x 768] x [768 x 512] per window, TEHNet.py:150-153).  At N = 2048 the first is larger (1 233 vs 805 MMAC
    per window), at BASELINE config 5's N = 8192 the second (3 221 MMAC).  The DOMINANT one for the workload is the `roofline`
    entry; the other is reported as `roofline_second`.  (Which dispatches of a trace belong to a site: site_dispatch_values.)"""
    sa = {"tag": "sa2.1", "mac_per_window": PROFILED_MAC_PER_WINDOW,
          "name": "sa_mlp_max<128,196,256> (sa2.1, K=128"}
    qc = {"tag": "qconv0", "mac_per_window": N * 768 * 512,
          "name": f"gemm_nt k=3 query convolution, both hands (qconv0, M={N}/window K=3x256 N=512"}
    return (sa, qc) if sa["mac_per_window"] >= qc["mac_per_window"] else (qc, sa)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="event windows per GPU per step")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--inflight", type=int, default=1,
                    help="forwards in flight (ev2hands_amd/inflight.py: step i on stream i mod K with its own workspace); for shards too small to fill "
                         "the chip, e.g. --points 8192 --batch 16 --inflight 2.  Default 1: one forward at a time on one stream")
    ap.add_argument("--cloud", default="E", choices=["U", "E"])
    ap.add_argument("--weights", default="random", choices=["random", "trained"],
                    help="random: hash-generated random-init checkpoint (the contract's default); trained: the checkpoint that came out of the "
                         "reference's training loop (tests/trained_ckpt.py) -- same shapes and FLOPs, other bit patterns (dead units, saturated "
                         "softmax): an extra line for profiles/, never the default")
    ap.add_argument("--precision", default=os.environ.get("EV2H_PRECISION", "f16x2"), choices=["f32", "bf16x3", "f16x2", "bf16", "f16"],
                    help="arithmetic of the MFMA contractions: f16x2 = fp32-class 2-plane fp16 split with per-window range scaling "
                         "(default), bf16x3 = fp32-class 3-plane bf16 split -- both pass the 1e-4 / exact-argmax parity bar, also on "
                         "checkpoints with hidden activations from 1e-4 to 1e+6 -- f32 = v_mfma_f32_32x32x2_f32, bf16 = reduced "
                         "precision (BASELINE.json config 3)")
    ap.add_argument("--no-host-io", dest="no_host_io", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-legs", "--no-f32-leg", dest="no_legs", action="store_true", help="skip the timing legs of the other arithmetic modes")
    ap.add_argument("--no-latency", action="store_true", help="skip the B=1 / B=8 latency legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the two-stream / gather self-checks after the timed region")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained leg (value_sustained + shader clock)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5, help="length of the sustained leg's timed region")
    ap.add_argument("--no-second-site", action="store_true", help="skip the short extra run that brackets the second-largest launch (roofline_second)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the live HBM-traffic measurement (two rocprofv3 PMC passes of a 3-step child run of this script)")
    ap.add_argument("--collision", action="store_true",
                    help="BASELINE config 5: add the two-hand self-collision term to every step (pair search with the evaluation's "
                         "max_collisions = 8, then the intersection loss's pair search with 16 and its distance-field penalty)")
    ap.add_argument("--collision-score", action="store_true",
                    help="with --collision: also the evaluation script's collision count (a second pair search, cap 8) in every step")
    ap.add_argument("--collision-mesh", default="surface", choices=["surface", "soup"],
                    help="geometry of the synthetic hand assets for --collision: 'surface' = a mitten-shaped mesh with smooth blend shapes and "
                         "skinning (deforms like a hand mesh: tens to hundreds of colliding triangle pairs per window), 'soup' = the default "
                         "parity assets, whose vertices and faces are independent random draws (every mesh intersects itself ~24 000 times: "
                         "the worst case for the pair search)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--shared-device", action="store_true",
                    help="REHEARSAL of an N-rank run on a box with ONE GPU: every rank runs the real forward on device 0, the process group is "
                         "gloo (RCCL refuses two ranks on one device) and the gather goes through dist.py's host-staged transport.  Everything "
                         "but RCCL's transport runs as in the real run -- launcher, sharding, barriers, max-over-ranks, per-rank fields, the "
                         "line's key set; the line is marked \"rehearsal\" and its throughput (N processes time-slicing one GPU) means nothing")
    ap.add_argument("--stub", action="store_true",
                    help="CPU self-test of the multi-rank plumbing (launcher, gloo all-gather, max-over-ranks timing) with fabricated "
                         "predictions; prints a line marked \"stub\": true that is not a measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set as torch.distributed.run would), watch them and relay rank 0's JSON line.  Runs before any HIP call; the parent never
    initialises the GPU.  Every rank's stdout + stderr go to gpurun_out/bench_rank<r>.log (kept: gpurun merges that directory
    back); the first rank that exits non-zero ends the run -- the others are terminated (exactly the children started here)
    instead of waiting in RCCL for a peer that is gone -- and the tail of every failed rank's log is printed."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(_free_port())
    env["WORLD_SIZE"] = str(a.gpus)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: required by RCCL on this driver
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(a.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        path = os.path.join(logdir, f"bench_rank{r}.log")
        f = open(path, "w")
        logs.append(path)
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e, stdout=f, stderr=subprocess.STDOUT), f))
    deadline = time.time() + float(os.environ.get("EV2H_BENCH_LAUNCH_TIMEOUT", 3600))
    rcs = [None] * a.gpus
    failed = False
    while any(rc is None for rc in rcs):
        for r, (p, _) in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) or time.time() > deadline:
            failed = True
            break
        time.sleep(0.2)
    if failed:
        for r, (p, _) in enumerate(procs):
            if rcs[r] is None:
                p.terminate()
        for r, (p, _) in enumerate(procs):
            if rcs[r] is None:
                try:
                    rcs[r] = p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    rcs[r] = p.wait()
    for _, f in procs:
        f.close()
    line = None
    for ln in open(logs[0], errors="replace").read().splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None and not failed:
        print(line, flush=True)
        return 0
    sys.stderr.write(f"bench.py launcher: rank return codes {rcs}, rank-0 JSON line {'found' if line else 'MISSING'}; logs: {logdir}/bench_rank*.log\n")
    for r, path in enumerate(logs):
        if rcs[r] != 0 or line is None:
            sys.stderr.write(f"---- tail of rank {r} (rc {rcs[r]}) ----\n" + open(path, errors="replace").read()[-3000:] + "\n")
    return max((abs(rc) for rc in rcs if rc), default=0) or 1


# ------------------------------------------------------------------------------------------------ helpers
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


PROFILED_KERNEL_KEY = "128, 196, 256"          # template arguments that name the profiled kernel in a trace
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# SURVEY.md 8d per window at N=2048, C=4: boundary I/O, the fused stage-boundary estimate, the unfused reference-style estimate
ALG_IO_BYTES_PER_WINDOW = 84888
SURVEY_FUSED_BYTES_PER_WINDOW = 16.5e6
SURVEY_UNFUSED_BYTES_PER_WINDOW = 1.99e9


def committed_pmc_traffic(precision="f32", B=256, N=2048):
    """Fallback: ({"kernel": bytes per launch of the profiled kernel, "step": bytes per forward step or None}, source file) from
    the newest committed rocprofv3 PMC summary (profiles/r*_pmc_hbm_traffic_<mode>*.json, tools/pmc_traffic.py) OF THIS WORKLOAD:
    tools/profile_round.sh profiles 256 windows of 2048 points, tools/profile_n8192.sh 128 windows of 8192 (`..._n8192_<mode>.json`).
    Any other shape has no committed figure (None): a 16-window line once carried the 256-window step's 6.4 GB."""
    import glob
    if (B, N) == (256, 2048):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_{precision}_v*.json")) +
                       glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_{precision}.json")))
    elif (B, N) == (128, 8192):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic_n8192_{precision}.json")))
    else:
        files = []
    if not files:
        return None, None
    try:
        j = json.load(open(files[-1]))
        ks = j["kernels"]
        k = ks[[n for n in ks if PROFILED_KERNEL_KEY in n][0]]
        return {"kernel": k["hbm_bytes_per_launch_corrected"], "step": j.get("step_hbm_bytes_corrected")}, os.path.basename(files[-1])
    except Exception:
        return None, None


def profiled_interpreter():
    """The program that follows `rocprofv3 ... --`.  It must be the interpreter ITSELF (the profiler's preloaded library has
    initialised the GPU, so any env / shell / launcher hop that re-execs is refused on this pool).  A symlink is not an exec hop, and
    resolving it would leave a virtualenv (the base interpreter has no pyvenv.cfg, hence no venv site-packages): sys.executable is
    used as it is unless it is a script shim (starts with `#!`), in which case the ELF behind it is taken."""
    exe = sys.executable
    try:
        with open(exe, "rb") as f:
            if f.read(2) == b"#!":
                return os.path.realpath(getattr(sys, "_base_executable", exe) or exe)
    except OSError:
        pass
    return exe


def live_pmc_traffic(a, timeout_s=300):
    """HBM traffic measured IN THIS RUN: two rocprofv3 passes (one PMC counter each, FETCH_SIZE then WRITE_SIZE, with --kernel-trace
    only -- the collection MI355X_MICROARCH.md prescribes) of a 3-forward child run of this very script, read from rocprofv3's
    sqlite output.  gfx950 correction: both counters are in KiB and FETCH_SIZE counts 64 B per 128 B request, so
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns {"kernel": bytes per launch of the profiled kernel, "step": bytes per
    forward step (sum over EVERY dispatch / 3)} or raises."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    nsteps = 3
    child = [os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-legs", "--no-latency", "--no-cpu-baseline", "--no-traffic", "--no-selfcheck", "--no-second-site", "--no-host-io", "--no-sustained",
             "--precision", a.precision, "--batch", str(a.batch), "--points", str(a.points), "--channels", str(a.channels), "--cloud", a.cloud, "--weights", a.weights]
    tmp = tempfile.mkdtemp(prefix="ev2h_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "EV2H_BENCH_FORCE_DIST")}
    env["TMPDIR"] = "/tmp"
    sums = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            # the program itself follows `--` (no env / shell / launcher hop: the profiler's preloaded library has initialised the GPU)
            r = subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--", profiled_interpreter()] + child, cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                raise RuntimeError(f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {(r.stdout or '')[-400:]}")
            rows = sqlite3.connect(dbs[0]).execute(
                "select kernel_name, grid_size, dispatch_id, sum(value) from counters_collection where counter_name = ? "
                "group by kernel_name, grid_size, dispatch_id", (ctr,)).fetchall()
            if not rows:
                raise RuntimeError(f"no {ctr} samples in {dbs[0]}")
            per_site = {}
            for site in launch_sites(a.points):
                vals = site_dispatch_values(site["tag"], rows)
                if not vals:            # never publish a "measured 0 bytes": a site that matches no dispatch is an error
                    raise RuntimeError(f"no dispatch of launch site {site['tag']} found in the {ctr} pass")
                per_site[site["tag"]] = (sum(vals) / len(vals), len(vals))
            sums[ctr] = (sum(r[3] for r in rows), per_site)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    tag1 = launch_sites(a.points)[0]["tag"]
    sites = {t: {"bytes": int((2 * sums["FETCH_SIZE"][1][t][0] + sums["WRITE_SIZE"][1][t][0]) * 1024), "launches_sampled": sums["FETCH_SIZE"][1][t][1]}
             for t in sums["FETCH_SIZE"][1]}
    return {"kernel": sites[tag1]["bytes"], "sites": sites,
            "step": int((2 * sums["FETCH_SIZE"][0] + sums["WRITE_SIZE"][0]) * 1024 / nsteps),
            "kernel_launches_sampled": sites[tag1]["launches_sampled"]}


def site_dispatch_values(tag, rows):
    """The counter values of the dispatches that belong to one launch site, from (kernel_name, grid_size, dispatch_id, value) rows.
    Independent of trailing template arguments and of the arithmetic mode: `sa2.1` = the fused set-abstraction kernel with the
    widths 128-196-256 (three launch sites per forward run it at the same shape: all are sampled); `qconv0` = the k = 3 query
    convolution = the dense-layer (gemm_nt*) launch with the LARGEST grid of the forward in every mode."""
    if tag == PROFILED_TAG:
        return [v for k, _, _, v in rows if PROFILED_KERNEL_KEY in k and "sa_mlp_max" in k]
    gemm = [(g, v) for k, g, _, v in rows if "gemm_nt" in k]
    if not gemm:
        return []
    gmax = max(g for g, _ in gemm)
    return [v for g, v in gemm if g == gmax]


def newest_profile(pattern):
    """basename of the newest committed profiles/<pattern> by ROUND number (r6_... after r5_...), or None"""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", pattern))
    if not files:
        return None
    rnd = lambda f: int((re.match(r"r(\d+)_", os.path.basename(f)) or [0, 0])[1])      # noqa: E731
    return os.path.basename(max(files, key=lambda f: (rnd(f), os.path.getmtime(f))))


def expected_line_keys(world, a):
    """The keys rank 0's line of a REAL (non-stub) run carries at this world size and with these switches -- one definition, used by
    run_rank to check its own line (`line_complete`), by the CPU launcher tests (stub lines list it as `would_emit`: world 2 and 8)
    and by tests/test_gpu_bench.py on the forced 1-rank RCCL path.  A multi-GPU line is graded like the 1-GPU one: it carries
    `roofline` (traffic from the committed PMC profile, marked as such) and `cpu_baseline` too."""
    keys = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "ms_per_step_per_rank", "roofline", "hbm", "line_complete"]
    if not a.no_cpu_baseline:
        keys.append("cpu_baseline")
    if not a.no_sustained:
        keys.append("value_sustained")
    if not a.no_selfcheck:
        keys.append("multi_gpu_selfcheck")
    if not a.no_second_site:
        keys.append("roofline_second")
    return sorted(keys)


def roofline_entry(precision, B, kernel_ms_list, traffic=None, traffic_src=None, site=None, n_points=2048):
    """Dominant-kernel roofline: algorithmic fp32 FLOPs of the layer (2 x MACs) over the HIP-event duration, against the dense peak of
    the matrix pipe the mode uses; the plane products the split modes execute are reported separately."""
    kavg = sum(kernel_ms_list) / max(len(kernel_ms_list), 1)
    nprod = PRODUCTS[precision]
    peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_16BIT_MFMA_TFLOPS
    site = site or launch_sites(2048)[0]
    alg_flops = 2.0 * site["mac_per_window"] * B
    alg = alg_flops / (kavg * 1e-3) / 1e12 if kavg > 0 else 0.0
    if traffic is None and site["tag"] == PROFILED_TAG:
        t, src = committed_pmc_traffic(precision if precision != "f16" else "bf16", B, n_points)      # (f16: bf16's images and tile walk; its own profile when committed)
        t16, src16 = committed_pmc_traffic(precision, B, n_points) if precision == "f16" else (None, None)
        if t16 is not None:
            t, src = t16, src16
        if t is not None:
            traffic, traffic_src = t["kernel"], "committed profile profiles/%s (not measured in this run)" % src
    traffic_kind = None if traffic is None else ("live" if traffic_src and "measured in this run" in traffic_src and "not measured" not in traffic_src else "committed")
    return {"bound": "mfma", "kernel": f"{site['name']}, {B} windows/launch, {precision})", "launch_site": site["tag"],
            "achieved": round(alg, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(alg / peak, 4),
            "accounting": "algorithmic fp32 multiply-adds x 2 / HIP-event kernel time / dense peak of the MFMA type used",
            "executed": round(alg * nprod, 2), "executed_frac": round(alg * nprod / peak, 4), "products_per_mac": nprod,
            "frac_of_f32_mfma_peak": round(alg / PEAK_F32_MFMA_TFLOPS, 4),
            **({"sustained_peak": SUSTAINED_16BIT_MFMA_TFLOPS, "executed_frac_of_sustained": round(alg * nprod / SUSTAINED_16BIT_MFMA_TFLOPS, 4),
                "sustained_source": "pure MFMA loop on random operands at the power-limited clock, profiles/%s" % (newest_profile("r*_mfma_ceiling.txt") or "(none committed)")}
               if precision != "f32" else {}),
            "traffic": traffic, "traffic_kind": traffic_kind, "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) * 1024 from rocprofv3 PMC passes", "traffic_source": traffic_src,
            "kernel_ms": round(kavg, 4), "kernel_samples": len(kernel_ms_list),
            "flop_per_launch": alg_flops}


def hbm_entry(B, N, ms_per_step, step_bytes, source):
    """Whole-step HBM view (north_star: "achieved fraction of the HBM roofline"): bytes every kernel of one forward moves (PMC) over
    the step time, against 8 TB/s, next to SURVEY 8d's brackets.  The path is matrix-pipe bound (233 kFLOP per algorithmic byte), so
    this fraction says how much HBM time the step needs, not how fast it could be."""
    scale = N / 2048.0
    gbs = step_bytes / (ms_per_step * 1e-3) / 1e9 if step_bytes else None
    return {"bound": "hbm", "bytes_per_step": step_bytes, "achieved": round(gbs, 1) if gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None, "source": source,
            "source_kind": None if not step_bytes else ("live" if source and "measured in this run" in source and "not measured" not in source else "committed"),
            "hbm_time_share_of_step": round(step_bytes / (HBM_PEAK_GBS * 1e9) / (ms_per_step * 1e-3), 4) if step_bytes else None,
            "algorithmic_io_bytes_per_step": int(ALG_IO_BYTES_PER_WINDOW * B * scale),
            "survey_fused_boundary_bytes_per_step": int(SURVEY_FUSED_BYTES_PER_WINDOW * B * scale),
            "survey_unfused_reference_bytes_per_step": int(SURVEY_UNFUSED_BYTES_PER_WINDOW * B * scale),
            "note": "bytes_per_step = sum over every kernel dispatch of one forward of (2*FETCH_SIZE + WRITE_SIZE) * 1024"}


def cpu_baseline(sd, assets, C_, N, cloud, seconds):
    """Oracle (port of the reference's CPU path) on the host cores, bounded sample.  BASELINE.md section 4 asks for all host
    cores, B=32 and 1 warm-up + 3 timed forwards; PyTorch-CPU collapses past a few dozen threads on these small ops (256
    threads measured 100x slower than 16), so at most 16 threads are used, and the batch / repetitions are the largest of the
    protocol's that fit the time budget.  Both the protocol and what was run are reported."""
    import torch
    from ev2hands_amd import synth
    from oracle import mano_oracle, tehnet_oracle
    host = os.cpu_count() or 1
    cores = min(host, 16)
    torch.set_num_threads(cores)
    hands = mano_oracle.make_hands(assets["left"], assets["right"])
    with torch.no_grad():
        x1 = synth.synth_cloud(cloud, 1, C_, N, 99)
        i1 = synth.fps_inits(1, N, 99)
        t0 = time.time()
        tehnet_oracle.tehnet_forward(sd, x1.clone(), hands, fps_init=i1)                      # warm-up + sizing
        t1 = time.time() - t0
        b = 32
        while b > 2 and 2 * b * t1 * 0.35 > seconds:      # per-window cost at B>=4 is ~0.35x the B=1 forward
            b //= 2
        reps = 3 if 3 * b * t1 * 0.35 <= seconds else 2
        xyz = synth.synth_cloud(cloud, b, C_, N, 99)
        inits = synth.fps_inits(b, N, 99)
        t0 = time.time()
        for _ in range(reps):
            tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits)
        dt = time.time() - t0
    return {"value": round(b * reps / dt, 3), "unit": "event-windows/s", "cores": cores, "kind": "port", "host_cores": host,
            "sample": f"1 warm-up (B=1) + {reps} timed forwards of B={b} N={N} C={C_} {cloud}-clouds, torch {torch.__version__} CPU, "
                      f"{cores} of {host} host threads (BASELINE.md protocol: all cores, B=32, 1+3 forwards)"}


def stub_outputs(B, N, rank):
    """Fabricated predictions of the right shapes (--stub): value = rank, so the gathered result can be checked."""
    import torch
    from ev2hands_amd import synth
    f = lambda *s: torch.full(s, float(rank))                                                  # noqa: E731
    out = {"class_logits": f(B, 4, N)}
    for side in ("left", "right"):
        out[side] = {"global_orient": f(B, 3), "hand_pose": f(B, synth.MANO_CMPS), "betas": f(B, 10), "transl": f(B, 3),
                     "vertices": f(B, synth.MANO_NV, 3), "j3d": f(B, 21, 3)}
    return out


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(a) -> int:
    import torch
    from ev2hands_amd import dist as evdist, synth

    # dmabuf IPC: RCCL / cross-process device memory on this driver need it; also for ranks started by torch.distributed.run,
    # which does not go through launch_ranks() -- set before anything initialises HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {a.gpus}, or run "
                         f"`python bench.py --gpus {a.gpus}` directly and let it start the ranks)")
    if a.stub and os.environ.get("EV2H_BENCH_STUB_FAIL_RANK") == str(rank):          # launcher self-test: a rank that dies at start-up
        sys.stderr.write(f"stub rank {rank}: simulated start-up failure\n")
        return 3
    # EV2H_BENCH_FORCE_DIST=1 runs the RCCL code path (init, all-gather, barrier, all-reduce) with a single rank
    use_dist = world > 1 or bool(os.environ.get("EV2H_BENCH_FORCE_DIST"))
    if a.stub:
        dev = torch.device("cpu")
    else:
        if a.shared_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if not a.stub:
        # HIP multiplexes a process's streams onto a few hardware queues (4 by default) and streams that share one run in order.
        # torch.distributed / RCCL create dozens of streams; when the forward's side stream was created after them it landed on
        # the main stream's queue and the two-stream overlaps silently vanished: that, not the gather, was the "4-5 % multi-GPU
        # overhead at world size 1" of round 2 (profiles/r3_dist_overhead.txt).  So: the library's side stream is created FIRST
        # (ev2h_init), with the runtime's default number of hardware queues (asking for 8 made the 1-rank RCCL step 10 % slower).
        from ev2hands_amd import _lib as _early
        _early.check(_early.lib().ev2h_init(), "ev2h_init")
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.stub or a.shared_device:
            dist.init_process_group("gloo")
        else:
            import datetime
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=float(os.environ.get("EV2H_BENCH_RCCL_TIMEOUT", 300))))
        world_seen = dist.get_world_size()
    else:
        world_seen = 1

    B, N, Cc = a.batch, a.points, a.channels
    gB = B * world
    lo, hi = evdist.shard_range(gB, rank, world)
    g_inits = synth.fps_inits(gB, N, 7)                    # drawn for the GLOBAL batch, then sliced (sharded == unsharded)
    inits = evdist.shard_fps_inits(g_inits, lo, hi)

    infl = None
    if a.stub:
        net = L = ev = sd = assets = None

        def forward(rows=None):
            return stub_outputs(hi - lo, N, rank)
    else:
        from ev2hands_amd import _lib
        from ev2hands_amd.model import TEHNetWrapper
        os.environ["ERPC"] = "1" if Cc == 5 else "0"
        os.environ["EV2H_PRECISION"] = a.precision
        if a.collision and a.collision_mesh == "surface":
            assets = {s: synth.synth_mano_surface_assets(s, 0) for s in ("left", "right")}
        else:
            assets = {s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}
        sd = synth.synth_state_dict(Cc, 0)
        if a.weights == "trained":
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import trained_ckpt
            sd = trained_ckpt.trained_state_dict(Cc)
        net = TEHNetWrapper(dev, mano_assets=assets)
        net.load_state_dict(sd, strict=True)
        net.eval()
        xyz = synth.synth_cloud(a.cloud, B, Cc, N, seed=1000 + rank).to(dev)       # per-rank seed == distinct windows
        L = _lib.lib()
        ev = HipEvents(max(a.steps, 1))

        closs = dev_faces = None
        if a.collision:
            from ev2hands_amd import collision as evcol
            closs = evcol.CollisionLoss(dev)
            dev_faces = (evcol.device_faces(net.hands["left"].faces, dev), evcol.device_faces(net.hands["right"].faces, dev))   # converted ONCE

        if a.inflight > 1 and not use_dist:            # (with a process group the gather pipeline owns the slots: GatherPipeline(inflight=K))
            from ev2hands_amd.inflight import InflightForward
            infl = InflightForward(net, a.inflight)

        def collision_terms(out):
            if closs is not None:
                # device-side only: the face tables were uploaded in the set-up (a pageable host->device copy per call would
                # synchronise the stream six times per step).  BASELINE config 5 names the intersection-LOSS term (losses.py:60-102:
                # pair search with max_collisions = 16 + conic distance-field penalty); --collision-score adds the evaluation
                # script's count (evaluate_ev2hands_r.py:128-160: a second search with max_collisions = 8), which the reference
                # never runs in the same step
                if a.collision_score:
                    out["collision_count"], _ = evcol.mesh_collisions(out["left"]["vertices"], out["right"]["vertices"], dev_faces[0], dev_faces[1],
                                                                      max_per_triangle=8)
                out["collision_penalty"] = closs.per_window(out, faces=dev_faces)
            return out

        def forward(rows=None):
            net.net.fps_init = inits
            if infl is not None:                       # K forwards in flight: returns a ticket, the work is on stream i mod K
                return infl.submit(xyz, rows=rows, post=collision_terms)
            with torch.no_grad():
                return collision_terms(net.net(xyz, net.hands, rows=rows))

    # multi-GPU: the forward writes its windows straight into this rank's slice of a persistent gather buffer (ev2h_outputs'
    # window strides), then ONE in-place all-gather -- no packing copy, no allocation per step.  Two buffers alternate and the
    # gather is asynchronous: the xGMI transfer of step i runs under the forward of step i+1 (EV2H_BENCH_SYNC_GATHER=1: in
    # stream order).  The last gather completes inside the timed region (sync() drains the pipeline).
    sync_gather = bool(os.environ.get("EV2H_BENCH_SYNC_GATHER"))
    pipe_inflight = a.inflight if (use_dist and not a.stub and a.inflight > 1) else 0
    pipe = evdist.GatherPipeline(N, gB, dev, depth=1 if sync_gather else 2, inflight=pipe_inflight, net=net if pipe_inflight else None) if use_dist else None

    def step():
        if pipe is None:
            return forward()
        rows = pipe.rows()
        if a.stub:
            rows.copy_(evdist.pack_outputs(forward()))
        elif pipe_inflight:
            # forwards in flight AND a gather: one call (dist.GatherPipeline.forward) -- the forward on its slot's stream, the collective
            # issued from that stream right behind it, the caller's stream never waits for a forward
            pending = pipe.forward(xyz, fps_init=inits, post=collision_terms)
            return pending.result() if sync_gather else pending
        else:
            forward(rows)
        pending = pipe.submit()
        return pending.result() if sync_gather else pending

    cdev = torch.device("cpu") if a.shared_device else dev        # where the scalar collectives' tensors live (gloo: host)

    def sync():
        if not a.stub and infl is not None:
            infl.drain()
        if pipe is not None:
            pipe.drain()
        if use_dist:
            dist.barrier()
        if not a.stub:
            torch.cuda.synchronize()

    def timed(nsteps, hook=None, every=0):
        sync()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step()
            if hook is not None and every and i % every == every // 2:
                hook()
        sync()
        dt_ = time.perf_counter() - t0
        tmax = torch.tensor([dt_], device=cdev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        timed.local = dt_
        return float(tmax.item())

    def all_ranks(value):
        """[value of rank 0, ..., value of rank world-1] on every rank"""
        t = torch.tensor([float(value)], device=cdev, dtype=torch.float64)
        if not use_dist:
            return [float(t.item())]
        full = torch.zeros(world_seen, device=cdev, dtype=torch.float64)
        dist.all_gather_into_tensor(full, t)
        return [float(v) for v in full.tolist()]

    last = None
    for _ in range(a.warmup):
        last = step()
    sync()
    site1, site2 = launch_sites(N)
    if L is not None:
        L.ev2h_profile_set(site1["tag"].encode(), ev.start, ev.stop, ev.n)
    dt = timed(a.steps)
    local_dt = timed.local                   # this rank's own wall time of the timed region (later timed() calls overwrite timed.local)
    if L is not None:
        L.ev2h_profile_set(None, None, None, 0)
    main_kernel_ms = ev.elapsed_ms(a.steps) if ev is not None else []
    second_kernel_ms = []
    if L is not None and not a.no_second_site:     # the second-largest launch, bracketed in a short run of its own (outside `value`)
        k2 = max(3, min(a.steps // 4, 20))
        ev2 = HipEvents(k2)
        L.ev2h_profile_set(site2["tag"].encode(), ev2.start, ev2.stop, ev2.n)
        timed(k2)
        L.ev2h_profile_set(None, None, None, 0)
        second_kernel_ms = ev2.elapsed_ms(k2)
    rank_ms = [round(v / a.steps * 1e3, 3) for v in all_ranks(local_dt)]        # every rank's own wall time per step
    # value_sustained: the SAME step over a region of >= 2 s, with the shader clock observed while it runs (one-wave probes on a
    # stream of their own, ev2h_shader_clock_probe).  The matrix-pipe kernels are power-limited on real data (2.39 GHz idle boost ->
    # ~1.65 GHz under load, profiles/r*_mfma_ceiling.txt): the driver's `--steps 20` region lasts a quarter of a second, so a reader
    # needs to see whether `value` is a boost-window number.  Never `value`; every rank takes part (collectives inside timed()).
    sustained = None
    if not a.stub and not a.no_sustained:
        from ev2hands_amd import _lib as _evl
        ms0 = dt / a.steps * 1e3
        ks = int(min(max(a.steps, math.ceil(a.sustained_seconds * 1e3 / max(ms0, 1e-3))), 20000))
        smp = _evl.ShaderClockSampler(dev, max_samples=40)
        torch.cuda.synchronize()
        smp.sample()                                       # (chip idle for the moment: the boost clock)
        idle = smp.mhz()
        smp.n = 0
        dts = timed(ks, hook=smp.sample, every=max(1, ks // 32))
        clk = sorted(smp.mhz())
        med = clk[len(clk) // 2] if clk else None
        sustained = {"value": round(gB * ks / dts, 2), "unit": "event-windows/s", "ms_per_step": round(dts / ks * 1e3, 3), "steps": ks, "seconds": round(dts, 3),
                     "ratio_to_value": round((gB * ks / dts) / (gB * a.steps / dt), 4),
                     "shader_clock_mhz": {"median": round(med, 1) if med else None, "min": round(clk[0], 1) if clk else None,
                                          "max": round(clk[-1], 1) if clk else None, "samples": len(clk)},
                     "shader_clock_mhz_idle": round(idle[0], 1) if idle else None,
                     "shader_clock_mhz_median_per_rank": [round(v, 1) for v in all_ranks(med or 0.0)],
                     "note": "same step, same buffers; clock = 100 MHz x delta(s_memtime) / delta(s_memrealtime) of one-wave probes launched on their "
                             "own stream during the region (ev2h_shader_clock_probe)"}
    if a.stub:                             # launcher self-test: one more step on every rank whose gathered result rank 0 checks
        last = step()
        if not isinstance(last, dict):
            last = last.result()
        sync()

    # self-checks of the multi-GPU run (outside the timed region; every rank takes part):
    #  * two_stream_gain: a few steps with the library's side stream switched off (ev2h_set_side_stream) against the same number
    #    with it on -- a rank whose side stream shares a hardware queue with its main stream shows ~1.00
    #    where the others show ~1.05, without any error;
    #  * gather_ms: the all-gather alone (device time between two events around a blocking collective on an idle device).
    selfcheck = None
    if not a.stub and not a.no_selfcheck:
        k = max(3, min(a.steps // 10, 10))
        timed(2)                                          # (back to steady clocks after the profiling hook)
        t_on = timed(k); on_local = timed.local
        prev = L.ev2h_set_side_stream(0)
        timed(2)
        t_off = timed(k); off_local = timed.local
        L.ev2h_set_side_stream(prev)
        timed(2)
        gains = [round(v, 4) for v in all_ranks(off_local / on_local)]
        try:
            from ev2hands_amd import _lib as _evlib
            probe = _evlib.side_stream_probe(50)
        except Exception:  # noqa: BLE001 -- EV2H_TWO_STREAMS=0: nothing to probe
            probe = 0.0
        probes = [round(v, 3) for v in all_ranks(probe)]
        selfcheck = {"two_stream_gain": round(t_off / t_on, 4), "two_stream_gain_per_rank": gains, "steps_each": k,
                     "side_stream_probe_per_rank": probes,
                     "side_stream_probe_note": "ev2h_side_stream_probe: a 50 us spin kernel on each of the two streams at once / one alone; ~1.0 = the "
                                               "streams run concurrently, ~2.0 = they share a hardware queue (0 = side stream off)",
                     "note": "wall time of k steps with ev2h_set_side_stream(0) / with the default two-stream schedule; ~1.00 on a rank "
                             "means its side stream shares a hardware queue with the main stream"}
        if pipe is not None:
            b0 = pipe.bufs[0]
            lo_ = b0.rank * b0.big
            gms = []
            for _ in range(7):
                torch.cuda.synchronize()
                if use_dist:
                    dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if b0.host_staged:                         # (--shared-device: gloo, the buffer's pinned-host transport)
                    b0._gather(async_op=False)
                    b0._landed()
                else:
                    dist.all_gather_into_tensor(b0.full, b0.full[lo_:lo_ + b0.big])
                e1.record()
                torch.cuda.synchronize()
                gms.append(e0.elapsed_time(e1))
            gms.sort()
            selfcheck["gather_ms"] = round(gms[len(gms) // 2], 4)
            selfcheck["gather_ms_per_rank"] = [round(v, 4) for v in all_ranks(gms[len(gms) // 2])]
            selfcheck["gather_bytes_received_per_rank_per_step"] = int((world_seen - 1) * b0.big * b0.full.shape[1] * 4)
            selfcheck["gather_buffer_bytes"] = int(b0.full.numel() * 4)

    # transparency legs: the same workload in the other arithmetic modes (not part of `value`)
    legs, leg_kernel_ms = {}, {}
    if not a.stub and not a.no_legs:
        for prec in ("f32", "bf16x3", "f16x2", "bf16", "f16"):
            if prec == a.precision:
                continue
            net.net.precision = prec
            for _ in range(max(1, min(a.warmup, 3))):
                step()
            k = max(2, min(a.steps // 2, 50))
            evl = HipEvents(k)
            L.ev2h_profile_set(PROFILED_TAG.encode(), evl.start, evl.stop, evl.n)
            dtf = timed(k)
            L.ev2h_profile_set(None, None, None, 0)
            leg_kernel_ms[prec] = evl.elapsed_ms(k)
            legs[prec] = {"value": round(gB * k / dtf, 2), "ms_per_step": round(dtf / k * 1e3, 3), "steps": k, "dtype": DTYPE[prec]}
        net.net.precision = a.precision

    # host-side boundary: the same step with its input handed over in (pinned) HOST memory and its predictions returned to host
    # memory, in stream order on the forward's stream (no copy/compute overlap: the pessimistic figure).  Reported beside `value`,
    # never as `value`: the path's own boundary takes device buffers (TEHNet.forward(xyz) with xyz on the device, demo.py:24-33)
    host_io = None
    if not a.stub and not a.no_host_io and world == 1 and not a.collision:
        try:
            k = max(4, min(a.steps // 4, 24))
            W = evdist.packed_width(N)
            hx = xyz.detach().cpu().pin_memory()
            dxs = [torch.empty_like(xyz) for _ in range(2)]
            drows = [torch.empty(B, W, device=dev, dtype=torch.float32) for _ in range(2)]
            hrows = [torch.empty(B, W, dtype=torch.float32).pin_memory() for _ in range(2)]
            main = torch.cuda.current_stream(dev)
            cin, cout = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

            def fwd(b):
                net.net.fps_init = inits
                with torch.no_grad():
                    net.net(dxs[b], net.hands, rows=drows[b])

            def in_order(i):
                dxs[0].copy_(hx, non_blocking=True)
                fwd(0)
                hrows[0].copy_(drows[0], non_blocking=True)

            ev_in = [torch.cuda.Event() for _ in range(2)]
            ev_fwd = [torch.cuda.Event() for _ in range(2)]
            ev_out = [torch.cuda.Event() for _ in range(2)]

            def overlapped(i):
                # input of step i and output of step i - 1 travel on their own streams under the forward of a neighbouring step
                b = i & 1
                with torch.cuda.stream(cin):
                    cin.wait_event(ev_fwd[b])                  # the forward that last read dxs[b] (step i - 2)
                    dxs[b].copy_(hx, non_blocking=True)
                    ev_in[b].record(cin)
                main.wait_event(ev_in[b])
                main.wait_event(ev_out[b])                     # drows[b] of step i - 2 has left
                fwd(b)
                ev_fwd[b].record(main)
                with torch.cuda.stream(cout):
                    cout.wait_event(ev_fwd[b])
                    hrows[b].copy_(drows[b], non_blocking=True)
                    ev_out[b].record(cout)

            def run(fn):
                for i in range(2):
                    fn(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(k):
                    fn(i)
                torch.cuda.synchronize()
                return time.perf_counter() - t0

            dt_io, dt_ov = run(in_order), run(overlapped)
            host_io = {"value": round(B * k / dt_ov, 2), "unit": "event-windows/s", "ms_per_step": round(dt_ov / k * 1e3, 3), "steps": k,
                       "in_stream_order": {"value": round(B * k / dt_io, 2), "ms_per_step": round(dt_io / k * 1e3, 3)},
                       "bytes_h2d_per_step": int(hx.numel() * 4), "bytes_d2h_per_step": int(B * W * 4),
                       "note": "pinned host input -> device, forward, packed predictions -> pinned host; `value`: copies on their own streams "
                               "under the neighbouring steps' forwards (two buffers); in_stream_order: copy, forward, copy on one stream"}
        except Exception as e:  # noqa: BLE001 -- never lose the throughput line to a side leg
            host_io = {"error": f"{type(e).__name__}: {e}"}

    latency = None
    if not a.stub and not a.no_latency and world == 1:
        try:
            latency = latency_legs(net, Cc, N, a.cloud, dev)
        except Exception as e:  # noqa: BLE001 -- a latency leg must never lose the throughput line
            latency = {"error": f"{type(e).__name__}: {e}"}

    line = None
    if rank == 0:
        res = {
            "metric": f"event-windows/sec at B={B} N={N}" + ("" if a.stub else f" ({a.precision})"),
            "value": round(gB * a.steps / dt, 2),
            "unit": "event-windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "stub" if a.stub else DTYPE[a.precision], "data": "synthetic",
            "config": {"workload": f"TEHNet.forward+MANO both hands, B={B}/GPU N={N} C={Cc} fp32, {a.cloud}-clouds, "
                                   + ("random-init 342-key checkpoint" if a.weights == "random" else
                                      "342-key checkpoint out of the reference's training loop on synthetic clouds (tests/trained_ckpt.py)")
                                   + ", synthetic MANO-shaped assets"
                                   + (" + intersection-loss term per window (pair search cap 16, list sized 2 x 1538 x 16: never truncated; conic "
                                      "distance-field penalty)" + (" + collision count for the score (second search, cap 8)" if a.collision_score else "") +
                                      "; hand meshes: " +
                                      ("mitten-shaped surfaces with smooth skinning (tens to hundreds of colliding pairs per window)" if a.collision_mesh == "surface"
                                       else "random triangle soup (every mesh intersects itself ~24 000 times: worst case for the pair search)")
                                      if a.collision else ""),
                       "global_batch": gB, "points": N, "channels": Cc, "precision": a.precision, "forwards_in_flight": a.inflight,
                       "world_size_seen": world_seen, "backend": ("gloo" if (a.stub or a.shared_device) else "nccl (RCCL)") if use_dist else None,
                       "parallelism": f"batch-shard x{world}" + ((" + in-place RCCL all-gather of predictions (the forward writes into the gather buffer; " +
                                                                       ("in stream order)" if sync_gather else "asynchronous, overlapped with the next forward)")) if use_dist else "")},
        }
        if a.shared_device:
            res["rehearsal"] = (f"{world} ranks time-slicing ONE GPU under a gloo process group (host-staged gather): the launcher, sharding, barriers, "
                                "max-over-ranks timing and every per-rank field run as in the real run, RCCL's transport does not; `value` is not a measurement")
        if a.stub:
            res["stub"] = True
            res["would_emit"] = expected_line_keys(world, a)      # what the real path prints at this world size (tests/test_bench_launcher.py)
            res["gathered_rows"] = int(last["class_logits"].shape[0]) if last is not None else None
            res["gathered_rank_ids"] = sorted({int(v) for v in last["class_logits"][:, 0, 0].tolist()}) if last is not None else None
        else:
            live, live_src = None, None
            if world == 1 and not a.no_traffic:
                try:
                    live = live_pmc_traffic(a)
                    live_src = "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace passes of a 3-forward child run"
                except Exception as e:  # noqa: BLE001 -- never lose the throughput line to the profiler
                    live_src = f"live PMC passes failed ({type(e).__name__}: {str(e)[:200]})"
            if live is None:
                t, src = committed_pmc_traffic(a.precision, B, N)
                if t is not None:
                    live, live_src = t, (live_src + "; " if live_src else "") + f"committed profile profiles/{src} (not measured in this run)"
            res["ms_per_step_per_rank"] = {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms}
            if selfcheck:
                res["multi_gpu_selfcheck"] = selfcheck
            if use_dist:
                try:
                    res["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:  # noqa: BLE001
                    res["config"]["rccl_version"] = None
            live_is_site1 = bool(live) and ("kernel_launches_sampled" in live or site1["tag"] == PROFILED_TAG)
            res["roofline"] = roofline_entry(a.precision, B, main_kernel_ms, live["kernel"] if live_is_site1 else None,
                                             live_src if live_is_site1 else None, site=site1, n_points=N)
            if second_kernel_ms:
                t2 = live["sites"][site2["tag"]]["bytes"] if live and "sites" in live and site2["tag"] in live["sites"] else None
                res["roofline_second"] = roofline_entry(a.precision, B, second_kernel_ms, t2, live_src if t2 is not None else None, site=site2, n_points=N)
            res["hbm"] = hbm_entry(B, N, dt / a.steps * 1e3, live.get("step") if live else None, live_src)
            f32_leg = legs.pop("f32", None)
            if f32_leg:
                res["f32_mfma_leg"] = f32_leg
                res["roofline_f32"] = roofline_entry("f32", B, leg_kernel_ms["f32"], n_points=N)
            if "bf16" in legs:                  # BASELINE.json config 3 names this arithmetic: its own roofline entry
                res["roofline_bf16"] = roofline_entry("bf16", B, leg_kernel_ms["bf16"], n_points=N)
            if "f16" in legs:                   # [r6] the reduced-precision mode that holds on a trained checkpoint (config 3 in practice)
                res["roofline_f16"] = roofline_entry("f16", B, leg_kernel_ms["f16"], n_points=N)
            if legs:
                res["other_modes"] = legs
            if latency:
                res["latency_ms"] = latency
            if host_io:
                res["pcie_inclusive"] = host_io
            if sustained:
                res["value_sustained"] = sustained
            if not a.no_cpu_baseline:
                # rank 0 times the CPU port at EVERY world size (the other ranks wait at the final barrier, inside the RCCL timeout):
                # a multi-GPU line without the key would be graded "unmeasured" 
                res["cpu_baseline"] = cpu_baseline(sd, assets, Cc, N, a.cloud, a.cpu_seconds)
                if world > 1:
                    res["cpu_baseline"]["note"] = f"timed on rank 0's host while the other {world - 1} ranks idle at the final barrier"
            want = expected_line_keys(world, a)
            missing = [k for k in want if k not in res and k != "line_complete"]
            res["line_complete"] = True if not missing else {"missing": missing}
        line = json.dumps(res)
    # RCCL writes its version banner to the C-level stdout, which -- when stdout is a pipe or a file -- sits in libc's buffer until
    # exit and would land AFTER the result (from any rank: a launcher merges the ranks' stdout).  Every rank flushes it, then a
    # barrier, then the group is destroyed, and only then rank 0 prints: the JSON line is the last thing on stdout.
    def flush_c_stdout():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    flush_c_stdout()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        flush_c_stdout()
        print(line, flush=True)
    return 0


def latency_legs(net, Cc, N, cloud, dev):
    """Latency of ONE forward at the reference's operating point (demo.py:24-33 times one batched forward between device
    synchronisations): B = 1 and B = 8, eager launches and hipGraph replay of the captured ev2h_forward."""
    import torch
    from ev2hands_amd import synth
    out = {}
    for b in (1, 8):
        x = synth.synth_cloud(cloud, b, Cc, N, seed=77).to(dev)
        inits = synth.fps_inits(b, N, 77)

        def once():
            net.net.fps_init = inits
            with torch.no_grad():
                return net(x)

        def med(fn, n=30):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            return round(ts[len(ts) // 2], 4)

        for _ in range(3):
            once()
        entry = {"eager": med(once)}
        if hasattr(net.net, "capture"):
            g = net.net.capture(x, net.hands, inits)
            for _ in range(3):
                g.replay()
            entry["hipgraph"] = med(g.replay)
        out[f"B={b}"] = entry
    out["note"] = ("median wall ms of one forward between device synchronisations, N=%d C=%d, mode %s" % (N, Cc, net.net.precision))
    return out


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a, argv)
    return run_rank(a)


if __name__ == "__main__":
    sys.exit(main())
