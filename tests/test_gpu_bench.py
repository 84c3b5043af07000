"""bench.py's one-line contract on a real GPU: a short run of the real (non-stub) path, checked field by field.

The CPU suite covers the launcher and the gather with `--stub`; the fields below only exist on the HIP path (HIP-event kernel
times, the second launch site, the per-rank step times, the host-buffer leg), so they are checked here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra, traffic=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "2", "--batch", "16", "--no-legs", "--no-latency",
           "--no-cpu-baseline", *([] if traffic else ["--no-traffic"]), *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE JSON line, and it is the last thing on stdout
    assert r.stdout.strip().splitlines()[-1] == lines[0]
    return json.loads(lines[0])


def test_bench_line_fields_on_the_hip_path():
    j = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 12 and j["warmup"] == 2 and j["unit"] == "event-windows/s" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    # value = windows of the timed region / its wall time
    assert abs(j["value"] - 16 * 1e3 / j["ms_per_step"]) <= 0.01 * j["value"]
    # the per-rank step time is the timed region's (it once reported the last short side run's)
    pr = j["ms_per_step_per_rank"]
    assert len(pr["all"]) == 1 and abs(pr["max"] - j["ms_per_step"]) <= 0.05 * j["ms_per_step"], (pr, j["ms_per_step"])
    for key in ("roofline", "roofline_second"):
        r = j[key]
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.0 < r["frac"] < 1.0 and r["kernel_ms"] > 0.0, r
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert j["roofline"]["launch_site"] != j["roofline_second"]["launch_site"]
    # --no-traffic at a shape without a committed PMC profile (16 windows): no figure, not another workload's
    assert j["roofline"]["traffic"] is None and j["hbm"]["bytes_per_step"] is None and j["hbm"]["frac"] is None, (j["roofline"], j["hbm"])
    io = j["pcie_inclusive"]
    assert "error" not in io, io
    assert 0.0 < io["value"] <= 1.10 * j["value"] and 0.0 < io["in_stream_order"]["value"] <= 1.10 * j["value"], (io, j["value"])
    assert io["bytes_h2d_per_step"] == 16 * 4 * 2048 * 4


def test_bench_forced_single_rank_rccl_path():
    """the multi-GPU code path (process group, gather pipeline, self-checks) with one rank"""
    env_key = "EV2H_BENCH_FORCE_DIST"
    old = os.environ.get(env_key)
    os.environ[env_key] = "1"
    try:
        j = _run("--no-host-io", "--sustained-seconds", "1.2")
    finally:
        if old is None:
            os.environ.pop(env_key, None)
        else:
            os.environ[env_key] = old
    assert j["config"]["backend"] == "nccl (RCCL)" and j["config"]["world_size_seen"] == 1
    # [r6] the line of the multi-GPU code path is complete by its own definition (bench.py: expected_line_keys) ...
    assert j["line_complete"] is True, j["line_complete"]
    # ... its roofline says where the traffic figure comes from (no PMC child run here: the committed profile) ...
    assert j["roofline"]["traffic_kind"] in ("committed", None) and "mfma_ceiling" in j["roofline"]["sustained_source"]
    if j["roofline"]["traffic"] is not None:
        assert "not measured in this run" in j["roofline"]["traffic_source"]
    # ... and the sustained leg ran for >= 1 s with the shader clock observed during it
    vs = j["value_sustained"]
    assert vs["seconds"] >= 1.0 and 0.5 < vs["ratio_to_value"] < 1.5 and vs["shader_clock_mhz"]["samples"] >= 8, vs
    assert 500.0 < vs["shader_clock_mhz"]["median"] < 3000.0 and 500.0 < vs["shader_clock_mhz_idle"] < 3000.0, vs
    assert len(vs["shader_clock_mhz_median_per_rank"]) == 1
    sc = j["multi_gpu_selfcheck"]
    assert sc["gather_ms"] >= 0.0 and len(sc["two_stream_gain_per_rank"]) == 1
    assert abs(j["ms_per_step_per_rank"]["max"] - j["ms_per_step"]) <= 0.05 * j["ms_per_step"]


def test_bench_live_hbm_traffic_counts_the_forwards_only():
    """the rocprofv3 PMC child passes must see the step's forwards and nothing else (a side leg's extra forwards once inflated the
    per-step bytes five-fold): per window the step moves ~33 MB at B = 256; at B = 16 the 18 MB of weights are shared by fewer windows"""
    j = _run(traffic=True)
    hbm = j["hbm"]
    if "measured in this run" not in str(hbm.get("source", "")):
        pytest.skip(f"live PMC passes unavailable on this box: {hbm.get('source')}")
    per_window = hbm["bytes_per_step"] / 16
    assert 10e6 < per_window < 90e6, per_window
    t = j["roofline"]["traffic"]
    assert t is None or 1e6 < t < 400e6, t


@pytest.mark.parametrize("precision", ["f16x2", "bf16"])
def test_bench_live_traffic_at_n8192_names_the_query_convolution(precision):
    """BASELINE config 5's window size: the dominant launch is the k = 3 query convolution, whose traced kernel name depends on
    the arithmetic mode and on template arguments added over the rounds -- bench.py once matched none of its dispatches and
    published a `measured` 0 bytes.  Both launch sites must now carry their own PMC traffic (or the source must say why not)."""
    j = _run("--points", "8192", "--batch", "4", "--precision", precision, "--no-host-io", traffic=True)
    assert j["roofline"]["launch_site"] == "qconv0" and j["roofline_second"]["launch_site"] == "sa2.1"
    if "measured in this run" not in str(j["hbm"].get("source", "")):
        pytest.skip(f"live PMC passes unavailable on this box: {j['hbm'].get('source')}")
    # the GEMM reads 4 windows x 8192 rows x 256 fp32 (33.5 MB) at least once and writes nothing but partial sums
    t1, t2 = j["roofline"]["traffic"], j["roofline_second"]["traffic"]
    assert t1 is not None and 20e6 < t1 < 400e6, t1
    assert t2 is not None and 1e6 < t2 < 400e6, t2


def test_bench_line_carries_the_cpu_baseline_on_the_multi_gpu_path():
    """the forced 1-rank RCCL path WITH the CPU leg (bounded to a few seconds): the key a SCALE line was missing"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EV2H_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--batch", "8", "--no-legs", "--no-latency", "--no-traffic",
           "--no-host-io", "--no-second-site", "--no-selfcheck", "--sustained-seconds", "0.3", "--cpu-seconds", "3"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert j["line_complete"] is True and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1


@pytest.mark.parametrize("world,inflight", [(2, 1), (4, 2)])
def test_multi_rank_rehearsal_on_one_gpu_prints_a_complete_line(world, inflight):
    """[r6] `bench.py --gpus N --shared-device`: the launcher starts N real ranks that all run the HIP forward on device 0 under a gloo
    process group (RCCL refuses two ranks on one device; the gather goes through dist.py's host-staged transport).  Everything the
    first real N-GPU run will execute except RCCL's transport: sharding of the global FPS starts, barriers, max-over-ranks timing,
    per-rank step times / shader clocks / side-stream probes, the CPU leg on rank 0 after the last barrier, the key set."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--shared-device", "--steps", "5", "--warmup", "1", "--batch", "8",
           "--inflight", str(inflight), "--no-legs", "--no-latency", "--no-traffic", "--no-host-io", "--no-second-site", "--sustained-seconds", "0.3",
           "--cpu-seconds", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-1000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["config"]["world_size_seen"] == world and j["config"]["backend"] == "gloo" and "rehearsal" in j
    assert j["line_complete"] is True, j["line_complete"]
    assert j["config"]["global_batch"] == 8 * world and j["config"]["forwards_in_flight"] == inflight
    assert len(j["ms_per_step_per_rank"]["all"]) == world and all(v > 0 for v in j["ms_per_step_per_rank"]["all"])
    assert abs(j["value"] - 8 * world * 1e3 / j["ms_per_step"]) <= 0.01 * j["value"]
    assert len(j["value_sustained"]["shader_clock_mhz_median_per_rank"]) == world
    sc = j["multi_gpu_selfcheck"]
    assert len(sc["two_stream_gain_per_rank"]) == world and len(sc["gather_ms_per_rank"]) == world, sc
    assert j["cpu_baseline"]["value"] > 0
    # 8 windows per rank: no committed PMC profile of THAT workload -- the line must say so instead of quoting the 256-window step's bytes
    assert j["roofline"]["traffic"] is None and j["roofline"]["traffic_kind"] is None and j["hbm"]["bytes_per_step"] is None
