"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus 2` invoked DIRECTLY (no torchrun) must start the two ranks
itself, run the gloo all-gather and the max-over-ranks timing, and relay rank 0's single JSON line (--stub: fabricated
predictions, no GPU).  Also: the same entry point under an external launcher environment, and the WORLD_SIZE mismatch error."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""          # the launcher path must not need a GPU
    return env


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_direct_invocation_launches_its_own_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--batch", "3", "--points", "128"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["stub"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["world_size_seen"] == 2 and d["config"]["global_batch"] == 6
    assert d["gathered_rows"] == 6 and d["gathered_rank_ids"] == [0, 1]       # both shards arrived through the all-gather
    assert d["value"] > 0 and d["scaling"] == "weak" and d["unit"] == "event-windows/s"
    assert "roofline" not in d and "cpu_baseline" not in d                     # a stub line can never pass for a measurement
    # ... but it lists what the REAL path prints at this world size (bench.py: expected_line_keys, checked by the real run itself
    # as `line_complete`): a multi-GPU line carries the roofline, the CPU baseline and the sustained leg like the 1-GPU line does
    for k in ("roofline", "roofline_second", "hbm", "cpu_baseline", "value_sustained", "multi_gpu_selfcheck", "ms_per_step_per_rank", "line_complete"):
        assert k in d["would_emit"], k


def test_eight_ranks_as_the_driver_will_run_it():
    """World size 8 (the driver's largest run) through the same plumbing: launcher, pipelined all-gather, max-over-ranks timing."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "4", "--warmup", "2", "--stub", "--batch", "2", "--points", "128"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 8 and d["config"]["world_size_seen"] == 8 and d["config"]["global_batch"] == 16
    assert d["gathered_rows"] == 16 and d["gathered_rank_ids"] == list(range(8))
    for k in ("roofline", "cpu_baseline", "value_sustained", "multi_gpu_selfcheck"):           # the first SCALE line must be gradeable
        assert k in d["would_emit"], k


def test_expected_line_keys_follow_the_switches():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--gpus", "8", "--no-cpu-baseline", "--no-sustained"])
    keys = bench.expected_line_keys(8, a)
    assert "cpu_baseline" not in keys and "value_sustained" not in keys and "roofline" in keys and "multi_gpu_selfcheck" in keys
    assert bench.newest_profile("r*_mfma_ceiling.txt") is not None            # roofline.sustained_source names the newest round's file


def test_under_torch_distributed_run_exactly_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`:
    the ranks exist already, bench.py must run as a rank (no launcher of its own) and exactly one JSON line must come out."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--batch", "2", "--points", "128"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["gathered_rows"] == 4 and d["gathered_rank_ids"] == [0, 1]


def test_in_order_gather_variant_of_the_step():
    """EV2H_BENCH_SYNC_GATHER=1: one gather buffer, the all-gather in stream order (the default alternates two buffers and issues it
    asynchronously); both variants must deliver every shard."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub", "--batch", "2", "--points", "128"],
                       env=dict(_clean_env(), EV2H_BENCH_SYNC_GATHER="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["gathered_rows"] == 4 and d["gathered_rank_ids"] == [0, 1] and "in stream order" in d["config"]["parallelism"]


def test_single_rank_stub_line_and_metric_string():
    p = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--stub", "--batch", "2", "--points", "256"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 1 and d["metric"] == "event-windows/sec at B=2 N=256"
    assert d["config"]["points"] == 256 and d["config"]["backend"] is None


def test_world_size_mismatch_is_a_clear_error():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--stub"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_a_rank_that_dies_ends_the_run_promptly_and_its_log_is_shown():
    """the launcher used to block on rank 0, which sits in the rendezvous until its timeout when a peer is gone.  Now the
    first non-zero exit terminates the other ranks and the failed rank's log tail is printed."""
    import time
    env = dict(_clean_env(), EV2H_BENCH_STUB_FAIL_RANK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub", "--batch", "2", "--points", "128"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and time.time() - t0 < 120
    assert "simulated start-up failure" in p.stderr and "rank 1" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]              # no result line from a failed run
    assert os.path.exists(os.path.join(ROOT, "gpurun_out", "bench_rank1.log"))
