#!/bin/bash
# One / two / three forwards in flight in the plain process and in the RCCL process (EV2H_BENCH_FORCE_DIST=1), and the headline shape:
# does every forward's side stream really run beside its caller stream, whatever the process created before?  (ev2h_bind_stream)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
Q="--no-cpu-baseline --no-legs --no-latency --no-traffic --no-second-site --no-host-io --no-sustained"
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=j.get('multi_gpu_selfcheck') or {}; print('$1', j['value'], j['ms_per_step'], 'two_stream_gain', sc.get('two_stream_gain'), 'probe', sc.get('side_stream_probe_per_rank'))"; }
for r in 1 2; do
  for k in 1 2 3; do
    python bench.py --points 8192 --batch 16 --steps 200 --inflight $k $Q 2>/dev/null | line "plain 16x8192 inflight=$k"
    EV2H_BENCH_FORCE_DIST=1 python bench.py --points 8192 --batch 16 --steps 200 --inflight $k $Q 2>/dev/null | line "rccl  16x8192 inflight=$k"
  done
  python bench.py --steps 100 $Q 2>/dev/null | line "plain 256x2048"
  EV2H_BENCH_FORCE_DIST=1 python bench.py --steps 100 $Q 2>/dev/null | line "rccl  256x2048"
done
