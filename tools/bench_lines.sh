#!/bin/bash
# The bench lines committed under profiles/ for a round:  bash tools/bench_lines.sh r5   (on the GPU box, from the repo root)
TAG=${1:-r6}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
Q="--no-cpu-baseline --no-legs --no-latency"
python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench_line_default_f16x2.json
python bench.py --precision bf16 $Q 2>/dev/null | tail -1 > $O/${TAG}_bench_line_bf16.json
python bench.py --precision f16 $Q 2>/dev/null | tail -1 > $O/${TAG}_bench_line_f16.json          # [r6] the reduced-precision mode to use (config 3)
python bench.py --precision bf16x3 $Q 2>/dev/null | tail -1 > $O/${TAG}_bench_line_bf16x3.json
python bench.py --cloud U $Q 2>/dev/null | tail -1 > $O/${TAG}_bench_line_U.json
python bench.py --points 8192 --batch 128 --steps 100 $Q 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192.json
python bench.py --points 8192 --batch 128 --steps 100 --collision $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_collision.json
python bench.py --points 8192 --batch 128 --steps 60 --collision --collision-mesh soup $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_collision_soup.json
# config 5 read as an 8-GPU shard (B = 128 over 8 ranks): 16 windows of 8192 points per rank
python bench.py --points 8192 --batch 16 --steps 200 $Q 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16.json
python bench.py --points 8192 --batch 16 --steps 200 --collision $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_collision.json
python bench.py --points 8192 --batch 16 --steps 200 --collision --inflight 2 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_collision_inflight2.json
EV2H_BENCH_FORCE_DIST=1 python bench.py $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_force_dist.json
# [r6] the multi-GPU code path as the driver's N > 1 run prints it (cpu_baseline included), and with forwards in flight inside the gather pipeline
EV2H_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 3 --no-legs --no-latency --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_force_dist_driver_args.json
EV2H_BENCH_FORCE_DIST=1 python bench.py --points 8192 --batch 16 --steps 200 --inflight 2 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_force_dist_inflight2.json
python bench.py --points 8192 --batch 16 --steps 200 --inflight 2 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_inflight2.json
python bench.py --points 8192 --batch 16 --steps 200 --inflight 3 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_inflight3.json
python bench.py --precision f16 --points 8192 --batch 16 --steps 200 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_f16.json
python bench.py --precision f16 --points 8192 --batch 16 --steps 200 --inflight 2 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_n8192_b16_f16_inflight2.json
python bench.py --inflight 2 $Q --no-traffic 2>/dev/null | tail -1 > $O/${TAG}_bench_line_default_f16x2_inflight2.json
for f in $O/${TAG}_bench_line_*.json; do python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', j['value'], j['ms_per_step'], j.get('pcie_inclusive',{}).get('value'), j['roofline']['frac'])"; done
