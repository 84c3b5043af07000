#!/bin/bash
# Config 5 as an 8-GPU shard (16 windows of 8192 points per rank): batch sweep at N = 8192 + kernel table at B = 16.
#   bash tools/n8192_sweep.sh r5 [label]
set -u
TAG=${1:-r5}
LBL=${2:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
PY=$(python3 -c "import sys; print(sys.executable)")
COMMON="--points 8192 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-sustained --no-host-io"
{
echo "# N = 8192 batch sweep (windows per GPU per step), f16x2, E clouds: python bench.py --points 8192 --batch B"
for b in 1 4 16 32 64 128; do
  s=$((b<32?200:40))
  python bench.py --batch $b --steps $s $COMMON 2>/dev/null | tail -1 > /tmp/bs_line.json
  python - "$b" <<'PY'
import json, sys
d = json.load(open("/tmp/bs_line.json"))
print("B=%4d  %9.1f windows/s  %8.3f ms/step" % (int(sys.argv[1]), d["value"], d["ms_per_step"]))
PY
done
} > $O/${TAG}_batch_sweep_n8192${LBL}.txt 2>&1
export EV2H_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt8k16 -o bench -- $PY bench.py --steps 20 --warmup 3 --batch 16 $COMMON > $O/${TAG}_ktlog_n8192_b16.txt 2>&1
python tools/rocpd_summary.py $(ls $O/${TAG}_kt8k16/*/*.db $O/${TAG}_kt8k16/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats_n8192_b16_f16x2_single_stream${LBL}.txt 2>&1
unset EV2H_TWO_STREAMS
rm -rf $O/${TAG}_kt8k16
cat $O/${TAG}_batch_sweep_n8192${LBL}.txt
