# single-stream kernel tables of a list of modes:  bash tools/debug/single_stream_modes.sh r6 "f16 bf16" [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; TAG=${1:-r6}; MODES=${2:-"f16 bf16"}; EXTRA=${3:-}
PY=$(python3 -c "import sys; print(sys.executable)")
BENCH="bench.py --steps 7 --warmup 3 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-sustained --no-host-io $EXTRA"
export EV2H_TWO_STREAMS=0
for m in $MODES; do
  rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_${m}_ss -o bench -- $PY $BENCH --precision $m > $O/${TAG}_ktlog_${m}_single_stream.txt 2>&1
  python tools/rocpd_summary.py $(ls $O/${TAG}_kt_${m}_ss/*/*.db $O/${TAG}_kt_${m}_ss/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats_${m}_single_stream.txt 2>&1
  rm -rf $O/${TAG}_kt_${m}_ss
done
