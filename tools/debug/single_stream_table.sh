cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; TAG=r5
PY=$(python3 -c "import sys; print(sys.executable)")
BENCH="bench.py --steps 7 --warmup 3 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-host-io --no-sustained"
export EV2H_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_f16x2_ss -o bench -- $PY $BENCH --precision f16x2 > $O/${TAG}_ktlog_f16x2_single_stream.txt 2>&1
python tools/rocpd_summary.py $(ls $O/${TAG}_kt_f16x2_ss/*/*.db $O/${TAG}_kt_f16x2_ss/*.db 2>/dev/null | head -1) --sites "128, 196, 256" 3 > $O/${TAG}_bench_kernel_stats_f16x2_single_stream.txt 2>&1
grep -o '"kernel_ms": [0-9.]*' $O/${TAG}_ktlog_f16x2_single_stream.txt | head -1 | sed 's/^/# bench.py HIP events around site sa2.1 (the table form) in the same command: /' >> $O/${TAG}_bench_kernel_stats_f16x2_single_stream.txt
rm -rf $O/${TAG}_kt_f16x2_ss
tail -8 $O/${TAG}_bench_kernel_stats_f16x2_single_stream.txt
