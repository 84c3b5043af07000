#!/bin/bash
# kernel table + PMC HBM traffic of ONE arithmetic mode (the per-mode part of tools/profile_round.sh):  bash tools/debug/profile_one_mode.sh bf16x3
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; TAG=r5
PY=$(python3 -c "import sys; print(sys.executable)")
BENCH="bench.py --steps 7 --warmup 3 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-sustained --no-host-io"
SMALL="bench.py --steps 2 --warmup 1 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-sustained --no-host-io"
prec=${1:-bf16x3}
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_$prec -o bench -- $PY $BENCH --precision $prec > $O/${TAG}_ktlog_$prec.txt 2>&1
python tools/rocpd_summary.py $(ls $O/${TAG}_kt_$prec/*/*.db $O/${TAG}_kt_$prec/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats_$prec.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch_$prec -o p -- $PY $SMALL --precision $prec > $O/${TAG}_pmc_fetch_$prec.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write_$prec -o p -- $PY $SMALL --precision $prec > $O/${TAG}_pmc_write_$prec.log 2>&1
python tools/pmc_traffic.py $(ls $O/${TAG}_pmc_fetch_$prec/*/*.db $O/${TAG}_pmc_fetch_$prec/*.db 2>/dev/null | head -1) $(ls $O/${TAG}_pmc_write_$prec/*/*.db $O/${TAG}_pmc_write_$prec/*.db 2>/dev/null | head -1) $prec $O/${TAG}_pmc_hbm_traffic_$prec.json ${TAG//[^0-9]/} 3 > /dev/null 2> $O/${TAG}_pmc_traffic_$prec.err
rm -rf $O/${TAG}_kt_* $O/${TAG}_pmc_fetch_* $O/${TAG}_pmc_write_*
head -12 $O/${TAG}_bench_kernel_stats_$prec.txt | cut -c1-160; head -c 600 $O/${TAG}_pmc_hbm_traffic_$prec.json; cat $O/${TAG}_pmc_traffic_$prec.err | tail -3
