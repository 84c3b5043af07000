#!/bin/bash
# Config 5 (N = 8192, B = 128 per GPU) kernel table + HBM traffic:  bash tools/profile_n8192.sh r4
set -u
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
PY=$(python3 -c "import sys; print(sys.executable)")   # the interpreter itself follows `--`; a symlink is not an exec hop and resolving it would leave a virtualenv
ARGS="--points 8192 --batch 128 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-host-io --no-sustained"
export EV2H_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt8k -o bench -- $PY bench.py --steps 7 --warmup 3 $ARGS > $O/${TAG}_ktlog_n8192.txt 2>&1
python tools/rocpd_summary.py $(ls $O/${TAG}_kt8k/*/*.db $O/${TAG}_kt8k/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats_n8192_f16x2_single_stream.txt 2>&1
unset EV2H_TWO_STREAMS
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pf8k -o p -- $PY bench.py --steps 2 --warmup 1 $ARGS > $O/${TAG}_pf8k.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pw8k -o p -- $PY bench.py --steps 2 --warmup 1 $ARGS > $O/${TAG}_pw8k.log 2>&1
python tools/pmc_traffic.py $(ls $O/${TAG}_pf8k/*/*.db $O/${TAG}_pf8k/*.db 2>/dev/null | head -1) $(ls $O/${TAG}_pw8k/*/*.db $O/${TAG}_pw8k/*.db 2>/dev/null | head -1) f16x2 $O/${TAG}_pmc_hbm_traffic_n8192_f16x2.json ${TAG//[^0-9]/} 3 > /dev/null 2>&1
rm -rf $O/${TAG}_kt8k $O/${TAG}_pf8k $O/${TAG}_pw8k
