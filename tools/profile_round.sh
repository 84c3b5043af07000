#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r2
# kernel-trace summaries for the three fp32-class modes, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md prescribes) for f16x2 and f32, SQ counters for f16x2.  Outputs under gpurun_out/<tag>_*; the summaries to
# commit are copied to profiles/ by hand.
set -u
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
BENCH="bench.py --steps 7 --warmup 3 --no-legs --no-latency --no-cpu-baseline"
SMALL="bench.py --steps 2 --warmup 1 --no-legs --no-latency --no-cpu-baseline"
for prec in f16x2 f32 bf16x3; do
  rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_$prec -o bench -- python3 $BENCH --precision $prec > $O/${TAG}_kt_$prec.log 2>&1
  python tools/rocpd_summary.py $(ls $O/${TAG}_kt_$prec/*.db | head -1) > $O/${TAG}_kernel_stats_$prec.txt 2>&1
done
for prec in f16x2 f32; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch_$prec -o p -- python3 $SMALL --precision $prec > $O/${TAG}_pmc_fetch_$prec.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write_$prec -o p -- python3 $SMALL --precision $prec > $O/${TAG}_pmc_write_$prec.log 2>&1
  python tools/pmc_traffic.py $(ls $O/${TAG}_pmc_fetch_$prec/*.db | head -1) $(ls $O/${TAG}_pmc_write_$prec/*.db | head -1) $prec $O/${TAG}_pmc_hbm_traffic_$prec.json ${TAG#r} > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/${TAG}_pmc_sq_a -o p -- python3 $SMALL > $O/${TAG}_pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/${TAG}_pmc_sq_b -o p -- python3 $SMALL > $O/${TAG}_pmc_sq_b.log 2>&1
{ python tools/pmc_report.py $(ls $O/${TAG}_pmc_sq_a/*.db | head -1) ; python tools/pmc_report.py $(ls $O/${TAG}_pmc_sq_b/*.db | head -1) ; } > $O/${TAG}_pmc_sq_f16x2.txt 2>&1
ls -la $O | grep ${TAG}_ | head -40
