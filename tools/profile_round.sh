#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r3
# kernel-trace summaries for the four arithmetic modes (f16x2 also single-stream: the table whose per-kernel averages are
# comparable with bench.py's HIP-event time), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes; per kernel launch and per whole step) for f16x2, bf16 and f32, SQ counters for f16x2 and bf16, and the evidence for
# the sustained matrix-pipe ceiling (mfma_power: pure MFMA loop on zeros vs random operands with the observed clock; kbench: the
# real SA kernels on zeros vs random data).  Outputs under gpurun_out/<tag>_*; the summaries to commit are copied to profiles/.
set -u
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
PY=$(python3 -c "import sys; print(sys.executable)")   # the interpreter itself follows `--`; a symlink is not an exec hop and resolving it would leave a virtualenv
BENCH="bench.py --steps 7 --warmup 3 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-host-io --no-sustained"
SMALL="bench.py --steps 2 --warmup 1 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-host-io --no-sustained"
for prec in f16x2 f32 bf16x3 bf16 f16; do
  rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_$prec -o bench -- $PY $BENCH --precision $prec > $O/${TAG}_ktlog_$prec.txt 2>&1
  python tools/rocpd_summary.py $(ls $O/${TAG}_kt_$prec/*/*.db $O/${TAG}_kt_$prec/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats_$prec.txt 2>&1
done
export EV2H_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_f16_ss -o bench -- $PY $BENCH --precision f16 > $O/${TAG}_ktlog_f16_single_stream.txt 2>&1
python tools/rocpd_summary.py $(ls $O/${TAG}_kt_f16_ss/*/*.db $O/${TAG}_kt_f16_ss/*.db 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats_f16_single_stream.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_f16x2_ss -o bench -- $PY $BENCH --precision f16x2 > $O/${TAG}_ktlog_f16x2_single_stream.txt 2>&1
python tools/rocpd_summary.py $(ls $O/${TAG}_kt_f16x2_ss/*/*.db $O/${TAG}_kt_f16x2_ss/*.db 2>/dev/null | head -1) --sites "128, 196, 256" 3 > $O/${TAG}_bench_kernel_stats_f16x2_single_stream.txt 2>&1
# the HIP-event time of site 0 (sa2.1) measured by bench.py in the SAME profiled command, next to the trace's per-site averages
grep -o '"kernel_ms": [0-9.]*' $O/${TAG}_ktlog_f16x2_single_stream.txt | head -1 | sed 's/^/# bench.py HIP events around site sa2.1 (the table form) in the same command: /' >> $O/${TAG}_bench_kernel_stats_f16x2_single_stream.txt
unset EV2H_TWO_STREAMS
for prec in f16x2 bf16 f32 f16; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch_$prec -o p -- $PY $SMALL --precision $prec > $O/${TAG}_pmc_fetch_$prec.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write_$prec -o p -- $PY $SMALL --precision $prec > $O/${TAG}_pmc_write_$prec.log 2>&1
  python tools/pmc_traffic.py $(ls $O/${TAG}_pmc_fetch_$prec/*/*.db $O/${TAG}_pmc_fetch_$prec/*.db 2>/dev/null | head -1) $(ls $O/${TAG}_pmc_write_$prec/*/*.db $O/${TAG}_pmc_write_$prec/*.db 2>/dev/null | head -1) $prec $O/${TAG}_pmc_hbm_traffic_$prec.json ${TAG//[^0-9]/} 3 > /dev/null 2>&1
done
for prec in f16x2 bf16 f16; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/${TAG}_pmc_sq_a_$prec -o p -- $PY $SMALL --precision $prec > $O/${TAG}_pmc_sq_a_$prec.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/${TAG}_pmc_sq_b_$prec -o p -- $PY $SMALL --precision $prec > $O/${TAG}_pmc_sq_b_$prec.log 2>&1
  { python tools/pmc_report.py $(ls $O/${TAG}_pmc_sq_a_$prec/*/*.db $O/${TAG}_pmc_sq_a_$prec/*.db 2>/dev/null | head -1) ; python tools/pmc_report.py $(ls $O/${TAG}_pmc_sq_b_$prec/*/*.db $O/${TAG}_pmc_sq_b_$prec/*.db 2>/dev/null | head -1) ; } > $O/${TAG}_pmc_sq_$prec.txt 2>&1
done
# sustained-ceiling evidence
{
  echo "## tools/ubench/mfma_power.hip: pure v_mfma_f32_32x32x16_f16 loop, every CU busy"
  hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/mfma_power.hip -o $O/mfma_power 2>/dev/null && $O/mfma_power
  echo
  echo "## KBENCH_PREC=f16x2 python tools/kbench.py sab   (the fused set-abstraction kernels at the bench shapes, random data)"
  KBENCH_PREC=f16x2 python tools/kbench.py sab
  echo
  echo "## KBENCH_ZERO=1 KBENCH_PREC=f16x2 python tools/kbench.py sab   (same instruction stream, all-zero data)"
  KBENCH_ZERO=1 KBENCH_PREC=f16x2 python tools/kbench.py sab
} > $O/${TAG}_mfma_ceiling.txt 2>&1
rm -f $O/mfma_power
# the raw rocprofv3 databases are tens of MiB each and gpurun copies back at most 64 MiB: keep the summaries only
rm -rf $O/${TAG}_kt_* $O/${TAG}_pmc_fetch_* $O/${TAG}_pmc_write_* $O/${TAG}_pmc_sq_a_* $O/${TAG}_pmc_sq_b_*
ls -la $O | grep ${TAG}_ | head -60
