#!/bin/bash
# Everything measured for the round, on ONE box (run through gpurun from the repo root; outputs under gpurun_out/, the summaries that
# are committed are copied to profiles/):  the float64 truth report and the precision report on the trained checkpoints, the whole GPU
# suite, the bench lines, the rocprofv3 kernel tables / PMC traffic / SQ counters for N = 2048 and N = 8192, the batch sweep at 8192
# points, forwards in flight.
python tests/trained_truth_report.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_trained_truth_report.txt
python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r5_gpu_tests.txt
bash tools/bench_lines.sh r5 > gpurun_out/r5_bench_lines_summary.txt 2>&1
bash tools/profile_round.sh r5 > gpurun_out/r5_profile_round.log 2>&1
bash tools/profile_n8192.sh r5 > /dev/null 2>&1
bash tools/n8192_sweep.sh r5 _final > /dev/null 2>&1
python tools/trained_precision_report.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_trained_precision_report.txt
{ python tools/debug/inflight.py 16 8192; python tools/debug/inflight.py 128 8192; python tools/debug/inflight.py 256 2048; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_inflight.txt
python bench.py --points 8192 --batch 16 --steps 200 --inflight 2 --no-cpu-baseline --no-legs --no-latency --no-traffic 2>/dev/null | tail -1 > gpurun_out/r5_bench_line_n8192_b16_inflight2.json
cat gpurun_out/r5_bench_lines_summary.txt; tail -3 gpurun_out/r5_gpu_tests.txt; cat gpurun_out/r5_batch_sweep_n8192_final.txt; cat gpurun_out/r5_inflight.txt
