cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
PY=$(python3 -c "import os,sys; print(os.path.realpath(sys.executable))")
SMALL="bench.py --steps 2 --warmup 1 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site"
for cfg in "f16x2:" "bf16:--precision bf16" "n8192_f16x2:--points 8192 --batch 128"; do
  tag=${cfg%%:*}; extra=${cfg#*:}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf_$tag -o p -- $PY $SMALL $extra > $O/pf_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw_$tag -o p -- $PY $SMALL $extra > $O/pw_$tag.log 2>&1
  python tools/pmc_traffic.py $(find $O/pf_$tag -name "*.db" | head -1) $(find $O/pw_$tag -name "*.db" | head -1) ${tag##*_} $O/r4_pmc_hbm_traffic_$tag.json 4 3 > /dev/null 2> $O/pmc_traffic_$tag.err
  rm -rf $O/pf_$tag $O/pw_$tag
done
python tests/fuzz_modes.py 300 404 > $O/r4_fuzz_modes.txt 2>&1; tail -2 $O/r4_fuzz_modes.txt
python tools/fuzz_gemm.py 400 > $O/r4_fuzz_gemm.txt 2>&1; tail -1 $O/r4_fuzz_gemm.txt
python tests/fuzz_events.py > $O/r4_fuzz_events.txt 2>&1; tail -1 $O/r4_fuzz_events.txt
A="--steps 60 --points 8192 --batch 128 --no-latency --no-cpu-baseline --no-traffic --no-legs --no-selfcheck --no-second-site"
for e in "" "EV2H_BALL_LDS_MAX=4096" "" "EV2H_BALL_LDS_MAX=4096"; do env $e python bench.py $A 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n8192 [$e]', j['value'], j['ms_per_step'])"; done
python bench.py $A --collision 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n8192 loss term', j['value'], j['ms_per_step'])"
python bench.py $A --collision --collision-score 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n8192 loss + score', j['value'], j['ms_per_step'])"
python bench.py $A 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n8192 none', j['value'], j['ms_per_step'])"
