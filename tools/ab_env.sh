#!/bin/bash
# Same-box A/B of an ENVIRONMENT switch on the whole step, interleaved:
#   PRECS="f16x2 bf16" bash tools/ab_env.sh EV2H_ATTN_UNFUSED_ZSUM [extra bench args...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
VAR=$1; shift
ARGS="--steps ${STEPS:-100} --warmup 5 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-host-io --no-sustained $*"
for r in $(seq 1 ${ROUNDS:-2}); do
  for prec in ${PRECS:-f16x2 bf16}; do
    for v in "" 1; do
      if [ -z "$v" ]; then E="-u $VAR"; else E="$VAR=$v"; fi       # (an empty value would still SET the switch)
      env $E EV2H_PRECISION=$prec python bench.py $ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$VAR=$v] $prec', j['value'], j['ms_per_step'], j['roofline']['launch_site'], j['roofline']['kernel_ms'], j['roofline_second']['launch_site'], j['roofline_second']['kernel_ms'])"
    done
  done
done
