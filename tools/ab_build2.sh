#!/bin/bash
# Same-box A/B of BUILDS on the whole step, for several precisions per build:
#   PRECS="bf16 f16x2" bash tools/ab_build2.sh "<EV2H_BUILD_DEFS A>" "<EV2H_BUILD_DEFS B>" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ARGS=${AB_ARGS:-"--steps 100 --warmup 5 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-sustained --no-host-io"}
for r in $(seq 1 ${ROUNDS:-1}); do
for defs in "$@"; do
  EV2H_BUILD_DEFS="$defs" python -m ev2hands_amd.build --force > /dev/null 2>&1 || echo "BUILD FAILED [$defs]"
  for prec in ${PRECS:-bf16 f16x2}; do
    for i in 1 2 3; do
      EV2H_PRECISION=$prec python bench.py $ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$defs] $prec', j['value'], j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])"
    done
  done
done
done
