#!/bin/bash
# Same-box A/B of two BUILDS of the library (compile-time switches), interleaved:
#   bash tools/ab_build.sh <rounds> "<EV2H_BUILD_DEFS A>" "<EV2H_BUILD_DEFS B>"      e.g.  bash tools/ab_build.sh 2 "" "-DEV2H_XPF_ALL"
ROUNDS=${1:-2}; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ARGS=${AB_ARGS:-"--steps 100 --warmup 5 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck --no-second-site --no-sustained --no-host-io"}
for r in $(seq 1 $ROUNDS); do
  for defs in "$@"; do
    EV2H_BUILD_DEFS="$defs" python -m ev2hands_amd.build --force > /dev/null 2>&1
    for i in 1 2; do
      python bench.py $ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r [$defs]', j['value'], j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])"
    done
  done
done
