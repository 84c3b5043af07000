#!/bin/bash
# Same-box A/B of bench.py configurations, interleaved (box-to-box spread is +-3 %, run-to-run on one box +-1 %):
#   bash tools/ab_bench.sh <rounds> "<env A>" "<env B>" [...more]      e.g.  bash tools/ab_bench.sh 3 "" "EV2H_COORD_OVERLAP=0"
# Every run is its own process (the switches are read once per process); prints windows/s per run and the per-config medians.
ROUNDS=${1:-3}; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
ARGS=${AB_ARGS:-"--steps 100 --warmup 5 --no-legs --no-latency --no-cpu-baseline --no-traffic --no-selfcheck"}
declare -A vals
for r in $(seq 1 $ROUNDS); do
  i=0
  for cfg in "$@"; do
    v=$(env $cfg python bench.py $ARGS 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "round $r  [${cfg:-default}]  $v"
    vals[$i]="${vals[$i]} $v"
    i=$((i+1))
  done
done
i=0
for cfg in "$@"; do
  med=$(echo ${vals[$i]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  echo "median [${cfg:-default}] = $med   (runs:${vals[$i]})"
  i=$((i+1))
done
