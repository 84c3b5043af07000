#!/bin/bash
# Throughput against the batch size (windows per GPU per step), default arithmetic:  bash tools/batch_sweep.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for b in 1 4 16 32 64 128 256 512; do
  s=$((b<32?400:100))
  python bench.py --batch $b --steps $s --no-legs --no-latency --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > /tmp/bs_line.json
  python - "$b" <<'PY'
import json, sys
d = json.load(open("/tmp/bs_line.json"))
print("B=%4d  %9.1f windows/s  %8.3f ms/step" % (int(sys.argv[1]), d["value"], d["ms_per_step"]))
PY
done
