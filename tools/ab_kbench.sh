#!/bin/bash
# Same-box A/B of BUILDS of the library on the set-abstraction micro-benchmark:
#   bash tools/ab_kbench.sh "<EV2H_BUILD_DEFS A>" "<EV2H_BUILD_DEFS B>" ...     (KBENCH_PREC=f16x2,bf16 by default)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export KBENCH_PREC=${KBENCH_PREC:-f16x2,bf16}
for defs in "$@"; do
  EV2H_BUILD_DEFS="$defs" python -m ev2hands_amd.build --force > /dev/null 2>&1 || echo "BUILD FAILED [$defs]"
  echo "=== [$defs]"
  python tools/kbench.py sab 2>/dev/null
  python tools/kbench.py sab 2>/dev/null | grep "128,196,256\|128,128,256"
done
