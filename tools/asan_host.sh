#!/bin/bash
# Host side of the library (weight packer, argument checks, workspace layout, error paths) under AddressSanitizer + UBSan, on the CPU:
#   bash tools/asan_host.sh            (no GPU needed; GPU ASan is not available on this pool)
# Builds the whole library with the sanitizers applied to the HOST compilation only (-Xarch_host) into /tmp and runs the CPU tests that
# call into it through ctypes, with the sanitizer runtime preloaded into python.
set -u
cd "$(dirname "$0")/.."
OUT=/tmp/ev2h_asan; mkdir -p $OUT
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
SRC=$(ls ev2hands_amd/csrc/*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared \
  -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined \
  -Iev2hands_amd/csrc -Iinclude $SRC -o $OUT/libev2hands_hip.so || exit 1
EV2H_LIB_PATH=$OUT/libev2hands_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_pack_abi.py tests/test_host_cpu.py tests/test_range_host.py -q -x -m "not gpu" -p no:cacheprovider "$@"
