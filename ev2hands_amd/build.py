"""Build libev2hands_hip.so for gfx950 in-tree:  python -m ev2hands_amd.build [--force]"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libev2hands_hip.so")
SOURCES = ["points.hip", "gemm.hip", "gemm_bf16.hip", "sa_mlp.hip", "sa_mlp_bf16.hip", "attention.hip", "mano.hip", "events.hip", "metrics.hip", "collision.hip", "pack.hip", "forward.hip"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "ev2hands_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    # -ffp-contract=off: hipcc would otherwise fuse a*b+c into fma and change the roundings the discrete
    # selections (FPS argmax, radius test, 3-NN) are sensitive to; every intended fma is an explicit fmaf.
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: packed fp32 VALU ops (v_pk_add_f32 / v_pk_fma_f32) that the SLP vectoriser forms out of adjacent
    # scalar operations run slower than the scalar pairs next to MFMAs on gfx950 (measured: SA kernels 5-7 % faster).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-I" + CSRC, "-I" + INCLUDE]
    cmd += os.environ.get("EV2H_BUILD_DEFS", "").split()      # e.g. -DEV2H_SAB_TIMELINE (tools/sa_timeline.py)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
