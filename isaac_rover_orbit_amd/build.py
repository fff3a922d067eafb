"""Build ``librover_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m isaac_rover_orbit_amd.build
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_PKG, "csrc", "rover_kernels.hip")]
HEADERS = [os.path.join(_PKG, "csrc", "rover_model.hpp"), os.path.join(os.path.dirname(_PKG), "include", "rover_hip.h")]
OUTPUT = os.path.join(_PKG, "librover_hip.so")
# fp32 parity with the CPU oracle: no contraction, no fast-math (correctly rounded div / sqrt are hipcc defaults)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(OUTPUT):
        return True
    t = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_extension(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc_path(), *FLAGS, "-o", OUTPUT, *SOURCES]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUTPUT


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
