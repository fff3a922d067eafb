"""Build ``librover_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m isaac_rover_orbit_amd.build
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("rover_kernels.hip", "terrain_kernels.hip", "policy_kernels.hip", "lift_kernels.hip")]
HEADERS = [os.path.join(_PKG, "csrc", "rover_model.hpp"), os.path.join(_PKG, "csrc", "rover_internal.hpp"),
           os.path.join(os.path.dirname(_PKG), "include", "rover_lift.h"),
           os.path.join(os.path.dirname(_PKG), "include", "rover_hip.h"),
           os.path.join(os.path.dirname(_PKG), "include", "rover_terrain.h"),
           os.path.join(os.path.dirname(_PKG), "include", "rover_policy.h"),
           os.path.join(os.path.dirname(_PKG), "include", "rover_debug.h")]
OBJ_DIR = os.path.join(os.path.dirname(_PKG), "build", "obj")
OUTPUT = os.path.join(_PKG, "librover_hip.so")
# fp32 parity with the CPU oracle: no contraction, no fast-math (correctly rounded div / sqrt are hipcc defaults)
# max-ilp: LLVM's AMDGPU scheduling strategy that schedules for instruction-level parallelism instead of occupancy -- these kernels run
# ONE wave per SIMD by construction (256 VGPRs, 160 KB of LDS), nothing but the wave's own instruction stream hides a latency.  Measured
# (round 5, profiles/r05_sched_strategy.txt): rover step 39.31 -> 39.06 us, policy pair 31.00 -> 30.44 us, lift 15.0 -> 14.9 us; same
# arithmetic (the flag moves instructions, it does not change them; every parity test bit-exact).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def source_digest() -> str:
    """sha256[:16] over the HIP sources and headers the library is built from -- the identity of a BUILD that survives a rebuild
    (two hipcc runs on the same sources differ in their code-object ids, so the binary's own hash does not).  bench.py prints the
    committed PMC summaries (profiles/*.json) only beside the sources they were measured on."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(OUTPUT):
        return True
    t = os.path.getmtime(OUTPUT)
    # an installed (non-editable) copy may lack include/*.h: a prebuilt library is then used as it is
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS if os.path.exists(f))


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """One object per source (rebuilt only when that source or a header changed), then one link."""
    if not (force or needs_build()):
        return OUTPUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc, objs = hipcc_path(), []
    t_hdr = max(os.path.getmtime(h) for h in HEADERS)
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
            cmd = [hipcc, *FLAGS, "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUTPUT, *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUTPUT


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
