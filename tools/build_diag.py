#!/usr/bin/env python3
"""Dev tool: build the diagnostic variants of librover_hip.so that the GPU-box measurement scripts load from build/abl/
(they travel to the GPU box with the snapshot):

    K1STAMP   -DRV_K1_STAMP   s_memtime phase stamps in the group step kernel  (tools/k1_stamps.py)
    POLSTAMP  -DPOL_STAMP     s_memtime phase stamps in the policy kernel      (tools/policy_stamps.py)
    LIFTSTAMP -DLF_STAMP      s_memtime phase stamps in the lift step kernel   (tools/lift_stamps.py)

    python tools/build_diag.py [STAMP K1STAMP POLSTAMP]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import build as b  # noqa: E402

VARIANTS = {"K1STAMP": ("rover_kernels.hip", "-DRV_K1_STAMP"),            # s_memtime phase stamps in the step kernels (tools/k1_stamps.py)
            "K1LITE": ("rover_kernels.hip", "-DRV_K1_LITE"),              # no-wait timeline of a step wave and its copy wave (tools/k1_lite.py)
            "K1LITEF": ("rover_kernels.hip", "-DRV_K1_LITE -DRV_K1_LITE_FINE"),
            "PREV": ("rover_kernels.hip", ""),                             # rover_kernels.hip of the last commit, or of $PREV_REV (A/B in one gpurun call)
            "LIFTSTAMP": ("lift_kernels.hip", "-DLF_STAMP"),               # ... in the lift step kernel (tools/lift_stamps.py)
            "POLSTAMP": ("policy_kernels.hip", "-DPOL_STAMP"),             # ... in the policy kernels (tools/policy_stamps.py [pair])
            "NOSLP": ("rover_kernels.hip", "-fno-slp-vectorize"),
            # the window copy limited to the rows each chunk column needs (bit-exact; fewer bytes, more instructions: measured slower)
            "SPANS": ("rover_kernels.hip", "-DRV_ROW_SPANS"),
            "BARRIERS": ("rover_kernels.hip", "-DRV_OWN_TILES=0"),        # the one-launch kernel's scan phase in its barrier form (B, C, D: rounds 4 - 5)
            # whole-file code generation switches tried on top of the product flags (profiles/r05_sched_strategy.txt; the strategy itself,
            # -amdgpu-sched-strategy=max-ilp, is a product flag since: isaac_rover_orbit_amd/build.py)
            "F_IFCVT": ("rover_kernels.hip", "-mllvm -amdgpu-early-ifcvt=1"), "F_NOLICM": ("rover_kernels.hip", "-mllvm -disable-machine-licm"),
            "F_PRELOAD": ("rover_kernels.hip", "-mllvm -amdgpu-kernarg-preload-count=16"),
            "UNROLL4": ("rover_kernels.hip", "-DRV_SOLVER_UNROLL=4"), "UNROLL2": ("rover_kernels.hip", "-DRV_SOLVER_UNROLL=2"),   # solver iterations per loop trip (default 8)
            # rounds (whole quads) of envs 1 / 2 / 3 cast by the step wave, the rest by its copy wave (tools/quick_bench.py)
            **{f"SH_{a}_{b_}_{c}": ("rover_kernels.hip", f"-DRV_SHARE_1={a} -DRV_SHARE_2={b_} -DRV_SHARE_3={c}")
               for (a, b_, c) in ((12, 16, 12), (12, 16, 16), (8, 16, 12), (12, 16, 8), (8, 16, 8), (16, 16, 12), (12, 12, 12))},
            # policy pair kernel: round 3's sequential form; queue depths of the weight fragments
            # timing experiment (WRONG results): the pair kernel's layer 1 does not wait for the observation tile -- the upper bound of what
            # starting layer 1 on the landed part of the tile could gain
            "P_NOTILEWAIT": ("policy_kernels.hip", "-DPOL_X_NOTILEWAIT"), "P_NOTILEWAIT_S": ("policy_kernels.hip", "-DPOL_X_NOTILEWAIT -DPOL_STAMP"),
            "P_SEQ": ("policy_kernels.hip", "-DPOL_PAIR_SEQUENTIAL"), "P_QD1_3": ("policy_kernels.hip", "-DPOL_STAMP -DPOL_QD1=3"),
            "P_QD4_8": ("policy_kernels.hip", "-DPOL_STAMP -DPOL_QD4=8 -DPOL_QD5=10")}
# (The round-2 / round-3 timing builds of the two-launch scan kernel -- RV_K2_EMPTY / PROLOGUE_ONLY / NO_COPY / NO_RAYS / STORE4 /
# NT_STORE / NOREDUCE / STAMP --, RV_K1_NOTERRAIN / CONSTS_IN_KERNEL and round 4's RV_X_NOLDS / NOSTORE / NOLINK / NODRAW have been
# removed from the sources together with their variants here: what they measured is recorded in docs/history.md sections 3.3 - 3.7 and 10.)


def main():
    b.build_extension()                      # the regular objects are reused for the untouched translation units
    out_dir = os.path.join(ROOT, "build", "abl")
    os.makedirs(out_dir, exist_ok=True)
    hipcc = b.hipcc_path()
    for tag in (sys.argv[1:] or list(VARIANTS)):
        src_name, define = VARIANTS[tag]
        obj = os.path.join(out_dir, f"{tag}.o")
        src = os.path.join(ROOT, "isaac_rover_orbit_amd", "csrc", src_name)
        if tag == "PREV":                    # A/B against the last commit: its rover_kernels.hip, same flags, same other objects
            src = os.path.join(out_dir, "prev_" + src_name)
            with open(src, "w") as f:
                f.write(subprocess.check_output(["git", "-C", ROOT, "show", os.environ.get("PREV_REV", "HEAD") + ":isaac_rover_orbit_amd/csrc/" + src_name], text=True))
            define += " -I" + os.path.join(ROOT, "isaac_rover_orbit_amd", "csrc") + " -I" + os.path.join(ROOT, "include")
        subprocess.check_call([hipcc, *b.FLAGS, *define.split(), "-c", "-o", obj, src])
        objs = [obj if os.path.basename(s) == src_name else os.path.join(b.OBJ_DIR, os.path.splitext(os.path.basename(s))[0] + ".o")
                for s in b.SOURCES]
        lib = os.path.join(out_dir, f"librover_abl{tag}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
        print(lib)


if __name__ == "__main__":
    main()
