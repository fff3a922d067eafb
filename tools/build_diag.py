#!/usr/bin/env python3
"""Dev tool: build the diagnostic variants of librover_hip.so that the GPU-box measurement scripts load from build/abl/
(they travel to the GPU box with the snapshot):

    STAMP     -DRV_K2_STAMP   s_memtime phase stamps in the scan kernel        (tools/k2_stamps.py)
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

VARIANTS = {"STAMP": ("rover_kernels.hip", "-DRV_K2_STAMP"), "K1STAMP": ("rover_kernels.hip", "-DRV_K1_STAMP"),
            "K1STAMP_INK": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_K1_CONSTS_IN_KERNEL"),
            "LIFTSTAMP": ("lift_kernels.hip", "-DLF_STAMP"), "NOREDUCE": ("rover_kernels.hip", "-DRV_K2_NOREDUCE"),
            "SKEL_STORE4": ("rover_kernels.hip", "-DRV_K2_NO_RAYS -DRV_K2_NO_COPY -DRV_K2_STORE4"),
            "SKEL_NOSTORE": ("rover_kernels.hip", "-DRV_K2_NO_RAYS -DRV_K2_NO_COPY -DRV_K2_NOSTORE"),
            "STORE4": ("rover_kernels.hip", "-DRV_K2_STORE4"), "NTSTORE": ("rover_kernels.hip", "-DRV_K2_NT_STORE"),
            "K2EMPTY": ("rover_kernels.hip", "-DRV_K2_EMPTY"), "K2PROLOGUE": ("rover_kernels.hip", "-DRV_K2_PROLOGUE_ONLY"),
            "NOCOPY": ("rover_kernels.hip", "-DRV_K2_NO_COPY"),
            "NORAYS": ("rover_kernels.hip", "-DRV_K2_NO_RAYS"), "NOCOPYRAYS": ("rover_kernels.hip", "-DRV_K2_NO_RAYS -DRV_K2_NO_COPY"),
            "K1_INK": ("rover_kernels.hip", "-DRV_K1_CONSTS_IN_KERNEL"),
            "NOSLP": ("rover_kernels.hip", "-fno-slp-vectorize"),
            "NOSLP_STAMP": ("rover_kernels.hip", "-fno-slp-vectorize -DRV_K1_STAMP"),
            "POLSTAMP": ("policy_kernels.hip", "-DPOL_STAMP"),
            # split of an env's sixteen ray rounds between a step wave and its copy wave (one-launch kernel)
            "SHARE_8_14": ("rover_kernels.hip", "-DRV_SHARE_FREE=8 -DRV_SHARE_COPY=14"),
            "SHARE_8_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=8 -DRV_SHARE_COPY=16"),
            "SHARE_7_14": ("rover_kernels.hip", "-DRV_SHARE_FREE=7 -DRV_SHARE_COPY=14"),
            "SHARE_6_13": ("rover_kernels.hip", "-DRV_SHARE_FREE=6 -DRV_SHARE_COPY=13"),
            "SHARE_16_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=16 -DRV_SHARE_COPY=16"),
            "SHARE_6_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=6 -DRV_SHARE_COPY=16"),
            "SHARE_10_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=10 -DRV_SHARE_COPY=16"),
            "SHARE_12_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=12 -DRV_SHARE_COPY=16"),
            # round 4: what bounds the pipelined cast (stamped builds; X_* produce wrong observations)
            "X_NOSTORE": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_X_NOSTORE"), "X_NOLDS": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_X_NOLDS"),
            "X_NOBOTH": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_X_NOLDS -DRV_X_NOSTORE"),
            "X_S16": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_SHARE_FREE=16 -DRV_SHARE_COPY=16"),
            "X_S8_12": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_SHARE_FREE=8 -DRV_SHARE_COPY=12"),
            "X_S12_12": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_SHARE_FREE=12 -DRV_SHARE_COPY=12"),
            "X_S8_16": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_SHARE_FREE=8 -DRV_SHARE_COPY=16"),
            "X_S4_12": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_SHARE_FREE=4 -DRV_SHARE_COPY=12"),
            "X_S12_16": ("rover_kernels.hip", "-DRV_K1_STAMP -DRV_SHARE_FREE=12 -DRV_SHARE_COPY=16"),
            # unstamped share variants for tools/quick_bench.py
            "Q_S4_12": ("rover_kernels.hip", "-DRV_SHARE_FREE=4 -DRV_SHARE_COPY=12"), "Q_S8_12": ("rover_kernels.hip", "-DRV_SHARE_FREE=8 -DRV_SHARE_COPY=12"),
            "Q_S4_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=4 -DRV_SHARE_COPY=16"), "Q_S8_16": ("rover_kernels.hip", "-DRV_SHARE_FREE=8 -DRV_SHARE_COPY=16"),
            "Q_S0_12": ("rover_kernels.hip", "-DRV_SHARE_FREE=0 -DRV_SHARE_COPY=12"), "Q_S4_8": ("rover_kernels.hip", "-DRV_SHARE_FREE=4 -DRV_SHARE_COPY=8"),
            "P_QD1_3": ("policy_kernels.hip", "-DPOL_STAMP -DPOL_QD1=3"), "P_QD4_8": ("policy_kernels.hip", "-DPOL_STAMP -DPOL_QD4=8 -DPOL_QD5=10"),
            "P_QD4_4": ("policy_kernels.hip", "-DPOL_STAMP -DPOL_QD4=4 -DPOL_QD5=3"), "P_QD1_1": ("policy_kernels.hip", "-DPOL_STAMP -DPOL_QD1=1"),
            "Q_NOLINK": ("rover_kernels.hip", "-DRV_X_NOLINK"), "Q_NODRAW": ("rover_kernels.hip", "-DRV_X_NODRAW"),
            "Q_NOSTORE": ("rover_kernels.hip", "-DRV_X_NOSTORE"), "Q_NOLDS": ("rover_kernels.hip", "-DRV_X_NOLDS"),
            "Q_NOBOTH": ("rover_kernels.hip", "-DRV_X_NOLDS -DRV_X_NOSTORE"), "Q_BASE": ("rover_kernels.hip", "-DRV_Q_BASE"),
            "Q_S8_8": ("rover_kernels.hip", "-DRV_SHARE_FREE=8 -DRV_SHARE_COPY=8"), "Q_S12_12": ("rover_kernels.hip", "-DRV_SHARE_FREE=12 -DRV_SHARE_COPY=12"),
            "Q_S12_8": ("rover_kernels.hip", "-DRV_SHARE_FREE=12 -DRV_SHARE_COPY=8"), "Q_S4_4": ("rover_kernels.hip", "-DRV_SHARE_FREE=4 -DRV_SHARE_COPY=4")}


def main():
    b.build_extension()                      # the regular objects are reused for the untouched translation units
    out_dir = os.path.join(ROOT, "build", "abl")
    os.makedirs(out_dir, exist_ok=True)
    hipcc = b.hipcc_path()
    for tag in (sys.argv[1:] or list(VARIANTS)):
        src_name, define = VARIANTS[tag]
        obj = os.path.join(out_dir, f"{tag}.o")
        subprocess.check_call([hipcc, *b.FLAGS, *define.split(), "-c", "-o", obj, os.path.join(ROOT, "isaac_rover_orbit_amd", "csrc", src_name)])
        objs = [obj if os.path.basename(s) == src_name else os.path.join(b.OBJ_DIR, os.path.splitext(os.path.basename(s))[0] + ".o")
                for s in b.SOURCES]
        lib = os.path.join(out_dir, f"librover_abl{tag}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
        print(lib)


if __name__ == "__main__":
    main()
