#!/usr/bin/env python3
"""Run the reference's OWN trainer script -- /root/reference/examples/02_train/train.py, unchanged -- in the GPU-less build
container and record its env-facing transcript.

    python tools/gen_trainer_transcript.py [--num_envs 64] [--timesteps 125]

What is real: train.py itself, rover_envs/utils/skrl_utils.py (SkrlVecEnvWrapper :15-41, SkrlSequentialLogTrainer.train
:96-148), rover_envs/learning/train/* (agent factory), rover_envs/envs/navigation/learning/skrl/models.py (actor / critic),
rover_envs/utils/config.py + rover_ppo.yaml, the reference's gym registration and cfg classes.
What is substituted: Isaac Sim / ORBIT by isaac_rover_orbit_amd.compat; skrl 1.1.0 (not installable) by tests/doubles/skrl;
the HIP env by tests/oracle_env.OracleRoverEnv (CPU oracle, bit-identical to the HIP path), which records every call.
Output: tests/golden/trainer_transcript.npz -- data only (actions the reference's policy produced, env outputs, call list).
"""
import argparse
import os
import runpy
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def run_reference_trainer(num_envs: int, timesteps: int, seed: int = 7, script_rel: str = "examples/02_train/train.py"):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "doubles")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import isaac_rover_orbit_amd.compat as compat
    import isaac_rover_orbit_amd.envs as envs_pkg
    from oracle_env import OracleRoverEnv
    compat.install()
    made = []

    class _Recording(OracleRoverEnv):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)

    real = envs_pkg.RoverEnv
    envs_pkg.RoverEnv = _Recording               # the entry point string "isaac_rover_orbit_amd.envs:RoverEnv" resolves at make()
    os.environ["SKRL_DOUBLE_MAX_TIMESTEPS"] = str(timesteps)
    os.environ.setdefault("EXP_PATH", "/nonexistent/isaac-sim/apps")      # train.py:27-29 only formats it into a string
    script = os.path.join(REF, *script_rel.split("/"))
    argv, cwd = sys.argv, os.getcwd()
    sys.path.insert(0, REF)
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                                # train.py writes logs/skrl/... relative to the cwd
        sys.argv = [script, "--headless", "--num_envs", str(num_envs), "--task", "AAURoverEnv-v0", "--agent", "PPO",
                    "--seed", str(seed)]
        try:
            runpy.run_path(script, run_name="__main__")
        finally:
            os.chdir(cwd)
            sys.argv = argv
            envs_pkg.RoverEnv = real
            os.environ.pop("SKRL_DOUBLE_MAX_TIMESTEPS", None)
    assert len(made) == 1, "the script must build exactly one env"
    return made[0]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=64)
    ap.add_argument("--timesteps", type=int, default=125)
    args = ap.parse_args()
    env = run_reference_trainer(args.num_envs, args.timesteps)
    import numpy as np
    t = env.transcript()
    out = os.path.join(ROOT, "tests", "golden", "trainer_transcript.npz")
    np.savez_compressed(out, **t)
    calls = list(t["calls"])
    print(f"{out}: {os.path.getsize(out)} bytes; calls: {calls[:3]} ... x{len(calls)}; steps {t['actions'].shape[0]}, "
          f"resets in rollout: {int(t['log'][:, 13].sum())}")
