"""BASELINE.json configurations at their FULL sizes on the MI355X (round-2 review, "configs"): config 2 on the real 2048 x 2048
bench terrain against the oracle, config 4 (1024 rays at 0.05 m, sigma_z 0.4 m) at N = 4096 by size-independent properties and
against the oracle, and the multi-process bench path of config 3 rehearsed with two gloo ranks on the one card."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close, oracle_config_from, oracle_terrain

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_env(config: int, n: int = 4096):
    """The workload bench.py builds for --config 2 / 4 (same generator call, same cfg edits)."""
    from isaac_rover_orbit_amd import terrain as T
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv
    ter = T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=0.15 if config == 2 else 0.4, n_rocks=400)
    ter.make_spawns(2 * n, seed=41)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    cfg.terrain.kind = "custom"
    if config == 4:
        cfg.height_scanner.resolution, cfg.height_scanner.size = 0.05, (1.55, 1.55)
    return RoverEnv(cfg, terrain=ter), ter


@pytest.mark.parametrize("config", [2, 4])
def test_full_size_bench_workload_matches_oracle(oracle, config):
    """N = 4096 on the 2048^2 bench terrain, 6 closed-loop random-action steps from the env's own reset: observations, rewards,
    flags and the whole state against the (OpenMP) oracle, bit for bit.  Config 4: 32 x 32 rays at 0.05 m on sigma_z 0.4 m."""
    env, ter = bench_env(config)
    n = env.num_envs
    assert env.num_rays == (961 if config == 2 else 1024) and env.terrain_data.shape == (2048, 2048)
    env.reset()
    ocfg = oracle_config_from(oracle, env._native_cfg)
    oter = oracle_terrain(oracle, ter)
    So = env.get_state().cpu().numpy().astype(np.float32).copy()
    So[::61, 51] = np.array([747], np.int32).view(np.float32)     # 68 envs time out in the third step: in-step resets on the big map
    env.set_state(torch.from_numpy(So))
    g = torch.Generator(device="cuda").manual_seed(0)
    resets = 0
    for k in range(6):
        a = torch.rand(n, 2, device="cuda", generator=g) * 2 - 1
        obs, rew, term, trunc, info = env.step(a)
        obs_o, rew_o, term_o, trunc_o, force_o, log_o = oracle.step(ocfg, oter, So, a.cpu().numpy())
        assert np.array_equal(term.cpu().numpy().astype(np.uint8), term_o) and np.array_equal(trunc.cpu().numpy().astype(np.uint8), trunc_o)
        assert_close(obs["policy"].cpu().numpy(), obs_o, 0, 0, f"config {config} obs step {k}")
        assert_close(rew.cpu().numpy(), rew_o, 0, 0, f"config {config} reward step {k}")
        resets += int(term_o.sum()) + int(trunc_o.sum())
    assert np.array_equal(env.get_state().cpu().numpy().view(np.int32), So.view(np.int32)), "state after 6 steps (bit exact)"
    assert resets > 0, "no in-step reset was exercised"
    env.close()


def test_config4_full_size_properties():
    """BASELINE config 4 at its full size: 60 steps at N = 4096, 1024 rays, rough terrain.  Size-independent invariants: finite
    observations except rays that leave the map (-inf), unit quaternions, speed cap, scan consistent with rover_height_scan,
    episode counters and reset bookkeeping consistent."""
    env, ter = bench_env(4)
    n = env.num_envs
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    ended = 0
    for k in range(60):
        a = torch.rand(n, 2, device="cuda", generator=g) * 2 - 1
        obs, rew, term, trunc, info = env.step(a)
        ended += int((term | trunc).sum())
    o = obs["policy"]
    assert o.shape == (n, 4 + 1024) and torch.isfinite(o[:, :4]).all() and torch.isfinite(rew).all()
    scan = o[:, 4:]
    assert (torch.isfinite(scan) | (scan == -float("inf"))).all() and torch.isfinite(scan).float().mean() > 0.999
    st = env.state
    assert ((st[3:7] ** 2).sum(0) - 1).abs().max() < 1e-4
    assert (st[7:10] ** 2).sum(0).sqrt().max() <= 1.5 + 1e-4                      # aau_rover_simple.py:25
    assert torch.equal(env.height_scan(), scan)                                      # the unit entry sees the same poses
    assert int(env.episode_length_buf.max()) <= 60 and ended > 0
    assert int(env.episode_log_vector[13]) >= 0
    # rough terrain: the scan really varies (sigma_z 0.4 m over a 1.55 m patch)
    fin = torch.where(torch.isfinite(scan), scan, torch.zeros_like(scan))
    assert fin.std(dim=1).median() > 0.02
    env.close()


def test_bench_two_ranks_gloo_rehearsal():
    """BASELINE config 3 cannot run here (one GPU): the multi-process path of bench.py is rehearsed with two ranks on the one
    card over gloo, so that it cannot rot before an 8-GPU node appears.  The line must carry the aggregate of both ranks,
    weak scaling, and say why the RCCL rollout gather did not run."""
    env = dict(os.environ, ROVER_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--num-envs", "1024", "--no-cpu-baseline", "--profile-steps", "5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_num_envs"] == 2048
    assert d["config"]["parallelism"] == "env-shard x2" and d["value"] > 0
    assert d["rollout_gather"] is None and "gloo" in d["rollout_gather_skipped"]
    # the record that proves (or here: disproves) one GPU per rank: both rehearsal ranks sit on the one card of this box
    rk = d["ranks"]
    assert rk["world"] == 2 and rk["backend"] == "gloo" and len(rk["devices"]) == 2
    assert [x["rank"] for x in rk["devices"]] == [0, 1]
    assert rk["distinct_devices"] == 1 and rk["one_gpu_per_rank"] is False
    assert abs(d["value"] - 2048 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
