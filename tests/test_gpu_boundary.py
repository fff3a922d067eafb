"""Boundary test on the GPU: a gymnasium / skrl-shaped consumer that touches exactly what the reference's training
loop touches (rover_envs/utils/skrl_utils.py:96-148, examples/02_train/train.py:115-145)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gym_make_and_trainer_shaped_loop():
    import isaac_rover_orbit_amd.compat as compat
    from isaac_rover_orbit_amd.cfg import AAURoverEnvCfg
    from isaac_rover_orbit_amd.envs import RLTaskEnv
    compat.install()
    compat.register_default_tasks()
    gym = compat.gym_api()
    cfg = AAURoverEnvCfg()
    cfg.scene.num_envs = 256
    cfg.terrain.shape = (1024, 1024)
    cfg.terrain.n_rocks = 100
    env = gym.make("AAURoverEnv-v0", cfg=cfg, headless=True, viewport=False)      # train.py:123
    assert isinstance(env.unwrapped, RLTaskEnv)                                     # skrl_utils.py:38
    num_obs = env.observation_manager.group_obs_dim["policy"][0]                    # train.py:131
    num_actions = env.action_manager.action_term_dim[0]                             # train.py:132
    assert (num_obs, num_actions) == (965, 2)
    assert env.observation_manager.group_obs_term_dim["policy"][-1][0] == 961       # get_models.py:39
    # the "policy": encoder slice quirk of models.py:94-95 (obs[:, 0:4] and obs[:, 3:-1])
    w = torch.randn(4 + 961, 2, device=env.device) * 0.01   # 4 + (965 - 1 - 3)
    states, infos = env.reset()                                                     # skrl_utils.py:114
    seen_episode_keys = set()
    for t in range(30):
        s = states["policy"]
        actions = torch.tanh(torch.cat([s[:, 0:4], s[:, 3:-1]], dim=1).clamp(-5, 5) @ w)     # agent.act
        next_states, rewards, terminated, truncated, infos = env.step(actions)      # skrl_utils.py:123
        # skrl's isaac-orbit wrapper views these as (N, 1)
        assert rewards.view(-1, 1).shape == (256, 1) and terminated.view(-1, 1).dtype == torch.bool
        assert torch.isfinite(rewards).all()
        if "episode" in infos:                                                      # skrl_utils.py:139-142
            for k, v in infos["episode"].items():
                if isinstance(v, torch.Tensor) and v.numel() == 1:
                    seen_episode_keys.add(k)
                    float(v.item())
        # `states` of this step must survive the next step (record_transition uses both)
        assert states["policy"].data_ptr() != next_states["policy"].data_ptr()
        states = next_states
    assert "Episode Reward/distance_to_target" in seen_episode_keys
    assert "Episode Termination/collision" in seen_episode_keys and "Metrics/target_pose/error_pos" in seen_episode_keys
    zero = torch.zeros(env.action_space.shape, device=env.unwrapped.device)        # 01_zero_agent.py:44-52
    env.step(zero)
    env.close()


def test_c_abi_from_plain_cpp_matches_python_binding(tmp_path):
    """examples/c_api_demo.cpp (no Python, no torch: hipMalloc + the C ABI) against RoverEnv on the same configuration:
    same mean reward, same bits in the last observation."""
    import os
    import shutil
    import subprocess
    import numpy as np
    import torch
    from isaac_rover_orbit_amd import terrain as T
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "c_api_demo")
    libdir = os.path.join(root, "isaac_rover_orbit_amd")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_api_demo.cpp"), "-o", exe, "-L" + libdir, "-lrover_hip",
                           "-Wl,-rpath," + libdir], timeout=600)
    n, steps = 192, 40
    out = subprocess.run([exe, str(n), str(steps)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    fields = out.stdout.strip().split("|")
    c_mean = float(fields[2].split()[-1])
    c_sum = fields[3].split()[-1]

    H = W = 1024
    i = np.arange(2 * n, dtype=np.float32)
    spawns = np.stack([np.float32(21.0) + np.float32(9.0) * i / np.float32(2 * n),
                       np.float32(30.0) - np.float32(9.0) * i / np.float32(2 * n), np.zeros(2 * n, np.float32)], 1).astype(np.float32)
    zero = np.zeros((H, W), np.uint8)
    ter = T.Terrain(ground=np.zeros((H, W), np.float32), obstacle=np.zeros((H, W), np.float32), rock_mask=zero,
                    safe_rock_mask=zero.copy(), spawn_locations=spawns)
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"; cfg.record_contact_forces = False
    cfg.use_int16_terrain = False
    env = RoverEnv(cfg, terrain=ter)
    env.reset()
    act = torch.tensor([0.8, 0.1], device="cuda").repeat(n, 1)
    tot = 0.0
    for _ in range(steps):
        obs, rew, *_ = env.step(act)
        tot += float(rew.cpu().numpy().astype(np.float64).sum())
    raw = obs["policy"].cpu().numpy().astype(np.float32).tobytes()
    h = 2166136261
    for b in raw:
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    assert f"{h:08x}" == c_sum, (f"{h:08x}", c_sum)
    assert abs(tot / (n * steps) - c_mean) <= 1e-9 * max(1.0, abs(c_mean))
    env.close()
