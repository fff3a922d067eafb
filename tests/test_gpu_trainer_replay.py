"""Replay of the reference trainer's env-facing call sequence on the MI355X.

tests/golden/trainer_transcript.npz was recorded in the build container while the reference's unchanged
``examples/02_train/train.py`` + ``SkrlVecEnvWrapper`` + ``SkrlSequentialLogTrainer.train`` drove the oracle-backed env
(tools/gen_trainer_transcript.py): the actions its PPO policy produced and everything the env returned.  Here the SAME call
sequence -- wrap_env(..., "isaac-orbit"), reset(), step(actions) x T with the trainer's ``states.copy_(next_states)`` and
``infos["episode"]`` reads -- runs against the HIP env; oracle and HIP path are bit-identical, so every recorded value must
come back exactly (extras["log"] means: rtol 1e-5, different summation order)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("log_values", ["host", "device"])   # "host" = what compat.convert gives the reference's trainer; "device" = native default
def test_replay_reference_trainer_transcript(golden_dir, log_values):
    sys.path.insert(0, os.path.join(ROOT, "tests", "doubles"))
    from skrl.envs.wrappers.torch import wrap_env
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RLTaskEnv, RoverEnv
    from isaac_rover_orbit_amd.envs.rover_env import LOG_KEYS
    g = np.load(f"{golden_dir}/trainer_transcript.npz")
    n, T = int(g["num_envs"]), g["actions"].shape[0]
    cfg = RoverEnvCfg()                     # = the reference's AAURoverEnvCfg field by field (tests/test_compat.py)
    cfg.scene.num_envs = n
    cfg.seed = int(g["seed"])
    cfg.log_values = log_values
    assert list(g["terrain"]) == [cfg.terrain.kind, str(tuple(cfg.terrain.shape)), str(cfg.terrain.seed), str(cfg.terrain.sigma_z),
                                  str(cfg.terrain.n_rocks)]
    env = RoverEnv(cfg, headless=True, viewport=False)                  # train.py:123
    assert isinstance(env.unwrapped, RLTaskEnv)                         # skrl_utils.py:38
    wenv = wrap_env(env, wrapper="isaac-orbit")                         # skrl_utils.py:41
    assert wenv.observation_manager.group_obs_dim["policy"][0] == 965 and wenv.action_manager.action_term_dim[0] == 2   # train.py:131-132
    calls = list(g["calls"])
    assert calls[0] == "reset" and calls[-1] == "close" and calls[1:-1] == ["step"] * T
    states, infos = wenv.reset()                                        # skrl_utils.py:114
    s = states.cpu().numpy()
    assert np.array_equal(s[:, :8], g["reset_obs_head"]) and np.array_equal(s.astype(np.float64).sum(1), g["reset_obs_rowsum"])
    for t in range(T):
        actions = torch.from_numpy(g["actions"][t]).to(env.device)
        next_states, rewards, terminated, truncated, infos = wenv.step(actions)            # :123
        assert rewards.shape == (n, 1) and terminated.shape == (n, 1) and truncated.shape == (n, 1)
        o = next_states.cpu().numpy()
        assert np.array_equal(o[:, :8], g["obs_head"][t]), f"step {t}: observation head"
        assert np.array_equal(o.astype(np.float64).sum(1), g["obs_rowsum"][t]), f"step {t}: height scan rows"
        assert np.array_equal(rewards.view(-1).cpu().numpy(), g["reward"][t]), f"step {t}: reward"
        assert np.array_equal(terminated.view(-1).cpu().numpy(), g["terminated"][t].astype(bool)), f"step {t}: terminated"
        assert np.array_equal(truncated.view(-1).cpu().numpy(), g["truncated"][t].astype(bool)), f"step {t}: truncated"
        assert "episode" in infos                                                          # :139
        for i, k in enumerate(LOG_KEYS):
            v = infos["episode"][k]
            assert isinstance(v, torch.Tensor) and v.numel() == 1                          # :141
            assert v.device.type == ("cpu" if log_values == "host" else "cuda")
            if g["log"][t][13] > 0:
                assert abs(v.item() - g["log"][t][i]) <= 1e-5 * max(1.0, abs(g["log"][t][i])), (t, k)
        states.copy_(next_states)                                                          # :148
    wenv.close()
