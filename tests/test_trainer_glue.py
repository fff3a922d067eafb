"""The reference's OWN trainer stack runs unchanged on top of this repository's boundary:
``examples/02_train/train.py`` -> ``parse_env_cfg`` / ``gym.make`` -> ``SkrlVecEnvWrapper`` (skrl_utils.py:15-41) ->
``get_agent`` (learning/train/ppo.py, get_models.py, learning/skrl/models.py) -> ``SkrlSequentialLogTrainer.train``
(skrl_utils.py:96-148).  Build container only (needs the reference checkout); skrl 1.1.0 is replaced by tests/doubles/skrl and
the HIP env by the oracle-backed OracleRoverEnv (same surface, CPU).  The env-facing transcript of such a run is the fixture
that tests/test_gpu_trainer_replay.py replays against the HIP env on the MI355X."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rover_envs")), reason="reference checkout not present")


@pytest.fixture()
def clean_modules():
    saved_path, saved_meta, before = list(sys.path), list(sys.meta_path), set(sys.modules)
    yield
    sys.path[:], sys.meta_path[:] = saved_path, saved_meta
    for name in set(sys.modules) - before:
        if name.split(".")[0] in ("rover_envs", "skrl", "omni", "carb", "pxr", "pymeshlab", "gym", "oracle_env"):
            sys.modules.pop(name, None)


def test_reference_train_script_runs_to_completion(clean_modules, golden_dir):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_trainer_transcript as gt
    n, steps = 64, 125
    env = gt.run_reference_trainer(n, steps)
    from skrl.agents.torch.ppo import PPO
    agent = PPO.instances[-1]
    # the trainer loop: one reset, `steps` steps, close (train.py:143)
    assert env.calls == ["reset"] + ["step"] * steps + ["close"]
    assert env.ctor_kwargs == {"headless": True, "viewport": False}            # train.py:123 gym.make(..., headless=, viewport=)
    # the cfg train.py built (parse_env_cfg -> the reference's AAURoverEnvCfg) reaches the env through compat.convert: that path
    # serves THIS trainer, whose loop .item()s every log entry after every step (skrl_utils.py:139-142) -> host mirror by default
    assert env.cfg.log_values == "host"
    # the reference's factories built ITS networks on the env's surface (train.py:131-139, get_models.py:39)
    assert type(agent.policy).__name__ == "GaussianNeuralNetwork" and type(agent.value).__name__ == "DeterministicNeuralNetwork"
    assert agent.policy.dense_encoder.encoder_layers[0].in_features == 961 and agent.policy.mlp[0].in_features == 64
    assert agent.memory.memory_size == 60 and agent.memory.num_envs == n       # rover_ppo.yaml:30
    assert agent.updates == steps // 60 and agent._initialised == 2            # init in __init__ (:92) and in train() (:110)
    # infos["episode"] was read and logged (skrl_utils.py:139-142): one entry per step and key
    tags = {k for _, d in agent.written for k in d} | set(agent.tracking_data)
    assert "EpisodeInfo / Episode Reward/distance_to_target" in tags and "EpisodeInfo / Episode Termination/is_success" in tags
    t = env.transcript()
    assert t["actions"].shape == (steps, n, 2) and np.abs(t["actions"]).max() <= 1.0      # tanh mean + clip_actions
    assert np.isfinite(t["reward"]).all() and t["log"][:, 13].sum() > 0                   # some envs reset inside the rollout
    # the committed fixture is such a run (same seeds); CPU matmul kernels may differ between hosts, so compare loosely
    g = np.load(f"{golden_dir}/trainer_transcript.npz")
    assert list(g["calls"]) == env.calls and g["actions"].shape == t["actions"].shape
    assert np.allclose(g["actions"][:5], t["actions"][:5], atol=1e-4)


def test_reference_eval_script_loads_the_pretrained_agent(clean_modules):
    """examples/03_inference_pretrained/eval.py, unchanged: ``agent.load(best_agent.pt)`` into the networks the reference's own
    factory builds (eval.py:146-149), then ``SkrlSequentialLogTrainer.eval`` (skrl_utils.py:150-175 -> skrl's single-agent
    evaluation loop) drives the env with the Isaac-Sim-trained policy."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_trainer_transcript as gt
    import torch
    n, steps = 64, 150
    env = gt.run_reference_trainer(n, steps, seed=3, script_rel="examples/03_inference_pretrained/eval.py")
    from skrl.agents.torch.ppo import PPO
    agent = PPO.instances[-1]
    assert env.calls == ["reset"] + ["step"] * steps + ["close"]              # eval.py:154-157
    assert agent.loaded == ["policy", "value", "optimizer"]                   # every module of the checkpoint was restored
    ckpt = torch.load(os.path.join(REF, "rover_envs/envs/navigation/robots/aau_rover/policies/best_agent.pt"),
                      map_location="cpu", weights_only=False)
    for k, v in ckpt["policy"].items():
        assert torch.equal(agent.policy.state_dict()[k].cpu(), v), k
    assert agent.updates == 0 and not agent.training                          # evaluation mode: nothing was learned
    t = env.transcript()
    assert np.isfinite(t["reward"]).all() and np.abs(t["actions"]).max() <= 1.0
    # the trained policy drives: it reaches targets within 150 steps, which random or zero actions never do
    log = t["log"]
    finished = log[:, 13].sum()
    successes = log[:, 8].sum()               # Episode Termination/is_success: episodes of the step's reset batch that ended so
    assert finished > 0 and successes > 0, (finished, successes)
