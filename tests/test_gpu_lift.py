"""FrankaCubeLift-v0 on the MI355X (BASELINE config 5): the HIP path through the C ABI (include/rover_lift.h)
against the reference fixture for the term arithmetic the reference owns, and against the separately written scalar CPU
oracle (oracle/lift_oracle.c; shares no source with the kernel) bit for bit -- state, observations, rewards, flags -- over
closed-loop rollouts with resets, for both lane mappings of the step kernel.  extras["log"] means: rtol 1e-5."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lo():
    from oracle import lift_oracle
    lift_oracle.build()
    lift_oracle.lib()
    return lift_oracle


def make_env(n, **over):
    from isaac_rover_orbit_amd.envs import FrankaCubeLiftEnv, LiftEnvCfg
    cfg = LiftEnvCfg()
    cfg.scene.num_envs = n
    for k, v in over.items():
        setattr(cfg, k, v)
    return FrankaCubeLiftEnv(cfg)


def oracle_cfg(lo, env):
    c = lo.Config()
    for name, _ in lo.Config._fields_:
        v = getattr(env._native_cfg, name)
        if hasattr(v, "__len__"):
            for i in range(len(v)):
                getattr(c, name)[i] = v[i]
        else:
            setattr(c, name, v)
    return c


def test_lift_terms_match_reference_fixture(lo, golden_dir):
    g = np.load(f"{golden_dir}/lift_terms.npz")
    env = make_env(64)
    lifted, reach, goal, fine, pos_b = [t.cpu().numpy() for t in env.terms(g["object_pos_w"], g["ee_pos_w"], g["robot_root_state_w"],
                                                                            g["command"])]
    assert_close(lifted, g["rew_object_is_lifted"], 0, 0, "object_is_lifted")
    assert_close(reach, g["rew_object_ee_distance"], 3e-7, 2e-6, "object_ee_distance")
    assert_close(goal, g["rew_object_goal_distance_03"], 3e-7, 2e-6, "object_goal_distance 0.3")
    assert_close(fine, g["rew_object_goal_distance_005"], 3e-7, 2e-6, "object_goal_distance 0.05")
    assert_close(pos_b, g["obs_object_position_in_robot_root_frame"], 3e-7, 2e-6, "object_position_in_robot_root_frame")
    # and bit-exact against the oracle's independent restatement
    o = lo.terms(g["object_pos_w"], g["ee_pos_w"], g["robot_root_state_w"], g["command"])
    for a, b, nm in zip((lifted, reach, goal, fine, pos_b), o, ("lifted", "reach", "goal", "fine", "pos_b")):
        assert_close(a, b, 0, 0, nm + " vs oracle")
    env.close()


@pytest.mark.parametrize("n,seed,lanes,log_every,pipe", [(64, 0, 8, 1, True), (333, 5, 8, 7, True), (61, 2, 16, 1, False),
                                                         (64, 3, 8, 0, False), (13, 6, 8, 1, True)])
def test_lift_rollout_matches_oracle(lo, n, seed, lanes, log_every, pipe):
    """300 closed-loop steps (every env times out once: in-step resets, command resampling), random actions incl. gripper;
    batch sizes that do not fill the last wave; eight lanes per env as two pipelined waves (the product form) and as one wave, and
    the shadowed sixteen-lane form.  extras["log"]: reduced on
    demand, read after every step / after every seventh step only (the deferred reduction must then hold what the per-step
    reduction of the oracle holds at that step), or reduced behind every step (log_every = 0, the C entry's default)."""
    env = make_env(n, seed=seed, log_reduction="every_step" if log_every == 0 else "on_demand")
    assert env._lib.rover_lift_debug_set_lanes(env._h, lanes) == 0 and env._lib.rover_lift_debug_set_pipeline(env._h, int(pipe)) == 0
    assert env.kernel_name() == f"lift_step_kernel<{lanes}, {'true' if pipe else 'false'}>"
    ocfg = oracle_cfg(lo, env)
    obs, info = env.reset()
    So = lo.new_state(n)
    obs_o = lo.reset(ocfg, So)
    assert_close(obs["policy"].cpu().numpy(), obs_o, 0, 0, "obs after reset")
    assert np.array_equal(env.get_state().cpu().numpy().view(np.int32), So.view(np.int32))
    rng = np.random.RandomState(seed)
    log_o = np.zeros(16, np.float32)
    for k in range(300):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        if k % 50 < 25:
            a[:, 7] = -1.0
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a).to(env.device))
        obs_o, rew_o, term_o, trunc_o, log_o = lo.step(ocfg, So, a, log=log_o)
        assert np.array_equal(term.cpu().numpy(), term_o.astype(bool)) and np.array_equal(trunc.cpu().numpy(), trunc_o.astype(bool)), k
        assert_close(obs["policy"].cpu().numpy(), obs_o, 0, 0, f"obs step {k}")
        assert_close(rew.cpu().numpy(), rew_o, 0, 0, f"reward step {k}")
        if log_every == 0:
            log = env._log.cpu().numpy()                                        # the raw vector: reduced by the step itself
        elif k % log_every == log_every - 1:
            log = np.array([float(v) for v in info["log"].values()] + [float(env._log[8])], np.float32)   # a read reduces
            assert list(info["log"].keys()) == list(env._log_dict.keys()) and info["episode"] is info["log"]
        else:
            continue
        assert log[8] == log_o[8], k
        assert_close(log[:8], log_o[:8], 1e-7, 1e-5, f"log step {k}")
    assert np.array_equal(env.get_state().cpu().numpy().view(np.int32), So.view(np.int32)), "final state bit exact"
    assert trunc_o.sum() == 0 and So[:, lo.EP_LEN].view(np.int32).max() < 60
    env.close()


def test_lift_grasp_and_contacts_match_oracle(lo):
    """States the random rollout rarely reaches: cube held between closing fingers (both pad row sets active), cube tilted on
    an edge / a corner (different corner sets per env), cube in free fall, joints at their limits with saturated actuators --
    40 steps each, bit exact."""
    n = 96
    env = make_env(n, seed=11)
    ocfg = oracle_cfg(lo, env)
    env.reset()
    So = lo.new_state(n)
    lo.reset(ocfg, So)
    rng = np.random.RandomState(3)
    S = So.copy()
    for e in range(n):
        kind = e % 4
        if kind == 0:        # in the hand
            tcp, R, _, _ = lo.hand_pose(ocfg, S[e, :9])
            S[e, lo.OBJ_POS:lo.OBJ_POS + 3] = tcp + rng.uniform(-0.004, 0.004, 3)
            S[e, 7:9] = 0.0205 + rng.uniform(0, 0.002, 2)
        elif kind == 1:      # tilted on the table
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            ang = rng.uniform(0.2, 0.9)
            S[e, lo.OBJ_QUAT:lo.OBJ_QUAT + 4] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
            S[e, lo.OBJ_POS + 2] = 0.03
            S[e, lo.OBJ_ANG:lo.OBJ_ANG + 3] = rng.uniform(-3, 3, 3)
        elif kind == 2:      # falling and spinning
            S[e, lo.OBJ_POS + 2] = rng.uniform(0.1, 0.4)
            S[e, lo.OBJ_LIN:lo.OBJ_LIN + 6] = rng.uniform(-1, 1, 6)
        else:                # arm far from the default pose, fast
            S[e, :7] = np.clip(S[e, :7] + rng.uniform(-1.5, 1.5, 7), [-2.8, -1.7, -2.8, -3.0, -2.8, 0.0, -2.8], [2.8, 1.7, 2.8, -0.1, 2.8, 3.7, 2.8])
            S[e, 9:16] = rng.uniform(-2, 2, 7)
    S = S.astype(np.float32)
    So[:] = S
    env.set_state(torch.from_numpy(S))
    pads_seen = 0
    for k in range(40):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        a[0::4, 7] = -1.0                     # the grasping envs keep closing
        a[3::4, :7] *= 3.0                    # far targets: actuators saturate
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a).to(env.device))
        obs_o, rew_o, term_o, trunc_o, _ = lo.step(ocfg, So, a)
        assert_close(obs["policy"].cpu().numpy(), obs_o, 0, 0, f"obs step {k}")
        assert_close(rew.cpu().numpy(), rew_o, 0, 0, f"reward step {k}")
        assert np.array_equal(term.cpu().numpy(), term_o.astype(bool))
        pads_seen += int((So[0::4, 7] < 0.03).sum())
    assert np.array_equal(env.get_state().cpu().numpy().view(np.int32), So.view(np.int32)), "final state bit exact"
    assert pads_seen > 100, "the grasp case never engaged the pads"
    env.close()


def test_lift_seed_and_checkpoint(lo):
    """reset(seed=) re-keys the resets (advisor finding of round 2); state_dict carries the key."""
    env = make_env(128, seed=1)
    o1 = env.reset()[0]["policy"].clone()
    o2 = env.reset(seed=77)[0]["policy"].clone()
    assert not torch.equal(o1[:, 18:21], o2[:, 18:21])
    ocfg = oracle_cfg(lo, env)
    assert ocfg.seed_lo == 77
    So = lo.new_state(128)
    So[:, lo.RESET_COUNT] = np.array([1], np.uint32).view(np.float32)           # second reset of every env
    assert_close(o2.cpu().numpy(), lo.reset(ocfg, So), 0, 0, "obs after reset(seed=77)")
    sd = env.state_dict()
    env2 = make_env(128, seed=5)
    env2.load_state_dict(sd)
    a = torch.zeros(128, 8, device=env.device)
    S = env.get_state(); S[:, 39] = torch.tensor([249], dtype=torch.int32).view(torch.float32).item()   # time out next step
    env.set_state(S); env2.set_state(S)
    assert torch.equal(env.step(a)[0]["policy"], env2.step(a)[0]["policy"])
    env.close(); env2.close()


def test_lift_boundary_surface():
    """What skrl / gym consumers touch (SURVEY 8b pattern): spaces, managers, extras, registration."""
    import isaac_rover_orbit_amd.compat as compat
    from isaac_rover_orbit_amd.envs import FrankaCubeLiftEnv, RLTaskEnv
    compat.register_default_tasks()
    gym = compat.gym_api()
    from isaac_rover_orbit_amd.envs import LiftEnvCfg
    cfg = LiftEnvCfg()
    cfg.scene.num_envs = 128
    env = gym.make("FrankaCubeLift-v0", cfg=cfg)
    assert isinstance(env.unwrapped, RLTaskEnv) and isinstance(env, FrankaCubeLiftEnv)
    assert env.observation_manager.group_obs_dim["policy"] == (36,) and env.action_manager.total_action_dim == 8
    assert env.action_space.shape == (128, 8) and env.single_observation_space["policy"].shape == (36,)
    obs, info = env.reset()
    assert obs["policy"].shape == (128, 36) and int(env.max_episode_length) == 250
    o, r, t, u, info = env.step(torch.zeros(128, 8, device=env.device))
    assert r.shape == (128,) and t.dtype == torch.bool and u.dtype == torch.bool
    assert set(info["episode"]) == {f"Episode Reward/{k}" for k in ("reaching_object", "lifting_object", "object_goal_tracking",
                                    "object_goal_tracking_fine_grained", "action_rate", "joint_vel")} | \
        {"Episode Termination/time_out", "Episode Termination/object_dropping"}
    assert env.command_manager.get_command("object_pose").shape == (128, 7)
    with pytest.raises(ValueError):
        env.step(torch.zeros(128, 7, device=env.device))
    env.close()


def test_lift_full_size_properties():
    """BASELINE config 5 size (2048 envs): finite everywhere over 260 random steps, timers and resets consistent."""
    n = 2048
    env = make_env(n, seed=3)
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(0)
    resets = 0
    for k in range(260):
        a = torch.rand(n, 8, device=env.device, generator=g) * 2 - 1
        obs, rew, term, trunc, info = env.step(a)
        resets += int(term.sum()) + int(trunc.sum())
    S = env.get_state()
    assert torch.isfinite(S[:, :62]).all() and torch.isfinite(obs["policy"]).all() and torch.isfinite(rew).all()
    assert resets >= n and int(env.episode_length_buf.max()) <= 12
    assert (S[:, 18 + 2] > -0.06).all()                       # nothing below the drop height survives a step
    env.close()
