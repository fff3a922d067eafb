import math

import pytest

from isaac_rover_orbit_amd.cfg import AAURoverEnvCfg, RoverEnvCfg, TermCfg


def test_cfg_defaults_follow_reference():
    c = AAURoverEnvCfg()
    assert c.max_episode_length == 750 and c.decimation == 6 and abs(c.sim.dt - 1 / 30) < 1e-12   # rover_env_cfg.py:269-271
    assert c.height_scanner.grid == (31, 31)                                                        # 961 rays
    n = c.to_native()
    assert n.max_episode_length == 750 and n.scan_nx * n.scan_ny == 961
    assert n.obs_scale_heading == pytest.approx(1 / math.pi) and n.obs_scale_distance == pytest.approx(0.11)
    assert n.heading_lo == pytest.approx(-math.pi) and n.resample_time == 150.0
    assert list(c.rewards) == ["distance_to_target", "reached_target", "oscillation", "angle_to_target",
                               "heading_soft_contraint", "collision", "far_from_target"]


def test_cfg_edits_reach_the_kernel_parameters():
    c = RoverEnvCfg()
    c.rewards["collision"].weight = -3.5
    c.actions.offset = (0.1, -0.2)
    c.height_scanner.resolution, c.height_scanner.size = 0.05, (1.55, 1.55)     # BASELINE config 4: 32 x 32 rays
    c.seed = (7 << 32) | 9
    c.reset_velocities = "zero"
    n = c.to_native()
    assert n.rew_weight[5] == -3.5 and n.offset_lin == pytest.approx(0.1) and n.offset_ang == pytest.approx(-0.2)
    assert (n.scan_nx, n.scan_ny) == (32, 32) and n.seed_lo == 9 and n.seed_hi == 7 and n.reset_mode == 1


def test_unknown_terms_are_rejected():
    c = RoverEnvCfg()
    c.rewards["my_term"] = TermCfg("my_func", weight=1.0)
    with pytest.raises(ValueError):
        c.to_native()
    c = RoverEnvCfg()
    c.terminations["is_success"].func = "something_else"
    with pytest.raises(ValueError):
        c.validate()
    # reward and termination thresholds are independent table entries (rover_env_cfg.py:136,162 vs :173,177) and reach the
    # parameter block as such
    c = RoverEnvCfg()
    c.rewards["reached_target"].params["threshold"] = 0.5
    c.terminations["far_from_target"].params["threshold"] = 12.5
    n = c.to_native()
    assert abs(n.rew_success_threshold - 0.5) < 1e-7 and abs(n.success_threshold - 0.18) < 1e-7
    assert abs(n.rew_far_threshold - 11.0) < 1e-7 and abs(n.far_threshold - 12.5) < 1e-7
    assert n.mass_model == 1 and n.solver_iterations == 32                   # aau_rover_simple.py:33; the USD link table's statics
    c.mass_model = "lumped"
    assert c.to_native().mass_model == 0
    c.mass_model = "point"
    with pytest.raises(ValueError):
        c.to_native()
