"""GPU parity of the fused policy / value forward pass.

  * HIP vs the C oracle (same k-ordered fmaf chains, same split-K cuts, same tanh sequence): BIT-EXACT (values; the sign
    of an exact zero is not compared)
  * HIP vs plain torch fp32 on the GPU (rocBLAS GEMMs, different accumulation order): |diff| <= 2e-5 on O(1) outputs
"""
import numpy as np
import pytest
import torch

from helpers import random_policy_weights, synthetic_obs, torch_policy_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def po():
    from oracle import policy_oracle
    policy_oracle.build()
    return policy_oracle


@pytest.mark.parametrize("out_dim,final_act", [(2, "tanh"), (1, "none")])
@pytest.mark.parametrize("n", [1, 16, 50, 1000])
def test_forward_bit_exact_vs_oracle(po, out_dim, final_act, n):
    from isaac_rover_orbit_amd.policy import RoverNet
    ws, bs = random_policy_weights(seed=10 + out_dim, out_dim=out_dim, scale=3.0)
    net = RoverNet(ws, bs, n_enc=2, final_act=final_act)
    obs = synthetic_obs(n, seed=n)
    got = net(torch.from_numpy(obs).cuda()).cpu().numpy()
    ref = po.forward(po.desc_from(net.desc), ws, bs, obs)
    assert got.shape == (n, out_dim)
    assert np.array_equal(got, ref), f"max |diff| {np.abs(got - ref).max()}"
    tor = torch_policy_reference(ws, bs, obs, final_act == "tanh", device="cuda")
    assert np.abs(got - tor).max() <= 2e-5


def test_full_batch_and_from_state_dict(po):
    """N = 4096 (the bench size) through the state_dict constructor, keys as in the reference's modules."""
    from isaac_rover_orbit_amd.policy import RoverNet
    ws, bs = random_policy_weights(seed=2, scale=2.0)
    sd = {}
    for i, j in enumerate((0, 2)):
        sd[f"dense_encoder.encoder_layers.{j}.weight"] = torch.from_numpy(ws[i])
        sd[f"dense_encoder.encoder_layers.{j}.bias"] = torch.from_numpy(bs[i])
    for i, j in enumerate((0, 2, 4, 6)):
        sd[f"mlp.{j}.weight"] = torch.from_numpy(ws[2 + i])
        sd[f"mlp.{j}.bias"] = torch.from_numpy(bs[2 + i])
    sd["log_std_parameter"] = torch.zeros(2)
    net = RoverNet.from_state_dict(sd)
    obs = synthetic_obs(4096, seed=1)
    got = net.act({"policy": torch.from_numpy(obs).cuda()}).cpu().numpy()
    ref = po.forward(po.desc_from(net.desc), ws, bs, obs)
    assert np.array_equal(got, ref)
    assert np.abs(got).max() <= 1.0 and np.abs(got).mean() > 0.01
    # linearity in the last layer's bias is a size-independent property: shifting the pre-activation moves tanh monotonically
    bs2 = [b.copy() for b in bs]
    bs2[5] += 0.25
    got2 = RoverNet(ws, bs2)(torch.from_numpy(obs).cuda()).cpu().numpy()
    assert (got2 >= got).all() and (got2 > got).mean() > 0.99


def test_closed_loop_with_the_env(po):
    """Policy kernel and env kernels back to back on one stream: obs -> action -> step, 20 steps, actions stay bounded and
    equal the oracle's on the same observations."""
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv
    from isaac_rover_orbit_amd.policy import RoverNet
    from helpers import small_procedural
    ter = small_procedural()
    ter.make_spawns(2 * 128)
    cfg = RoverEnvCfg(); cfg.scene.num_envs = 128; cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=ter)
    ws, bs = random_policy_weights(seed=4, scale=3.0)
    net = RoverNet(ws, bs)
    obs, _ = env.reset()
    for _ in range(20):
        a = net.act(obs)
        o = obs["policy"].cpu().numpy()
        if np.isfinite(o).all():
            assert np.array_equal(a.cpu().numpy(), po.forward(po.desc_from(net.desc), ws, bs, o))
        obs, rew, term, trunc, info = env.step(a)
    assert torch.isfinite(rew).all()
    env.close()


def test_bad_arguments_are_reported():
    from isaac_rover_orbit_amd.policy import RoverNet
    ws, bs = random_policy_weights(seed=0)
    net = RoverNet(ws, bs)
    with pytest.raises(ValueError):
        net(torch.zeros(4, 964, device="cuda"))
    from isaac_rover_orbit_amd._lib import RoverHipError
    with pytest.raises(RoverHipError, match="chain"):
        RoverNet(ws[:2] + [ws[2][:, :60]] + ws[3:], bs)   # MLP input narrower than prop + encoder output: wrong obs split
