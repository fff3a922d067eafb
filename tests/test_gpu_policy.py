"""GPU parity of the fused policy / value forward pass.

  * HIP vs the C oracle (same k-ordered fmaf chains, same split-K cuts, same tanh sequence): BIT-EXACT (values; the sign
    of an exact zero is not compared)
  * HIP vs plain torch fp32 on the GPU (rocBLAS GEMMs, different accumulation order): |diff| <= 2e-5 on O(1) outputs
"""
import numpy as np
import pytest
import torch

from helpers import policy_forward_fixture, random_policy_weights, synthetic_obs, torch_policy_reference

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


def test_forward_and_pair_match_the_reference_networks_fixture(po, golden_dir):
    """SURVEY 8f-3 pin on the GPU: rover_policy_forward and rover_policy_forward_pair against the outputs of the reference's OWN
    GaussianNeuralNetwork / DeterministicNeuralNetwork (tests/golden/policy_forward.npz; models.py:89-102, 151-163), |diff| <= 2e-5
    (fp32 accumulation order: MFMA k-chains vs torch's CPU GEMM), and bit-exact against the oracle on the same rows."""
    from isaac_rover_orbit_amd.policy import RoverNet, forward_pair
    nets, obs, mean, value = policy_forward_fixture(golden_dir)
    actor = RoverNet(*nets["policy"], n_enc=2, final_act="tanh")
    critic = RoverNet(*nets["value"], n_enc=2, final_act="none")
    o = torch.from_numpy(obs).cuda()
    a1, v1 = actor(o).cpu().numpy(), critic(o).cpu().numpy()
    a2, v2 = (t.cpu().numpy() for t in forward_pair(actor, critic, o))
    vtol = 2e-5 * max(1.0, float(np.abs(value).max()))
    for a, v in ((a1, v1), (a2, v2)):
        assert np.abs(a - mean).max() <= 2e-5 and np.abs(v - value).max() <= vtol
    assert np.array_equal(a1, a2) and np.array_equal(v1, v2)
    assert np.array_equal(a1, po.forward(po.desc_from(actor.desc), *nets["policy"], obs))
    assert np.array_equal(v1, po.forward(po.desc_from(critic.desc), *nets["value"], obs))


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


def _generic_case(po, shapes, n_enc, final_act, obs_dim, prop_dim, n, seed, force_split=None):
    """Random network of arbitrary shape through RoverNet and through the oracle with the same descriptor."""
    from isaac_rover_orbit_amd.policy import RoverNet
    rng = np.random.RandomState(seed)
    ws = [(rng.uniform(-1, 1, s) / np.sqrt(s[1]) * 2.0).astype(np.float32) for s in shapes]
    bs = [(rng.uniform(-1, 1, (s[0],)) * 0.1).astype(np.float32) for s in shapes]
    net = RoverNet(ws, bs, n_enc=n_enc, final_act=final_act, obs_dim=obs_dim, prop_dim=prop_dim)
    if force_split is not None:
        for i, f in enumerate(force_split):
            net.desc.layers[i].split_k = f
    obs = rng.standard_normal((n, obs_dim)).astype(np.float32)
    got = net(torch.from_numpy(obs).cuda()).cpu().numpy()
    ref = po.forward(po.desc_from(net.desc), ws, bs, obs)
    assert np.array_equal(got, ref), f"max |diff| {np.abs(got - ref).max()}"
    return got


def test_generic_architectures(po):
    """The kernel is not specialised to 961-80-60 / 256-160-128: odd widths, no encoder, K not a multiple of 16, more than
    six column tiles in a split-K layer (several passes), a tile-parallel layer with more than 32 tiles."""
    # MLP only (no encoder): obs 10 -> 33 -> 7, linear head
    _generic_case(po, [(33, 10), (7, 33)], 0, "none", 10, 10, 37, seed=1)
    # encoder over obs[:, 5:-1] of a 70-wide row (64 inputs), prop 6; widths that are not multiples of 16
    _generic_case(po, [(20, 64), (9, 20), (50, 15), (3, 50)], 2, "tanh", 70, 6, 100, seed=2)
    # split-K layer with 13 column tiles (three passes of <= 6) and a 600-wide K; then a 24-tile tile-parallel layer
    _generic_case(po, [(200, 600), (384, 204), (5, 384)], 1, "none", 604, 4, 48, seed=3, force_split=[1, 0, 1])
    # a network that does not fit the LDS is refused with an error code, not a crash
    from isaac_rover_orbit_amd._lib import RoverHipError
    with pytest.raises(RoverHipError, match="LDS"):
        _generic_case(po, [(200, 600), (1024, 204), (5, 1024)], 1, "none", 604, 4, 16, seed=3)
    # every layer forced to split-K, and every layer forced to a single chain
    shapes = [(80, 961), (60, 80), (256, 64), (160, 256), (128, 160), (2, 128)]
    a = _generic_case(po, shapes, 2, "tanh", 965, 4, 33, seed=4, force_split=[1] * 6)
    b = _generic_case(po, shapes, 2, "tanh", 965, 4, 33, seed=4, force_split=[0] * 6)
    assert np.abs(a - b).max() < 1e-5 and not np.array_equal(a, b)   # same network, different (documented) summation cuts


def test_forward_pair_equals_two_forwards():
    """rover_policy_forward_pair (actor + critic of a rollout step on one staged tile, one launch) is bit-identical to two
    rover_policy_forward calls -- for full batches, a ragged last tile and a single row; other architectures fall back."""
    from isaac_rover_orbit_amd.policy import RoverNet, forward_pair
    wa, ba = random_policy_weights(seed=3, out_dim=2)
    wc, bc = random_policy_weights(seed=4, out_dim=1)
    actor = RoverNet(wa, ba, n_enc=2, final_act="tanh")
    critic = RoverNet(wc, bc, n_enc=2, final_act="none")
    for n in (4096, 333, 1):
        obs = torch.from_numpy(synthetic_obs(n, seed=n)).cuda()
        a1, v1 = actor(obs), critic(obs)
        a2, v2 = forward_pair(actor, critic, obs)
        assert a2.shape == (n, 2) and v2.shape == (n, 1)
        assert torch.equal(a1, a2) and torch.equal(v1, v2), n
    # an actor with a wider head (6 outputs: its padded last-layer bias is 8 floats, the critic's 4) and two weight
    # replicas -- the critic's replica stride is its own, not the actor's
    wa6, ba6 = random_policy_weights(seed=9, out_dim=6)
    actor6 = RoverNet(wa6, ba6, n_enc=2, final_act="tanh", n_copies=2)
    critic2 = RoverNet(wc, bc, n_enc=2, final_act="none", n_copies=2)
    obs = torch.from_numpy(synthetic_obs(333, seed=5)).cuda()
    a1, v1 = actor6(obs), critic2(obs)
    a2, v2 = forward_pair(actor6, critic2, obs)
    assert a2.shape == (333, 6) and torch.equal(a1, a2) and torch.equal(v1, v2) and torch.equal(v1, critic(obs))
    # different hidden-layer slopes: refused by the pair entry (ROVER_ERR_UNSUPPORTED), the binding falls back to two launches
    leaky = RoverNet(wc, bc, n_enc=2, final_act="none", leaky_slope=0.2)
    s1, s2 = forward_pair(actor, leaky, obs)
    assert torch.equal(s1, actor(obs)) and torch.equal(s2, leaky(obs)) and not torch.equal(s2, critic(obs))
    # a non-reference architecture takes the two-call path (same results by construction)
    rng = np.random.RandomState(0)
    ws = [rng.uniform(-1, 1, (24, 4)).astype(np.float32) / 2, rng.uniform(-1, 1, (3, 24)).astype(np.float32) / 5]
    bs = [np.zeros(24, np.float32), np.zeros(3, np.float32)]
    small = RoverNet(ws, bs, n_enc=0, final_act="none")          # an MLP on the four proprioceptive columns only
    obs = torch.from_numpy(synthetic_obs(64, seed=1)).cuda()
    s1, v1 = small(obs), critic(obs)
    s2, v2 = forward_pair(small, critic, obs)
    assert torch.equal(s1, s2) and torch.equal(v1, v2)
