"""CPU tests of the policy row: the C oracle of the forward pass against plain torch fp32 (random weights
of the reference architecture, and the reference's own checkpoint where /root/reference is mounted), the host-side
weight packing of librover_hip.so, and the descriptor defaults.  No GPU work."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import policy_forward_fixture, random_policy_weights, synthetic_obs, torch_policy_reference

CKPT = "/root/reference/rover_envs/envs/navigation/robots/aau_rover/policies/best_agent.pt"


@pytest.fixture(scope="module")
def po():
    from oracle import policy_oracle
    policy_oracle.build()
    return policy_oracle


@pytest.mark.parametrize("out_dim,final_tanh", [(2, True), (1, False)])
def test_oracle_matches_torch_fp32(po, out_dim, final_tanh):
    """Tolerance: fp32 accumulation order differs (fmaf chain vs BLAS blocking): |diff| <= 2e-5 on O(1) outputs."""
    ws, bs = random_policy_weights(seed=out_dim, out_dim=out_dim, scale=3.0)
    obs = synthetic_obs(300, seed=7)
    ref = torch_policy_reference(ws, bs, obs, final_tanh)
    got = po.forward(po.default_desc(out_dim, final_tanh), ws, bs, obs)
    assert got.shape == ref.shape == (300, out_dim)
    assert np.abs(ref).max() > 0.05 and np.abs(got - ref).max() <= 2e-5


def test_oracle_matches_the_reference_networks_fixture(po, golden_dir):
    """SURVEY 8f-3 pin: the oracle against outputs recorded from the reference's OWN model classes (models.py:89-102, 151-163)
    -- not against this repository's torch restatement.  Tolerance: fp32 accumulation order (fmaf chain vs torch's CPU GEMM)
    on O(1) outputs, |diff| <= 2e-5."""
    nets, obs, mean, value = policy_forward_fixture(golden_dir)
    got_a = po.forward(po.default_desc(2, True), *nets["policy"], obs)
    got_v = po.forward(po.default_desc(1, False), *nets["value"], obs)
    assert got_a.shape == mean.shape == (320, 2) and got_v.shape == value.shape == (320, 1)
    assert np.abs(mean).max() > 0.5 and np.abs(value).max() > 1.0                     # the fixture is not degenerate
    assert np.abs(got_a - mean).max() <= 2e-5, np.abs(got_a - mean).max()
    assert np.abs(got_v - value).max() <= 2e-5 * max(1.0, float(np.abs(value).max())), np.abs(got_v - value).max()
    # the builder's torch restatement (used by the other tests as the fp32 reference) agrees with the reference's classes too
    assert np.abs(torch_policy_reference(*nets["policy"], obs, True) - mean).max() <= 2e-6


@pytest.mark.skipif(not os.path.exists(CKPT), reason="reference checkpoint not mounted (build container only)")
def test_oracle_matches_the_reference_networks_with_the_shipped_checkpoint(po, golden_dir):
    """best_agent.pt loaded into the reference's own classes (as eval.py:146-149 does) vs the oracle on the same weights."""
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved_path, before = list(sys.path), set(sys.modules)
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import gen_golden as gg
        models = gg._load_reference_models()
        ck = torch.load(CKPT, map_location="cpu", weights_only=False)
        nets = gg.reference_networks(models, {})
        _, obs, _, _ = policy_forward_fixture(golden_dir)
        for role, out_dim, final_tanh in (("policy", 2, True), ("value", 1, False)):
            nets[role].load_state_dict(ck[role])
            with torch.no_grad():
                ref = nets[role].compute({"states": torch.from_numpy(obs)}, role=role)[0].numpy()
            sd = ck[role]
            ws = [sd[f"dense_encoder.encoder_layers.{i}.weight"].numpy() for i in (0, 2)] + [sd[f"mlp.{i}.weight"].numpy() for i in (0, 2, 4, 6)]
            bs = [sd[f"dense_encoder.encoder_layers.{i}.bias"].numpy() for i in (0, 2)] + [sd[f"mlp.{i}.bias"].numpy() for i in (0, 2, 4, 6)]
            got = po.forward(po.default_desc(out_dim, final_tanh), ws, bs, obs)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), (role, np.abs(got - ref).max())
    finally:
        sys.path[:] = saved_path
        for name in set(sys.modules) - before:
            if name.split(".")[0] in ("rover_envs", "skrl", "omni", "carb", "pxr", "pymeshlab", "cv2", "gen_golden", "reference_stubs", "_ref_skrl_models"):
                sys.modules.pop(name, None)


@pytest.mark.skipif(not os.path.exists(CKPT), reason="reference checkpoint not mounted (build container only)")
def test_oracle_matches_torch_on_reference_checkpoint(po):
    import torch
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    obs = synthetic_obs(128, seed=3)
    for role, out_dim, final_tanh in (("policy", 2, True), ("value", 1, False)):
        sd = ck[role]
        ws = [sd[f"dense_encoder.encoder_layers.{i}.weight"].numpy() for i in (0, 2)] + [sd[f"mlp.{i}.weight"].numpy() for i in (0, 2, 4, 6)]
        bs = [sd[f"dense_encoder.encoder_layers.{i}.bias"].numpy() for i in (0, 2)] + [sd[f"mlp.{i}.bias"].numpy() for i in (0, 2, 4, 6)]
        assert [w.shape for w in ws] == [(80, 961), (60, 80), (256, 64), (160, 256), (128, 160), (out_dim, 128)]
        ref = torch_policy_reference(ws, bs, obs, final_tanh)
        got = po.forward(po.default_desc(out_dim, final_tanh), ws, bs, obs)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_tanh_sequence_accuracy(po):
    xs = np.concatenate([np.linspace(-12, 12, 4001), [0.0, 1e-8, -1e-8, 0.6249, 0.625, 50.0, -50.0]]).astype(np.float32)
    got = np.array([po.tanhf(float(x)) for x in xs], dtype=np.float32)
    assert np.abs(got - np.tanh(xs.astype(np.float64))).max() <= 3e-7


def test_default_desc_and_packing_of_the_library(po):
    from isaac_rover_orbit_amd import _lib, policy
    lib = _lib.load()
    for out_dim, tanh in ((2, 1), (1, 0)):
        d = _lib.PolicyDesc()
        assert lib.rover_policy_default_desc(C.byref(d), out_dim, tanh) == 0
        o = po.default_desc(out_dim, bool(tanh))
        for name in ("obs_dim", "prop_dim", "enc_offset", "enc_dim", "n_enc", "n_mlp"):
            assert getattr(d, name) == getattr(o, name), name
        assert abs(d.leaky_slope - 0.01) < 1e-9
        for i in range(6):
            assert (d.layers[i].K, d.layers[i].N, d.layers[i].act, d.layers[i].split_k) == \
                   (o.layers[i].K, o.layers[i].N, o.layers[i].act, o.layers[i].split_k)
        # the python mirror derives the same descriptor from the weight shapes alone
        m = policy.make_desc([(l.N, l.K) for l in list(d.layers)[:6]], 2, "tanh" if tanh else "none", 965, 4, 0.01)
        for i in range(6):
            assert (m.layers[i].K, m.layers[i].N, m.layers[i].act, m.layers[i].split_k) == \
                   (d.layers[i].K, d.layers[i].N, d.layers[i].act, d.layers[i].split_k)
    # packing: fragment order of the f32 16x16x4 MFMA B operand, zero padding, offsets
    ws, bs = random_policy_weights(seed=5)
    d = _lib.PolicyDesc()
    lib.rover_policy_default_desc(C.byref(d), 2, 1)
    n = lib.rover_policy_packed_floats(C.byref(d))
    packed = np.full(n, np.nan, dtype=np.float32)
    wp = (C.c_void_p * 6)(*[w.ctypes.data for w in ws])
    bp = (C.c_void_p * 6)(*[b.ctypes.data for b in bs])
    assert lib.rover_policy_pack(C.byref(d), wp, bp, packed.ctypes.data) == 0
    assert not np.isnan(packed).any()
    off = 0
    for li in range(6):
        K, N = ws[li].shape[1], ws[li].shape[0]
        G, T = -(-K // 16), -(-N // 16)
        assert d.layers[li].w_off == off
        blk = packed[off:off + T * G * 256].reshape(T, G, 64, 4)
        full = np.zeros((T * 16, G * 16), np.float32)
        full[:N, :K] = ws[li]
        lane = np.arange(64)
        for j in range(4):
            expect = np.stack([full[16 * t + (lane & 15), 16 * g + 4 * j + (lane >> 4)] for t in range(T) for g in range(G)])
            assert np.array_equal(blk[:, :, :, j].reshape(T * G, 64), expect)
        off += T * G * 256
        assert d.layers[li].b_off == off
        nb = (N + 3) & ~3
        assert np.array_equal(packed[off:off + N], bs[li]) and not packed[off + N:off + nb].any()
        off += nb
    assert off == n


def test_pack_rejects_inconsistent_descriptors():
    from isaac_rover_orbit_amd import _lib
    lib = _lib.load()
    d = _lib.PolicyDesc()
    lib.rover_policy_default_desc(C.byref(d), 2, 1)
    d.layers[2].K = 63
    assert lib.rover_policy_packed_floats(C.byref(d)) > 0
    assert lib.rover_policy_pack(C.byref(d), None, None, None) != 0
    assert b"chain" in lib.rover_last_error()
