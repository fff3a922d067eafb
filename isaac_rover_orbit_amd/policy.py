"""Fused policy / value network inference on the MI355X: host-side mirror of the reference's models.

The reference builds its actor and critic in ``rover_envs/learning/train/get_models.py:36-62`` from
``rover_envs/envs/navigation/learning/skrl/models.py`` (``GaussianNeuralNetwork`` :39-103, ``DeterministicNeuralNetwork``
:106-163): encoder 961 -> 80 -> 60 on ``obs[:, 3:-1]``, MLP on ``cat(obs[:, 0:4], enc)`` 64 -> 256 -> 160 -> 128 ->
{2 + tanh, 1}, LeakyReLU(0.01).  ``RoverNet`` takes such a module's ``state_dict`` (or a skrl checkpoint such as the
shipped ``best_agent.pt``), packs the weights once into the MFMA fragment order and evaluates the whole network in ONE
kernel launch through the C ABI of ``include/rover_policy.h``.  Inference only (no autograd); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Mapping, Sequence

import numpy as np
import torch

from . import _lib

ENCODER_KEY = "dense_encoder.encoder_layers.{}.{}"   # models.py:28-30: Linear at even indices, activation at odd ones
MLP_KEY = "mlp.{}.{}"                                 # models.py:76-86


class RoverNet:
    """One network (policy mean or value) resident on the GPU in packed form.

    ``weights[i]`` is the torch ``Linear.weight`` of layer i, shape (N, K); the first ``n_enc`` layers are the height-scan
    encoder, the rest the MLP.  ``final_act``: "tanh" (policy, models.py:86) or "none" (value).
    """

    def __init__(self, weights: Sequence, biases: Sequence, n_enc: int = 2, final_act: str = "tanh", obs_dim: int = 965,
                 prop_dim: int = 4, leaky_slope: float = 0.01, device="cuda", n_copies: int = 4):
        if not torch.cuda.is_available():
            raise _lib.RoverHipError("RoverNet needs a ROCm GPU (no CPU fallback)")
        self._lib = _lib.load()
        self.device = torch.device(device)
        ws = [np.ascontiguousarray(torch.as_tensor(w).detach().cpu().numpy(), dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(torch.as_tensor(b).detach().cpu().numpy(), dtype=np.float32) for b in biases]
        self.desc = make_desc([w.shape for w in ws], n_enc, final_act, obs_dim, prop_dim, leaky_slope)
        nl = len(ws)
        n_floats = int(self._lib.rover_policy_packed_floats(C.byref(self.desc)))
        packed = np.empty(n_floats, dtype=np.float32)
        wp = (C.c_void_p * nl)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * nl)(*[b.ctypes.data for b in bs])
        _lib.check(self._lib.rover_policy_pack(C.byref(self.desc), wp, bp, packed.ctypes.data), "rover_policy_pack")
        # replicas of the packed weights: all workgroups stream the same bytes in lock step, copies spread them over L2 channels
        self.n_copies = int(os.environ.get("ROVER_POLICY_COPIES", n_copies))
        self.packed = torch.from_numpy(np.tile(packed, self.n_copies)).to(self.device)
        self.out_dim = int(ws[-1].shape[0])
        self.obs_dim = obs_dim

    # ---- constructors from the reference's artefacts
    @classmethod
    def from_state_dict(cls, sd: Mapping[str, torch.Tensor], final_act: str = "tanh", **kw) -> "RoverNet":
        """``sd`` = ``state_dict()`` of a reference model (keys ``dense_encoder.encoder_layers.<2i>.*``, ``mlp.<2i>.*``)."""
        ws, bs, n_enc = [], [], 0
        i = 0
        while ENCODER_KEY.format(2 * i, "weight") in sd:
            ws.append(sd[ENCODER_KEY.format(2 * i, "weight")]); bs.append(sd[ENCODER_KEY.format(2 * i, "bias")])
            i += 1
        n_enc = i
        i = 0
        while MLP_KEY.format(2 * i, "weight") in sd:
            ws.append(sd[MLP_KEY.format(2 * i, "weight")]); bs.append(sd[MLP_KEY.format(2 * i, "bias")])
            i += 1
        if i == 0:
            raise ValueError("state_dict has no mlp.<i>.weight entries")
        return cls(ws, bs, n_enc=n_enc, final_act=final_act, **kw)

    @classmethod
    def from_checkpoint(cls, path: str, role: str = "policy", **kw) -> "RoverNet":
        """skrl agent checkpoint (``agent.save``): ``{"policy": state_dict, "value": state_dict, ...}`` (eval.py:146-155)."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        return cls.from_state_dict(ck[role], final_act="tanh" if role == "policy" else "none", **kw)

    # ---- inference
    @torch.no_grad()
    def forward(self, obs: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if obs.dim() != 2 or obs.shape[1] != self.obs_dim or obs.dtype != torch.float32 or not obs.is_cuda:
            raise ValueError(f"obs must be a float32 cuda tensor of shape (n, {self.obs_dim})")
        obs = obs.contiguous()
        n = int(obs.shape[0])
        if out is None:
            out = torch.empty((n, self.out_dim), dtype=torch.float32, device=obs.device)
        if obs.device != self.packed.device or out.device != obs.device:
            raise ValueError(f"obs / out must live on the network's device {self.packed.device}")
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        with torch.cuda.device(obs.device):     # the entry point launches on the thread's current device
            _lib.check(self._lib.rover_policy_forward(C.byref(self.desc), self.packed.data_ptr(), self.n_copies, obs.data_ptr(),
                                                      n, out.data_ptr(), stream), "rover_policy_forward")
        return out

    __call__ = forward

    def act(self, obs) -> torch.Tensor:
        """Deterministic action = the Gaussian mean (skrl evaluation mode); accepts the env's ``{"policy": obs}`` dict."""
        if isinstance(obs, dict):
            obs = obs["policy"]
        return self.forward(obs)


@torch.no_grad()
def forward_pair(net_a: RoverNet, net_b: RoverNet, obs: torch.Tensor, out_a: torch.Tensor | None = None,
                 out_b: torch.Tensor | None = None):
    """Both networks on the same observation rows in ONE launch (``rover_policy_forward_pair``): what a PPO rollout step asks
    for -- ``policy.act(states)`` and ``value.act(states)`` (skrl_utils.py:114-135) -- with the rows fetched once.  Bit-identical
    to ``net_a(obs), net_b(obs)``; falls back to exactly those two calls when a network is not the reference architecture."""
    if isinstance(obs, dict):
        obs = obs["policy"]
    if obs.dim() != 2 or obs.shape[1] != net_a.obs_dim or obs.dtype != torch.float32 or not obs.is_cuda:
        raise ValueError(f"obs must be a float32 cuda tensor of shape (n, {net_a.obs_dim})")
    obs = obs.contiguous()
    n = int(obs.shape[0])
    if out_a is None:
        out_a = torch.empty((n, net_a.out_dim), dtype=torch.float32, device=obs.device)
    if out_b is None:
        out_b = torch.empty((n, net_b.out_dim), dtype=torch.float32, device=obs.device)
    if net_a.n_copies != net_b.n_copies or obs.device != net_a.packed.device or obs.device != net_b.packed.device:
        return net_a(obs, out_a), net_b(obs, out_b)
    stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
    with torch.cuda.device(obs.device):
        rc = net_a._lib.rover_policy_forward_pair(C.byref(net_a.desc), net_a.packed.data_ptr(), C.byref(net_b.desc), net_b.packed.data_ptr(),
                                                  net_a.n_copies, obs.data_ptr(), n, out_a.data_ptr(), out_b.data_ptr(), stream)
    if rc == 4:      # ROVER_ERR_UNSUPPORTED: not the reference architecture
        return net_a(obs, out_a), net_b(obs, out_b)
    _lib.check(rc, "rover_policy_forward_pair")
    return out_a, out_b


def make_desc(shapes, n_enc: int, final_act: str, obs_dim: int, prop_dim: int, leaky_slope: float) -> "_lib.PolicyDesc":
    nl = len(shapes)
    if nl > _lib.POLICY_MAX_LAYERS or nl - n_enc < 1:
        raise ValueError("unsupported number of layers")
    d = _lib.PolicyDesc()
    d.obs_dim, d.prop_dim, d.leaky_slope, d.n_enc, d.n_mlp = obs_dim, prop_dim, leaky_slope, n_enc, nl - n_enc
    if n_enc > 0:
        d.enc_dim = int(shapes[0][1])
        d.enc_offset = prop_dim - 1          # models.py:95  states[:, self.mlp_input_size - 1:-1]
        if d.enc_offset + d.enc_dim != obs_dim - 1:
            raise ValueError(f"encoder width {d.enc_dim} does not match obs[:, {d.enc_offset}:-1] of a {obs_dim}-wide row")
    for i, (n, k) in enumerate(shapes):
        lay = d.layers[i]
        lay.K, lay.N, lay.act = int(k), int(n), _lib.ACT_LEAKY_RELU
        # numerics contract (rover_policy.h): wide-K layers with few column tiles split K over the four waves
        tiles = (int(n) + 15) // 16
        lay.split_k = 1 if tiles <= 6 and (int(k) >= 512 or (tiles < 4 and int(k) >= 128)) else 0
    d.layers[nl - 1].act = {"tanh": _lib.ACT_TANH, "none": _lib.ACT_NONE}[final_act]
    return d
