"""ctypes binding of oracle/policy_oracle.c (CPU restatement of the policy / value forward pass).  TEST INFRASTRUCTURE ONLY:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ROVER_ORACLE_DIR: load the library from another directory (the sanitizer builds of `make -C oracle sanitized`)
_LIB_PATH = os.path.join(os.environ.get("ROVER_ORACLE_DIR") or _HERE, "libpolicy_oracle.so")
MAX_LAYERS = 8
ACT_NONE, ACT_LEAKY_RELU, ACT_TANH = 0, 1, 2


class Layer(C.Structure):
    _fields_ = [("K", C.c_int32), ("N", C.c_int32), ("act", C.c_int32), ("split_k", C.c_int32),
                ("w_off", C.c_uint32), ("b_off", C.c_uint32)]


class Desc(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("prop_dim", C.c_int32), ("enc_offset", C.c_int32), ("enc_dim", C.c_int32),
                ("n_enc", C.c_int32), ("n_mlp", C.c_int32), ("leaky_slope", C.c_float), ("layers", Layer * MAX_LAYERS)]


def build(force: bool = False) -> str:
    if os.environ.get("ROVER_ORACLE_DIR"):
        return _LIB_PATH
    src = os.path.join(_HERE, "policy_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpolicy_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.rvo_policy_forward.argtypes = [C.POINTER(Desc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                            C.c_int, C.c_void_p]
        _lib.rvo_tanhf.argtypes = [C.c_float]
        _lib.rvo_tanhf.restype = C.c_float
    return _lib


def default_desc(out_dim: int = 2, final_tanh: bool = True) -> Desc:
    """The reference architecture (get_models.py:36-62) with the split-K choice of the HIP kernel."""
    d = Desc()
    d.obs_dim, d.prop_dim, d.enc_offset, d.enc_dim, d.n_enc, d.n_mlp, d.leaky_slope = 965, 4, 3, 961, 2, 4, 0.01
    K, N = [961, 80, 64, 256, 160, 128], [80, 60, 256, 160, 128, out_dim]
    for i in range(6):
        d.layers[i].K, d.layers[i].N, d.layers[i].act, d.layers[i].split_k = K[i], N[i], ACT_LEAKY_RELU, 0
    d.layers[0].split_k = 1
    d.layers[5].split_k = 1
    d.layers[5].act = ACT_TANH if final_tanh else ACT_NONE
    return d


def desc_from(other) -> Desc:
    """Field-by-field copy of a descriptor with the same layout (the product's ctypes mirror of rover_policy_desc)."""
    d = Desc()
    for name in ("obs_dim", "prop_dim", "enc_offset", "enc_dim", "n_enc", "n_mlp", "leaky_slope"):
        setattr(d, name, getattr(other, name))
    for i in range(MAX_LAYERS):
        for name in ("K", "N", "act", "split_k"):
            setattr(d.layers[i], name, getattr(other.layers[i], name))
    return d


def forward(desc: Desc, weights, biases, obs: np.ndarray) -> np.ndarray:
    nl = desc.n_enc + desc.n_mlp
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    assert len(ws) == len(bs) == nl
    for i in range(nl):
        assert ws[i].shape == (desc.layers[i].N, desc.layers[i].K) and bs[i].shape == (desc.layers[i].N,)
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    assert obs.ndim == 2 and obs.shape[1] == desc.obs_dim
    out = np.empty((obs.shape[0], desc.layers[nl - 1].N), dtype=np.float32)
    wp = (C.c_void_p * nl)(*[w.ctypes.data for w in ws])
    bp = (C.c_void_p * nl)(*[b.ctypes.data for b in bs])
    rc = lib().rvo_policy_forward(C.byref(desc), wp, bp, obs.ctypes.data, obs.shape[0], out.ctypes.data)
    assert rc == 0
    return out


def tanhf(x: float) -> float:
    return float(lib().rvo_tanhf(C.c_float(x)))
