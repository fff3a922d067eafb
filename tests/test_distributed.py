"""Multi-process path on CPU: world_size-2 gloo.  The step path has no collective (envs are independent, RNG keyed by
global id); what is exercised here is the sharding arithmetic, the rollout all_gather and the log reduction, plus
shard-invariance of the simulated trajectories (through the oracle, since the HIP path needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from helpers import oracle_terrain, small_procedural
    from isaac_rover_orbit_amd import distributed as rd
    from oracle import rover_oracle as ro
    r, w, lr = rd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.is_initialized()
    n_global, T_roll = 96, 5
    shard = rd.shard_envs(n_global, rank, world)
    ter = small_procedural()
    cfg, t = ro.default_config(seed_lo=5), oracle_terrain(ro, ter, 2 * n_global)
    S = ro.new_state(shard.local_num_envs)
    ro.reset_all(cfg, t, S, env_id_offset=shard.env_id_offset)
    rng = np.random.RandomState(3)
    acts = rng.uniform(-1, 1, (T_roll, n_global, 2)).astype(np.float32)
    lo, hi = shard.env_id_offset, shard.env_id_offset + shard.local_num_envs
    obs_roll = torch.zeros(T_roll, shard.local_num_envs, 965)
    log = np.zeros(16, np.float32)
    for k in range(T_roll):
        o, rew, term, trunc, force, log = ro.step(cfg, t, S, acts[k, lo:hi], env_id_offset=lo, log=log)
        obs_roll[k] = torch.from_numpy(o)
    gat = rd.RolloutGatherer()
    full = gat.gather(obs_roll)                                    # (world, T, n_local, 965)
    glog = rd.reduce_log(torch.from_numpy(log.copy()))
    if rank == 0:
        torch.save({"full": full, "glog": glog}, os.path.join(tmp, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_rollout_gather(tmp_path, oracle):
    from helpers import oracle_terrain, small_procedural
    from isaac_rover_orbit_amd import distributed as rd
    world, n_global, T_roll = 2, 96, 5
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"))
    # single-process reference of the same global batch
    ro = oracle
    ter = small_procedural()
    cfg, t = ro.default_config(seed_lo=5), oracle_terrain(ro, ter, 2 * n_global)
    S = ro.new_state(n_global)
    ro.reset_all(cfg, t, S)
    rng = np.random.RandomState(3)
    acts = rng.uniform(-1, 1, (T_roll, n_global, 2)).astype(np.float32)
    ref = np.zeros((T_roll, n_global, 965), np.float32)
    log = np.zeros(16, np.float32)
    for k in range(T_roll):
        ref[k], _, _, _, _, log = ro.step(cfg, t, S, acts[k], log=log)
    full = got["full"].numpy()
    assert full.shape == (2, T_roll, 48, 965)
    assert np.array_equal(np.concatenate([full[0], full[1]], axis=1), ref)       # sharding does not change results
    assert float(got["glog"][13]) == float(log[13]) or float(log[13]) == 0
    sh = [rd.shard_envs(10, r, 3) for r in range(3)]
    assert [(s.env_id_offset, s.local_num_envs) for s in sh] == [(0, 4), (4, 3), (7, 3)]
    ws = rd.weak_shard(4096, 5, 8)
    assert ws.env_id_offset == 5 * 4096 and ws.global_num_envs == 32768
    with pytest.raises(ValueError):
        rd.shard_envs(2, 0, 4)
