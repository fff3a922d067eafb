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


def test_config3_arithmetic_eight_ranks_of_4096(oracle):
    """BASELINE config 3 (num_envs = 32768 sharded over 8 GPUs) has never met eight devices; its only arithmetic that no other test
    touches is pinned here on the CPU: `weak_shard(4096, r, 8)`, a spawn table of 2 x 32768 rows
    (terrain_utils.py:123-124: n_spawns = 2 * num_envs, GLOBAL), and the per-call affine spawn permutation -- every rank derives the
    same (a, b) from (seed, call counter) without communication, and the rows of the 32768 global env ids are pairwise distinct
    across ALL ranks, for the reset call and for a later step call."""
    from isaac_rover_orbit_amd import distributed as rd
    from helpers import flat
    ro = oracle
    world, per_rank = 8, 4096
    shards = [rd.weak_shard(per_rank, r, world) for r in range(world)]
    assert [s.env_id_offset for s in shards] == [r * per_rank for r in range(world)]
    assert all(s.global_num_envs == 32768 and s.local_num_envs == per_rank for s in shards)
    n_global = shards[0].global_num_envs
    n_spawns = 2 * n_global
    # a spawn table whose x coordinate IS the row index: the row an env drew can be read back from its position
    table = np.stack([np.arange(n_spawns, dtype=np.float32) * 0.001 + 20.0, np.full(n_spawns, 25.0, np.float32),
                      np.zeros(n_spawns, np.float32)], 1)
    ter = flat()
    t = ro.TerrainData(ter.height, ter.obstacle, ter.safe_mask if hasattr(ter, "safe_mask") else ter.safe_rock_mask, ter.resolution,
                       ter.min_x, ter.min_y, table)
    for counter in (0, 7):                      # the reset call of a fresh env; some later call
        rows_all = []
        for s in shards[::3] + [shards[-1]]:    # ranks 0, 3, 6, 7: a 32768-env oracle reset per rank would only repeat the arithmetic
            cfg = ro.default_config(seed_lo=11)
            cfg.counter_lo = counter
            S = ro.new_state(512)               # the first 512 ids of the rank's range (the bijection is over ids, not over batches)
            ro.reset_all(cfg, t, S, env_id_offset=s.env_id_offset)
            rows = np.rint((S[:, 0].astype(np.float64) - 20.0) / 0.001).astype(np.int64)
            assert rows.min() >= 0 and rows.max() < n_spawns
            rows_all.append((s.env_id_offset, rows))
        # the same (a, b) on every rank: row = (a * gid + b) mod n_spawns -- recover (a, b) from rank 0 and predict the others
        off0, r0 = rows_all[0]
        b = int(r0[0])                                      # gid 0
        a = int((r0[1] - r0[0]) % n_spawns)                 # gid 1
        assert np.gcd(a, n_spawns) == 1                     # a bijection of Z / n_spawns
        for off, rows in rows_all:
            gid = off + np.arange(rows.size, dtype=np.int64)
            assert np.array_equal(rows, (a * gid + b) % n_spawns), (counter, off)
        # hence distinct over the WHOLE global batch (n_spawns >= global num_envs: gid -> row is injective)
        all_rows = (a * np.arange(n_global, dtype=np.int64) + b) % n_spawns
        assert np.unique(all_rows).size == n_global
    # the product refuses the configuration that would break it: a spawn table shorter than the global batch (checked on the host
    # in RoverEnv and in rover_set_terrain; here: the rule itself)
    assert n_spawns >= shards[-1].env_id_offset + per_rank


def test_bench_sharding_arguments_for_eight_ranks():
    """bench.py's rank -> shard -> cfg path with WORLD_SIZE = 8 (no process group, no GPU): ids, global env count and the size of
    the spawn table it would build."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from isaac_rover_orbit_amd import distributed as rd
    for rank in (0, 5, 7):
        shard = rd.weak_shard(4096, rank, 8)
        cfg, n_spawns = bench.shard_cfg(4096, shard, config=2)
        assert cfg.scene.num_envs == 4096 and cfg.env_id_offset == 4096 * rank and cfg.global_num_envs == 32768
        assert n_spawns == 2 * 32768
