#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of AAURoverEnv-v0 at num_envs=4096 per GPU (BASELINE.json metric, config 2).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one ``RoverEnv.step()`` over the rank's batch of envs: the two HIP kernels of the hot path, launched through
the C ABI exactly as a trainer would (random actions pre-generated in HBM, in-kernel resets included).  Weak scaling:
every rank simulates ``--num-envs`` envs (global ids sharded by rank, terrain replicated, no data-path collective).
Rank 0 prints ONE JSON line.  Extra legs (not in the timed region):
  * roofline      -- HIP-event duration of the dominant kernel vs its algorithmic bytes (SURVEY 8d / DESIGN.md)
  * cpu_baseline  -- the CPU oracle ("port") timed on a bounded sample of the same workload (N=1 only)
  * rollout_gather (N>1) -- one RCCL all_gather of a 60-step rollout shard (BASELINE config 3)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
# algorithmic bytes per env-step: obs write 4*(4+R), heightfield reads 4*R, state r+w 2*4*52, action 8,
# reward+flags 6, wheel samples 6*6*4
def algorithmic_bytes(rays: int):
    scan = 4 * (4 + rays) + 4 * rays
    dyn = 2 * 4 * 52 + 8 + 6 + 6 * 6 * 4
    return scan, dyn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forces", action="store_true", help="do not materialise contact_sensor.force_matrix_w")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from isaac_rover_orbit_amd import distributed as rd
    from isaac_rover_orbit_amd import terrain as T
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv

    rank, world, local_rank = rd.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a ROCm GPU (the rover hot path has no CPU fallback)")
    dev_index = local_rank % n_dev          # one rank per GPU in production; ranks share a card only in gloo rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    nccl = world > 1 and dist.get_backend() == "nccl"
    n = args.num_envs
    shard = rd.weak_shard(n, rank, world)

    # ---- workload: SURVEY 8d config 2 (procedural 2048^2 heightfield @ 0.05 m, fBm sigma_z 0.15 m seed 1234, ~400 rocks)
    ter = T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=0.15, n_rocks=400)
    ter.make_spawns(2 * shard.global_num_envs, seed=41)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    cfg.sim.device = str(dev)
    cfg.terrain.kind = "custom"
    cfg.env_id_offset = shard.env_id_offset
    cfg.global_num_envs = shard.global_num_envs
    cfg.record_contact_forces = not args.no_forces
    env = RoverEnv(cfg, terrain=ter)

    total = args.steps + args.warmup
    g = torch.Generator(device=dev).manual_seed(rank)          # torch's CUDA generator is Philox; seed 0 on rank 0
    n_act = min(total, 2048)
    actions = torch.rand(n_act, n, 2, device=dev, generator=g) * 2 - 1
    env.reset()
    for k in range(args.warmup):
        env.step(actions[k % n_act])

    def barrier():
        if world > 1:
            dist.barrier()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        env.step(actions[k % n_act])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if nccl else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = shard.global_num_envs * args.steps / elapsed

    # ---- roofline leg: HIP-event duration of each kernel (on the stream the kernels are launched on)
    ms1 = ms2 = 0.0
    ps = max(args.profile_steps, 1)
    for k in range(ps):
        a, b = env.profile_step(actions[k % n_act])
        ms1 += a
        ms2 += b
    ms1, ms2 = ms1 / ps, ms2 / ps
    scan_b, dyn_b = algorithmic_bytes(env.num_rays)
    kernels = {
        "rover_step_kernel": {"ms": ms1, "algorithmic_bytes": dyn_b * n, "GB/s": dyn_b * n / (ms1 * 1e-3) / 1e9},
        "rover_scan_obs_kernel": {"ms": ms2, "algorithmic_bytes": scan_b * n, "GB/s": scan_b * n / (ms2 * 1e-3) / 1e9},
    }
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    traffic = None
    tr_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tr_path):
        try:
            traffic = json.load(open(tr_path)).get(dom, {}).get("bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["GB/s"] / HBM_PEAK_GBS, "traffic": traffic,
                "whole_step_GB/s": (scan_b + dyn_b) * value / world / 1e9,
                "whole_step_frac": (scan_b + dyn_b) * value / world / 1e9 / HBM_PEAK_GBS, "kernels": kernels}

    out = {
        "metric": "env-steps/sec AAURoverEnv-v0 @ num_envs=4096; 1/2/4/8 MI355X", "value": value, "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "AAURoverEnv-v0 num_envs=%d per GPU, procedural heightfield 2048x2048 @0.05 m "
                               "(fBm sigma_z 0.15 m seed 1234 + 400 rocks), random U(-1,1) actions, in-step resets" % n,
                   "num_envs_per_gpu": n, "global_num_envs": shard.global_num_envs, "rays": env.num_rays,
                   "decimation": cfg.decimation, "sim_dt": cfg.sim.dt, "solver_iterations": cfg.solver_iterations,
                   "contact_forces_materialised": cfg.record_contact_forces, "parallelism": f"env-shard x{world}"},
        "roofline": roofline,
    }

    # ---- RCCL rollout gather (BASELINE config 3): one 60-step rollout shard of observations, not in `value`
    if world > 1 and nccl:
        try:
            T_roll = 60
            roll = torch.empty(T_roll, n, env.obs_dim, device=dev)
            roll.normal_()
            gat = rd.RolloutGatherer()
            outbuf = torch.empty((world,) + tuple(roll.shape), device=dev)
            gat.gather(roll, outbuf)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                gat.gather(roll, outbuf)
            torch.cuda.synchronize()
            barrier()
            dt = (time.perf_counter() - t0) / reps
            out["rollout_gather"] = {"rollout_steps": T_roll, "shard_bytes": roll.numel() * 4, "ms": dt * 1e3,
                                     "per_rank_recv_GB/s": roll.numel() * 4 * (world - 1) / dt / 1e9,
                                     "ms_per_env_step_equiv": dt * 1e3 / T_roll}
            del roll, outbuf
        except Exception as e:  # the extra leg must never take the headline number down
            out["rollout_gather"] = {"failed": repr(e)}

    # ---- CPU baseline: the oracle (a port, test infrastructure) on a bounded sample of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import rover_oracle as ro
            ro.build()
            ocfg = ro.Config()
            for name, _ in ro.Config._fields_:
                v = getattr(env._native_cfg, name)
                if name == "rew_weight":
                    for i in range(7):
                        ocfg.rew_weight[i] = v[i]
                else:
                    setattr(ocfg, name, v)
            oter = ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, ter.resolution, ter.min_x, ter.min_y,
                                  ter.spawn_locations)
            S = ro.new_state(n)
            ro.reset_all(ocfg, oter, S)
            acts = actions[:64].cpu().numpy()
            ro.step(ocfg, oter, S, acts[0])            # warm-up (thread pool, page faults)
            ro.step(ocfg, oter, S, acts[1])
            t0 = time.perf_counter()
            m = 0
            while m < 4000 and (m < 4 or time.perf_counter() - t0 < args.cpu_seconds):   # bounded sample: ~cpu_seconds of CPU work
                ro.step(ocfg, oter, S, acts[m % 64])
                m += 1
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n * m / dt, "unit": "env-steps/s", "cores": int(ro.lib().rvo_num_threads()),
                                   "kind": "port",
                                   "sample": f"{m} steps of the same N={n} workload on the C oracle (OpenMP), {dt:.1f} s",
                                   "host_cpus": os.cpu_count()}
        except Exception as e:  # the baseline must never take the bench down
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    env.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
