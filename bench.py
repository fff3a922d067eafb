#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of AAURoverEnv-v0 at num_envs=4096 per GPU (BASELINE.json metric, config 2).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one ``RoverEnv.step()`` over the rank's batch of envs: the hot path's HIP kernel -- ONE launch per step at the
benchmark's 4096 envs per GPU (step + height scan), two launches below 2048 envs -- launched
through the C ABI exactly as a trainer would (random actions pre-generated in HBM, in-kernel resets included).  Weak scaling:
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


def bench_lift(args):
    """BASELINE config 5: FrankaCubeLift-v0, num_envs = 2048, 1 x MI355X; env-steps/s through FrankaCubeLiftEnv.step()."""
    import numpy as np
    import torch
    from isaac_rover_orbit_amd.envs import FrankaCubeLiftEnv, LiftEnvCfg
    n = args.num_envs if args.num_envs != 4096 else 2048
    torch.cuda.set_device(0)
    cfg = LiftEnvCfg()
    cfg.scene.num_envs = n
    env = FrankaCubeLiftEnv(cfg)
    total = args.steps + args.warmup
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = torch.rand(min(total, 512), n, 8, device=env.device, generator=g) * 2 - 1
    env.reset()
    for k in range(args.warmup):
        env.step(acts[k % acts.shape[0]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        env.step(acts[k % acts.shape[0]])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "env-steps/sec FrankaCubeLift-v0 (BASELINE config 5)", "value": n * args.steps / dt, "unit": "env-steps/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"FrankaCubeLift-v0 num_envs={n}, random U(-1,1) actions, 100 Hz x decimation 2, in-step resets "
                                  "(BASELINE config 5)", "baseline_config": 5, "num_envs_per_gpu": n}}
    # ---- roofline leg.  Algorithmic bytes per env-step of this path: state read + write 2 x 4 x 64 B,
    #      observation row 36 x 4 B, action 8 x 4 B, reward + flags 6 B = 694 B.  The kernel is bound by the length of its waves'
    #      instruction streams (512 waves at 2048 envs), not by HBM: the fraction is structurally tiny and says so.
    ps = max(args.profile_steps, 1)
    ms_raw = ms_ev = 0.0
    for k in range(ps):
        a, b = env.profile_step(acts[k % acts.shape[0]])
        ms_raw += a
        ms_ev += b
    ms_raw, ms_ev = ms_raw / ps, ms_ev / ps
    ms_k = max(ms_raw - ms_ev, 1e-6)
    alg = (2 * 4 * 64 + 36 * 4 + 8 * 4 + 6) * n
    name = env.kernel_name()
    out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": alg / (ms_k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "event_pair_overhead_ms": ms_ev,
                       "kernels": {name: {"ms": ms_k, "ms_raw_events": ms_raw, "algorithmic_bytes": alg,
                                          "GB/s": alg / (ms_k * 1e-3) / 1e9}},
                       "note": "latency-bound by construction: 8 lanes per env, two pipelined waves per 8 envs = 512 waves on 1024 SIMDs "
                               "at 2048 envs; the time is the critical path arm substep 0 -> cube substeps 0, 1 -> managers "
                               "(~35 k cycles), see DESIGN.md section 9"}
    out["config"]["parity"] = ("reward / observation term functions pinned by the reference fixture (lift_terms.npz); arm / cube / gripper "
                               "simulator is a documented model (PhysX in the reference), parity unpinned; HIP == the separately "
                               "written scalar oracle bit for bit")
    if not args.no_cpu_baseline:
        try:
            from oracle import lift_oracle as lo
            lo.build()
            oc = lo.default_config()
            S = lo.new_state(n)
            lo.reset(oc, S)
            a = np.random.RandomState(0).uniform(-1, 1, (16, n, 8)).astype(np.float32)
            lo.step(oc, S, a[0])
            t0 = time.perf_counter()
            m = 0
            while m < 2000 and (m < 4 or time.perf_counter() - t0 < args.cpu_seconds):
                lo.step(oc, S, a[m % 16])
                m += 1
            d = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n * m / d, "unit": "env-steps/s", "cores": int(lo.max_threads()), "kind": "port",
                                   "host_cpus": os.cpu_count(),
                                   "sample": f"{m} steps of the same N={n} workload on the C oracle of the task (OpenMP, "
                                             f"{int(lo.max_threads())} threads), {d:.1f} s"}
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
    env.close()
    print(json.dumps(out))


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
    ap.add_argument("--with-policy", action="store_true",
                    help="extra leg (outside `value`): the rollout loop a trainer runs -- actor + critic forward passes "
                         "(fused MFMA kernel, reference architecture, random-init weights) + env.step() on one stream, per-kernel us")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="BASELINE.json config: 2 = headline (31x31 rays @0.1 m, sigma_z 0.15 m); "
                         "4 = dense scanner stress (32x32 rays @0.05 m, sigma_z 0.4 m); "
                         "5 = manipulation task FrankaCubeLift-v0 (default num_envs 2048)")
    args = ap.parse_args()
    if args.config == 5:
        return bench_lift(args)

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
    #      or config 4 (same map generator at sigma_z 0.4 m, 32 x 32 rays at 0.05 m spacing)
    sigma_z = 0.15 if args.config == 2 else 0.4
    ter = T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=sigma_z, n_rocks=400)
    ter.make_spawns(2 * shard.global_num_envs, seed=41)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    cfg.sim.device = str(dev)
    cfg.terrain.kind = "custom"
    cfg.env_id_offset = shard.env_id_offset
    cfg.global_num_envs = shard.global_num_envs
    cfg.record_contact_forces = not args.no_forces
    if args.config == 4:
        cfg.height_scanner.resolution, cfg.height_scanner.size = 0.05, (1.55, 1.55)
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

    # ---- roofline leg: HIP-event duration of each kernel (on the stream the kernels are launched on).  An event pair
    #      with nothing between them measures the fixed cost an interval carries (event processing + the dispatch gap);
    #      it is subtracted so that the kernel durations agree with the rocprofv3 trace and sum to <= ms_per_step.
    ms1 = ms2 = 0.0
    ps = max(args.profile_steps, 1)
    for k in range(ps):
        a, b = env.profile_step(actions[k % n_act])
        ms1 += a
        ms2 += b
    ms1_raw, ms2_raw = ms1 / ps, ms2 / ps
    ev_ms = env.profile_event_overhead(200)
    ms1, ms2 = max(ms1_raw - ev_ms, 1e-6), max(ms2_raw - ev_ms, 1e-6)
    scan_b, dyn_b = algorithmic_bytes(env.num_rays)
    k1_name, k2_name = env.kernel_names()       # exactly what rocprofv3's kernel trace prints (rover_kernel_names)
    if k1_name.startswith("rover_step_scan_kernel"):
        # one launch per step: the height scan is the last phase of the step kernel's waves, so that kernel moves ALL the
        # algorithmic bytes of an env step; the second interval is the log reduction (none when extras["log"] is on demand)
        kernels = {k1_name: {"ms": ms1, "ms_raw_events": ms1_raw, "algorithmic_bytes": (dyn_b + scan_b) * n,
                             "GB/s": (dyn_b + scan_b) * n / (ms1 * 1e-3) / 1e9}}
        if k2_name:
            kernels[k2_name] = {"ms": ms2, "ms_raw_events": ms2_raw, "algorithmic_bytes": 0, "GB/s": 0.0}
    else:
        kernels = {
            k1_name: {"ms": ms1, "ms_raw_events": ms1_raw, "algorithmic_bytes": dyn_b * n,
                      "GB/s": dyn_b * n / (ms1 * 1e-3) / 1e9},
            k2_name: {"ms": ms2, "ms_raw_events": ms2_raw, "algorithmic_bytes": scan_b * n,
                      "GB/s": scan_b * n / (ms2 * 1e-3) / 1e9},
        }
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    traffic = None
    traffic_build = None
    tr_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if args.config == 2 and n == 4096 and os.path.exists(tr_path):
        try:
            tr = json.load(open(tr_path))
            traffic = tr.get(dom, {}).get("bytes_per_launch")
            traffic_build = tr.get("_build")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["GB/s"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_build": traffic_build,
                "event_pair_overhead_ms": ev_ms,
                "whole_step_GB/s": (scan_b + dyn_b) * value / world / 1e9,
                "whole_step_frac": (scan_b + dyn_b) * value / world / 1e9 / HBM_PEAK_GBS, "kernels": kernels}

    out = {
        "metric": "env-steps/sec AAURoverEnv-v0 @ num_envs=4096; 1/2/4/8 MI355X", "value": value, "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "AAURoverEnv-v0 num_envs=%d per GPU, procedural heightfield 2048x2048 @0.05 m "
                               "(fBm sigma_z %.2f m seed 1234 + 400 rocks), %dx%d rays @%.2f m, random U(-1,1) actions, "
                               "in-step resets (BASELINE config %d)"
                               % (n, sigma_z, env.cfg.height_scanner.grid[0], env.cfg.height_scanner.grid[1],
                                  env.cfg.height_scanner.resolution, args.config),
                   "baseline_config": args.config,
                   "num_envs_per_gpu": n, "global_num_envs": shard.global_num_envs, "rays": env.num_rays,
                   "decimation": cfg.decimation, "sim_dt": cfg.sim.dt, "solver_iterations": cfg.solver_iterations,
                   "contact_forces_materialised": cfg.record_contact_forces, "parallelism": f"env-shard x{world}",
                   "scan_surface": env.cfg.height_scanner.surface, "spawn_draw": env.cfg.spawn_draw,
                   "parity": "MDP terms / reset / Ackermann / terrain look-ups pinned by reference fixtures; rover dynamics + contact "
                             "(PhysX in the reference) and the mesh ray-caster (Warp) are documented models, parity unpinned "
                             "(DESIGN.md section 4)"},
        "roofline": roofline,
    }

    # ---- RCCL rollout gather (BASELINE config 3): one 60-step rollout shard of observations, not in `value`
    if world > 1 and not nccl:
        out["rollout_gather"] = None
        out["rollout_gather_skipped"] = f"backend is {dist.get_backend()}, not nccl (RCCL): CPU rehearsal"
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

    # ---- CPU baseline: the oracle (a port, test infrastructure) on a bounded sample of the same workload.  Its action
    #      rows are its own (host RNG), independent of --steps / --warmup.
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
            acts = np.random.RandomState(1234).uniform(-1.0, 1.0, (64, n, 2)).astype(np.float32)

            def cpu_leg(threads, n_sub, seconds, max_steps):
                ro.set_num_threads(threads)
                S = ro.new_state(n_sub)
                ro.reset_all(ocfg, oter, S)
                ro.step(ocfg, oter, S, acts[0][:n_sub])            # warm-up (thread pool, page faults)
                ro.step(ocfg, oter, S, acts[1][:n_sub])
                t0 = time.perf_counter()
                m = 0
                while m < max_steps and (m < 4 or time.perf_counter() - t0 < seconds):
                    ro.step(ocfg, oter, S, acts[m % len(acts)][:n_sub])
                    m += 1
                return m, time.perf_counter() - t0

            host = os.cpu_count() or 1
            n_thr = int(ro.max_threads())
            n1 = min(n, 512)                                   # single thread: the first 512 envs of the workload
            m1, dt1 = cpu_leg(1, n1, args.cpu_seconds / 3.0, 400)
            mN, dtN = cpu_leg(n_thr, n, args.cpu_seconds * 2.0 / 3.0, 4000)
            out["cpu_baseline"] = {"value": n * mN / dtN, "unit": "env-steps/s", "cores": n_thr, "kind": "port",
                                   "sample": f"{mN} steps of the same N={n} workload on the C oracle (OpenMP, {n_thr} threads), "
                                             f"{dtN:.1f} s; single thread: {m1} steps of its first {n1} envs, {dt1:.1f} s",
                                   "single_thread_value": n1 * m1 / dt1, "host_cpus": host}
        except Exception as e:  # the baseline must never take the bench down
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    # ---- rollout-loop leg: what a trainer runs per env step -- policy mean, value, env.step -- on ONE stream.
    #      Random-init weights of the reference architecture (get_models.py:36-62); outside `value`.
    if args.with_policy and env.num_rays == 961:
        try:
            from isaac_rover_orbit_amd.policy import RoverNet
            rs = np.random.RandomState(7)
            K, Nn = [961, 80, 64, 256, 160, 128], [80, 60, 256, 160, 128]
            def net(out_dim, act):
                ws = [(rs.uniform(-1, 1, (nn, kk)) / np.sqrt(kk)).astype(np.float32) for kk, nn in zip(K, Nn + [out_dim])]
                bs = [(rs.uniform(-1, 1, nn) / np.sqrt(kk)).astype(np.float32) for kk, nn in zip(K, Nn + [out_dim])]
                return RoverNet(ws, bs, n_enc=2, final_act=act, device=dev)
            actor, critic = net(2, "tanh"), net(1, "none")
            if os.environ.get("ROVER_SCAN_FORM"):   # measurement hook: 5 / 6 = XCD-aware pair dealing of the scan kernel off / on
                import ctypes as C
                fn = C.CDLL(env._lib._name).rover_debug_set_scan_form
                fn.argtypes = [C.c_void_p, C.c_int]
                assert fn(env._h, int(os.environ["ROVER_SCAN_FORM"])) == 0
            obs = env.obs_buf["policy"]
            # per-kernel event times in the RUNNING loop (one event set per iteration, one synchronisation at the end: a
            # synchronisation per iteration makes the actor the first kernel on an idle GPU and adds ~8 us to it)
            t_act = t_val = t_env = 0.0
            reps = max(args.profile_steps, 20)
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps + 5)]
            for ev in evs:
                ev[0].record(); a_pol = actor(obs); ev[1].record(); critic(obs); ev[2].record()
                obs = env.step(a_pol)[0]["policy"]; ev[3].record()
            torch.cuda.synchronize()
            for ev in evs[5:]:
                t_act += ev[0].elapsed_time(ev[1]); t_val += ev[1].elapsed_time(ev[2]); t_env += ev[2].elapsed_time(ev[3])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                a_pol = actor(obs); critic(obs)
                obs = env.step(a_pol)[0]["policy"]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            # the same loop with both networks in ONE launch on one staged tile (rover_policy_forward_pair)
            from isaac_rover_orbit_amd.policy import forward_pair
            t_pair = 0.0
            for ev in evs:
                ev[0].record(); a_pol, _v = forward_pair(actor, critic, obs); ev[1].record()
                obs = env.step(a_pol)[0]["policy"]
            torch.cuda.synchronize()
            for ev in evs[5:]:
                t_pair += ev[0].elapsed_time(ev[1])
            t0 = time.perf_counter()
            for k in range(args.steps):
                a_pol, _v = forward_pair(actor, critic, obs)
                obs = env.step(a_pol)[0]["policy"]
            torch.cuda.synchronize()
            dt_pair = (time.perf_counter() - t0) / args.steps
            flops = 2.0 * n * sum(kk * nn for kk, nn in zip(K, Nn + [2]))
            out["with_policy"] = {"ms_per_step": dt * 1e3, "env_steps_per_s": n / dt,
                                  "kernels_us_events": {"rover_policy_kernel (actor)": t_act / reps * 1e3 - ev_ms * 1e3,
                                                        "rover_policy_kernel (critic)": t_val / reps * 1e3 - ev_ms * 1e3,
                                                        "env.step (K1 + K2)": t_env / reps * 1e3 - ev_ms * 1e3},
                                  "pair": {"ms_per_step": dt_pair * 1e3, "env_steps_per_s": n / dt_pair,
                                           "rover_policy_ref_pair_kernel_us_events": t_pair / reps * 1e3 - ev_ms * 1e3},
                                  "observations_finite": bool(torch.isfinite(obs).all()),
                                  "actor_TFLOPs_f32": flops / ((t_act / reps - ev_ms) * 1e-3) / 1e12, "f32_mfma_peak_TFLOPs": 157.0,
                                  "note": "closed loop: the actor's mean action drives the env (observations with -inf rays are "
                                          "what the kernel reads; the reference feeds them to torch the same way)"}
        except Exception as e:  # the extra leg must never take the headline number down
            out["with_policy"] = {"failed": repr(e)}

    env.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
