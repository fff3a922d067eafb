#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of AAURoverEnv-v0 at num_envs=4096 per GPU (BASELINE.json metric, config 2).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one ``RoverEnv.step()`` over the rank's batch of envs: the hot path's HIP kernel -- ONE launch per step at the
benchmark's 4096 envs per GPU (step + height scan), two launches below 2048 envs -- launched
through the C ABI (random actions pre-generated in HBM, in-kernel resets included).  The timed loop is the metric's
random-action rollout of ``step()``: it does not read ``extras["log"]``, so the on-demand log reduction never runs inside it.
The reference's TRAINER reads the log after every step (``skrl_utils.py:139-142``: ``.item()`` on every entry of
``infos["episode"]``): that loop is the ``extra.rollout_loop.trainer_loop`` leg, reported beside the headline, never instead
of it.  Weak scaling: every rank simulates ``--num-envs`` envs (global ids sharded by rank, terrain replicated, no data-path
collective).  Rank 0 prints ONE JSON line.  Extra legs (not in the timed region):
  * roofline      -- duration of the dominant kernel vs its algorithmic bytes (SURVEY 8d / DESIGN.md); ``roofline.issue`` = what
                     actually bounds the kernel (VALU issue of one to two waves per SIMD), from the committed PMC summary
  * cpu_baseline  -- the CPU oracle ("port") timed on a bounded sample of the same workload (N=1 only)
  * extra         -- (N=1, config 2, unless --no-extra) short runs of BASELINE configs 4 and 5, of the rollout loop with both
                     networks and the trainer's per-step log read, of the env alone with the log reduced behind EVERY step, and of the
                     step at rounds 1-4's settings (16 solver iterations; 16 iterations + lumped mass)
  * ranks / rollout_gather (N>1) -- which devices the ranks ran on; one RCCL all_gather of a 60-step rollout shard, plain and
                     overlapped with the next rollout on a side stream (BASELINE config 3)
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
PARITY_NOTE = ("MDP terms / reset / Ackermann / terrain look-ups pinned by reference fixtures; rover dynamics + contact "
               "(PhysX in the reference) and the mesh ray-caster (Warp) are documented models, parity unpinned (DESIGN.md section 4)")


# algorithmic bytes per env-step: obs write 4*(4+R), heightfield reads 4*R, state r+w 2*4*52, action 8,
# reward+flags 6, wheel samples 6*6*4
def algorithmic_bytes(rays: int):
    scan = 4 * (4 + rays) + 4 * rays
    dyn = 2 * 4 * 52 + 8 + 6 + 6 * 6 * 4
    return scan, dyn


def _load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def library_digest():
    """Identity of the build this process runs: sha256[:16] over the HIP sources + headers (isaac_rover_orbit_amd.build.source_digest:
    stable across rebuilds) and the file name of the loaded library (a tools/build_diag.py variant carries another name) -- what
    tools/pmc_issue.py / tools/pmc_traffic.py record beside the counters they collect (`_lib_sha256`): a committed counter summary is
    printed as measured only for the build it was measured on."""
    from isaac_rover_orbit_amd import _lib, build
    try:
        return build.source_digest() + ":" + os.path.basename(_lib.LIB_PATH)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ config 5 (lift)
def lift_line(num_envs, steps, warmup, profile_steps, cpu_seconds):
    """BASELINE config 5: FrankaCubeLift-v0, num_envs = 2048, 1 x MI355X; env-steps/s through FrankaCubeLiftEnv.step()."""
    import numpy as np
    import torch
    from isaac_rover_orbit_amd.envs import FrankaCubeLiftEnv, LiftEnvCfg
    n = num_envs
    torch.cuda.set_device(0)
    cfg = LiftEnvCfg()
    cfg.scene.num_envs = n
    env = FrankaCubeLiftEnv(cfg)
    total = steps + warmup
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = torch.rand(min(total, 512), n, 8, device=env.device, generator=g) * 2 - 1
    env.reset()
    for k in range(warmup):
        env.step(acts[k % acts.shape[0]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warmup, total):
        env.step(acts[k % acts.shape[0]])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "env-steps/sec FrankaCubeLift-v0 (BASELINE config 5)", "value": n * steps / dt, "unit": "env-steps/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"FrankaCubeLift-v0 num_envs={n}, random U(-1,1) actions, 100 Hz x decimation 2, in-step resets "
                                  "(BASELINE config 5)", "baseline_config": 5, "num_envs_per_gpu": n}}
    # ---- roofline leg.  Algorithmic bytes per env-step of this path (docs/history.md section 9, f-4): state read + write 2 x 4 x 64 B,
    #      observation row 36 x 4 B, action 8 x 4 B, reward + flags 6 B = 694 B.  The kernel is bound by the length of its waves'
    #      instruction streams (512 waves at 2048 envs), not by HBM: the fraction is structurally tiny and says so.
    #      Duration = ms_per_step: one kernel per step, so the step time bounds the kernel from above (event intervals minus the
    #      empty pair read BELOW it and rocprofv3's trace, with its per-dispatch instrumentation, above it: both are kept beside).
    ps = max(profile_steps, 1)
    ms_raw = ms_ev = 0.0
    for k in range(ps):
        a, b = env.profile_step(acts[k % acts.shape[0]])
        ms_raw += a
        ms_ev += b
    ms_raw, ms_ev = ms_raw / ps, ms_ev / ps
    ms_k = out["ms_per_step"]
    alg = (2 * 4 * 64 + 36 * 4 + 8 * 4 + 6) * n
    name = env.kernel_name()
    out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": alg / (ms_k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "duration_basis": "ms_per_step",
                       "event_pair_overhead_ms": ms_ev,
                       "kernels": {name: {"ms": ms_k, "ms_raw_events": ms_raw, "ms_events_minus_empty_pair": max(ms_raw - ms_ev, 1e-6),
                                          "algorithmic_bytes": alg, "GB/s": alg / (ms_k * 1e-3) / 1e9}},
                       "note": "latency-bound by construction: 8 lanes per env, two pipelined waves per 8 envs = 512 waves on 1024 SIMDs "
                               "at 2048 envs; the time is the critical path arm substep 0 -> cube substeps 0, 1 -> managers "
                               "(~35 k cycles), see docs/history.md section 9"}
    out["config"]["parity"] = ("reward / observation term functions pinned by the reference fixture (lift_terms.npz); arm / cube / gripper "
                               "simulator is a documented model (PhysX in the reference), parity unpinned; HIP == the separately "
                               "written scalar oracle bit for bit")
    if cpu_seconds > 0:
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
            while m < 2000 and (m < 4 or time.perf_counter() - t0 < cpu_seconds):
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
    return out


# ------------------------------------------------------------------------------------------------ rover legs
def shard_cfg(n, shard, config, dev="cuda:0", solver_iterations=None, no_forces=False, stream_obs=False, mass_model=None,
              log_reduction=None):
    """The env cfg of one rank and the size of the spawn table it needs: 2 x the GLOBAL env count (terrain_utils.py:123-124), so that
    the per-call spawn permutation is a bijection of all global env ids (no GPU needed: tests/test_distributed.py runs it for
    eight ranks)."""
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    cfg.sim.device = str(dev)
    cfg.terrain.kind = "custom"
    n_global = shard.global_num_envs if shard else n
    if shard:
        cfg.env_id_offset = shard.env_id_offset
        cfg.global_num_envs = shard.global_num_envs
    cfg.record_contact_forces = not no_forces
    cfg.stream_observations = bool(stream_obs)
    if solver_iterations is not None:
        cfg.solver_iterations = int(solver_iterations)
    if mass_model is not None:
        cfg.mass_model = mass_model
    if log_reduction is not None:
        cfg.log_reduction = log_reduction
    if config == 4:
        cfg.height_scanner.resolution, cfg.height_scanner.size = 0.05, (1.55, 1.55)
    return cfg, 2 * n_global


def make_rover(dev, n, config, shard=None, terrain_cache=None, **cfg_kw):
    """SURVEY 8d config 2 (procedural 2048^2 heightfield @ 0.05 m, fBm sigma_z 0.15 m seed 1234, ~400 rocks) or config 4
    (same map generator at sigma_z 0.4 m, 32 x 32 rays at 0.05 m spacing)."""
    from isaac_rover_orbit_amd import terrain as T
    from isaac_rover_orbit_amd.envs import RoverEnv
    sigma_z = 0.15 if config == 2 else 0.4
    cfg, n_spawns = shard_cfg(n, shard, config, dev=dev, **cfg_kw)
    key = (sigma_z, n_spawns)
    if terrain_cache is not None and key in terrain_cache:
        ter = terrain_cache[key]
    else:
        ter = T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=sigma_z, n_rocks=400)
        ter.make_spawns(n_spawns, seed=41)
        if terrain_cache is not None:
            terrain_cache[key] = ter
    return RoverEnv(cfg, terrain=ter), ter, cfg, sigma_z


def short_rover_run(dev, config, steps=300, warmup=360, terrain_cache=None, **cfg_kw):
    """One short run of a rover configuration (an ``extra`` leg): value, ms_per_step, kernel name(s)."""
    import torch
    env, _, cfg, _ = make_rover(dev, 4096, config, terrain_cache=terrain_cache, **cfg_kw)
    g = torch.Generator(device=dev).manual_seed(0)
    acts = torch.rand(128, 4096, 2, device=dev, generator=g) * 2 - 1
    env.reset()
    for k in range(warmup):
        env.step(acts[k % 128])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % 128])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    scan_b, dyn_b = algorithmic_bytes(env.num_rays)
    out = {"value": 4096 / dt, "unit": "env-steps/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": warmup, "num_envs": 4096,
           "rays": env.num_rays, "solver_iterations": cfg.solver_iterations, "mass_model": cfg.mass_model,
           "log_reduction": cfg.log_reduction, "kernel": env.kernel_names()[0], "second_kernel": env.kernel_names()[1],
           "whole_step_frac": (scan_b + dyn_b) * 4096 / dt / 1e9 / HBM_PEAK_GBS}
    env.close()
    return out


def rollout_loop_leg(env, dev, steps, profile_steps, ev_ms):
    """what a trainer runs per env step -- policy mean, value, env.step -- on ONE stream (random-init weights of the
    reference architecture, get_models.py:36-62), as two launches, as the pair kernel, and as the reference's trainer loop:
    the pair + env.step + ``.item()`` on every entry of ``infos["episode"]`` after every step (skrl_utils.py:139-142), which runs
    the on-demand log reduction behind every step and synchronises the host thirteen times."""
    import numpy as np
    import torch
    from isaac_rover_orbit_amd.policy import RoverNet, forward_pair
    n = env.num_envs
    rs = np.random.RandomState(7)
    K, Nn = [961, 80, 64, 256, 160, 128], [80, 60, 256, 160, 128]

    def net(out_dim, act):
        ws = [(rs.uniform(-1, 1, (nn, kk)) / np.sqrt(kk)).astype(np.float32) for kk, nn in zip(K, Nn + [out_dim])]
        bs = [(rs.uniform(-1, 1, nn) / np.sqrt(kk)).astype(np.float32) for kk, nn in zip(K, Nn + [out_dim])]
        return RoverNet(ws, bs, n_enc=2, final_act=act, device=dev)
    actor, critic = net(2, "tanh"), net(1, "none")
    obs = env.obs_buf["policy"]
    # per-kernel event times in the RUNNING loop (one event set per iteration, one synchronisation at the end: a
    # synchronisation per iteration makes the actor the first kernel on an idle GPU and adds ~8 us to it)
    t_act = t_val = t_env = 0.0
    reps = max(profile_steps, 20)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps + 5)]
    for ev in evs:
        ev[0].record(); a_pol = actor(obs); ev[1].record(); critic(obs); ev[2].record()
        obs = env.step(a_pol)[0]["policy"]; ev[3].record()
    torch.cuda.synchronize()
    for ev in evs[5:]:
        t_act += ev[0].elapsed_time(ev[1]); t_val += ev[1].elapsed_time(ev[2]); t_env += ev[2].elapsed_time(ev[3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        a_pol = actor(obs); critic(obs)
        obs = env.step(a_pol)[0]["policy"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the same loop with both networks in ONE launch on one staged tile (rover_policy_forward_pair)
    t_pair = 0.0
    for ev in evs:
        ev[0].record(); a_pol, _v = forward_pair(actor, critic, obs); ev[1].record()
        obs = env.step(a_pol)[0]["policy"]
    torch.cuda.synchronize()
    for ev in evs[5:]:
        t_pair += ev[0].elapsed_time(ev[1])
    t0 = time.perf_counter()
    for k in range(steps):
        a_pol, _v = forward_pair(actor, critic, obs)
        obs = env.step(a_pol)[0]["policy"]
    torch.cuda.synchronize()
    dt_pair = (time.perf_counter() - t0) / steps
    # the pair loop with the log reduced behind every step on the device, no host read (what a trainer that logs on the device pays)
    t0 = time.perf_counter()
    for k in range(steps):
        a_pol, _v = forward_pair(actor, critic, obs)
        obs = env.step(a_pol)[0]["policy"]
        env.flush_log()
    torch.cuda.synchronize()
    dt_flush = (time.perf_counter() - t0) / steps
    # the reference's trainer loop: read every scalar of infos["episode"] on the host after every step
    tracked = 0.0
    t_steps = max(min(steps, 300), 20)
    t0 = time.perf_counter()
    for k in range(t_steps):
        a_pol, _v = forward_pair(actor, critic, obs)
        o, _r, _te, _tr, infos = env.step(a_pol)
        obs = o["policy"]
        if "episode" in infos:
            for kk, vv in infos["episode"].items():
                if isinstance(vv, torch.Tensor) and vv.numel() == 1:
                    tracked += vv.item()
    torch.cuda.synchronize()
    dt_trainer = (time.perf_counter() - t0) / t_steps
    # the same loop, unchanged, with cfg.log_values = "host": the dictionary's entries are views of a pinned host mirror that the
    # first read after a step refreshes with ONE copy and ONE synchronisation
    env.set_log_values("host")
    tracked_h = 0.0
    t0 = time.perf_counter()
    for k in range(t_steps):
        a_pol, _v = forward_pair(actor, critic, obs)
        o, _r, _te, _tr, infos = env.step(a_pol)
        obs = o["policy"]
        if "episode" in infos:
            for kk, vv in infos["episode"].items():
                if isinstance(vv, torch.Tensor) and vv.numel() == 1:
                    tracked_h += vv.item()
    torch.cuda.synchronize()
    dt_trainer_h = (time.perf_counter() - t0) / t_steps
    env.set_log_values("device")
    flops = 2.0 * n * sum(kk * nn for kk, nn in zip(K, Nn + [2]))
    return {"ms_per_step": dt * 1e3, "env_steps_per_s": n / dt,
            "kernels_us_events": {"rover_policy_kernel (actor)": t_act / reps * 1e3 - ev_ms * 1e3,
                                  "rover_policy_kernel (critic)": t_val / reps * 1e3 - ev_ms * 1e3,
                                  "env.step": t_env / reps * 1e3 - ev_ms * 1e3},
            "pair": {"ms_per_step": dt_pair * 1e3, "env_steps_per_s": n / dt_pair,
                     "rover_policy_ref_pair_kernel_us_events": t_pair / reps * 1e3 - ev_ms * 1e3},
            "pair_with_device_log": {"ms_per_step": dt_flush * 1e3, "env_steps_per_s": n / dt_flush,
                                     "note": "pair + env.step + rover_flush_log behind every step (the log vector stays on the device)"},
            "trainer_loop": {"ms_per_step": dt_trainer_h * 1e3, "env_steps_per_s": n / dt_trainer_h, "steps": t_steps,
                             "log_values": "host",
                             "note": "pair + env.step + .item() on every entry of infos['episode'] after every step, as "
                                     "skrl_utils.py:139-142 does, with cfg.log_values = 'host' -- what compat.convert gives the "
                                     "reference's own trainer (the gym-redirect path): the entries are views of a pinned host mirror, "
                                     "ONE copy + ONE synchronisation per step, .item() is then free", "checksum": tracked_h},
            "trainer_loop_device_log": {"ms_per_step": dt_trainer * 1e3, "env_steps_per_s": n / dt_trainer, "steps": t_steps,
                                        "log_values": "device",
                                        "note": "the same loop on ORBIT-style 0-d DEVICE tensors (the native RoverEnvCfg default): the log "
                                                "reduction launches behind every step and the host synchronises once per entry (13 per step)",
                                        "checksum": tracked},
            "observations_finite": bool(torch.isfinite(obs).all()),
            "actor_TFLOPs_f32": flops / ((t_act / reps - ev_ms) * 1e-3) / 1e12, "f32_mfma_peak_TFLOPs": 157.0,
            "note": "closed loop: the actor's mean action drives the env (observations with -inf rays are what the kernel reads; "
                    "the reference feeds them to torch the same way); none of these loops is `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll", type=int, default=1500,
                    help="untimed env steps BEFORE the warm-up steps, part of the initialisation like the construction and the reset: the rollout "
                         "reaches its steady state (episodes of mixed age: the first time-outs come at step 750; resets in every step) and the "
                         "device its sustained state -- measured (profiles/r05_preroll.txt): the first 200-step windows after construction run "
                         "42.5 / 40.5 / 39.1 / 37.8 / 37.7 us per step, and a 20-step region behind 400 / 1500 / 4000 such steps reads "
                         "40.5 / 39.3 / 39.3 us per step")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` block (configs 4 / 5, rollout loop, log every step, 16 iterations)")
    ap.add_argument("--no-forces", action="store_true", help="do not materialise contact_sensor.force_matrix_w")
    ap.add_argument("--with-policy", action="store_true",
                    help="top-level `with_policy` leg (outside `value`): the rollout loop a trainer runs -- actor + critic forward "
                         "passes (fused MFMA kernel, reference architecture, random-init weights) + env.step() on one stream; the "
                         "default run carries the same leg as extra.rollout_loop")
    ap.add_argument("--solver-iterations", type=int, default=None,
                    help="override cfg.solver_iterations (default: the reference's 32, aau_rover_simple.py:33); a line measured with an "
                         "override says so in config.solver_iterations and is NOT the headline")
    ap.add_argument("--mass-model", default=None, choices=("lumped", "subtree_weights"), help="override cfg.mass_model (default subtree_weights)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="BASELINE.json config: 2 = headline (31x31 rays @0.1 m, sigma_z 0.15 m); "
                         "4 = dense scanner stress (32x32 rays @0.05 m, sigma_z 0.4 m); "
                         "5 = manipulation task FrankaCubeLift-v0 (default num_envs 2048)")
    args = ap.parse_args()
    if args.config == 5:
        n5 = args.num_envs if args.num_envs != 4096 else 2048
        print(json.dumps(lift_line(n5, args.steps, args.warmup, args.profile_steps, 0.0 if args.no_cpu_baseline else args.cpu_seconds)))
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    from isaac_rover_orbit_amd import distributed as rd

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
    terrain_cache = {}
    env, ter, cfg, sigma_z = make_rover(dev, n, args.config, shard=shard, no_forces=args.no_forces, terrain_cache=terrain_cache,
                                        solver_iterations=args.solver_iterations, mass_model=args.mass_model)

    total = args.steps + args.warmup
    g = torch.Generator(device=dev).manual_seed(rank)          # torch's CUDA generator is Philox; seed 0 on rank 0
    n_act = min(total, 2048)
    actions = torch.rand(n_act, n, 2, device=dev, generator=g) * 2 - 1
    env.reset()
    for k in range(args.preroll):              # initialisation, like the env construction above: not part of W, not timed
        env.step(actions[(k * 7) % n_act])
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
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline leg.  HIP-event duration of each kernel (on the stream the kernels are launched on), raw and minus the cost of
    #      an empty event pair.  One launch per step: the roofline figure uses ms_per_step, which bounds the kernel's duration from
    #      above (the corrected event interval reads below both the step time and rocprofv3's trace -- round-3 review); two
    #      launches: the corrected event intervals, which sum to <= ms_per_step.
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
    one_launch = k1_name.startswith("rover_step_scan")
    if one_launch:
        # the height scan is the last phase of the step kernel's waves, so that kernel moves ALL the algorithmic bytes of an env
        # step; the second interval is the log reduction (none when extras["log"] is on demand)
        ms_k = ms_per_step if world == 1 else max(ms1_raw, ms1)
        kernels = {k1_name: {"ms": ms_k, "ms_raw_events": ms1_raw, "ms_events_minus_empty_pair": ms1,
                             "algorithmic_bytes": (dyn_b + scan_b) * n, "GB/s": (dyn_b + scan_b) * n / (ms_k * 1e-3) / 1e9}}
        if k2_name:
            kernels[k2_name] = {"ms": ms2, "ms_raw_events": ms2_raw, "algorithmic_bytes": 0, "GB/s": 0.0}
        basis = "ms_per_step (one kernel per step: an upper bound of its duration; profiles/*_kernel_stats.csv holds rocprofv3's)"
    else:
        kernels = {
            k1_name: {"ms": ms1, "ms_raw_events": ms1_raw, "algorithmic_bytes": dyn_b * n,
                      "GB/s": dyn_b * n / (ms1 * 1e-3) / 1e9},
            k2_name: {"ms": ms2, "ms_raw_events": ms2_raw, "algorithmic_bytes": scan_b * n,
                      "GB/s": scan_b * n / (ms2 * 1e-3) / 1e9},
        }
        basis = "HIP events minus the empty event pair"
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    # committed PMC summaries (profiles/hbm_traffic.json, profiles/issue_counters.json) belong to ONE build and ONE configuration:
    # they are printed as this run's figures only when the loaded library's digest and the solver / mass settings are the ones they
    # were collected with; otherwise the line says so instead of showing another build's counters beside a fresh ms_per_step
    digest = library_digest()

    def current(summary):
        return (summary is not None and summary.get("_lib_sha256") == digest and
                summary.get("_solver_iterations") == cfg.solver_iterations and summary.get("_mass_model") == cfg.mass_model)
    traffic = traffic_build = None
    tr = _load_json("hbm_traffic.json") if (args.config == 2 and n == 4096) else None
    traffic_stale = tr is not None and not current(tr)
    if tr and not traffic_stale:
        traffic = tr.get(dom, {}).get("bytes_per_launch")
        traffic_build = tr.get("_build")
    elif tr:
        traffic_build = {"stale": True, "counters_of_build": tr.get("_lib_sha256"), "this_build": digest, "note": tr.get("_build")}
    roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["GB/s"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_build": traffic_build,
                "duration_basis": basis, "event_pair_overhead_ms": ev_ms,
                "whole_step_GB/s": (scan_b + dyn_b) * value / world / 1e9,
                "whole_step_frac": (scan_b + dyn_b) * value / world / 1e9 / HBM_PEAK_GBS, "kernels": kernels}
    # what bounds the kernel: at ~12 % of the HBM roofline the honest bound is instruction issue -- one step wave (+ one copy wave)
    # per SIMD, a wave64 VALU instruction occupies the SIMD's 16 lanes for 4 cycles.  From the committed PMC summary of this build.
    ic = _load_json("issue_counters.json") if (args.config == 2 and n == 4096) else None
    if ic and not current(ic):
        roofline["issue"] = {"stale": True, "counters_of_build": ic.get("_lib_sha256"), "this_build": digest, "build": ic.get("_build")}
    elif ic and dom in ic:
        c = ic[dom]
        simds = float(c.get("simds", 1024))
        wave_cycles = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
        roofline["issue"] = {"valu_insts_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
                             "valu_insts_per_simd": c["SQ_INSTS_VALU"] / simds,
                             "wave_cycles": wave_cycles,
                             "cycles_per_valu": wave_cycles / (c["SQ_INSTS_VALU"] / simds),
                             "valu_busy_frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (simds * wave_cycles),
                             "waves_per_simd": c["SQ_WAVES"] / simds,
                             "bound": "instruction issue of the one step wave per SIMD: a wave issues an instruction every ~4.4 (4-byte encoding) / ~5.4 cycles (8-byte) "
                                      "whatever it is, while the SIMD takes a full-rate fp32 op every ~2 cycles and a packed / DPP / convert / select op every "
                                      "~3.2 (profiles/r05_exec_probe.txt); valu_busy_frac = SQ_ACTIVE_INST_VALU (wave-cycles inside VALU instructions) per SIMD "
                                      "and cycle of the wave life", "build": ic.get("_build"),
                             "lib_sha256": ic.get("_lib_sha256"),
                             "source": "profiles/issue_counters.json (rocprofv3 --pmc, tools/r05_measure.sh)"}

    out = {
        "metric": "env-steps/sec AAURoverEnv-v0 @ num_envs=4096; 1/2/4/8 MI355X", "value": value, "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "AAURoverEnv-v0 num_envs=%d per GPU, procedural heightfield 2048x2048 @0.05 m "
                               "(fBm sigma_z %.2f m seed 1234 + 400 rocks), %dx%d rays @%.2f m, random U(-1,1) actions, "
                               "in-step resets (BASELINE config %d)"
                               % (n, sigma_z, env.cfg.height_scanner.grid[0], env.cfg.height_scanner.grid[1],
                                  env.cfg.height_scanner.resolution, args.config),
                   "baseline_config": args.config,
                   "num_envs_per_gpu": n, "global_num_envs": shard.global_num_envs, "rays": env.num_rays,
                   "decimation": cfg.decimation, "sim_dt": cfg.sim.dt, "solver_iterations": cfg.solver_iterations,
                   "mass_model": cfg.mass_model, "lib_sha256": digest,
                   "contact_forces_materialised": cfg.record_contact_forces, "parallelism": f"env-shard x{world}",
                   "scan_surface": env.cfg.height_scanner.surface, "spawn_draw": env.cfg.spawn_draw,
                   "log_reduction": getattr(cfg, "log_reduction", "on_demand"), "preroll_steps": args.preroll,
                   "parity": PARITY_NOTE},
        "roofline": roofline,
    }

    # ---- which devices did the ranks run on?  (BASELINE config 3 needs one GPU per rank; a rehearsal shares one card)
    if world > 1:
        try:
            props = torch.cuda.get_device_properties(dev)
            me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": props.name,
                  "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xFF,
                                                    getattr(props, "pci_device_id", 0) & 0xFF),
                  "uuid": str(getattr(props, "uuid", ""))}
            gathered = [None] * world
            dist.all_gather_object(gathered, me)
            ids = [(d["pci_bus_id"], d["uuid"]) for d in gathered]
            out["ranks"] = {"world": world, "backend": dist.get_backend(), "devices": gathered,
                            "distinct_devices": len(set(ids)), "one_gpu_per_rank": len(set(ids)) == world}
        except Exception as e:
            out["ranks"] = {"world": world, "backend": dist.get_backend(), "failed": repr(e)}

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
            # the same gather on a side stream, overlapped with the next rollout's 60 env steps: how much of it stays exposed?
            side = rd.RolloutGatherer(side_stream=True)
            def rollout():
                for k in range(T_roll):
                    env.step(actions[k % n_act])
            rollout()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            rollout()
            torch.cuda.synchronize()
            barrier()
            t_roll = time.perf_counter() - t0
            t0 = time.perf_counter()
            side.gather(roll, outbuf)
            rollout()
            side.wait()
            torch.cuda.synchronize()
            barrier()
            t_both = time.perf_counter() - t0
            out["rollout_gather"]["overlapped"] = {"rollout_ms": t_roll * 1e3, "rollout_plus_gather_ms": t_both * 1e3,
                                                   "exposed_ms": max(t_both - t_roll, 0.0) * 1e3,
                                                   "hidden_ms": max(dt - max(t_both - t_roll, 0.0), 0.0) * 1e3}
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

    # ---- rollout-loop leg on request (the default run carries it inside `extra`)
    if args.with_policy and env.num_rays == 961:
        try:
            out["with_policy"] = rollout_loop_leg(env, dev, args.steps, args.profile_steps, ev_ms)
        except Exception as e:  # the extra leg must never take the headline number down
            out["with_policy"] = {"failed": repr(e)}

    # ---- extra block: the numbers that used to exist only in builder-run files, each a short run outside `value`
    if rank == 0 and world == 1 and args.config == 2 and n == 4096 and not args.no_extra:
        extra = {"note": "short runs outside `value` (300 steps after 360 untimed steps each unless stated)"}
        try:
            if "with_policy" in out and "failed" not in out["with_policy"]:
                extra["rollout_loop"] = out["with_policy"]
            else:
                extra["rollout_loop"] = rollout_loop_leg(env, dev, 300, 60, ev_ms)
        except Exception as e:
            extra["rollout_loop"] = {"failed": repr(e)}
        env.close()
        env = None
        for key, fn in (("config4", lambda: short_rover_run(dev, 4, terrain_cache=terrain_cache)),
                        ("stream_observations", lambda: short_rover_run(dev, 2, terrain_cache=terrain_cache, stream_obs=True)),
                        ("log_every_step", lambda: short_rover_run(dev, 2, terrain_cache=terrain_cache, log_reduction="every_step")),
                        ("solver_iterations_16", lambda: short_rover_run(dev, 2, solver_iterations=16, terrain_cache=terrain_cache)),
                        ("round4_model", lambda: short_rover_run(dev, 2, solver_iterations=16, mass_model="lumped", terrain_cache=terrain_cache)),
                        ("config5", lambda: {k: v for k, v in lift_line(2048, 300, 360, 40, 0.0).items()
                                             if k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "roofline")})):
            try:
                extra[key] = fn()
            except Exception as e:
                extra[key] = {"failed": repr(e)}
        if isinstance(extra.get("stream_observations"), dict) and "ms_per_step" in extra["stream_observations"]:
            extra["stream_observations"]["note"] = ("config 2 with cfg.stream_observations = True (non-temporal stores of the observation rows): faster "
                                                    "when nothing on the device reads the rows next, as in this random-action rollout; the headline "
                                                    "keeps the default (plain stores: the policy kernel behind a step finds the rows in L2)")
        if isinstance(extra.get("log_every_step"), dict) and "ms_per_step" in extra["log_every_step"]:
            extra["log_every_step"]["note"] = ("the env ALONE (random actions, no policy) with cfg.log_reduction = 'every_step': the one-launch step kernel "
                                               "followed by rover_log_kernel behind EVERY step -- the reference rebuilds extras['log'] inside _reset_idx in "
                                               "every step that resets (rover_env.py:36-39; ~4 envs per step at 4096 envs).  The difference to `value` "
                                               "is one dependent dispatch (~2.3 us boundary, profiles/r03_boundary_ubench.txt) + the reduction kernel")
            extra["log_every_step"]["delta_ms_vs_value"] = extra["log_every_step"]["ms_per_step"] - ms_per_step
        if isinstance(extra.get("solver_iterations_16"), dict) and "ms_per_step" in extra["solver_iterations_16"]:
            extra["solver_iterations_16"]["note"] = ("the headline runs the reference's 32 position iterations (aau_rover_simple.py:33); rounds 1-4 "
                                                     "ran 16: this line is that setting on this round's mass model")
        if isinstance(extra.get("round4_model"), dict) and "ms_per_step" in extra["round4_model"]:
            extra["round4_model"]["note"] = ("16 iterations + cfg.mass_model = 'lumped': the model BENCH_r04's `value` was measured on, on this "
                                             "round's kernels (like-for-like with 110.2 M driver-timed / 120 M long-run of round 4)")
        out["extra"] = extra

    if env is not None:
        env.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
