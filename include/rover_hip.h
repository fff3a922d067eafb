/*
 * rover_hip.h -- C ABI of the MI355X-native AAURoverEnv-v0 hot path (librover_hip.so).
 *
 * The reference (abmoRobotics/isaac_rover_orbit) has NO native / FFI boundary on this path: its "operator API" is the
 * Python object protocol between the skrl / gymnasium consumers and the ORBIT `RLTaskEnv` subclass
 *     rover_envs/envs/navigation/entrypoints/rover_env.py:12-102   (RoverEnv.__init__ / _reset_idx / step)
 * Each entry point below names the reference interface it replaces.  The Python class that reproduces the reference
 * protocol on top of this ABI is isaac_rover_orbit_amd/envs/rover_env.py (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - plain C: opaque handle, pointers + sizes, int return codes (0 = ROVER_OK); rover_last_error() gives the text
 *     of the last failure on the calling thread.  No exceptions, no torch types.
 *   - every buffer is CALLER-OWNED DEVICE memory (e.g. torch-ROCm tensors); the library never allocates outputs and
 *     keeps only the pointers it is given.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all calls are asynchronous on `stream` and never synchronise the host (the reference's host syncs at
 *     rover_env.py:89-90 and terrain_importer.py:143 are gone).
 *   - one handle per GPU per process; calls on one handle must be serialised by the caller.
 *
 * State layout: SoA, fp32 words, state[word * num_envs + env]; word indices = ROVER_* below (int fields are the raw
 * 32-bit pattern).  get/set_state of the reference does not exist (env state is never checkpointed there); here the
 * state tensor is simply owned by the caller.
 */
#ifndef ROVER_HIP_H
#define ROVER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ROVER_OK = 0,
    ROVER_ERR_INVALID = 1,     /* bad argument / shape                                   */
    ROVER_ERR_STATE = 2,       /* call order (terrain or buffers not bound)              */
    ROVER_ERR_HIP = 3,         /* a HIP runtime call failed                              */
    ROVER_ERR_UNSUPPORTED = 4  /* configuration outside what the kernels were built for  */
};

/* per-env state words (identical numbering to the CPU oracle's AoS words) */
enum {
    ROVER_POS = 0, ROVER_QUAT = 3, ROVER_LINVEL = 7, ROVER_ANGVEL = 10,
    ROVER_BOGIE_Q = 13, ROVER_STEER_Q = 16, ROVER_WHEEL_Q = 20,
    ROVER_BOGIE_QD = 26, ROVER_STEER_QD = 29, ROVER_WHEEL_QD = 33,
    ROVER_TARGET_W = 39, ROVER_HEADING_CMD_W = 42, ROVER_ENV_ORIGIN = 43,
    ROVER_ACTION = 46, ROVER_PREV_ACTION = 48, ROVER_TIME_LEFT = 50, ROVER_EP_LEN = 51,
    ROVER_CMD_B = 52, ROVER_HEADING_CMD_B = 55, ROVER_EP_SUM = 56,
    ROVER_METRIC_POS = 63, ROVER_METRIC_HEAD = 64, ROVER_LAMBDA_N = 65, ROVER_RESET_COUNT = 71,
    ROVER_STATE_WORDS = 72
};
enum { ROVER_NUM_REW = 7, ROVER_NUM_TERM = 4, ROVER_NUM_BODIES = 13, ROVER_LOG_WORDS = 16 };

/* Declarative MDP parameters.  Replaces the @configclass tables of
 *   rover_envs/envs/navigation/rover_env_cfg.py:97-278 (observations, rewards, terminations, commands, timing),
 *   rover_envs/mdp/actions/actions_cfg.py:9-51 + robots/aau_rover/env_cfg.py:21-31 (Ackermann geometry / offset),
 *   rover_envs/envs/navigation/mdp/randomizations.py:12 (z_offset).                                              */
typedef struct rover_config {
    float scale_lin, scale_ang, offset_lin, offset_ang;
    float wheel_radius, d_fr, d_mw, wheelbase;
    float sim_dt;
    int32_t decimation;
    int32_t max_episode_length;
    float max_episode_length_s;
    float success_threshold, far_threshold, target_distance;
    float heading_lo, heading_hi, resample_time;
    float rew_weight[ROVER_NUM_REW]; /* distance_to_target, reached_target, oscillation, angle_to_target,
                                        heading_soft_contraint, collision, far_from_target */
    float obs_scale_distance, obs_scale_heading;
    float scan_resolution, scan_size_x, scan_size_y, scan_height_offset;
    int32_t scan_nx, scan_ny;
    float reset_z_offset;
    int32_t reset_mode;      /* 0 = reference behaviour (root pose only); 1 = also zero velocities and joints */
    uint32_t seed_lo, seed_hi;
    float friction_mu;
    int32_t solver_iterations;
    int32_t max_target_tries;
    int32_t step_mapping;    /* mapping of the step kernel: 0 = auto = sixteen lanes per env (wave-cooperative, spill-free; ONE
                                launch per step wherever the terrain / ray pattern allow it), 2 = the same, stated; 1 = one env
                                per lane: faster than the two-launch form of the group mapping from ~65536 envs per GPU on, but
                                its kernels carry 728 - 984 B of scratch per lane -- only on request.  Bit-identical results */
    int32_t spawn_draw;      /* how a reset picks its spawn table row (randomizations.py:22 `randperm(len(table))[:k]`):
                                1 = DISTINCT rows inside one reset batch, like the reference's randperm prefix: row =
                                (a * global_env_id + b) mod n_spawns, (a, b) redrawn for every reset()/step() call from
                                (seed, call counter), gcd(a, n_spawns) = 1 -- a bijection of the env ids, so no two envs
                                that reset together share a row (needs n_spawns >= global num_envs; the reference's
                                table has 2 x num_envs rows); 0 = independent uniform row per env (with replacement) */
    uint32_t counter_lo, counter_hi; /* initial value of the call counter (rover_set_counter / checkpoint resume) */
    int32_t scan_surface;    /* surface the vertical rays of the height scanner hit (RayCasterCfg mesh_prim_paths,
                                rover_env_cfg.py:84): 0 = the TRIANGLE MESH of the heightfield (every 0.05 m cell split
                                along its (i, j) - (i+1, j+1) diagonal: what a mesh ray-cast of that terrain returns),
                                1 = the bilinear patch (smooth; the wheels' contact surface) */
    int32_t mass_model;      /* where the rover's 25 kg act (rover_envs/assets/robots/aau_rover_simple/rover_instance.usd link table,
                                tests/golden/rover_model.json): 1 (default) = the weight of each bogie SUBTREE (beam + steer links +
                                wheels: 7 + 7 + 9 kg) acts at its own centre of mass -- a generalised gravity force on the bogie
                                coordinate: the statics of the articulated rover (centre : front wheel load 0.74 : 1); 0 = the
                                model of rounds 1 - 4: all mass lumped at the chassis centre of mass (1.96 : 1 by lever arms) */
    float rew_success_threshold, rew_far_threshold; /* thresholds of the reached_target / far_from_target REWARD terms
                                (rover_env_cfg.py:136,162) -- table entries of their own beside the terminations' (:173,177) */
} rover_config;

typedef struct rover_sim rover_sim;

/* Fills `cfg` with the AAURoverEnv-v0 defaults (the values of the cfg files cited above). */
int rover_default_config(rover_config *cfg);

/* Replaces RoverEnv.__init__ -> RLTaskEnv.__init__ (rover_env.py:18-25): creates the per-process simulation handle
 * for `num_envs` environments whose GLOBAL ids are env_id_offset .. env_id_offset + num_envs - 1 (multi-GPU sharding;
 * the RNG is keyed by global id so results do not depend on the sharding).  `device` is the HIP device ordinal. */
int rover_create(const rover_config *cfg, int32_t num_envs, int32_t env_id_offset, int32_t device, rover_sim **out);
int rover_destroy(rover_sim *sim);

/* Replaces RoverTerrainImporter / TerrainManager.__init__ (terrain_importer.py:127-132, terrain_utils.py:92-127):
 * binds the shared read-only terrain data (device pointers, row-major [y][x]).
 *   height     (H, W) fp32  merged surface (ground + obstacles)  -- wheels + ray-caster
 *   obstacle   (H, W) fp32  obstacle layer (0 = no rock)          -- contact-sensor filter (rover_env_cfg.py:72-75)
 *   safe_mask  (H, W) u8    1 = target not allowed                -- safe_rock_mask (terrain_utils.py:311)
 *   spawns     (n_spawns, 3) fp32 spawn table                     -- terrain_utils.py:330-385                   */
int rover_set_terrain(rover_sim *sim, const float *height, const float *obstacle, const uint8_t *safe_mask, int32_t H,
                      int32_t W, float resolution, float min_x, float min_y, const float *spawns, int32_t n_spawns);

/* Optional: a separate (H, W) fp32 heightmap for HeightmapManager.get_height_at (the z of a sampled target,
 * terrain_importer.py:153-155 -> terrain_utils.py:62-84).  The reference builds that heightmap by bounding-box rasterisation
 * (terrain_utils.py:23-57), which is NOT the surface its physics and ray-caster see (the mesh itself); terrains ingested
 * from a mesh therefore carry both.  NULL = use `height` (procedural terrains: one and the same array). */
int rover_set_terrain_lookup(rover_sim *sim, const float *lookup_height);

/* Optional: an EXACT 16-bit copy of `height` for the ray-caster kernel -- height[i] == height_q[i] * q_scale for every
 * cell, q_scale a positive power of two (checked; anything else is ROVER_ERR_INVALID).  The caller guarantees the
 * equality: isaac_rover_orbit_amd/terrain.py quantises generated terrain to 2^-13 m and looks for an exact quantum
 * (2^-13 ... 2^-8 m) for ingested terrain.  Halves the bytes the scan kernel stages per env; results are bit-identical
 * to the fp32 array.  NULL switches back to fp32.  No reference counterpart (the reference ray-casts a mesh with Warp). */
int rover_set_terrain_q16(rover_sim *sim, const int16_t *height_q, float q_scale);

/* Scratch the library needs from the caller (per-wave log partials); bytes. */
size_t rover_workspace_bytes(const rover_sim *sim);

/* Binds the caller-owned persistent buffers: state (ROVER_STATE_WORDS x num_envs fp32, SoA) and workspace. */
int rover_bind(rover_sim *sim, float *state, void *workspace, size_t workspace_bytes);

/* Replaces env.reset() (ORBIT RLTaskEnv.reset -> _reset_idx(all), rover_env.py:27-39): resets every env and writes
 * the first observation.  obs: (num_envs, 4 + scan_nx * scan_ny) fp32 row-major. */
int rover_reset(rover_sim *sim, float *obs, void *stream);

/* RLTaskEnv._reset_idx(env_ids) with the reference's recorded torch draws INJECTED in place of the Philox draws (same
 * code path as the in-step reset): the envs whose `mask` byte is non-zero (NULL = every env) take spawn table row
 * spawn_row[e] (randomizations.py:22 `randperm(len(table))[:k]`), yaw = yaw_u[e] * 2 pi (:30), the target angles
 * theta_u[e * max_target_tries + j] * 2 pi of rejection round j (terrain_importer.py:143-169) and heading command
 * heading_u[e] * (hi - lo) + lo (:93-95); then _update_command and the observation rows of ALL envs.  This is the parity
 * protocol for reset outcomes (torch's RNG stream cannot be reproduced; tests/golden/reset.npz holds the draws).
 * All arrays are device pointers with one entry (row) per env. */
int rover_reset_with_draws(rover_sim *sim, const uint8_t *mask, const int32_t *spawn_row, const float *yaw_u,
                           const float *theta_u, const float *heading_u, float *obs, void *stream);

/* Replaces env.seed(seed) / reset(seed=...) (gymnasium contract): new key of the counter-based RNG used by the resets
 * that follow.  Host only, takes effect with the next launch. */
int rover_set_seed(rover_sim *sim, uint32_t seed_lo, uint32_t seed_hi);

/* The call counter: number of rover_reset / rover_step launches so far (= env.common_step_counter + resets).  It keys the
 * per-batch spawn permutation (spawn_draw = 1); save it with the state words to resume a rollout bit for bit. */
int rover_get_counter(const rover_sim *sim, uint64_t *counter);
int rover_set_counter(rover_sim *sim, uint64_t counter);

/* Replaces RoverEnv.step (rover_env.py:42-102), including the in-step reset of finished envs (_reset_idx :27-39,
 * reset_root_state_rover randomizations.py:12-39, TerrainBasedPositionCommand._resample_command
 * terrain_importer.py:74-95) and the command / observation managers.
 *   action      (num_envs, 2) fp32 in            -- (lin, ang), not clipped (as in the reference)
 *   obs         (num_envs, 4 + rays) fp32 out    -- [last_action(2), distance*0.11, heading/pi, height_scan]
 *   reward      (num_envs,) fp32 out
 *   terminated  (num_envs,) u8 out               -- termination_manager.terminated
 *   truncated   (num_envs,) u8 out               -- termination_manager.time_outs
 *   force       (13 * 3, num_envs) fp32 out or NULL -- contact_sensor.data.force_matrix_w (obstacle filter), stored
 *                  SoA: force[(body * 3 + xyz) * num_envs + env]; a (num_envs, 13, 1, 3) strided view of it satisfies
 *                  the `.view(num_envs, -1, 3)` of rewards.py:119 / terminations.py:59
 *   log         (ROVER_LOG_WORDS,) fp32 in/out   -- extras["log"]: [0..6] Episode Reward/<term>, [7..10] Episode
 *                  Termination/<term> (time_limit, is_success, far_from_target, collision), [11..12] Metrics/
 *                  target_pose/{error_pos,error_heading}, [13] number of envs reset in this step; entries 0..12 are
 *                  only rewritten when [13] > 0 (the reference rebuilds extras["log"] only inside _reset_idx).      */
int rover_step(rover_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated, uint8_t *truncated,
               float *force, float *log, void *stream);

/* The env step in two halves -- the slow path for USER-WRITTEN reward / termination terms.  The reference's term tables
 * (rover_env_cfg.py:126-183) hold arbitrary `func=` callables with the signature func(env, **params) (mdp/rewards.py:14-137,
 * mdp/terminations.py:14-64); ORBIT's managers evaluate them on the state the physics left, BEFORE _reset_idx
 * (rover_env.py:82-91).  rover_step does physics, terms and reset in one launch, so a caller with its own terms uses
 *   rover_step_begin   rover_env.py:62-86: action, 6 physics steps, counters, the built-in terms and their episodic sums;
 *                      stores state, reward (built-in terms), terminated / truncated (built-in terms), force.  No reset.
 *   ... the caller evaluates its terms on `state` / `force` (device tensors), adds to `reward`, ORs into the flags ...
 *   rover_step_finish  rover_env.py:89-99 for `reset_mask` (u8 per env: built-in OR user terminations): episodic log of the
 *                      masked envs, reset, command update, observation rows; `log` is reduced eagerly.
 * Same arithmetic as rover_step (first half: sixteen lanes per env, second half: one env per lane -- the mappings agree bit for
 * bit): with a reset mask equal to the built-in flags the two halves produce the bits rover_step produces (the log vector up to
 * the order of its sums).  `force` is required (the built-in collision term of the second half reads it). */
int rover_step_begin(rover_sim *sim, const float *action, float *reward, uint8_t *terminated, uint8_t *truncated, float *force,
                     void *stream);
int rover_step_finish(rover_sim *sim, const uint8_t *reset_mask, float *obs, float *force, float *log, void *stream);

/* extras["log"] on demand.  The reference fills the dictionary inside _reset_idx (rover_env.py:27-39) and its consumers read it
 * when they choose to: the reference's own trainer reads every entry after EVERY step (skrl_utils.py:139-142: .item() in the
 * training loop), a random-action rollout never does.  With rover_set_log_deferred(sim, 1) rover_step leaves `log` alone; rover_flush_log
 * produces, at any later point on the same stream, exactly the vector the per-step reduction would hold there: [0..12] from the
 * latest step in which an env reset, [13] = the number of envs reset in the latest step (0 when the latest resets are older).
 * Flush at most once per step (a second flush without a step in between reports [13] = 0).  With deferral rover_step is ONE
 * kernel launch wherever the one-launch form applies -- the height scan as the last phase of the step kernel's waves (group
 * mapping, int16 terrain copy, at most 1024 rays, a batch of at least eight envs per compute unit); without it rover_log_kernel
 * follows that launch.  Where the form does not apply a step is two launches (the scan kernel's first workgroup reduces the log).
 * Default: not deferred. */
int rover_set_log_deferred(rover_sim *sim, int32_t deferred);
/* Observation rows with streaming (non-temporal) stores in the one-launch kernels: 1 = on.  Worth ~2 % of the step when nothing on
 * the device reads the rows next (a host-side consumer, a random-action rollout); leave it off (default) when a policy kernel
 * follows -- its read of the rows then hits L2.  No reference counterpart (the reference's observation tensor is a torch.cat). */
int rover_set_obs_streaming(rover_sim *sim, int32_t streaming);
int rover_flush_log(rover_sim *sim, float *log, void *stream);

/* Profiling twin of rover_step: identical launches, bracketed by HIP events recorded on `stream`; returns the device
 * time of the two kernels in milliseconds (one-launch form: of the fused kernel and of the log reduction, if any).  Synchronises the host -- measurement only (bench.py roofline leg). */
int rover_profile_step(rover_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated,
                       uint8_t *truncated, float *force, float *log, void *stream, float *ms_step_kernel,
                       float *ms_scan_kernel);

/* Average elapsed time (ms) of an event pair with nothing recorded between the two events: the fixed cost each interval of
 * rover_profile_step carries; bench.py subtracts it.  Synchronises -- measurement only. */
int rover_profile_event_overhead(rover_sim *sim, void *stream, int32_t reps, float *ms);

/* Tracing (the reference has none on this path -- ORBIT's own timers live upstream): with markers enabled
 * rover_step pushes one roctx range "rover_step" per call with the ranges "K1 <kernel>" and "K2 scan + observation rows"
 * around its two launches, so that `rocprofv3 --marker-trace --kernel-trace` output lines up with env steps.  The roctx
 * library is resolved with dlopen at the first enabling call (ROVER_ERR_UNSUPPORTED when none is installed).  Host only. */
int rover_set_markers(rover_sim *sim, int32_t enabled);

/* (One-launch forms: step_kernel = "rover_step_scan_kernel<true|false>" / "rover_step_scan1_kernel<...>", scan_kernel =
 * "rover_log_kernel", or "" when the log is deferred.)
 * Names of the two kernels rover_step launches for the current configuration / terrain, exactly as rocprofv3's kernel trace
 * prints them minus the "(anonymous namespace)::" qualifier and the parameter list (e.g. "rover_step_kernel_group",
 * "rover_scan_step_kernel<true, true, 1024, 2>"): the keys of bench.py's roofline block and of profiles/hbm_traffic.json.
 * Both buffers hold `cap` bytes.  Host only. */
int rover_kernel_names(const rover_sim *sim, char *step_kernel, char *scan_kernel, size_t cap);

/* Unit entry points used by the parity tests (same kernels' device functions, one env per lane):
 *   rover_ackermann      -- AckermannAction2.process_actions + ackermann (ackermann_actions.py:226-322)
 *                           steer (n,4) [FL,RL,RR,FR], wheel (n,6) [ML,FL,RL,RR,MR,FR] as the reference stacks them
 *   rover_height_scan    -- RayCaster + height_scan_rover (rover_env_cfg.py:78-86, observations.py:35-45) for the
 *                           bound state; scan (num_envs, rays)
 *   rover_physics        -- `substeps` x (write_data_to_sim, sim.step, scene.update) (rover_env.py:64-72) on the bound
 *                           state with explicit joint targets in MODEL order (steer FL,FR,RL,RR; wheels FL,FR,CL,CR,RL,RR)
 *   rover_mdp_terms      -- the term functions of the step kernel's tail on caller-supplied rows (device pointers):
 *                           3 observation / 7 reward / 3 termination terms + mdp.time_out (observations.py:15-32,
 *                           rewards.py:14-137, terminations.py:14-64); cmd_b (n,3), action / prev_action (n,2),
 *                           ep_len (n,) int32, force (n,13,3) -> obs_distance, obs_angle (n,), rew (n,7) unweighted,
 *                           term (n,4) u8 [time_out, is_success, far_from_target, collision] */
int rover_ackermann(rover_sim *sim, int32_t n, const float *raw, float *processed, float *steer, float *wheel, void *stream);
int rover_mdp_terms(rover_sim *sim, int32_t n, const float *cmd_b, const float *action, const float *prev_action,
                    const int32_t *ep_len, const float *force, float *obs_distance, float *obs_angle, float *rew,
                    uint8_t *term, void *stream);
int rover_height_scan(rover_sim *sim, float *scan, void *stream);
int rover_physics(rover_sim *sim, const float *steer_target, const float *wheel_target, int32_t substeps, float *force,
                  void *stream);

/* Model constant table (same order as the oracle's rvo_model_constants); returns the count. Host only. */
int rover_model_constants(float *out, int32_t cap);
int rover_state_words(void);
size_t rover_config_bytes(void); /* sizeof(rover_config): lets a binding verify its struct mirror */
const char *rover_last_error(void);
const char *rover_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_HIP_H */
