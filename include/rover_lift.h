/*
 * rover_lift.h -- C ABI of the MI355X-native FrankaCubeLift-v0 step()/reset() path (librover_hip.so), SURVEY 8(f-4),
 * BASELINE config 5 ("manipulation task ... articulated-arm integrator + contact, num_envs=2048").
 *
 * Replaces, for gym id "FrankaCubeLift-v0" (rover_envs/envs/manipulation/config/franka/__init__.py:6-14, entry point
 * omni.isaac.orbit.envs:RLTaskEnv with FrankaCubeLiftEnvCfg), the ORBIT RLTaskEnv.step / reset of
 *     rover_envs/envs/manipulation/manipulation_env_cfg.py:93-235, config/franka/joint_pos_env_cfg.py:25-82,
 *     mdp/rewards.py:20-67, mdp/observations.py:19-31.
 * Same conventions as rover_hip.h: plain C, opaque handle, caller-owned DEVICE buffers, int codes + rover_last_error(),
 * asynchronous on the caller's stream.
 */
#ifndef ROVER_LIFT_H
#define ROVER_LIFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-env state words (SoA on the GPU: state[word * num_envs + env]) */
enum {
    LIFT_Q = 0,            /* 9  joint positions: 7 arm + 2 fingers                                   */
    LIFT_QD = 9,           /* 9  joint velocities                                                     */
    LIFT_OBJ_POS = 18,     /* 3  cube centre, world (= env frame: robot root at the origin, identity) */
    LIFT_OBJ_QUAT = 21,    /* 4  cube orientation (w, x, y, z)                                        */
    LIFT_OBJ_LIN = 25,     /* 3                                                                       */
    LIFT_OBJ_ANG = 28,     /* 3  world frame                                                          */
    LIFT_CMD = 31,         /* 7  UniformPoseCommand: position + quaternion in the robot base frame    */
    LIFT_TIME_LEFT = 38,   /* 1  command resampling timer                                             */
    LIFT_EP_LEN = 39,      /* 1  int32                                                                */
    LIFT_ACTION = 40,      /* 8  action_manager.action                                                */
    LIFT_PREV_ACTION = 48, /* 8  action_manager.prev_action                                           */
    LIFT_EP_SUM = 56,      /* 6  per-term episodic reward sums                                        */
    LIFT_RESET_COUNT = 62, /* 1  uint32                                                               */
    LIFT_STATE_WORDS = 64
};
enum { LIFT_NUM_REW = 6, LIFT_NUM_TERM = 2, LIFT_OBS = 36, LIFT_ACT = 8, LIFT_LOG_WORDS = 16 };

/* FrankaCubeLiftEnvCfg in the shape the kernels consume */
typedef struct lift_config {
    float sim_dt;                 /* manipulation_env_cfg.py:232 (1/100)                 */
    int32_t decimation;           /* :233                                                */
    int32_t max_episode_length;   /* ceil(5 s / (0.01 * 2)) = 250, :234                  */
    float max_episode_length_s;
    float action_scale;           /* joint_pos_env_cfg.py:36 (0.5, use_default_offset)   */
    float finger_open, finger_close; /* :41-42                                           */
    float rew_weight[LIFT_NUM_REW]; /* reaching 1, lifting 15, goal 16, goal fine 5, action_rate 1e-3, joint_vel 1e-4 (:120-144)
                                       (the reference's weights of the two penalties are POSITIVE, manipulation_env_cfg.py:137-143) */
    float reach_std, goal_std, goal_fine_std, minimal_height; /* :121-135             */
    float drop_height;            /* :153 (-0.05)                                        */
    float cmd_lo[3], cmd_hi[3];   /* :170 pos_x (0.3, 0.7), pos_y (0.3, 0.7), pos_z (0, 0) */
    float cmd_resample_time;      /* :166 (5.0)                                          */
    float obj_init[3];            /* joint_pos_env_cfg.py:51 (0.5, 0, 0.055)             */
    float obj_range_lo[3], obj_range_hi[3]; /* manipulation_env_cfg.py:185             */
    float ee_offset_z;            /* joint_pos_env_cfg.py:78 (0.1034)                    */
    uint32_t seed_lo, seed_hi;
    int32_t solver_iterations;    /* Gauss-Seidel sweeps of the cube contact solver      */
    float mu_table, mu_pad;
} lift_config;

typedef struct rover_lift_sim rover_lift_sim;

/* FrankaCubeLiftEnvCfg defaults (the cfg files cited above). */
int rover_lift_default_config(lift_config *cfg);
size_t rover_lift_config_bytes(void);
int rover_lift_state_words(void);
/* Model constant table (same order as the oracle's lfo_model_constants); returns the count.  Host only. */
int rover_lift_model_constants(float *out, int32_t cap);

/* RLTaskEnv.__init__: handle for `num_envs` envs with GLOBAL ids env_id_offset .. (RNG keyed by global id). */
int rover_lift_create(const lift_config *cfg, int32_t num_envs, int32_t env_id_offset, int32_t device, rover_lift_sim **out);
int rover_lift_destroy(rover_lift_sim *sim);
size_t rover_lift_workspace_bytes(const rover_lift_sim *sim);
/* state: LIFT_STATE_WORDS x num_envs fp32 (SoA); workspace: rover_lift_workspace_bytes() bytes, 128-byte aligned.  Clears the
 * workspace's two counters with a synchronous hipMemset (init-time call). */
int rover_lift_bind(rover_lift_sim *sim, float *state, void *workspace, size_t workspace_bytes);

/* env.seed(seed) / reset(seed=...) (gymnasium contract): new key of the counter-based RNG used by the resets that follow.
 * Host only, takes effect with the next launch. */
int rover_lift_set_seed(rover_lift_sim *sim, uint32_t seed_lo, uint32_t seed_hi);

/* env.reset(): reset_scene_to_default + reset_root_state_uniform(object) + manager resets + command resample for every env
 * (manipulation_env_cfg.py:176-190); obs (num_envs, 36) */
int rover_lift_reset(rover_lift_sim *sim, float *obs, void *stream);

/* RLTaskEnv.step: action (num_envs, 8) = 7 arm joint-position offsets (x 0.5 + default pose, joint_pos_env_cfg.py:35-37) +
 * 1 binary gripper command (< 0 closes, :38-43) -> 2 x (implicit-PD arm dynamics, cube / table / finger contact) ->
 * terminations (time_out, object_dropping) -> 6 rewards x dt -> in-step reset -> command -> observations.
 *   obs (num_envs, 36): joint_pos_rel 9, joint_vel_rel 9, object position in the robot root frame 3, command 7, last action 8
 *   reward (num_envs,), terminated / truncated (num_envs,) u8
 *   log (16,): [0..5] Episode Reward/<term> (mean over the envs reset in this step / episode length in s), [6] time_out count,
 *              [7] object_dropping count, [8] number of envs reset; entries 0..7 rewritten only when [8] > 0 */
int rover_lift_step(rover_lift_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated, uint8_t *truncated,
                    float *log, void *stream);

/* extras["log"] on demand.  The reference fills the log dictionary inside _reset_idx and its consumers read it now and then
 * (skrl_utils.py:139-142); the reduction behind `log` is a second, dependent launch per step (~1.6 us of 19).  With
 * rover_lift_set_log_deferred(sim, 1) rover_lift_step leaves `log` alone and rover_lift_flush_log produces, at any later point
 * on the same stream, exactly the vector the per-step reduction would hold there: [0..7] from the latest step in which an env
 * reset, [8] = the number of envs reset in the latest step (0 when the latest resets are older).  Flush at most once per step
 * (a second flush without a step in between reports [8] = 0).  Default: not deferred. */
int rover_lift_set_log_deferred(rover_lift_sim *sim, int32_t deferred);
int rover_lift_flush_log(rover_lift_sim *sim, float *log, void *stream);

/* Profiling twin of rover_lift_step: the same launch bracketed by HIP events on `stream`; device time of the step kernel in
 * milliseconds and, next to it, what an event pair with nothing in between reports (the fixed cost the first figure
 * carries).  Synchronises -- measurement only (bench.py --config 5). */
int rover_lift_profile_step(rover_lift_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated,
                            uint8_t *truncated, float *log, void *stream, float *ms_step_kernel, float *ms_event_overhead);
/* Name of the step kernel as rocprofv3 prints it (minus qualifiers / parameter list), e.g. "lift_step_kernel<8>". */
int rover_lift_kernel_name(const rover_lift_sim *sim, char *step_kernel, size_t cap);

/* Unit entry for the parity tests: the reference's own term functions (rewards.py:20-67, observations.py:19-31) on caller rows:
 * obj_pos, ee_pos (n,3), root_state (n,13), cmd (n,7) -> lifted, reach, goal, goal_fine (n,), obj_pos_b (n,3); device pointers */
int rover_lift_terms(rover_lift_sim *sim, int32_t n, const float *obj_pos, const float *ee_pos, const float *root_state,
                     const float *cmd, float *lifted, float *reach, float *goal, float *goal_fine, float *obj_pos_b, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_LIFT_H */
