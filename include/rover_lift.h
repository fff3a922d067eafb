/*
 * rover_lift.h -- C ABI of the MI355X-native FrankaCubeLift-v0 step()/reset() path (librover_hip.so), SURVEY 8(f-4),
 * BASELINE config 5 ("manipulation task ... articulated-arm integrator + contact, num_envs=2048").
 *
 * Replaces, for gym id "FrankaCubeLift-v0" (rover_envs/envs/manipulation/config/franka/__init__.py:6-14, entry point
 * omni.isaac.orbit.envs:RLTaskEnv with FrankaCubeLiftEnvCfg), the ORBIT RLTaskEnv.step / reset of
 *     rover_envs/envs/manipulation/manipulation_env_cfg.py:93-235, config/franka/joint_pos_env_cfg.py:25-82,
 *     mdp/rewards.py:20-67, mdp/observations.py:19-31.
 * Same conventions as rover_hip.h: plain C, opaque handle, caller-owned DEVICE buffers, int codes + rover_last_error(),
 * asynchronous on the caller's stream.  State: SoA fp32 words, state[word * num_envs + env], LIFT_* indices and the
 * lift_config struct are defined in isaac_rover_orbit_amd/csrc/lift_model.h (the model definition; plain C).
 */
#ifndef ROVER_LIFT_H
#define ROVER_LIFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct lift_config;
typedef struct rover_lift_sim rover_lift_sim;

/* FrankaCubeLiftEnvCfg defaults (the cfg files cited above). */
int rover_lift_default_config(struct lift_config *cfg);
size_t rover_lift_config_bytes(void);
int rover_lift_state_words(void);

/* RLTaskEnv.__init__: handle for `num_envs` envs with GLOBAL ids env_id_offset .. (RNG keyed by global id). */
int rover_lift_create(const struct lift_config *cfg, int32_t num_envs, int32_t env_id_offset, int32_t device, rover_lift_sim **out);
int rover_lift_destroy(rover_lift_sim *sim);
size_t rover_lift_workspace_bytes(const rover_lift_sim *sim);
/* state: LIFT_STATE_WORDS x num_envs fp32 (SoA); workspace: rover_lift_workspace_bytes() bytes, 128-byte aligned */
int rover_lift_bind(rover_lift_sim *sim, float *state, void *workspace, size_t workspace_bytes);

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

/* Unit entry for the parity tests: the reference's own term functions (rewards.py:20-67, observations.py:19-31) on caller rows:
 * obj_pos, ee_pos (n,3), root_state (n,13), cmd (n,7) -> lifted, reach, goal, goal_fine (n,), obj_pos_b (n,3); device pointers */
int rover_lift_terms(rover_lift_sim *sim, int32_t n, const float *obj_pos, const float *ee_pos, const float *root_state,
                     const float *cmd, float *lifted, float *reach, float *goal, float *goal_fine, float *obj_pos_b, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_LIFT_H */
