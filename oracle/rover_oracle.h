/*
 * rover_oracle.h -- TEST INFRASTRUCTURE ONLY (see rover_oracle.c header).
 *
 * Plain-C CPU restatement of the AAURoverEnv-v0 step()/reset() hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path (isaac_rover_orbit_amd) never does.
 */
#ifndef ROVER_ORACLE_H
#define ROVER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-env state words (AoS here: state[env * RVO_STATE_WORDS + word]) ------------------------------ */
enum {
    RVO_POS = 0,            /* 3  root (Body link) position, world                                        */
    RVO_QUAT = 3,           /* 4  root orientation (w, x, y, z)                                           */
    RVO_LINVEL = 7,         /* 3  centre-of-mass linear velocity, world                                   */
    RVO_ANGVEL = 10,        /* 3  angular velocity, world                                                 */
    RVO_BOGIE_Q = 13,       /* 3  FL_Boogie, FR_Boogie, R_Boogie                                          */
    RVO_STEER_Q = 16,       /* 4  FL, FR, RL, RR                                                          */
    RVO_WHEEL_Q = 20,       /* 6  FL, FR, CL, CR, RL, RR                                                  */
    RVO_BOGIE_QD = 26,      /* 3                                                                          */
    RVO_STEER_QD = 29,      /* 4                                                                          */
    RVO_WHEEL_QD = 33,      /* 6                                                                          */
    RVO_TARGET_W = 39,      /* 3  pos_command_w                                                           */
    RVO_HEADING_CMD_W = 42, /* 1                                                                          */
    RVO_ENV_ORIGIN = 43,    /* 3                                                                          */
    RVO_ACTION = 46,        /* 2  action_manager.action                                                   */
    RVO_PREV_ACTION = 48,   /* 2  action_manager.prev_action                                              */
    RVO_TIME_LEFT = 50,     /* 1  command resampling timer                                                */
    RVO_EP_LEN = 51,        /* 1  int32 episode_length_buf                                                */
    /* ---- manager state beyond the 52-word physical/MDP state ---- */
    RVO_CMD_B = 52,         /* 3  pos_command_b as left by the previous command_manager.compute (B-13)    */
    RVO_HEADING_CMD_B = 55, /* 1                                                                          */
    RVO_EP_SUM = 56,        /* 7  per-term episodic reward sums                                           */
    RVO_METRIC_POS = 63,    /* 1  metrics["error_pos"]                                                    */
    RVO_METRIC_HEAD = 64,   /* 1  metrics["error_heading"]                                                */
    RVO_LAMBDA_N = 65,      /* 6  cached normal impulses (contact warm start)                             */
    RVO_RESET_COUNT = 71,   /* 1  uint32 number of resets so far (RNG counter)                            */
    RVO_STATE_WORDS = 72
};

enum { RVO_NUM_REW = 7, RVO_NUM_TERM = 4, RVO_NUM_BODIES = 13 };

/* log vector written by rvo_step when >= 1 env was reset in the step (extras["log"]):
 *   [0..6]  Episode Reward/<term>           (mean of episodic sums over reset envs / max_episode_length_s)
 *   [7..10] Episode Termination/<term>      (counts: time_limit, is_success, far_from_target, collision)
 *   [11]    Metrics/target_pose/error_pos   [12] Metrics/target_pose/error_heading   (means over reset envs)
 *   [13]    number of envs reset in this step (0 => entries 0..12 were left untouched)                    */
enum { RVO_LOG_WORDS = 16 };

typedef struct {
    /* action term: rover_envs/mdp/actions/actions_cfg.py:17-33, robots/aau_rover/env_cfg.py:21-31 */
    float scale_lin, scale_ang, offset_lin, offset_ang;
    float wheel_radius, d_fr, d_mw, wheelbase;
    /* timing: rover_env_cfg.py:269-271 */
    float sim_dt;
    int32_t decimation;
    int32_t max_episode_length;
    float max_episode_length_s;
    /* commands / terminations / rewards: rover_env_cfg.py:126-200, terrain_importer.py:132 */
    float success_threshold, far_threshold, target_distance;
    float heading_lo, heading_hi, resample_time;
    float rew_weight[RVO_NUM_REW];
    float obs_scale_distance, obs_scale_heading;
    /* ray caster: rover_env_cfg.py:78-86; observations.py:45 */
    float scan_resolution, scan_size_x, scan_size_y, scan_height_offset;
    int32_t scan_nx, scan_ny;
    /* reset: randomizations.py:12-39 */
    float reset_z_offset;
    int32_t reset_mode;          /* 0 = reference (root pose only, B-17); 1 = also zero velocities + joints */
    uint32_t seed_lo, seed_hi;
    /* contact model */
    float friction_mu;
    int32_t solver_iterations;
    int32_t max_target_tries;
    int32_t step_mapping;        /* GPU-side kernel mapping selector; ignored by the oracle */
    int32_t spawn_draw;          /* 1 = distinct rows per reset batch (affine bijection of the env ids), 0 = independent   */
    uint32_t counter_lo, counter_hi; /* call counter keying the per-batch spawn permutation (the caller increments it)  */
    int32_t scan_surface;        /* 0 = triangle mesh of the heightfield (cells split along (i,j)-(i+1,j+1)), 1 = bilinear */
    int32_t mass_model;          /* 1 (default) = the weight of each bogie SUBTREE (beam + steer links + wheels: 7 + 7 + 9 of the USD's
                                    25 kg) acts at its own centre of mass: a generalised gravity force on the bogie coordinate -- the
                                    statics of the articulated rover (centre : front wheel load 0.74 : 1); 0 = rounds 1-4: everything
                                    lumped at the chassis centre of mass (2 : 1) */
    float rew_success_threshold, rew_far_threshold; /* thresholds of the reached_target / far_from_target REWARD terms
                                    (rover_env_cfg.py:136,162: table entries of their own, beside the terminations' :173,177) */
} rvo_config;

typedef struct {
    const float *height;          /* (H, W) merged surface [row = y, col = x], metres                        */
    const float *obstacle;        /* (H, W) obstacle layer: height of the rock above the ground (0 = none)    */
    const uint8_t *safe_mask;     /* (H, W) 1 = target not allowed (safe_rock_mask, terrain_utils.py:311)     */
    int32_t H, W;
    float resolution, min_x, min_y;
    const float *spawns;          /* (n_spawns, 3) spawn table (terrain_utils.py:330-385)                     */
    int32_t n_spawns;
    const float *lookup;          /* (H, W) heightmap of get_height_at (target z); NULL = `height`              */
} rvo_terrain;

void rvo_default_config(rvo_config *cfg);
int rvo_state_words(void);
/* model constant table, same order as rover_model_constants() of the HIP library */
int rvo_model_constants(float *out, int cap);

/* ---- exact-arithmetic layer (pinned by tests/golden) -------------------------------------------------- */
void rvo_ackermann(const rvo_config *cfg, int n, const float *raw /* n x 2 */, float *processed /* n x 2 */,
                   float *steer /* n x 4 [FL,RL,RR,FR] */, float *wheel /* n x 6 [ML,FL,RL,RR,MR,FR] */);
void rvo_mdp_terms(const rvo_config *cfg, int n, const float *cmd_b /* n x 3 */, const float *action,
                   const float *prev_action, const int32_t *ep_len, const float *force /* n x 13 x 3 */,
                   float *obs_distance, float *obs_angle, float *rew /* n x 7 unweighted */,
                   uint8_t *term /* n x 4: time_out, success, far, collision */);
void rvo_height_scan_term(const rvo_config *cfg, int n, int rays, const float *pos_z, const float *hit_z, float *out);
void rvo_get_height_at(const rvo_terrain *t, int n, const float *xy, float *out);
void rvo_target_invalid(const rvo_terrain *t, int n, const float *xy, uint8_t *out);
void rvo_update_command(int n, const float *root_pos, const float *root_quat, const float *target_w,
                        const float *heading_cmd_w, float *cmd_b, float *heading_b);
void rvo_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);

/* ---- modelled layer (physics / ray caster: parity UNPINNED) ------------------------------ */
void rvo_terrain_sample(const rvo_terrain *t, int n, const float *xy, float *h, float *gx, float *gy, float *obst);
void rvo_height_scan(const rvo_config *cfg, const rvo_terrain *t, int n, const float *state, float *scan);
void rvo_physics_step(const rvo_config *cfg, const rvo_terrain *t, int n, float *state,
                      const float *steer_target /* n x 4 model order FL,FR,RL,RR */,
                      const float *wheel_target /* n x 6 model order FL,FR,CL,CR,RL,RR */,
                      int substeps, float *force /* n x 13 x 3 or NULL */);

/* ---- whole path ---------------------------------------------------------------------------------------- */
void rvo_reset_all(const rvo_config *cfg, const rvo_terrain *t, int n, int env_id_offset, float *state, float *obs);
/* _reset_idx with injected draws (theta_u: n x max_target_tries uniforms, consumed in order per env) */
void rvo_reset_with_draws(const rvo_config *cfg, const rvo_terrain *t, int n, int env_id_offset, float *state,
                          const uint8_t *mask, const int32_t *spawn_row, const float *yaw_u, const float *theta_u,
                          const float *heading_u, float *obs);
void rvo_step(const rvo_config *cfg, const rvo_terrain *t, int n, int env_id_offset, float *state,
              const float *action /* n x 2 */, float *obs /* n x (4 + rays) */, float *reward, uint8_t *terminated,
              uint8_t *truncated, float *force /* n x 13 x 3 */, float *log_out /* RVO_LOG_WORDS */);
int rvo_num_threads(void);
void rvo_set_num_threads(int n); /* 0 = OpenMP default */
/* study variants of the dynamics model (rover_oracle.c: bit 0 triangle-surface wheel contact, bit 1 coupled 9 x 9 mass matrix + PGS);
 * 0 = the model the HIP path implements -- tests, smoke and bench never set anything else */
void rvo_set_split(float split_c, float split_b);   /* study hook: mass-splitting factors (0 = the product's 3 / 2) */
void rvo_set_model_variant(int v);
int rvo_get_model_variant(void);

#ifdef __cplusplus
}
#endif
#endif
