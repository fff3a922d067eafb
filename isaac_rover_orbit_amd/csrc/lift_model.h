/*
 * lift_model.h -- the reduced FrankaCubeLift-v0 model (BASELINE config 5): articulated-arm integrator +
 * cube / table / gripper contact + the task's MDP, written once in plain C (float, fmaf, explicit polynomials) and compiled
 * BOTH into the HIP kernel (csrc/lift_kernels.hip, one env per lane) and into the CPU checker (oracle/lift_oracle.c).
 *
 * Reference behaviour being replaced (file:line in /root/reference):
 *   rover_envs/envs/manipulation/manipulation_env_cfg.py:93-235  actions / observations / rewards / terminations / commands /
 *                                                                 randomization tables, dt = 0.01 s, decimation 2, 5 s episodes
 *   rover_envs/envs/manipulation/config/franka/joint_pos_env_cfg.py:25-82  Franka + 0.8-scale DexCube, joint-position action
 *                                                                 (scale 0.5, default offset), binary gripper, ee frame offset
 *   rover_envs/envs/manipulation/mdp/rewards.py:20-67, mdp/observations.py:19-31   the task's own term functions
 *   ORBIT (third-party, absent): RLTaskEnv.step ordering, mdp.joint_pos_rel / joint_vel_rel / generated_commands /
 *   last_action / action_rate_l2 / joint_vel_l2 / time_out / base_height / reset_scene_to_default /
 *   reset_root_state_uniform, UniformPoseCommand, JointPositionAction, BinaryJointPositionAction (restated from their
 *   documented behaviour, SURVEY App. C style) and PhysX (replaced by the model below).
 *
 * PARITY: the term arithmetic the reference owns is pinned by tests/golden/lift_terms.npz; everything PhysX / ORBIT did is
 * a documented MODEL (parity unpinned): Franka kinematics from the public modified-DH table, link inertias approximating the
 * public identification (Gaz et al. 2019), ORBIT's FRANKA_PANDA_CFG actuator gains as published (stiffness 80 / damping 4,
 * effort 87 | 12 N m; hand 2000 / 100), remote Nucleus assets (table, cube) replaced by a plane at z = 0 and a 4 cm cube.
 *
 * Model.  Arm: 7 revolute joints, full joint-space dynamics  M(q) qdd + c(q, qd) = tau  with M and c from the recursive
 * Newton-Euler algorithm (c = RNE(q, qd, 0) incl. gravity; column j of M = RNE(q, 0, e_j) without gravity); implicit PD
 * actuators:  (M + h Kd + h^2 Kp) qd+ = M qd + h (Kp (q* - q) - c),  Cholesky 7x7; joints whose PD torque exceeds the effort
 * limit are re-solved with the limit as a constant torque; velocity and position limits by clamping.  Gripper: two prismatic
 * fingers (1 DOF each, implicit PD).  Cube: 6-DOF rigid body; contacts = its 8 corners against the table plane (normal +
 * 2 friction rows each) and the two finger pads (normal + 2 friction + 1 torsional row each, the hand acting as a moving
 * kinematic body); projected Gauss-Seidel on velocities with Baumgarte stabilisation.  No arm-table / arm-cube collisions.
 */
#ifndef LIFT_MODEL_H
#define LIFT_MODEL_H

#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define LM_FN __host__ __device__ __forceinline__ static
#else
#define LM_FN static inline
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
    int32_t solver_iterations;    /* PGS sweeps of the cube contact solver               */
    float mu_table, mu_pad;
} lift_config;

/* ---------------------------------------------------------------------------------------------------- constants */
#define LM_PI 3.14159265358979323846f
#define LM_HALF_PI 1.57079632679489661923f
#define LM_G 9.81f
#define LM_CUBE_HALF 0.02f          /* 0.8 x DexCube ~ 4 cm edge (joint_pos_env_cfg.py:53-54) */
#define LM_CUBE_MASS 0.064f         /* 1000 kg/m^3                                            */
#define LM_PAD_X 0.010f             /* finger pad half length (hand x)                        */
#define LM_PAD_Z 0.009f             /* finger pad half height (hand z)                        */
#define LM_FINGER_MASS 0.05f
#define LM_FINGER_KP 2000.0f        /* FRANKA_PANDA_CFG panda_hand actuator                   */
#define LM_FINGER_KD 100.0f
#define LM_FINGER_EFFORT 200.0f
#define LM_FINGER_VLIM 0.2f
#define LM_ARM_KP 80.0f             /* FRANKA_PANDA_CFG panda_shoulder / panda_forearm        */
#define LM_ARM_KD 4.0f
#define LM_BAUMGARTE 0.2f
#define LM_TORSION_R 0.008f         /* effective radius of the pad patch for torsional friction */

/* modified DH (Craig) of the Franka Emika Panda: a_{i-1}, d_i, alpha_{i-1} */
#define LM_DH_A {0.0f, 0.0f, 0.0f, 0.0825f, -0.0825f, 0.0f, 0.088f}
#define LM_DH_D {0.333f, 0.0f, 0.316f, 0.0f, 0.384f, 0.0f, 0.0f}
#define LM_DH_SA {0.0f, -1.0f, 1.0f, 1.0f, -1.0f, 1.0f, 1.0f}   /* sin(alpha): alpha in {0, -pi/2, +pi/2} */
#define LM_DH_CA {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}     /* cos(alpha)                               */
#define LM_FLANGE_D 0.107f
/* link inertial parameters in the link (modified-DH) frames; link 7 carries the hand and the fingers */
#define LM_LINK_M {4.97f, 0.647f, 3.228f, 3.588f, 1.226f, 1.667f, 1.495f}
#define LM_LINK_C {{0.0039f, 0.0021f, -0.0476f}, {-0.0031f, -0.0287f, 0.0035f}, {0.0275f, 0.0392f, -0.0665f},           \
                   {-0.0532f, 0.1044f, 0.0275f}, {-0.0118f, 0.0411f, -0.0384f}, {0.0601f, -0.0141f, -0.0105f},          \
                   {0.0054f, -0.0021f, 0.1050f}}
#define LM_LINK_I {{0.70f, 0.71f, 0.0091f}, {0.0080f, 0.0281f, 0.0260f}, {0.0372f, 0.0362f, 0.0108f},                    \
                   {0.0259f, 0.0196f, 0.0283f}, {0.0355f, 0.0295f, 0.0086f}, {0.0020f, 0.0043f, 0.0054f},                \
                   {0.0260f, 0.0240f, 0.0060f}}
#define LM_ARMATURE 0.02f
#define LM_Q_LO {-2.8973f, -1.7628f, -2.8973f, -3.0718f, -2.8973f, -0.0175f, -2.8973f}
#define LM_Q_HI {2.8973f, 1.7628f, 2.8973f, -0.0698f, 2.8973f, 3.7525f, 2.8973f}
#define LM_QD_LIM {2.175f, 2.175f, 2.175f, 2.175f, 2.61f, 2.61f, 2.61f}
#define LM_EFFORT {87.0f, 87.0f, 87.0f, 87.0f, 12.0f, 12.0f, 12.0f}
/* FRANKA_PANDA_CFG.init_state.joint_pos */
#define LM_Q_DEFAULT {0.0f, -0.569f, 0.0f, -2.810f, 0.0f, 3.037f, 0.741f, 0.04f, 0.04f}

/* ---------------------------------------------------------------------------------------------------- small math */
LM_FN float lm_clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
LM_FN void lm_cross(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
LM_FN float lm_dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
/* same Cody-Waite + Cephes sequence as rv_sincosf of the rover path */
LM_FN void lm_sincosf(float x, float *s, float *c)
{
    const float k = floorf(x * 0.63661977236758134f + 0.5f);
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188e-8f;
    const int q = ((int)k) & 3;
    const float z = r * r;
    const float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * (-1.9515295891e-4f)));
    const float cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const float ss = (q & 1) ? cp : sp;
    const float cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
/* expf / tanhf as explicit fp32 sequences (Cephes expf: range reduction by ln 2, degree-5 polynomial), ~1 ulp */
LM_FN float lm_expf(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    const float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * z + r + 1.0f;
    const int32_t e = (int32_t)n;
    union { float f; int32_t i; } u;
    u.i = (e + 127) << 23;
    return p * u.f;
}
LM_FN float lm_tanhf(float x)
{
    const float a = fabsf(x);
    float t;
    if (a < 0.625f) {   /* Cephes tanhf small-argument polynomial */
        const float z = a * a;
        t = ((((-5.70498872745e-3f * z + 2.06390887954e-2f) * z - 5.37397155531e-2f) * z + 1.33314422036e-1f) * z - 3.33332819422e-1f) * z * a + a;
    } else {
        t = 1.0f - 2.0f / (lm_expf(2.0f * a) + 1.0f);
    }
    return x < 0.0f ? -t : t;
}
/* ORBIT utils.math.quat_apply (w, x, y, z):  v + 2 w (q_v x v) + 2 q_v x (q_v x v) */
LM_FN void lm_quat_apply(const float *q, const float *v, float *o)
{
    float t[3], u[3];
    lm_cross(q + 1, v, t);
    t[0] *= 2.0f; t[1] *= 2.0f; t[2] *= 2.0f;
    lm_cross(q + 1, t, u);
    o[0] = v[0] + q[0] * t[0] + u[0];
    o[1] = v[1] + q[0] * t[1] + u[1];
    o[2] = v[2] + q[0] * t[2] + u[2];
}
LM_FN void lm_quat_to_mat(const float *q, float R[3][3])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z); R[0][1] = 2.0f * (x * y - w * z); R[0][2] = 2.0f * (x * z + w * y);
    R[1][0] = 2.0f * (x * y + w * z); R[1][1] = 1.0f - 2.0f * (x * x + z * z); R[1][2] = 2.0f * (y * z - w * x);
    R[2][0] = 2.0f * (x * z - w * y); R[2][1] = 2.0f * (y * z + w * x); R[2][2] = 1.0f - 2.0f * (x * x + y * y);
}
LM_FN void lm_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
LM_FN float lm_u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

LM_FN void lift_default_config(lift_config *c)
{
    c->sim_dt = 0.01f; c->decimation = 2; c->max_episode_length = 250; c->max_episode_length_s = 5.0f;
    c->action_scale = 0.5f; c->finger_open = 0.04f; c->finger_close = 0.0f;
    const float w[LIFT_NUM_REW] = {1.0f, 15.0f, 16.0f, 5.0f, 1.0e-3f, 1.0e-4f};
    for (int i = 0; i < LIFT_NUM_REW; ++i) c->rew_weight[i] = w[i];
    c->reach_std = 0.1f; c->goal_std = 0.3f; c->goal_fine_std = 0.05f; c->minimal_height = 0.06f;
    c->drop_height = -0.05f;
    c->cmd_lo[0] = 0.3f; c->cmd_hi[0] = 0.7f; c->cmd_lo[1] = 0.3f; c->cmd_hi[1] = 0.7f; c->cmd_lo[2] = 0.0f; c->cmd_hi[2] = 0.0f;
    c->cmd_resample_time = 5.0f;
    c->obj_init[0] = 0.5f; c->obj_init[1] = 0.0f; c->obj_init[2] = 0.055f;
    c->obj_range_lo[0] = -0.1f; c->obj_range_hi[0] = 0.1f; c->obj_range_lo[1] = -0.25f; c->obj_range_hi[1] = 0.25f;
    c->obj_range_lo[2] = 0.0f; c->obj_range_hi[2] = 0.0f;
    c->ee_offset_z = 0.1034f;
    c->seed_lo = 0u; c->seed_hi = 0u;
    c->solver_iterations = 8;
    c->mu_table = 0.6f; c->mu_pad = 0.9f;
}

/* ---------------------------------------------------------------------------------------------------- arm dynamics */
typedef struct {
    float R[7][3][3];   /* rotation link i -> parent (i - 1) */
    float p[7][3];      /* origin of link i in parent coordinates */
} lm_chain;

LM_FN void lm_chain_build(const float *q, lm_chain *ch)
{
    const float A[7] = LM_DH_A, D[7] = LM_DH_D, SA[7] = LM_DH_SA, CA[7] = LM_DH_CA;
    for (int i = 0; i < 7; ++i) {
        float s, c;
        lm_sincosf(q[i], &s, &c);
        /* Rx(alpha) Rz(theta) */
        ch->R[i][0][0] = c;          ch->R[i][0][1] = -s;         ch->R[i][0][2] = 0.0f;
        ch->R[i][1][0] = s * CA[i];  ch->R[i][1][1] = c * CA[i];  ch->R[i][1][2] = -SA[i];
        ch->R[i][2][0] = s * SA[i];  ch->R[i][2][1] = c * SA[i];  ch->R[i][2][2] = CA[i];
        ch->p[i][0] = A[i];
        ch->p[i][1] = -SA[i] * D[i];
        ch->p[i][2] = CA[i] * D[i];
    }
}
LM_FN void lm_rt_mul(const float R[3][3], const float *v, float *o) /* o = R^T v */
{
    for (int i = 0; i < 3; ++i) o[i] = R[0][i] * v[0] + R[1][i] * v[1] + R[2][i] * v[2];
}
LM_FN void lm_r_mul(const float R[3][3], const float *v, float *o) /* o = R v */
{
    for (int i = 0; i < 3; ++i) o[i] = R[i][0] * v[0] + R[i][1] * v[1] + R[i][2] * v[2];
}
/* recursive Newton-Euler: tau = M(q) qdd + c(q, qd) [+ g(q) when gravity != 0] */
LM_FN void lm_rne(const lm_chain *ch, const float *qd, const float *qdd, float gravity, float *tau)
{
    const float M[7] = LM_LINK_M, C[7][3] = LM_LINK_C, I[7][3] = LM_LINK_I;
    float w[7][3], wd[7][3], F[7][3], N[7][3];
    float wp[3] = {0.0f, 0.0f, 0.0f}, wdp[3] = {0.0f, 0.0f, 0.0f}, ap[3] = {0.0f, 0.0f, gravity};
    for (int i = 0; i < 7; ++i) {
        float rw[3], rwd[3], t[3], t2[3], acc[3], a[3];
        lm_rt_mul(ch->R[i], wp, rw);
        w[i][0] = rw[0]; w[i][1] = rw[1]; w[i][2] = rw[2] + qd[i];
        lm_rt_mul(ch->R[i], wdp, rwd);
        /* rw x (qd z) = (rw.y qd, -rw.x qd, 0) */
        wd[i][0] = rwd[0] + rw[1] * qd[i];
        wd[i][1] = rwd[1] - rw[0] * qd[i];
        wd[i][2] = rwd[2] + qdd[i];
        lm_cross(wdp, ch->p[i], t);
        lm_cross(wp, ch->p[i], t2);
        lm_cross(wp, t2, acc);
        acc[0] += ap[0] + t[0]; acc[1] += ap[1] + t[1]; acc[2] += ap[2] + t[2];
        lm_rt_mul(ch->R[i], acc, a);
        /* centre of mass acceleration */
        lm_cross(wd[i], C[i], t);
        lm_cross(w[i], C[i], t2);
        lm_cross(w[i], t2, acc);
        for (int k = 0; k < 3; ++k) F[i][k] = M[i] * (a[k] + t[k] + acc[k]);
        const float Iw[3] = {I[i][0] * w[i][0], I[i][1] * w[i][1], I[i][2] * w[i][2]};
        lm_cross(w[i], Iw, t);
        for (int k = 0; k < 3; ++k) N[i][k] = I[i][k] * wd[i][k] + t[k];
        for (int k = 0; k < 3; ++k) { wp[k] = w[i][k]; wdp[k] = wd[i][k]; ap[k] = a[k]; }
    }
    float f[3] = {0.0f, 0.0f, 0.0f}, n[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 6; i >= 0; --i) {
        float fi[3], ni[3], t[3];
        if (i < 6) {
            float rf[3], rn[3];
            lm_r_mul(ch->R[i + 1], f, rf);
            lm_r_mul(ch->R[i + 1], n, rn);
            lm_cross(ch->p[i + 1], rf, t);
            for (int k = 0; k < 3; ++k) { fi[k] = rf[k] + F[i][k]; ni[k] = N[i][k] + rn[k] + t[k]; }
        } else {
            for (int k = 0; k < 3; ++k) { fi[k] = F[i][k]; ni[k] = N[i][k]; }
        }
        lm_cross(C[i], F[i], t);
        for (int k = 0; k < 3; ++k) { ni[k] += t[k]; f[k] = fi[k]; n[k] = ni[k]; }
        tau[i] = n[2];
    }
}
/* joint-space mass matrix, column by column: column j = the inverse dynamics of a unit acceleration of joint j at rest and
 * without gravity.  That is lm_rne(ch, 0, e_j, 0) with everything that is identically zero left out -- no angular velocity,
 * so no centripetal / gyroscopic terms; links below j do not move; rows above j follow from symmetry -- about a third of
 * the arithmetic of seven full passes. */
LM_FN void lm_mass_matrix(const lm_chain *ch, float Mm[7][7])
{
    const float M[7] = LM_LINK_M, C[7][3] = LM_LINK_C, I[7][3] = LM_LINK_I;
    for (int j = 0; j < 7; ++j) {
        float F[7][3], N[7][3];
        float wd[3] = {0.0f, 0.0f, 1.0f}, a[3] = {0.0f, 0.0f, 0.0f};
        for (int i = j; i < 7; ++i) {
            float t[3];
            if (i > j) {
                float wdn[3], acc[3];
                lm_cross(wd, ch->p[i], t);
                for (int k = 0; k < 3; ++k) acc[k] = a[k] + t[k];
                lm_rt_mul(ch->R[i], acc, a);
                lm_rt_mul(ch->R[i], wd, wdn);
                for (int k = 0; k < 3; ++k) wd[k] = wdn[k];
            }
            lm_cross(wd, C[i], t);
            for (int k = 0; k < 3; ++k) { F[i][k] = M[i] * (a[k] + t[k]); N[i][k] = I[i][k] * wd[k]; }
        }
        float f[3] = {0.0f, 0.0f, 0.0f}, n[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 6; i >= j; --i) {
            float fi[3], ni[3], t[3];
            if (i < 6) {
                float rf[3], rn[3];
                lm_r_mul(ch->R[i + 1], f, rf);
                lm_r_mul(ch->R[i + 1], n, rn);
                lm_cross(ch->p[i + 1], rf, t);
                for (int k = 0; k < 3; ++k) { fi[k] = rf[k] + F[i][k]; ni[k] = N[i][k] + rn[k] + t[k]; }
            } else {
                for (int k = 0; k < 3; ++k) { fi[k] = F[i][k]; ni[k] = N[i][k]; }
            }
            lm_cross(C[i], F[i], t);
            for (int k = 0; k < 3; ++k) { ni[k] += t[k]; f[k] = fi[k]; n[k] = ni[k]; }
            Mm[i][j] = n[2];
            Mm[j][i] = n[2];
        }
    }
}
/* in-place Cholesky solve of the SPD system A x = b (lower triangle of A is overwritten) */
LM_FN void lm_chol_solve7(float A[7][7], float *b)
{
    for (int j = 0; j < 7; ++j) {
        float d = A[j][j];
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
        d = sqrtf(d > 1.0e-9f ? d : 1.0e-9f);
        A[j][j] = d;
        const float inv = 1.0f / d;
        for (int i = j + 1; i < 7; ++i) {
            float s = A[i][j];
            for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
            A[i][j] = s * inv;
        }
    }
    for (int i = 0; i < 7; ++i) {
        float s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i][k] * b[k];
        b[i] = s / A[i][i];
    }
    for (int i = 6; i >= 0; --i) {
        float s = b[i];
        for (int k = i + 1; k < 7; ++k) s -= A[k][i] * b[k];
        b[i] = s / A[i][i];
    }
}
/* forward kinematics of the hand: rotation Rh (hand -> world), tool centre point, linear / angular velocity of the TCP */
typedef struct {
    float R[3][3];
    float tcp[3];
    float v[3], w[3];
} lm_hand;
LM_FN void lm_hand_fk(const lm_chain *ch, const float *qd, float ee_offset_z, lm_hand *h)
{
    float R0[3][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};
    float p0[3] = {0.0f, 0.0f, 0.0f}, v0[3] = {0.0f, 0.0f, 0.0f}, w0[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < 7; ++i) {
        float pw[3], t[3], Rn[3][3];
        lm_r_mul(R0, ch->p[i], pw);
        lm_cross(w0, pw, t);
        for (int k = 0; k < 3; ++k) { p0[k] += pw[k]; v0[k] += t[k]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rn[r][c] = R0[r][0] * ch->R[i][0][c] + R0[r][1] * ch->R[i][1][c] + R0[r][2] * ch->R[i][2][c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R0[r][c] = Rn[r][c];
        for (int k = 0; k < 3; ++k) w0[k] += R0[k][2] * qd[i];   /* joint axis = local z */
    }
    /* flange + hand frame (rotated -45 deg about z) + tool offset along the hand z axis */
    const float cz = 0.70710678118654752f;
    for (int r = 0; r < 3; ++r) {
        h->R[r][0] = (R0[r][0] - R0[r][1]) * cz;
        h->R[r][1] = (R0[r][0] + R0[r][1]) * cz;
        h->R[r][2] = R0[r][2];
    }
    const float off = LM_FLANGE_D + ee_offset_z;
    float rel[3], t[3];
    for (int k = 0; k < 3; ++k) { rel[k] = R0[k][2] * off; h->tcp[k] = p0[k] + rel[k]; h->w[k] = w0[k]; }
    lm_cross(w0, rel, t);
    for (int k = 0; k < 3; ++k) h->v[k] = v0[k] + t[k];
}

/* one physics substep of the arm (7 joints): implicit PD on the full joint-space dynamics */
LM_FN void lm_arm_substep(float h, const float *target, float *q, float *qd, lm_chain *ch)
{
    const float LO[7] = LM_Q_LO, HI[7] = LM_Q_HI, VL[7] = LM_QD_LIM, EF[7] = LM_EFFORT;
    float Mm[7][7], c[7], zero[7] = {0, 0, 0, 0, 0, 0, 0};
    lm_chain_build(q, ch);
    lm_rne(ch, qd, zero, LM_G, c);
    lm_mass_matrix(ch, Mm);
    for (int i = 0; i < 7; ++i) Mm[i][i] += LM_ARMATURE;
    float A[7][7], b[7], v[7];
    int sat[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 7; ++i) {
            float mv = 0.0f;
            for (int j = 0; j < 7; ++j) { A[i][j] = Mm[i][j]; mv += Mm[i][j] * qd[j]; }
            if (sat[i] == 0) {
                A[i][i] += h * LM_ARM_KD + h * h * LM_ARM_KP;
                b[i] = mv + h * (LM_ARM_KP * (target[i] - q[i]) - c[i]);
            } else {
                b[i] = mv + h * ((float)sat[i] * EF[i] - c[i]);
            }
        }
        lm_chol_solve7(A, b);
        for (int i = 0; i < 7; ++i) v[i] = b[i];
        if (pass == 1) break;
        int any = 0;
        for (int i = 0; i < 7; ++i) {
            const float tau = LM_ARM_KP * (target[i] - (q[i] + h * v[i])) - LM_ARM_KD * v[i];
            if (tau > EF[i]) { sat[i] = 1; any = 1; }
            if (tau < -EF[i]) { sat[i] = -1; any = 1; }
        }
        if (!any) break;
    }
    for (int i = 0; i < 7; ++i) {
        float vi = lm_clampf(v[i], -VL[i], VL[i]);
        float x = q[i] + h * vi;
        if (x > HI[i]) { x = HI[i]; vi = 0.0f; }
        if (x < LO[i]) { x = LO[i]; vi = 0.0f; }
        q[i] = x;
        qd[i] = vi;
    }
}

/* ---------------------------------------------------------------------------------------------------- cube contact */
/* Contact rows live in FIXED slots -- 8 cube corners against the table (normal + 2 friction rows each) and 2 finger pads
 * (normal, 2 friction rows, torsion) -- visited in that order by every Gauss-Seidel sweep and skipped when inactive.  Static
 * slots keep the solver in registers on the GPU; a dynamically indexed row list lives in scratch memory, and that cost 2/3
 * of the step kernel's time. */
typedef struct {
    int active;
    float r[3];          /* corner relative to the cube centre (world axes)          */
    float meff[3];       /* rows: normal (z), friction x, friction y                 */
    float target;        /* normal row: stabilisation / speculative-contact velocity */
    float lam[3];
} lm_corner;
typedef struct {
    int active;
    float n[3][3];       /* row directions: normal, pad axis (x), approach axis (z)  */
    float rxn[4][3];     /* r x n of the three rows; row 3 (torsion) = closing axis  */
    float meff[4], target[4], lam[4];
} lm_pad;
#if defined(__HIPCC__)
#define LM_UNROLL _Pragma("unroll")
#else
#define LM_UNROLL _Pragma("GCC unroll 8")
#endif

/* one physics substep of cube + fingers.  obj = S + LIFT_OBJ_POS (pos 3, quat 4, lin 3, ang 3 contiguous). */
LM_FN void lm_cube_substep(const lift_config *cfg, float h, const lm_hand *hand, const float *finger_target, float *fq, float *fqd,
                           float *obj)
{
    float *pos = obj, *quat = obj + 3, *lin = obj + 7, *ang = obj + 10;
    const float inv_m = 1.0f / LM_CUBE_MASS;
    const float inv_I = 1.0f / (LM_CUBE_MASS * (2.0f * LM_CUBE_HALF) * (2.0f * LM_CUBE_HALF) / 6.0f);
    /* fingers: implicit PD folded into an effective mass and a free velocity */
    float fm_inv[2], fv[2];
    for (int k = 0; k < 2; ++k) {
        const float meff = LM_FINGER_MASS + h * LM_FINGER_KD + h * h * LM_FINGER_KP;
        fm_inv[k] = 1.0f / meff;
        float force = LM_FINGER_KP * (finger_target[k] - fq[k]);
        force = lm_clampf(force, -LM_FINGER_EFFORT, LM_FINGER_EFFORT);
        fv[k] = (LM_FINGER_MASS * fqd[k] + h * force) * fm_inv[k];
    }
    lin[2] -= LM_G * h;
    float Rc[3][3];
    lm_quat_to_mat(quat, Rc);
    /* ---- table: the cube's corners against the plane z = 0.  With n a unit axis the row's r x n has a closed form:
     * z: (r1, -r0, 0), x: (0, r2, -r1), y: (-r2, 0, r0) */
    lm_corner cr[8];
    LM_UNROLL
    for (int c = 0; c < 8; ++c) {
        const float lc[3] = {(c & 1) ? LM_CUBE_HALF : -LM_CUBE_HALF, (c & 2) ? LM_CUBE_HALF : -LM_CUBE_HALF,
                             (c & 4) ? LM_CUBE_HALF : -LM_CUBE_HALF};
        lm_r_mul(Rc, lc, cr[c].r);
        const float *r = cr[c].r;
        const float z = pos[2] + r[2];
        cr[c].active = z < 0.004f;
        cr[c].meff[0] = 1.0f / (inv_m + inv_I * (r[1] * r[1] + r[0] * r[0]));
        cr[c].meff[1] = 1.0f / (inv_m + inv_I * (r[2] * r[2] + r[1] * r[1]));
        cr[c].meff[2] = 1.0f / (inv_m + inv_I * (r[2] * r[2] + r[0] * r[0]));
        const float push = LM_BAUMGARTE * (-z) / h;
        cr[c].target = z < 0.0f ? (push < 1.0f ? push : 1.0f) : -z / h;
        cr[c].lam[0] = 0.0f; cr[c].lam[1] = 0.0f; cr[c].lam[2] = 0.0f;
    }
    /* ---- finger pads.  Hand frame: x along the pads, y = closing axis, z = approach axis; finger k sits at y = +-fq[k] */
    lm_pad pd[2];
    {
        float d[3], ch_[3];
        for (int k = 0; k < 3; ++k) d[k] = pos[k] - hand->tcp[k];
        lm_rt_mul(hand->R, d, ch_);
        /* support of the cube along the hand y axis */
        float ey = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float p = hand->R[0][1] * Rc[0][a] + hand->R[1][1] * Rc[1][a] + hand->R[2][1] * Rc[2][a];
            ey += fabsf(p) * LM_CUBE_HALF;
        }
        const int between = fabsf(ch_[0]) < LM_CUBE_HALF + LM_PAD_X && fabsf(ch_[2]) < LM_CUBE_HALF + LM_PAD_Z;
        float yh[3], xh[3], zh[3];
        for (int i = 0; i < 3; ++i) { xh[i] = hand->R[i][0]; yh[i] = hand->R[i][1]; zh[i] = hand->R[i][2]; }
        /* contact point: on the pad plane, under the cube centre (clamped to the pad) */
        const float px = lm_clampf(ch_[0], -LM_PAD_X, LM_PAD_X), pz = lm_clampf(ch_[2], -LM_PAD_Z, LM_PAD_Z);
        LM_UNROLL
        for (int k = 0; k < 2; ++k) {
            const float sgn = k == 0 ? 1.0f : -1.0f;             /* finger 0 at +y pushes the cube towards -y */
            const float gap = fq[k] - (sgn * ch_[1] + ey);
            lm_pad *P = &pd[k];
            P->active = between && gap < 0.002f;
            float cp[3], r[3], vpad[3], t[3];
            for (int i = 0; i < 3; ++i) {
                cp[i] = hand->tcp[i] + xh[i] * px + yh[i] * (sgn * fq[k]) + zh[i] * pz;
                r[i] = cp[i] - pos[i];
                d[i] = cp[i] - hand->tcp[i];
            }
            lm_cross(hand->w, d, t);
            for (int i = 0; i < 3; ++i) vpad[i] = hand->v[i] + t[i];
            for (int i = 0; i < 3; ++i) { P->n[0][i] = -sgn * yh[i]; P->n[1][i] = xh[i]; P->n[2][i] = zh[i]; }
            for (int q = 0; q < 3; ++q) lm_cross(r, P->n[q], P->rxn[q]);
            for (int i = 0; i < 3; ++i) P->rxn[3][i] = yh[i];    /* torsional friction about the closing axis (pure couple) */
            P->meff[0] = 1.0f / (inv_m + inv_I * lm_dot(P->rxn[0], P->rxn[0]) + fm_inv[k]);
            P->meff[1] = 1.0f / (inv_m + inv_I * lm_dot(P->rxn[1], P->rxn[1]));
            P->meff[2] = 1.0f / (inv_m + inv_I * lm_dot(P->rxn[2], P->rxn[2]));
            P->meff[3] = 1.0f / inv_I;
            const float push = LM_BAUMGARTE * (-gap) / h;
            P->target[0] = lm_dot(vpad, P->n[0]) + (gap < 0.0f ? (push < 0.5f ? push : 0.5f) : -gap / h);
            P->target[1] = lm_dot(vpad, P->n[1]);
            P->target[2] = lm_dot(vpad, P->n[2]);
            P->target[3] = lm_dot(hand->w, yh);
            for (int q = 0; q < 4; ++q) P->lam[q] = 0.0f;
        }
    }
    /* ---- projected Gauss-Seidel on the velocities: corners 0..7 (normal, x, y), then pads 0, 1 (normal, x, z, torsion) */
    for (int it = 0; it < cfg->solver_iterations; ++it) {
        LM_UNROLL
        for (int c = 0; c < 8; ++c) {
            if (!cr[c].active) continue;
            const float r0 = cr[c].r[0], r1 = cr[c].r[1], r2 = cr[c].r[2];
            {   /* normal */
                const float u = lin[2] + (r1 * ang[0] - r0 * ang[1]);
                float lam = cr[c].lam[0] + (cr[c].target - u) * cr[c].meff[0];
                if (lam < 0.0f) lam = 0.0f;
                const float dl = lam - cr[c].lam[0];
                cr[c].lam[0] = lam;
                lin[2] += dl * inv_m;
                ang[0] += r1 * dl * inv_I; ang[1] -= r0 * dl * inv_I;
            }
            const float lim = cfg->mu_table * cr[c].lam[0];
            {   /* friction along x */
                const float u = lin[0] + (r2 * ang[1] - r1 * ang[2]);
                const float lam = lm_clampf(cr[c].lam[1] + (0.0f - u) * cr[c].meff[1], -lim, lim);
                const float dl = lam - cr[c].lam[1];
                cr[c].lam[1] = lam;
                lin[0] += dl * inv_m;
                ang[1] += r2 * dl * inv_I; ang[2] -= r1 * dl * inv_I;
            }
            {   /* friction along y */
                const float u = lin[1] + (r0 * ang[2] - r2 * ang[0]);
                const float lam = lm_clampf(cr[c].lam[2] + (0.0f - u) * cr[c].meff[2], -lim, lim);
                const float dl = lam - cr[c].lam[2];
                cr[c].lam[2] = lam;
                lin[1] += dl * inv_m;
                ang[2] += r0 * dl * inv_I; ang[0] -= r2 * dl * inv_I;
            }
        }
        LM_UNROLL
        for (int k = 0; k < 2; ++k) {
            lm_pad *P = &pd[k];
            if (!P->active) continue;
            LM_UNROLL
            for (int q = 0; q < 4; ++q) {
                float u = lm_dot(P->rxn[q], ang);
                if (q < 3) u += lm_dot(P->n[q], lin);
                if (q == 0) u += fv[k];                         /* the finger closes along the row direction */
                float lam = P->lam[q] + (P->target[q] - u) * P->meff[q];
                if (q == 0) {
                    if (lam < 0.0f) lam = 0.0f;
                } else {
                    const float lim = (q == 3 ? cfg->mu_pad * LM_TORSION_R : cfg->mu_pad) * P->lam[0];
                    lam = lm_clampf(lam, -lim, lim);
                }
                const float dl = lam - P->lam[q];
                P->lam[q] = lam;
                if (q < 3)
                    for (int i = 0; i < 3; ++i) lin[i] += P->n[q][i] * dl * inv_m;
                for (int i = 0; i < 3; ++i) ang[i] += P->rxn[q][i] * dl * inv_I;
                if (q == 0) fv[k] += dl * fm_inv[k];            /* the reaction opens the finger */
            }
        }
    }
    /* ---- integrate */
    for (int k = 0; k < 2; ++k) {
        float v = lm_clampf(fv[k], -LM_FINGER_VLIM, LM_FINGER_VLIM);
        float x = fq[k] + h * v;
        if (x > 0.04f) { x = 0.04f; v = 0.0f; }
        if (x < 0.0f) { x = 0.0f; v = 0.0f; }
        fq[k] = x;
        fqd[k] = v;
    }
    for (int k = 0; k < 3; ++k) pos[k] += h * lin[k];
    {
        const float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3], hh = 0.5f * h;
        const float nw = qw + hh * (-ang[0] * qx - ang[1] * qy - ang[2] * qz);
        const float nx = qx + hh * (ang[0] * qw + ang[1] * qz - ang[2] * qy);
        const float ny = qy + hh * (ang[1] * qw + ang[2] * qx - ang[0] * qz);
        const float nz = qz + hh * (ang[2] * qw + ang[0] * qy - ang[1] * qx);
        const float inv = 1.0f / sqrtf(nw * nw + nx * nx + ny * ny + nz * nz);
        quat[0] = nw * inv; quat[1] = nx * inv; quat[2] = ny * inv; quat[3] = nz * inv;
    }
}

/* ---------------------------------------------------------------------------------------------------- MDP terms */
/* rewards.py:20-67 + ORBIT mdp.action_rate_l2 / joint_vel_l2; terminations: mdp.time_out, mdp.base_height.
 * root = robot root state (pos 3, quat 4): the task keeps it at the origin with identity orientation. */
LM_FN void lift_terms_one(const lift_config *c, const float *obj_pos, const float *ee_pos, const float *root_pos, const float *root_quat,
                          const float *cmd_pos_b, const float *action, const float *prev_action, const float *qd9, int32_t ep_len,
                          float *rew, uint8_t *term, float *obj_pos_b)
{
    /* object_ee_distance :29-46 */
    const float dx = obj_pos[0] - ee_pos[0], dy = obj_pos[1] - ee_pos[1], dz = obj_pos[2] - ee_pos[2];
    const float dist_ee = sqrtf(dx * dx + dy * dy + dz * dz);
    rew[0] = 1.0f - lm_tanhf(dist_ee / c->reach_std);
    /* object_is_lifted :20-26 */
    const int lifted = obj_pos[2] > c->minimal_height;
    rew[1] = lifted ? 1.0f : 0.0f;
    /* object_goal_distance :49-67: des_pos_w = root_pos + quat_apply(root_quat, des_pos_b) */
    float des_w[3];
    lm_quat_apply(root_quat, cmd_pos_b, des_w);
    des_w[0] += root_pos[0]; des_w[1] += root_pos[1]; des_w[2] += root_pos[2];
    const float gx = des_w[0] - obj_pos[0], gy = des_w[1] - obj_pos[1], gz = des_w[2] - obj_pos[2];
    const float dist_goal = sqrtf(gx * gx + gy * gy + gz * gz);
    rew[2] = (lifted ? 1.0f : 0.0f) * (1.0f - lm_tanhf(dist_goal / c->goal_std));
    rew[3] = (lifted ? 1.0f : 0.0f) * (1.0f - lm_tanhf(dist_goal / c->goal_fine_std));
    /* ORBIT mdp.action_rate_l2: sum((action - prev_action)^2); mdp.joint_vel_l2: sum(joint_vel^2) */
    float ar = 0.0f, jv = 0.0f;
    for (int i = 0; i < LIFT_ACT; ++i) { const float d = action[i] - prev_action[i]; ar += d * d; }
    for (int i = 0; i < 9; ++i) jv += qd9[i] * qd9[i];
    rew[4] = ar;
    rew[5] = jv;
    term[0] = ep_len >= c->max_episode_length;          /* mdp.time_out */
    term[1] = obj_pos[2] < c->drop_height;              /* mdp.base_height(minimum_height = -0.05) */
    /* observations.py:19-31: subtract_frame_transforms(root_pos, root_quat, object_pos) */
    const float qi[4] = {root_quat[0], -root_quat[1], -root_quat[2], -root_quat[3]};
    const float rel[3] = {obj_pos[0] - root_pos[0], obj_pos[1] - root_pos[1], obj_pos[2] - root_pos[2]};
    lm_quat_apply(qi, rel, obj_pos_b);
}

/* reset_scene_to_default + reset_root_state_uniform (object) + manager resets + UniformPoseCommand resample */
LM_FN void lift_resample_command(const lift_config *c, float *S, uint32_t gid, uint32_t count, uint32_t stream)
{
    uint32_t r[4];
    lm_philox(gid, count, 1u, stream, c->seed_lo, c->seed_hi, r);
    for (int k = 0; k < 3; ++k) S[LIFT_CMD + k] = lm_u01(r[k]) * (c->cmd_hi[k] - c->cmd_lo[k]) + c->cmd_lo[k];
    S[LIFT_CMD + 3] = 1.0f; S[LIFT_CMD + 4] = 0.0f; S[LIFT_CMD + 5] = 0.0f; S[LIFT_CMD + 6] = 0.0f;   /* euler (0, 0, 0) */
    S[LIFT_TIME_LEFT] = c->cmd_resample_time;
}
LM_FN void lift_reset_one(const lift_config *c, float *S, uint32_t gid)
{
    const float QDEF[9] = LM_Q_DEFAULT;
    union { float f; uint32_t u; } cnt;
    cnt.f = S[LIFT_RESET_COUNT];
    const uint32_t count = cnt.u;
    uint32_t r[4];
    lm_philox(gid, count, 0u, 0u, c->seed_lo, c->seed_hi, r);
    for (int i = 0; i < 9; ++i) { S[LIFT_Q + i] = QDEF[i]; S[LIFT_QD + i] = 0.0f; }
    for (int k = 0; k < 3; ++k)
        S[LIFT_OBJ_POS + k] = c->obj_init[k] + (lm_u01(r[k]) * (c->obj_range_hi[k] - c->obj_range_lo[k]) + c->obj_range_lo[k]);
    S[LIFT_OBJ_QUAT] = 1.0f; S[LIFT_OBJ_QUAT + 1] = 0.0f; S[LIFT_OBJ_QUAT + 2] = 0.0f; S[LIFT_OBJ_QUAT + 3] = 0.0f;
    for (int k = 0; k < 6; ++k) S[LIFT_OBJ_LIN + k] = 0.0f;
    for (int i = 0; i < LIFT_ACT; ++i) { S[LIFT_ACTION + i] = 0.0f; S[LIFT_PREV_ACTION + i] = 0.0f; }
    for (int i = 0; i < LIFT_NUM_REW; ++i) S[LIFT_EP_SUM + i] = 0.0f;
    lift_resample_command(c, S, gid, count, 0u);
    union { float f; int32_t i; } z;
    z.i = 0;
    S[LIFT_EP_LEN] = z.f;
    cnt.u = count + 1u;
    S[LIFT_RESET_COUNT] = cnt.f;
}
/* ObservationCfg.PolicyCfg, manipulation_env_cfg.py:105-110: [joint_pos_rel 9, joint_vel_rel 9, object pos in root frame 3,
 * command 7, last_action 8] */
LM_FN void lift_write_obs(const lift_config *c, const float *S, float *obs)
{
    const float QDEF[9] = LM_Q_DEFAULT;
    (void)c;
    for (int i = 0; i < 9; ++i) { obs[i] = S[LIFT_Q + i] - QDEF[i]; obs[9 + i] = S[LIFT_QD + i]; }
    const float root_pos[3] = {0.0f, 0.0f, 0.0f}, qi[4] = {1.0f, -0.0f, -0.0f, -0.0f};
    const float rel[3] = {S[LIFT_OBJ_POS] - root_pos[0], S[LIFT_OBJ_POS + 1] - root_pos[1], S[LIFT_OBJ_POS + 2] - root_pos[2]};
    lm_quat_apply(qi, rel, obs + 18);
    for (int i = 0; i < 7; ++i) obs[21 + i] = S[LIFT_CMD + i];
    for (int i = 0; i < LIFT_ACT; ++i) obs[28 + i] = S[LIFT_ACTION + i];
}

/* RLTaskEnv.step of FrankaCubeLift-v0 for ONE env (ORBIT ordering: action -> decimation x physics -> counters ->
 * terminations -> rewards -> reset -> command -> observations).  lg: 10 log contributions (6 episodic sums, 2 termination
 * flags, 1 reset flag, 1 pad) valid when the env resets. */
LM_FN void lift_step_one(const lift_config *c, float *S, const float *action, uint32_t gid, float *obs, float *reward,
                         uint8_t *terminated, uint8_t *truncated, float *lg)
{
    const float QDEF[9] = LM_Q_DEFAULT;
    /* ActionManager.process_action */
    for (int i = 0; i < LIFT_ACT; ++i) { S[LIFT_PREV_ACTION + i] = S[LIFT_ACTION + i]; S[LIFT_ACTION + i] = action[i]; }
    float target[7], ftarget[2];
    for (int i = 0; i < 7; ++i) target[i] = QDEF[i] + c->action_scale * action[i];      /* JointPositionAction */
    const float g = action[7] < 0.0f ? c->finger_close : c->finger_open;                /* BinaryJointPositionAction */
    ftarget[0] = g; ftarget[1] = g;
    lm_chain ch;
    lm_hand hand;
    for (int s = 0; s < c->decimation; ++s) {
#ifndef LM_ABL_NO_ARM   /* ablation builds (tools/build_diag.py) time the halves of a substep; never defined in the product */
        lm_arm_substep(c->sim_dt, target, S + LIFT_Q, S + LIFT_QD, &ch);
#endif
        lm_chain_build(S + LIFT_Q, &ch);
        lm_hand_fk(&ch, S + LIFT_QD, c->ee_offset_z, &hand);
#ifndef LM_ABL_NO_CUBE
        lm_cube_substep(c, c->sim_dt, &hand, ftarget, S + LIFT_Q + 7, S + LIFT_QD + 7, S + LIFT_OBJ_POS);
#endif
    }
    if (c->decimation <= 0) {
        lm_chain_build(S + LIFT_Q, &ch);
        lm_hand_fk(&ch, S + LIFT_QD, c->ee_offset_z, &hand);
    }
    union { float f; int32_t i; } el;
    el.f = S[LIFT_EP_LEN];
    el.i += 1;
    S[LIFT_EP_LEN] = el.f;
    float rew[LIFT_NUM_REW], obj_b[3];
    uint8_t term[LIFT_NUM_TERM];
    const float root_pos[3] = {0.0f, 0.0f, 0.0f}, root_quat[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    lift_terms_one(c, S + LIFT_OBJ_POS, hand.tcp, root_pos, root_quat, S + LIFT_CMD, S + LIFT_ACTION, S + LIFT_PREV_ACTION, S + LIFT_QD,
                   el.i, rew, term, obj_b);
    const float step_dt = c->sim_dt * (float)c->decimation;
    float total = 0.0f;
    for (int i = 0; i < LIFT_NUM_REW; ++i) {
        if (c->rew_weight[i] != 0.0f) {
            const float val = rew[i] * c->rew_weight[i] * step_dt;
            total += val;
            S[LIFT_EP_SUM + i] += val;
        }
    }
    *reward = total;
    *truncated = term[0];
    *terminated = term[1];
    const int do_reset = term[0] | term[1];
    for (int i = 0; i < 10; ++i) lg[i] = 0.0f;
    if (do_reset) {
        for (int i = 0; i < LIFT_NUM_REW; ++i) lg[i] = S[LIFT_EP_SUM + i];
        lg[6] = (float)term[0];
        lg[7] = (float)term[1];
        lg[8] = 1.0f;
        lift_reset_one(c, S, gid);
    }
    /* CommandTerm.compute: timer -> resample (5 s = the episode length: fires together with the time-out reset) */
    S[LIFT_TIME_LEFT] -= step_dt;
    if (S[LIFT_TIME_LEFT] <= 0.0f) {
        union { float f; uint32_t u; } cnt;
        cnt.f = S[LIFT_RESET_COUNT];
        lift_resample_command(c, S, gid, cnt.u, 1u);
    }
    lift_write_obs(c, S, obs);
}

#endif /* LIFT_MODEL_H */
