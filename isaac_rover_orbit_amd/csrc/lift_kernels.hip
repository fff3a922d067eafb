// lift_kernels.hip -- FrankaCubeLift-v0 step()/reset() on gfx950 (BASELINE config 5) + its C ABI (include/rover_lift.h).
//
// Reference behaviour being replaced (file:line in /root/reference):
//   rover_envs/envs/manipulation/manipulation_env_cfg.py:93-235   actions / observations / rewards / terminations / commands /
//                                                                  randomization tables, dt = 0.01 s, decimation 2, 5 s episodes
//   rover_envs/envs/manipulation/config/franka/joint_pos_env_cfg.py:25-82   Franka + 0.8-scale DexCube, joint-position action
//                                                                  (scale 0.5, default offset), binary gripper, ee frame offset
//   rover_envs/envs/manipulation/mdp/rewards.py:20-67, mdp/observations.py:19-31   the task's own term functions
//   ORBIT (third-party, absent): RLTaskEnv.step ordering and the generic mdp terms; PhysX: replaced by the model of docs/history.md
//   section 9 (PARITY UNPINNED).  The CPU oracle (oracle/lift_oracle.c) is a separately written scalar restatement of the same
//   specification; the two agree bit for bit because every floating-point operation order is part of the specification.
//
// Mapping: EIGHT LANES PER ENV (LPE = 8; 16 with the upper half-row shadowing the lower, for small batches).  A 7-DOF arm
// needs 7 mass-matrix columns + 1 bias vector per substep = eight inverse-dynamics passes of the recursive Newton-Euler
// algorithm over the SAME chain with different (qd, qdd, g): one pass per lane, identical instruction stream.  The cube has
// eight corners: one corner's contact rows are set up per lane.  Joint sines / cosines: one joint per lane.  The lanes of
// an env exchange their results through a wave-private LDS slot (write 8 values, read 64: two ds_write_b128 + sixteen
// ds_read_b128), after which the short serial parts (7 x 7 Cholesky, hand kinematics, Gauss-Seidel sweep over the contact
// rows, MDP terms) run replicated in the eight lanes -- a wave is lockstep, so replicating costs nothing that idling
// would not.  At 2048 envs this is 256 waves (one per SIMD on a quarter of the chip): the kernel is bound by the length of
// ONE wave's instruction stream, which the lane split cuts from ~35 k to ~8 k instructions.  No scratch memory: every array
// is fully unrolled into registers (checked: .private_segment_fixed_size 0, tests/test_abi.py).  No MFMA: 7 x 7 systems
// per env, nothing batch-contractible.  All arithmetic fp32, -ffp-contract=off, explicit fmaf.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/rover_hip.h"
#include "../../include/rover_lift.h"
#include "../../include/rover_debug.h"
#include "rover_internal.hpp"

namespace {

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, #expr ": %s", hipGetErrorString(_e));         \
    } while (0)

// ---------------------------------------------------------------------------------------------------- model constants
constexpr float K_GRAV = 9.81f;
constexpr float K_CUBE_HALF = 0.02f;        // 0.8 x DexCube (joint_pos_env_cfg.py:53-54): 4 cm edge
constexpr float K_CUBE_MASS = 0.064f;
constexpr float K_PAD_HALF_X = 0.010f, K_PAD_HALF_Z = 0.009f;
constexpr float K_FINGER_MASS = 0.05f, K_FINGER_KP = 2000.0f, K_FINGER_KD = 100.0f, K_FINGER_EFFORT = 200.0f;   // FRANKA_PANDA_CFG panda_hand
constexpr float K_FINGER_VLIM = 0.2f, K_FINGER_TRAVEL = 0.04f;
constexpr float K_ARM_KP = 80.0f, K_ARM_KD = 4.0f, K_ARMATURE = 0.02f;     // FRANKA_PANDA_CFG panda_shoulder / panda_forearm
constexpr float K_BAUMGARTE = 0.2f, K_TORSION_R = 0.008f, K_FLANGE_D = 0.107f, K_TABLE_MARGIN = 0.004f, K_PAD_MARGIN = 0.002f;

// modified DH (Craig) of the Franka Emika Panda; kind = 0 / +1 / -1 for alpha = 0 / +pi/2 / -pi/2
#define LF_KIND {0, -1, 1, 1, -1, 1, 1}
#define LF_DH_A {0.0f, 0.0f, 0.0f, 0.0825f, -0.0825f, 0.0f, 0.088f}
#define LF_DH_D {0.333f, 0.0f, 0.316f, 0.0f, 0.384f, 0.0f, 0.0f}
#define LF_MASS {4.97f, 0.647f, 3.228f, 3.588f, 1.226f, 1.667f, 1.495f}
#define LF_COM {{0.0039f, 0.0021f, -0.0476f}, {-0.0031f, -0.0287f, 0.0035f}, {0.0275f, 0.0392f, -0.0665f},                \
                {-0.0532f, 0.1044f, 0.0275f}, {-0.0118f, 0.0411f, -0.0384f}, {0.0601f, -0.0141f, -0.0105f},               \
                {0.0054f, -0.0021f, 0.1050f}}
#define LF_INERTIA {{0.70f, 0.71f, 0.0091f}, {0.0080f, 0.0281f, 0.0260f}, {0.0372f, 0.0362f, 0.0108f},                    \
                    {0.0259f, 0.0196f, 0.0283f}, {0.0355f, 0.0295f, 0.0086f}, {0.0020f, 0.0043f, 0.0054f},                 \
                    {0.0260f, 0.0240f, 0.0060f}}
#define LF_Q_LO {-2.8973f, -1.7628f, -2.8973f, -3.0718f, -2.8973f, -0.0175f, -2.8973f}
#define LF_Q_HI {2.8973f, 1.7628f, 2.8973f, -0.0698f, 2.8973f, 3.7525f, 2.8973f}
#define LF_QD_LIM {2.175f, 2.175f, 2.175f, 2.175f, 2.61f, 2.61f, 2.61f}
#define LF_EFFORT {87.0f, 87.0f, 87.0f, 87.0f, 12.0f, 12.0f, 12.0f}
#define LF_Q_DEFAULT {0.0f, -0.569f, 0.0f, -2.810f, 0.0f, 3.037f, 0.741f, 0.04f, 0.04f}   // FRANKA_PANDA_CFG.init_state.joint_pos

#define LF_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------------------------- scalar helpers
LF_DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// Friction box of the contact sweep: clampf(x, -lim, lim) with lim >= +0 as ONE v_med3_f32.  gfx950 orders -0 < +0 in
// min / max / med3 (tools/ubench/semantics_probe.hip); with lo = -lim <= hi = lim that gives the value the compare-and-select
// form above gives for every finite x (the only zero-sign case, lim = +0: x = -0 -> -0, x = +0 -> +0 in both).
LF_DEV float clamp_sym(float x, float lim) { return __builtin_amdgcn_fmed3f(x, -lim, lim); }
// l < 0 ? 0 : l as v_max_f32: equal for every l except -0 (the accumulated impulses are sums ending in "+ lam", never -0)
LF_DEV float nonneg(float l) { return __builtin_fmaxf(l, 0.0f); }

// Cody-Waite reduction by pi/2 + Cephes minimax polynomials
LF_DEV void sincos_poly(float x, float &s_out, float &c_out)
{
    const float k = floorf(x * 0.63661977236758134f + 0.5f);
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188e-8f;
    const int quadrant = ((int)k) & 3;
    const float z = r * r;
    const float ps = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * (-1.9515295891e-4f)));
    const float pc = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const float a = (quadrant & 1) ? pc : ps, b = (quadrant & 1) ? ps : pc;
    s_out = (quadrant & 2) ? -a : a;
    c_out = ((quadrant + 1) & 2) ? -b : b;
}
LF_DEV uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
LF_DEV float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
// 1 / sqrt(x): exponent-halving first guess + three Newton steps
LF_DEV float rsqrt_newton(float x)
{
    float y = u2f(0x5f3759dfu - (f2u(x) >> 1));
    const float hx = 0.5f * x;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const float t = hx * y;
        y = y * fmaf(-t, y, 1.5f);
    }
    return y;
}
LF_DEV float exp_poly(float x)
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
    return p * u2f((uint32_t)(((int32_t)n + 127) << 23));
}
LF_DEV float tanh_poly(float x)
{
    const float a = fabsf(x);
    float t;
    if (a < 0.625f) {
        const float z = a * a;
        t = ((((-5.70498872745e-3f * z + 2.06390887954e-2f) * z - 5.37397155531e-2f) * z + 1.33314422036e-1f) * z - 3.33332819422e-1f) * z * a + a;
    } else {
        t = 1.0f - 2.0f / (exp_poly(2.0f * a) + 1.0f);
    }
    return x < 0.0f ? -t : t;
}
LF_DEV void cross3(const float *a, const float *b, float *o)
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
LF_DEV float dot3(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
// u k1 - v k2 with model CONSTANTS k1, k2: terms with a zero constant are left out (folds at compile time after unrolling)
LF_DEV float dopc(float u, float k1, float v, float k2)
{
    if (k1 != 0.0f && k2 != 0.0f) return fmaf(u, k1, -(v * k2));
    if (k1 != 0.0f) return u * k1;
    if (k2 != 0.0f) return -(v * k2);
    return 0.0f;
}
LF_DEV void cross_vk(const float *v, const float *k, float *o)   // v x k
{
    o[0] = dopc(v[1], k[2], v[2], k[1]);
    o[1] = dopc(v[2], k[0], v[0], k[2]);
    o[2] = dopc(v[0], k[1], v[1], k[0]);
}
LF_DEV void cross_kv(const float *k, const float *v, float *o)   // k x v
{
    o[0] = dopc(v[2], k[1], v[1], k[2]);
    o[1] = dopc(v[0], k[2], v[2], k[0]);
    o[2] = dopc(v[1], k[0], v[0], k[1]);
}
// link rotation R = Rx(alpha) Rz(theta) for alpha in {0, +-pi/2}: R^T v and R v written out per kind
LF_DEV void to_child(int kind, float s, float c, const float *v, float *o)
{
    if (kind == 0) {
        o[0] = fmaf(c, v[0], s * v[1]);
        o[1] = fmaf(c, v[1], -(s * v[0]));
        o[2] = v[2];
    } else if (kind > 0) {
        o[0] = fmaf(c, v[0], s * v[2]);
        o[1] = fmaf(c, v[2], -(s * v[0]));
        o[2] = -v[1];
    } else {
        o[0] = fmaf(c, v[0], -(s * v[2]));
        o[1] = -fmaf(c, v[2], s * v[0]);
        o[2] = v[1];
    }
}
LF_DEV void to_parent(int kind, float s, float c, const float *v, float *o)
{
    const float x = fmaf(c, v[0], -(s * v[1])), y = fmaf(s, v[0], c * v[1]);
    o[0] = x;
    if (kind == 0) { o[1] = y; o[2] = v[2]; }
    else if (kind > 0) { o[1] = -v[2]; o[2] = y; }
    else { o[1] = v[2]; o[2] = -y; }
}
LF_DEV void link_offset(int i, float *p)   // (a, -sin(alpha) d, cos(alpha) d)
{
    constexpr int KIND[7] = LF_KIND;
    constexpr float A[7] = LF_DH_A, D[7] = LF_DH_D;
    p[0] = A[i];
    p[1] = KIND[i] == 0 ? 0.0f : (KIND[i] > 0 ? -D[i] : D[i]);
    p[2] = KIND[i] == 0 ? D[i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------- arm dynamics
// one inverse-dynamics pass (recursive Newton-Euler), base at rest with acceleration (0, 0, gravity)
LF_DEV void newton_euler(const float *sn, const float *cs, const float *qd, const float *qdd, float gravity, float *tau)
{
    constexpr int KIND[7] = LF_KIND;
    constexpr float MASS[7] = LF_MASS, COM[7][3] = LF_COM, INERTIA[7][3] = LF_INERTIA;
    float F[7][3], N[7][3];
    float w_par[3] = {0.0f, 0.0f, 0.0f}, wd_par[3] = {0.0f, 0.0f, 0.0f}, a_par[3] = {0.0f, 0.0f, gravity};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float p[3], rw[3], rwd[3], w[3], wd[3], t1[3], t2[3], t3[3], acc[3], a[3];
        link_offset(i, p);
        to_child(KIND[i], sn[i], cs[i], w_par, rw);
        w[0] = rw[0]; w[1] = rw[1]; w[2] = rw[2] + qd[i];
        to_child(KIND[i], sn[i], cs[i], wd_par, rwd);
        wd[0] = fmaf(rw[1], qd[i], rwd[0]);
        wd[1] = fmaf(-rw[0], qd[i], rwd[1]);
        wd[2] = rwd[2] + qdd[i];
        cross_vk(wd_par, p, t1);
        cross_vk(w_par, p, t2);
        cross3(w_par, t2, t3);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = (a_par[k] + t1[k]) + t3[k];
        to_child(KIND[i], sn[i], cs[i], acc, a);
        cross_vk(wd, COM[i], t1);
        cross_vk(w, COM[i], t2);
        cross3(w, t2, t3);
#pragma unroll
        for (int k = 0; k < 3; ++k) F[i][k] = MASS[i] * ((a[k] + t1[k]) + t3[k]);
        const float Iw[3] = {INERTIA[i][0] * w[0], INERTIA[i][1] * w[1], INERTIA[i][2] * w[2]};
        cross3(w, Iw, t1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            N[i][k] = fmaf(INERTIA[i][k], wd[k], t1[k]);
            w_par[k] = w[k]; wd_par[k] = wd[k]; a_par[k] = a[k];
        }
    }
    float f[3] = {0.0f, 0.0f, 0.0f}, n[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 6; i >= 0; --i) {
        float fi[3], ni[3], t[3];
        if (i < 6) {
            float rf[3], rn[3], p[3];
            link_offset(i + 1, p);
            to_parent(KIND[i + 1], sn[i + 1], cs[i + 1], f, rf);
            to_parent(KIND[i + 1], sn[i + 1], cs[i + 1], n, rn);
            cross_kv(p, rf, t);
#pragma unroll
            for (int k = 0; k < 3; ++k) { fi[k] = rf[k] + F[i][k]; ni[k] = (N[i][k] + rn[k]) + t[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { fi[k] = F[i][k]; ni[k] = N[i][k]; }
        }
        cross_kv(COM[i], F[i], t);
#pragma unroll
        for (int k = 0; k < 3; ++k) { n[k] = ni[k] + t[k]; f[k] = fi[k]; }
        tau[i] = n[2];
    }
}

// Cholesky solve, lower triangle of A given (A[i][j], i >= j); reciprocal square roots by Newton
LF_DEV void cholesky_solve7(const float (&A)[7][7], const float *b, float *x)
{
    float L[7][7], dinv[7], y[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        float d2 = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d2 = fmaf(-L[j][k], L[j][k], d2);
        d2 = d2 > 1.0e-9f ? d2 : 1.0e-9f;
        dinv[j] = rsqrt_newton(d2);
#pragma unroll
        for (int i = j + 1; i < 7; ++i) {
            float s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fmaf(-L[i][k], L[j][k], s);
            L[i][j] = s * dinv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = fmaf(-L[i][k], y[k], s);
        y[i] = s * dinv[i];
    }
#pragma unroll
    for (int i = 6; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int k = i + 1; k < 7; ++k) s = fmaf(-L[k][i], x[k], s);
        x[i] = s * dinv[i];
    }
}

// The implicit-PD velocity solve and the integration of a substep, given the eight passes: cols[j][i] = tau_i of pass j
// (column j of the mass matrix), bias = tau of the velocity / gravity pass.  `second` reports whether some joint saturated
// (the caller decides wave-uniformly whether to run the second solve; running it for an unsaturated env reproduces the
// first solve bit for bit, because A and b are then unchanged).
template <bool WAVE>
LF_DEV void arm_solve_integrate(float h, const float (&cols)[8][8], const float *target, float *q, float *qd)
{
    constexpr float Q_LO[7] = LF_Q_LO, Q_HI[7] = LF_Q_HI, QD_LIM[7] = LF_QD_LIM, EFFORT[7] = LF_EFFORT;
    float Mv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float m = i >= j ? cols[j][i] : cols[i][j];
            acc = j == 0 ? m * qd[0] : fmaf(m, qd[j], acc);
        }
        Mv[i] = acc;
    }
    const float imp = h * K_ARM_KD + (h * h) * K_ARM_KP;
    int sat[7] = {0, 0, 0, 0, 0, 0, 0};
    float v[7];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float A[7][7], b[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) A[i][j] = cols[j][i];
            A[i][i] = A[i][i] + K_ARMATURE;
            const float b_pd = fmaf(h, fmaf(K_ARM_KP, target[i] - q[i], -cols[7][i]), Mv[i]);
            const float b_sat = fmaf(h, (sat[i] > 0 ? EFFORT[i] : -EFFORT[i]) - cols[7][i], Mv[i]);
            A[i][i] = sat[i] == 0 ? A[i][i] + imp : A[i][i];
            b[i] = sat[i] == 0 ? b_pd : b_sat;
        }
        cholesky_solve7(A, b, v);
        if (pass == 1) break;
        bool any = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float tq = fmaf(K_ARM_KP, target[i] - fmaf(h, v[i], q[i]), -(K_ARM_KD * v[i]));
            if (tq > EFFORT[i]) { sat[i] = 1; any = true; }
            if (tq < -EFFORT[i]) { sat[i] = -1; any = true; }
        }
        if (!(WAVE ? __builtin_amdgcn_ballot_w64(any) != 0ull : any)) break;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float vi = clampf(v[i], -QD_LIM[i], QD_LIM[i]);
        float x = fmaf(h, vi, q[i]);
        if (x > Q_HI[i]) { x = Q_HI[i]; vi = 0.0f; }
        if (x < Q_LO[i]) { x = Q_LO[i]; vi = 0.0f; }
        q[i] = x;
        qd[i] = vi;
    }
}

// ---------------------------------------------------------------------------------------------------- hand kinematics
struct HandPose {
    float R[3][3];      // hand frame -> world (x along the pads, y closing axis, z approach axis)
    float tcp[3], v[3], w[3];
};
LF_DEV void hand_kinematics(const float *sn, const float *cs, const float *qd, float ee_offset_z, HandPose &H)
{
    constexpr int KIND[7] = LF_KIND;
    float X[3] = {1.0f, 0.0f, 0.0f}, Y[3] = {0.0f, 1.0f, 0.0f}, Z[3] = {0.0f, 0.0f, 1.0f};
    float pos[3] = {0.0f, 0.0f, 0.0f}, vel[3] = {0.0f, 0.0f, 0.0f}, om[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        float p[3], pw[3], t[3];
        link_offset(i, p);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float acc = 0.0f;
            bool have = false;
            if (p[0] != 0.0f) { acc = X[k] * p[0]; have = true; }
            if (p[1] != 0.0f) { acc = have ? fmaf(Y[k], p[1], acc) : Y[k] * p[1]; have = true; }
            if (p[2] != 0.0f) { acc = have ? fmaf(Z[k], p[2], acc) : Z[k] * p[2]; have = true; }
            pw[k] = acc;
        }
        cross3(om, pw, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) { pos[k] = pos[k] + pw[k]; vel[k] = vel[k] + t[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float U = KIND[i] == 0 ? Y[k] : (KIND[i] > 0 ? Z[k] : -Z[k]);
            const float Zn = KIND[i] == 0 ? Z[k] : (KIND[i] > 0 ? -Y[k] : Y[k]);
            const float Xn = fmaf(cs[i], X[k], sn[i] * U);
            const float Yn = fmaf(cs[i], U, -(sn[i] * X[k]));
            X[k] = Xn; Y[k] = Yn; Z[k] = Zn;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) om[k] = fmaf(Z[k], qd[i], om[k]);
    }
    const float r2 = 0.70710678118654752f, reach = K_FLANGE_D + ee_offset_z;
    float rel[3], t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        H.R[k][0] = (X[k] - Y[k]) * r2;
        H.R[k][1] = (X[k] + Y[k]) * r2;
        H.R[k][2] = Z[k];
        rel[k] = Z[k] * reach;
        H.tcp[k] = pos[k] + rel[k];
        H.w[k] = om[k];
    }
    cross3(om, rel, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) H.v[k] = vel[k] + t[k];
}

// ---------------------------------------------------------------------------------------------------- cube + gripper
struct CubeConsts {
    float inv_m, inv_I, inv_h, f_minv;
};
LF_DEV CubeConsts cube_consts(float h)
{
    CubeConsts k;
    k.inv_m = 1.0f / K_CUBE_MASS;
    k.inv_I = 1.0f / (K_CUBE_MASS * (2.0f * K_CUBE_HALF) * (2.0f * K_CUBE_HALF) / 6.0f);
    k.inv_h = 1.0f / h;
    k.f_minv = 1.0f / (K_FINGER_MASS + h * K_FINGER_KD + (h * h) * K_FINGER_KP);
    return k;
}
LF_DEV void quat_to_matrix(const float *quat, float (&Rc)[3][3])
{
    const float w = quat[0], x = quat[1], y = quat[2], z = quat[3];
    Rc[0][0] = 1.0f - 2.0f * (y * y + z * z); Rc[0][1] = 2.0f * (x * y - w * z); Rc[0][2] = 2.0f * (x * z + w * y);
    Rc[1][0] = 2.0f * (x * y + w * z); Rc[1][1] = 1.0f - 2.0f * (x * x + z * z); Rc[1][2] = 2.0f * (y * z - w * x);
    Rc[2][0] = 2.0f * (x * z - w * y); Rc[2][1] = 2.0f * (y * z + w * x); Rc[2][2] = 1.0f - 2.0f * (x * x + y * y);
}
// contact rows of cube corner c against the table plane: out = {r0, r1, r2, meff_n, meff_x, meff_y, target, active}; the
// effective masses of an inactive corner are ZERO, which turns its three row updates into exact no-ops (impulses stay +0)
LF_DEV void corner_rows(int c, const float (&Rc)[3][3], float pos_z, const CubeConsts &K, float *out)
{
    const float sx = (c & 1) ? K_CUBE_HALF : -K_CUBE_HALF, sy = (c & 2) ? K_CUBE_HALF : -K_CUBE_HALF, sz = (c & 4) ? K_CUBE_HALF : -K_CUBE_HALF;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = fmaf(Rc[k][2], sz, fmaf(Rc[k][1], sy, Rc[k][0] * sx));
    const float z = pos_z + r[2];
    const bool active = z < K_TABLE_MARGIN;
    const float m0 = 1.0f / fmaf(K.inv_I, fmaf(r[1], r[1], r[0] * r[0]), K.inv_m);
    const float m1 = 1.0f / fmaf(K.inv_I, fmaf(r[2], r[2], r[1] * r[1]), K.inv_m);
    const float m2 = 1.0f / fmaf(K.inv_I, fmaf(r[2], r[2], r[0] * r[0]), K.inv_m);
    const float push = (K_BAUMGARTE * -z) * K.inv_h;
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
    out[3] = active ? m0 : 0.0f; out[4] = active ? m1 : 0.0f; out[5] = active ? m2 : 0.0f;
    out[6] = z < 0.0f ? (push < 1.0f ? push : 1.0f) : -z * K.inv_h;
    out[7] = active ? 1.0f : 0.0f;
}
struct PadRows {
    float n[3][3], rxn[4][3], meff[4], target[4], lam[4];
    bool active;
};

// One substep of cube + fingers given the corner rows (cr[c] = the eight values of corner_rows).  any_active(x) answers
// "does this row set need processing": the per-env flag on the CPU-style path, a wave ballot in the step kernel.
template <bool WAVE>
LF_DEV bool any_active(bool x)
{
    return WAVE ? __builtin_amdgcn_ballot_w64(x) != 0ull : x;
}
template <bool WAVE, class CFG>
LF_DEV void cube_substep(const CFG &cfg, float h, const CubeConsts &K, const HandPose &H, const float (&Rc)[3][3],
                         const float (&cr)[8][8], const float *finger_target, float *fq, float *fqd, float *pos, float *quat, float *lin,
                         float *ang)
{
    float fv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float force = clampf(K_FINGER_KP * (finger_target[k] - fq[k]), -K_FINGER_EFFORT, K_FINGER_EFFORT);
        fv[k] = fmaf(h, force, K_FINGER_MASS * fqd[k]) * K.f_minv;
    }
    // (gravity has been applied to lin[2] by the caller, before the corner rows were formed -- same order as the oracle)
    PadRows pd[2];
    {
        float d[3], cl[3], xh[3], yh[3], zh[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { d[k] = pos[k] - H.tcp[k]; xh[k] = H.R[k][0]; yh[k] = H.R[k][1]; zh[k] = H.R[k][2]; }
        cl[0] = dot3(xh, d); cl[1] = dot3(yh, d); cl[2] = dot3(zh, d);
        float ext = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float col[3] = {Rc[0][a], Rc[1][a], Rc[2][a]};
            ext = fmaf(fabsf(dot3(yh, col)), K_CUBE_HALF, ext);
        }
        const bool between = fabsf(cl[0]) < K_CUBE_HALF + K_PAD_HALF_X && fabsf(cl[2]) < K_CUBE_HALF + K_PAD_HALF_Z;
        const float px = clampf(cl[0], -K_PAD_HALF_X, K_PAD_HALF_X), pz = clampf(cl[2], -K_PAD_HALF_Z, K_PAD_HALF_Z);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float sgn = k == 0 ? 1.0f : -1.0f;
            const float gap = fq[k] - (fmaf(sgn, cl[1], ext));
            PadRows &P = pd[k];
            P.active = between && gap < K_PAD_MARGIN;
        }
        // The rows of pads nobody touches are never read: form them only when some env of the wave is being grasped
        // (random actions: almost never -- ~150 instructions per pad and substep).
        const bool form_pads = any_active<WAVE>(pd[0].active || pd[1].active);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!form_pads) continue;
            const float sgn = k == 0 ? 1.0f : -1.0f;
            const float gap = fq[k] - (fmaf(sgn, cl[1], ext));
            PadRows &P = pd[k];
            float arm[3], r[3], vpad[3], t[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                arm[i] = fmaf(zh[i], pz, fmaf(yh[i], sgn * fq[k], xh[i] * px));
                r[i] = (H.tcp[i] + arm[i]) - pos[i];
            }
            cross3(H.w, arm, t);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                vpad[i] = H.v[i] + t[i];
                P.n[0][i] = -sgn * yh[i]; P.n[1][i] = xh[i]; P.n[2][i] = zh[i];
            }
#pragma unroll
            for (int row = 0; row < 3; ++row) cross3(r, P.n[row], P.rxn[row]);
#pragma unroll
            for (int i = 0; i < 3; ++i) P.rxn[3][i] = yh[i];
            const float m0 = 1.0f / (fmaf(K.inv_I, dot3(P.rxn[0], P.rxn[0]), K.inv_m) + K.f_minv);
            const float m1 = 1.0f / fmaf(K.inv_I, dot3(P.rxn[1], P.rxn[1]), K.inv_m);
            const float m2 = 1.0f / fmaf(K.inv_I, dot3(P.rxn[2], P.rxn[2]), K.inv_m);
            const float m3 = 1.0f / K.inv_I;
            // an inactive pad: zero effective masses = exact no-op rows (see corner_rows)
            P.meff[0] = P.active ? m0 : 0.0f; P.meff[1] = P.active ? m1 : 0.0f; P.meff[2] = P.active ? m2 : 0.0f; P.meff[3] = P.active ? m3 : 0.0f;
            const float push = (K_BAUMGARTE * -gap) * K.inv_h;
            P.target[0] = dot3(vpad, P.n[0]) + (gap < 0.0f ? (push < 0.5f ? push : 0.5f) : -gap * K.inv_h);
            P.target[1] = dot3(vpad, P.n[1]);
            P.target[2] = dot3(vpad, P.n[2]);
            P.target[3] = dot3(H.w, yh);
#pragma unroll
            for (int row = 0; row < 4; ++row) P.lam[row] = 0.0f;
        }
    }
    float lam[8][3];
#pragma unroll
    for (int c = 0; c < 8; ++c) lam[c][0] = lam[c][1] = lam[c][2] = 0.0f;
    const float inv_m = K.inv_m, inv_I = K.inv_I;
    // Which row sets this wave has to sweep does not change during the iterations: decided ONCE, wave-uniformly, per face
    // of the cube (corners 0..3 = the -z face, 4..7 = the +z face) and for the two pads together.  A row set that is swept
    // although the env's own rows are inactive has zero effective masses = exact no-ops, so the result is the one the
    // oracle gets by skipping inactive rows one by one.  (A test per corner inside the loop cost ten VALU -> branch round
    // trips per iteration: 12.5 k of the kernel's 60 k cycles went into sweeps of ~1.4 k instructions.)
    const bool sweep_lo = any_active<WAVE>(cr[0][7] != 0.0f || cr[1][7] != 0.0f || cr[2][7] != 0.0f || cr[3][7] != 0.0f);
    const bool sweep_hi = any_active<WAVE>(cr[4][7] != 0.0f || cr[5][7] != 0.0f || cr[6][7] != 0.0f || cr[7][7] != 0.0f);
    const bool sweep_pads = any_active<WAVE>(pd[0].active || pd[1].active);
    auto corner = [&](int c) {
        const float r0 = cr[c][0], r1 = cr[c][1], r2 = cr[c][2];
        {
            const float u = lin[2] + fmaf(r1, ang[0], -(r0 * ang[1]));
            const float l = nonneg(fmaf(cr[c][6] - u, cr[c][3], lam[c][0]));
            const float dl = l - lam[c][0];
            lam[c][0] = l;
            lin[2] = fmaf(dl, inv_m, lin[2]);
            const float di = dl * inv_I;
            ang[0] = fmaf(r1, di, ang[0]);
            ang[1] = fmaf(-r0, di, ang[1]);
        }
        const float lim = cfg.mu_table * lam[c][0];
        {
            const float u = lin[0] + fmaf(r2, ang[1], -(r1 * ang[2]));
            const float l = clamp_sym(fmaf(-u, cr[c][4], lam[c][1]), lim);
            const float dl = l - lam[c][1];
            lam[c][1] = l;
            lin[0] = fmaf(dl, inv_m, lin[0]);
            const float di = dl * inv_I;
            ang[1] = fmaf(r2, di, ang[1]);
            ang[2] = fmaf(-r1, di, ang[2]);
        }
        {
            const float u = lin[1] + fmaf(r0, ang[2], -(r2 * ang[0]));
            const float l = clamp_sym(fmaf(-u, cr[c][5], lam[c][2]), lim);
            const float dl = l - lam[c][2];
            lam[c][2] = l;
            lin[1] = fmaf(dl, inv_m, lin[1]);
            const float di = dl * inv_I;
            ang[2] = fmaf(r0, di, ang[2]);
            ang[0] = fmaf(-r2, di, ang[0]);
        }
    };
    auto pads = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            PadRows &P = pd[k];
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                float u = dot3(P.rxn[row], ang);
                if (row < 3) u = u + dot3(P.n[row], lin);
                if (row == 0) u = u + fv[k];
                float l = fmaf(P.target[row] - u, P.meff[row], P.lam[row]);
                if (row == 0) {
                    l = nonneg(l);
                } else {
                    const float lim = (row == 3 ? cfg.mu_pad * K_TORSION_R : cfg.mu_pad) * P.lam[0];
                    l = clamp_sym(l, lim);
                }
                const float dl = l - P.lam[row];
                P.lam[row] = l;
                const float dm = dl * inv_m, di = dl * inv_I;
                if (row < 3) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) lin[i] = fmaf(P.n[row][i], dm, lin[i]);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) ang[i] = fmaf(P.rxn[row][i], di, ang[i]);
                if (row == 0) fv[k] = fmaf(dl, K.f_minv, fv[k]);
            }
        }
    };
    // two copies of the iteration loop: the pad rows (66 registers) are only alive in the one that sweeps them
    if (!sweep_pads) {
        for (int it = 0; it < cfg.solver_iterations; ++it) {
            if (sweep_lo) {
#pragma unroll
                for (int c = 0; c < 4; ++c) corner(c);
            }
            if (sweep_hi) {
#pragma unroll
                for (int c = 4; c < 8; ++c) corner(c);
            }
        }
    } else {
        for (int it = 0; it < cfg.solver_iterations; ++it) {
            if (sweep_lo) {
#pragma unroll
                for (int c = 0; c < 4; ++c) corner(c);
            }
            if (sweep_hi) {
#pragma unroll
                for (int c = 4; c < 8; ++c) corner(c);
            }
            pads();
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float v = clampf(fv[k], -K_FINGER_VLIM, K_FINGER_VLIM);
        float x = fmaf(h, v, fq[k]);
        if (x > K_FINGER_TRAVEL) { x = K_FINGER_TRAVEL; v = 0.0f; }
        if (x < 0.0f) { x = 0.0f; v = 0.0f; }
        fq[k] = x;
        fqd[k] = v;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[k] = fmaf(h, lin[k], pos[k]);
    {
        const float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3], hh = 0.5f * h;
        const float nw = fmaf(hh, -fmaf(ang[2], qz, fmaf(ang[1], qy, ang[0] * qx)), qw);
        const float nx = fmaf(hh, fmaf(-ang[2], qy, fmaf(ang[1], qz, ang[0] * qw)), qx);
        const float ny = fmaf(hh, fmaf(-ang[0], qz, fmaf(ang[2], qx, ang[1] * qw)), qy);
        const float nz = fmaf(hh, fmaf(-ang[1], qx, fmaf(ang[0], qy, ang[2] * qw)), qz);
        const float inv = rsqrt_newton(fmaf(nz, nz, fmaf(ny, ny, fmaf(nx, nx, nw * nw))));
        quat[0] = nw * inv; quat[1] = nx * inv; quat[2] = ny * inv; quat[3] = nz * inv;
    }
}

// ---------------------------------------------------------------------------------------------------- MDP terms
// ORBIT utils.math.quat_apply (w, x, y, z) in ORBIT's operation order
LF_DEV void quat_apply(const float *q, const float *v, float *o)
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float tx = 2.0f * (y * v[2] - z * v[1]), ty = 2.0f * (z * v[0] - x * v[2]), tz = 2.0f * (x * v[1] - y * v[0]);
    o[0] = v[0] + w * tx + (y * tz - z * ty);
    o[1] = v[1] + w * ty + (z * tx - x * tz);
    o[2] = v[2] + w * tz + (x * ty - y * tx);
}
// rewards.py:20-26 object_is_lifted, :29-46 object_ee_distance, :49-67 object_goal_distance; observations.py:19-31
LF_DEV void reference_terms(const lift_config &c, const float *obj, const float *ee, const float *root_pos, const float *root_quat,
                            const float *cmd_pos_b, float &lifted, float &reach, float &goal, float &goal_fine, float *obj_b)
{
    lifted = obj[2] > c.minimal_height ? 1.0f : 0.0f;
    const float d[3] = {obj[0] - ee[0], obj[1] - ee[1], obj[2] - ee[2]};
    reach = 1.0f - tanh_poly(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / c.reach_std);
    float des[3];
    quat_apply(root_quat, cmd_pos_b, des);
    const float g[3] = {des[0] + root_pos[0] - obj[0], des[1] + root_pos[1] - obj[1], des[2] + root_pos[2] - obj[2]};
    const float dist = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    goal = lifted * (1.0f - tanh_poly(dist / c.goal_std));
    goal_fine = lifted * (1.0f - tanh_poly(dist / c.goal_fine_std));
    const float qi[4] = {root_quat[0], -root_quat[1], -root_quat[2], -root_quat[3]};
    const float rel[3] = {obj[0] - root_pos[0], obj[1] - root_pos[1], obj[2] - root_pos[2]};
    quat_apply(qi, rel, obj_b);
}

LF_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
LF_DEV float uniform01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

// UniformPoseCommand._resample_command (manipulation_env_cfg.py:163-172)
LF_DEV void resample_command(const lift_config &c, float *S, uint32_t gid, uint32_t count, uint32_t stream)
{
    uint32_t r[4];
    philox4x32_10(gid, count, 1u, stream, c.seed_lo, c.seed_hi, r);
#pragma unroll
    for (int k = 0; k < 3; ++k) S[LIFT_CMD + k] = fmaf(uniform01(r[k]), c.cmd_hi[k] - c.cmd_lo[k], c.cmd_lo[k]);
    S[LIFT_CMD + 3] = 1.0f; S[LIFT_CMD + 4] = 0.0f; S[LIFT_CMD + 5] = 0.0f; S[LIFT_CMD + 6] = 0.0f;
    S[LIFT_TIME_LEFT] = c.cmd_resample_time;
}
// RLTaskEnv._reset_idx: reset_scene_to_default (:179), reset_root_state_uniform on the object (:181-190), manager resets
LF_DEV void reset_env(const lift_config &c, float *S, uint32_t gid)
{
    constexpr float QDEF[9] = LF_Q_DEFAULT;
    const uint32_t count = f2u(S[LIFT_RESET_COUNT]);
    uint32_t r[4];
    philox4x32_10(gid, count, 0u, 0u, c.seed_lo, c.seed_hi, r);
#pragma unroll
    for (int i = 0; i < 9; ++i) { S[LIFT_Q + i] = QDEF[i]; S[LIFT_QD + i] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        S[LIFT_OBJ_POS + k] = c.obj_init[k] + fmaf(uniform01(r[k]), c.obj_range_hi[k] - c.obj_range_lo[k], c.obj_range_lo[k]);
    S[LIFT_OBJ_QUAT] = 1.0f; S[LIFT_OBJ_QUAT + 1] = 0.0f; S[LIFT_OBJ_QUAT + 2] = 0.0f; S[LIFT_OBJ_QUAT + 3] = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) S[LIFT_OBJ_LIN + k] = 0.0f;
#pragma unroll
    for (int i = 0; i < LIFT_ACT; ++i) { S[LIFT_ACTION + i] = 0.0f; S[LIFT_PREV_ACTION + i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < LIFT_NUM_REW; ++i) S[LIFT_EP_SUM + i] = 0.0f;
    resample_command(c, S, gid, count, 0u);
    S[LIFT_EP_LEN] = u2f(0u);
    S[LIFT_RESET_COUNT] = u2f(count + 1u);
}
// ObservationsCfg.PolicyCfg (manipulation_env_cfg.py:101-116)
LF_DEV void write_observation(const float *S, float *o)
{
    constexpr float QDEF[9] = LF_Q_DEFAULT;
#pragma unroll
    for (int i = 0; i < 9; ++i) { o[i] = S[LIFT_Q + i] - QDEF[i]; o[9 + i] = S[LIFT_QD + i]; }
    const float root_pos[3] = {0.0f, 0.0f, 0.0f}, qi[4] = {1.0f, -0.0f, -0.0f, -0.0f};
    const float rel[3] = {S[LIFT_OBJ_POS] - root_pos[0], S[LIFT_OBJ_POS + 1] - root_pos[1], S[LIFT_OBJ_POS + 2] - root_pos[2]};
    quat_apply(qi, rel, o + 18);
#pragma unroll
    for (int i = 0; i < 7; ++i) o[21 + i] = S[LIFT_CMD + i];
#pragma unroll
    for (int i = 0; i < LIFT_ACT; ++i) o[28 + i] = S[LIFT_ACTION + i];
}

// ---------------------------------------------------------------------------------------------------- lane exchange
// Wave-private all-gather inside an env's eight lanes: lane `role` contributes K floats, every lane receives all 8 x K.
// The slots of the eight lane groups of a wave are XSTRIDE floats apart: 8 x 8 payload + 4 floats of skew, so that the
// 16-byte reads of different groups fall into different LDS banks.  A 64-thread workgroup is one wave: __syncthreads() is
// a compiler / memory fence only (no s_barrier is emitted for a single-wave workgroup).
constexpr int XSTRIDE = 68;
// Synchronisation of ONE wave with itself around its private LDS slot.  Single-wave workgroup: __syncthreads() (no s_barrier is
// emitted).  Two-wave (pipelined) workgroup: the waves run different code between their common barriers, so the wave-private
// exchanges must not use the workgroup barrier: a wavefront-scope fence (LDS operations of one wave complete in order).
template <bool WAVE_ONLY>
__device__ __forceinline__ void slot_sync()
{
    if constexpr (WAVE_ONLY) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
template <int K, bool WAVE_ONLY = false>
__device__ __forceinline__ void gather8(float *xbuf, int slot, int role, bool shadow, const float (&mine)[K], float (&all)[8][K])
{
    float *base = xbuf + slot * XSTRIDE;
    if (!shadow) {
        if constexpr (K == 8) {
            reinterpret_cast<float4 *>(base + role * 8)[0] = make_float4(mine[0], mine[1], mine[2], mine[3]);
            reinterpret_cast<float4 *>(base + role * 8)[1] = make_float4(mine[4], mine[5], mine[6], mine[7]);
        } else if constexpr (K == 4) {
            reinterpret_cast<float4 *>(base + role * 4)[0] = make_float4(mine[0], mine[1], mine[2], mine[3]);
        } else {
            reinterpret_cast<float2 *>(base + role * 2)[0] = make_float2(mine[0], mine[1]);
        }
    }
    slot_sync<WAVE_ONLY>();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if constexpr (K == 8) {
            const float4 a = reinterpret_cast<const float4 *>(base + r * 8)[0], b = reinterpret_cast<const float4 *>(base + r * 8)[1];
            all[r][0] = a.x; all[r][1] = a.y; all[r][2] = a.z; all[r][3] = a.w;
            all[r][4] = b.x; all[r][5] = b.y; all[r][6] = b.z; all[r][7] = b.w;
        } else if constexpr (K == 4) {
            const float4 a = reinterpret_cast<const float4 *>(base + r * 4)[0];
            all[r][0] = a.x; all[r][1] = a.y; all[r][2] = a.z; all[r][3] = a.w;
        } else {
            const float2 a = reinterpret_cast<const float2 *>(base + r * 2)[0];
            all[r][0] = a.x; all[r][1] = a.y;
        }
    }
    slot_sync<WAVE_ONLY>();
}
// Element `role` of an 8-vector that every lane holds, with the lane's one-hot bit masks (sel[i] = all ones iff role == i):
// eight v_and_or_b32 on register values.  (A chain of selects between the elements would be turned by the optimiser into
// ONE load through a selected address -- a dynamically indexed private array, which hipcc then parks in LDS or scratch
// instead of registers; and compare masks kept in SGPR pairs across the substep loop cost SGPR spills.)
__device__ __forceinline__ float pick8(const float *v, const uint32_t *sel)
{
    uint32_t r = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) r = (f2u(v[i]) & sel[i]) | r;
    return u2f(r);
}

// s_memtime phase stamps of wave 0 of every workgroup: diagnostic builds only (tools/build_diag.py LIFTSTAMP, tools/lift_stamps.py)
#ifdef LF_STAMP
__device__ unsigned long long *g_lift_stamps;
__device__ __forceinline__ void lift_stamp(int slot)
{
    if ((threadIdx.x & 63) == 0) {      // lane 0 of either wave: [workgroup][wave][slot]
        unsigned long long t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        g_lift_stamps[((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * 32 + slot] = t;
    }
}
#define LIFT_STAMP(k) lift_stamp(k)
#else
#define LIFT_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------- kernels
__global__ __launch_bounds__(64) void lift_reset_kernel(lift_config c, int n, int env_id_offset, float *__restrict__ state,
                                                        float *__restrict__ obs)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n) return;
    float S[LIFT_STATE_WORDS];
#pragma unroll
    for (int i = 0; i < LIFT_STATE_WORDS; ++i) S[i] = state[(size_t)i * n + e];
    reset_env(c, S, (uint32_t)(env_id_offset + e));
#pragma unroll
    for (int i = 0; i < LIFT_STATE_WORDS; ++i) state[(size_t)i * n + e] = S[i];
    float o[LIFT_OBS];
    write_observation(S, o);
#pragma unroll
    for (int i = 0; i < LIFT_OBS; ++i) obs[(size_t)e * LIFT_OBS + i] = o[i];
}

// The configuration words the substep loop reads, by value (kernel argument SGPRs).  Everything else of lift_config
// (reward weights, ranges, seeds ...) is read from a device copy at the point of use, after the substeps: as kernel
// arguments those ~27 words would stay live in SGPRs through the whole kernel and push other scalars into spill lanes.
struct LiftHot {
    float sim_dt;
    int32_t decimation, solver_iterations;
    float mu_table, mu_pad, ee_offset_z, action_scale, finger_open, finger_close;
};
static LiftHot hot_of(const lift_config &c)
{
    return LiftHot{c.sim_dt, c.decimation, c.solver_iterations, c.mu_table, c.mu_pad, c.ee_offset_z, c.action_scale, c.finger_open,
                   c.finger_close};
}

// RLTaskEnv.step of FrankaCubeLift-v0, LPE lanes per env (8, or 16 with lanes 8..15 of a row shadowing lanes 0..7).
// PIPE: two waves per eight envs.  The arm does not feel the cube in this model (one-way coupling: pads -> cube), so wave 0
// integrates the arm -- inverse-dynamics passes, solves, hand kinematics -- one substep AHEAD of wave 1, which runs the cube /
// finger contact of the substep from the hand pose wave 0 left in LDS, and then the managers.  One workgroup barrier per
// substep; the arithmetic and its order are those of the single-wave form (bit-identical results).
template <int LPE, bool PIPE>
__global__ __launch_bounds__(PIPE ? 128 : 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lift_step_kernel(
    LiftHot hc, const lift_config *__restrict__ cfg_dev, int n, int env_id_offset, float *__restrict__ state,
    const float *__restrict__ action, float *__restrict__ obs,
    float *__restrict__ reward, uint8_t *__restrict__ terminated, uint8_t *__restrict__ truncated, float *__restrict__ lg_out,
    unsigned *__restrict__ counters, uint32_t serial)
{
    static_assert(LPE == 8 || LPE == 16, "eight lanes per env, optionally shadowed");
    static_assert(!PIPE || LPE == 8, "the pipelined form has eight lanes per env");
    constexpr float QDEF[9] = LF_Q_DEFAULT;
    constexpr int HAND_WORDS = 20, ARM_WORDS = 16;
    __shared__ __attribute__((aligned(16))) float xbuf_all[(PIPE ? 2 : 1) * 8 * XSTRIDE];
    __shared__ __attribute__((aligned(16))) float hand_buf[PIPE ? 2 * 8 * HAND_WORDS : 4];   // [substep parity][env slot]: R, tcp, v, w
    __shared__ __attribute__((aligned(16))) float arm_buf[PIPE ? 8 * ARM_WORDS : 4];          // [env slot]: final q[0..6], qd[0..6]
    const int wave = PIPE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const bool do_arm = !PIPE || wave == 0, do_cube = !PIPE || wave == 1;
    float *xbuf = xbuf_all + wave * 8 * XSTRIDE;
    const int lane = threadIdx.x & 63, role = lane & 7;
    const int slot = LPE == 8 ? lane >> 3 : (lane >> 4) * 2;
    const bool shadow = LPE == 16 && (lane & 8) != 0;
    int e = blockIdx.x * (64 / LPE) + lane / LPE;
    const bool valid = e < n;
    if (!valid) e = n - 1;                        // lanes past the batch recompute the last env and store nothing
    const bool writer = valid && !shadow && role == 0;
    const uint32_t gid = (uint32_t)(env_id_offset + e);

    // State I/O is lane-distributed: lane r of an env moves the words r, r + 8, ..., r + 56 (one wave instruction moves 64
    // DIFFERENT words: 8 envs x 8 consecutive words) and the lanes exchange them through the LDS slot.  The physics half
    // (words 0 .. 31: joints, cube) is loaded now, the manager half (command, timers, actions, episodic sums) only after the
    // substeps -- replicated it would sit in 32 registers through the solver; distributed it is 4 (loaded up front).
    const unsigned lane_off = ((unsigned)role * (unsigned)n + (unsigned)e) * 4u;      // byte offset of word `role` of env e
    auto column = [&](float *base, int k) -> float & {                                 // word 8 k + role of env e
        return *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (size_t)(8 * k) * (size_t)(unsigned)n * 4u + lane_off);
    };
    float S[LIFT_STATE_WORDS], late[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) late[k] = column(state, 4 + k);      // in flight during the physics, exchanged after it
    {
        float mine[4], all[8][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) mine[k] = column(state, k);
        gather8<4, PIPE>(xbuf, slot, role, shadow, mine, all);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 8; ++r) S[8 * k + r] = all[r][k];
    }
    float target[7], finger_target[2], a[LIFT_ACT];
    {   // JointPositionAction (:35-37), BinaryJointPositionAction (:38-43)
        const float4 a0 = reinterpret_cast<const float4 *>(action + (size_t)e * LIFT_ACT)[0];
        const float4 a1 = reinterpret_cast<const float4 *>(action + (size_t)e * LIFT_ACT)[1];
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
#pragma unroll
        for (int i = 0; i < 7; ++i) target[i] = QDEF[i] + hc.action_scale * a[i];
        finger_target[0] = finger_target[1] = a[7] < 0.0f ? hc.finger_close : hc.finger_open;
    }

    const float h = hc.sim_dt;
    const CubeConsts K = cube_consts(h);
    // one-hot lane masks in VGPRs: sel[i] = ~0 in the lane with role i (pass i of the inverse dynamics / joint i / corner i)
    uint32_t sel[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sel[i] = role == i ? 0xFFFFFFFFu : 0u;
        asm volatile("" : "+v"(sel[i]));          // keep them as register values (not re-derived compare masks)
    }
    float *q = S + LIFT_Q, *qd = S + LIFT_QD;
    float sn[8], cs[8];
    auto joint_trig = [&]() {       // sin / cos of joint `role` in lane `role`, then shared
        float mine[2], all[8][2];
        sincos_poly(pick8(q, sel), mine[0], mine[1]);      // lane 7 evaluates finger joint q[7]: unused
        gather8<2, PIPE>(xbuf, slot, role, shadow, mine, all);
#pragma unroll
        for (int i = 0; i < 8; ++i) { sn[i] = all[i][0]; cs[i] = all[i][1]; }
    };
    LIFT_STAMP(0);
    if (do_arm) joint_trig();
    HandPose hand;
    LIFT_STAMP(1);
    // hand-off of the pipelined form: the arm wave leaves the hand pose of the substep (and, every time, the arm's joint state:
    // the last one written is the final one) in LDS; the cube wave picks them up behind the substep's barrier
    auto hand_put = [&](int par) {
        if (role == 0) {
            float4 *hb = reinterpret_cast<float4 *>(hand_buf + (par * 8 + slot) * HAND_WORDS);
            hb[0] = make_float4(hand.R[0][0], hand.R[0][1], hand.R[0][2], hand.R[1][0]);
            hb[1] = make_float4(hand.R[1][1], hand.R[1][2], hand.R[2][0], hand.R[2][1]);
            hb[2] = make_float4(hand.R[2][2], hand.tcp[0], hand.tcp[1], hand.tcp[2]);
            hb[3] = make_float4(hand.v[0], hand.v[1], hand.v[2], hand.w[0]);
            hb[4] = make_float4(hand.w[1], hand.w[2], 0.0f, 0.0f);
            float4 *ab = reinterpret_cast<float4 *>(arm_buf + slot * ARM_WORDS);
            ab[0] = make_float4(q[0], q[1], q[2], q[3]);
            ab[1] = make_float4(q[4], q[5], q[6], qd[0]);
            ab[2] = make_float4(qd[1], qd[2], qd[3], qd[4]);
            ab[3] = make_float4(qd[5], qd[6], 0.0f, 0.0f);
        }
    };
    auto hand_get = [&](int par) {
        const float4 *hb = reinterpret_cast<const float4 *>(hand_buf + (par * 8 + slot) * HAND_WORDS);
        const float4 h0 = hb[0], h1 = hb[1], h2 = hb[2], h3 = hb[3], h4 = hb[4];
        hand.R[0][0] = h0.x; hand.R[0][1] = h0.y; hand.R[0][2] = h0.z; hand.R[1][0] = h0.w;
        hand.R[1][1] = h1.x; hand.R[1][2] = h1.y; hand.R[2][0] = h1.z; hand.R[2][1] = h1.w;
        hand.R[2][2] = h2.x; hand.tcp[0] = h2.y; hand.tcp[1] = h2.z; hand.tcp[2] = h2.w;
        hand.v[0] = h3.x; hand.v[1] = h3.y; hand.v[2] = h3.z; hand.w[0] = h3.w;
        hand.w[1] = h4.x; hand.w[2] = h4.y;
    };
    for (int s = 0; s < hc.decimation; ++s) {
        // ---- arm: pass `role` of the eight inverse-dynamics passes, exchanged, then the replicated 7 x 7 solve
        if (do_arm) {
            {
                float qd_l[7], qdd_l[7], tau[8], cols[8][8];
#pragma unroll
                for (int i = 0; i < 7; ++i) { qd_l[i] = u2f(f2u(qd[i]) & sel[7]); qdd_l[i] = u2f(0x3f800000u & sel[i]); }
                newton_euler(sn, cs, qd_l, qdd_l, u2f(f2u(K_GRAV) & sel[7]), tau);
                tau[7] = 0.0f;
                LIFT_STAMP(2 + 6 * (s & 1));
                gather8<8, PIPE>(xbuf, slot, role, shadow, tau, cols);
                arm_solve_integrate<true>(h, cols, target, q, qd);
            }
            LIFT_STAMP(3 + 6 * (s & 1));
            joint_trig();
            hand_kinematics(sn, cs, qd, hc.ee_offset_z, hand);
            if constexpr (PIPE) hand_put(s & 1);
        }
        if constexpr (PIPE) __syncthreads();       // the ONE workgroup barrier of a substep: hand pose s is in LDS
        LIFT_STAMP(4 + 6 * (s & 1));
        // ---- cube: gravity, rows of corner `role`, exchanged, then the replicated Gauss-Seidel sweeps
        if (do_cube) {
            if constexpr (PIPE) hand_get(s & 1);
            float *pos = S + LIFT_OBJ_POS, *quat = S + LIFT_OBJ_QUAT, *lin = S + LIFT_OBJ_LIN, *ang = S + LIFT_OBJ_ANG;
            lin[2] = fmaf(-K_GRAV, h, lin[2]);
            float Rc[3][3], mine[8], cr[8][8];
            quat_to_matrix(quat, Rc);
            corner_rows(role, Rc, pos[2], K, mine);      // corner index = role: three per-lane sign selects, uniform code otherwise
            gather8<8, PIPE>(xbuf, slot, role, shadow, mine, cr);
            LIFT_STAMP(5 + 6 * (s & 1));
            cube_substep<true>(hc, h, K, hand, Rc, cr, finger_target, q + 7, qd + 7, pos, quat, lin, ang);
        }
        LIFT_STAMP(7 + 6 * (s & 1));
    }
    if (hc.decimation <= 0) {
        if (do_arm) {
            hand_kinematics(sn, cs, qd, hc.ee_offset_z, hand);
            if constexpr (PIPE) hand_put(0);
        }
        if constexpr (PIPE) {
            __syncthreads();
            if (do_cube) hand_get(0);
        }
    }
    if constexpr (PIPE) {
        if (!do_cube) return;                      // the arm wave is done: everything below is the cube wave's
        const float4 *ab = reinterpret_cast<const float4 *>(arm_buf + slot * ARM_WORDS);
        const float4 a0 = ab[0], a1 = ab[1], a2 = ab[2], a3 = ab[3];
        q[0] = a0.x; q[1] = a0.y; q[2] = a0.z; q[3] = a0.w; q[4] = a1.x; q[5] = a1.y; q[6] = a1.z;
        qd[0] = a1.w; qd[1] = a2.x; qd[2] = a2.y; qd[3] = a2.z; qd[4] = a2.w; qd[5] = a3.x; qd[6] = a3.y;
    }

    // ---- manager words; ActionManager.process_action (prev_action <- action <- the raw action)
    {
        float all[8][4];
        gather8<4, PIPE>(xbuf, slot, role, shadow, late, all);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 8; ++r) S[32 + 8 * k + r] = all[r][k];
    }
#pragma unroll
    for (int i = 0; i < LIFT_ACT; ++i) { S[LIFT_PREV_ACTION + i] = S[LIFT_ACTION + i]; S[LIFT_ACTION + i] = a[i]; }
    LIFT_STAMP(14);
    // ---- counters, terminations, rewards (replicated in the eight lanes); from here on `c` is the full configuration
    const lift_config &c = *cfg_dev;
    const int32_t ep_len = (int32_t)f2u(S[LIFT_EP_LEN]) + 1;
    S[LIFT_EP_LEN] = u2f((uint32_t)ep_len);
    const float root_pos[3] = {0.0f, 0.0f, 0.0f}, root_quat[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    float rew[LIFT_NUM_REW], obj_b[3];
    reference_terms(c, S + LIFT_OBJ_POS, hand.tcp, root_pos, root_quat, S + LIFT_CMD, rew[1], rew[0], rew[2], rew[3], obj_b);
    float rate = 0.0f, jvel = 0.0f;
#pragma unroll
    for (int i = 0; i < LIFT_ACT; ++i) { const float d = S[LIFT_ACTION + i] - S[LIFT_PREV_ACTION + i]; rate += d * d; }
#pragma unroll
    for (int i = 0; i < 9; ++i) jvel += S[LIFT_QD + i] * S[LIFT_QD + i];
    rew[4] = rate;
    rew[5] = jvel;
    const bool time_out = ep_len >= c.max_episode_length;      // mdp.time_out
    const bool dropped = S[LIFT_OBJ_POS + 2] < c.drop_height;   // mdp.base_height(minimum_height = -0.05)
    const float step_dt = c.sim_dt * (float)c.decimation;
    float total = 0.0f;
#pragma unroll
    for (int i = 0; i < LIFT_NUM_REW; ++i) {
        if (c.rew_weight[i] != 0.0f) {
            const float val = rew[i] * c.rew_weight[i] * step_dt;
            total += val;
            S[LIFT_EP_SUM + i] += val;
        }
    }
    const bool do_reset = time_out || dropped;
    LIFT_STAMP(15);
    float lg[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) lg[i] = 0.0f;
    if (__builtin_amdgcn_ballot_w64(do_reset) != 0ull) {
        float R[LIFT_STATE_WORDS];
#pragma unroll
        for (int i = 0; i < LIFT_STATE_WORDS; ++i) R[i] = S[i];
        reset_env(c, R, gid);
#pragma unroll
        for (int i = 0; i < LIFT_NUM_REW; ++i) lg[i] = do_reset ? S[LIFT_EP_SUM + i] : 0.0f;
        lg[6] = do_reset && time_out ? 1.0f : 0.0f;
        lg[7] = do_reset && dropped ? 1.0f : 0.0f;
        lg[8] = do_reset ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < LIFT_STATE_WORDS; ++i) S[i] = do_reset ? R[i] : S[i];
    }
    // ---- CommandTerm.compute: timer, resample
    S[LIFT_TIME_LEFT] -= step_dt;
    const bool resample = S[LIFT_TIME_LEFT] <= 0.0f;
    if (__builtin_amdgcn_ballot_w64(resample) != 0ull) {
        float R[LIFT_STATE_WORDS];
#pragma unroll
        for (int i = LIFT_CMD; i <= LIFT_TIME_LEFT; ++i) R[i] = S[i];
        resample_command(c, R, gid, f2u(S[LIFT_RESET_COUNT]), 1u);
#pragma unroll
        for (int i = LIFT_CMD; i <= LIFT_TIME_LEFT; ++i) S[i] = resample ? R[i] : S[i];
    }
    LIFT_STAMP(16);
    // ---- stores.  State: lane 0 lays the 64 words out transposed in the LDS slot ([word % 8][word / 8]), every lane picks
    // up its eight words with two 16-byte reads and stores them (64 different words per wave instruction again).
    {
        float *base = xbuf + slot * XSTRIDE;
        if (role == 0 && !shadow) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                reinterpret_cast<float4 *>(base + r * 8)[0] = make_float4(S[r], S[8 + r], S[16 + r], S[24 + r]);
                reinterpret_cast<float4 *>(base + r * 8)[1] = make_float4(S[32 + r], S[40 + r], S[48 + r], S[56 + r]);
            }
        }
        slot_sync<PIPE>();
        const float4 v0 = reinterpret_cast<const float4 *>(base + role * 8)[0], v1 = reinterpret_cast<const float4 *>(base + role * 8)[1];
        slot_sync<PIPE>();
        if (valid && !shadow) {
            column(state, 0) = v0.x; column(state, 1) = v0.y; column(state, 2) = v0.z; column(state, 3) = v0.w;
            column(state, 4) = v1.x; column(state, 5) = v1.y; column(state, 6) = v1.z; column(state, 7) = v1.w;
        }
    }
    if (writer) {
        float o[LIFT_OBS];
        write_observation(S, o);
        float4 *orow = reinterpret_cast<float4 *>(obs + (size_t)e * LIFT_OBS);
#pragma unroll
        for (int i = 0; i < LIFT_OBS / 4; ++i) orow[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
        reward[e] = total;
        terminated[e] = dropped ? 1 : 0;
        truncated[e] = time_out ? 1 : 0;
        if (do_reset) {     // rows tagged with the launch serial: the reduction (this step's or a deferred one) sums the LATEST tagged step
#pragma unroll
            for (int i = 0; i < 8; ++i) lg_out[(size_t)i * n + e] = lg[i];
            lg_out[(size_t)8 * n + e] = u2f(serial);
            atomicAdd(&counters[1], 1u);
            counters[2] = serial;                           // every writer of a launch stores the same value; later launches follow in stream order
        }
    }
    LIFT_STAMP(17);
}

// extras["log"]: means over the envs that reset in the latest step that had resets.  One workgroup; when nobody reset since the
// last reduction -- the common case -- it reads one counter and leaves.  Otherwise a fixed-order reduction over the rows tagged
// with that step's launch serial: env e is summed by thread e % 256, the threads are combined by a fixed tree => deterministic.
// log_out[0..7] keep their values between reductions; log_out[8] = number of envs that reset in the step `serial_now` (0 when the
// latest resets are older).  Run after every step (rover_lift_step, default) or on demand (rover_lift_set_log_deferred +
// rover_lift_flush_log: the same numbers at every flush, bit for bit, without a second launch per step).
// (Measured alternatives: the round-2 form, one workgroup walking all 10 x n floats every step, took 21 us -- as long as
// the step kernel; folding the reduction into the step kernel behind a "last wave" ticket cost ~5 us per step, because an
// agent-scope release on this multi-XCD part is an L2 write-back and every wave waits for its atomic's round trip.)
__global__ __launch_bounds__(256) void lift_log_kernel(lift_config c, int n, const float *__restrict__ lg, unsigned *counters,
                                                       float *__restrict__ log_out, uint32_t serial_now)
{
    __shared__ float part[4][9];
    const int t = threadIdx.x;
    if (counters[1] == 0u) {
        if (t == 0) log_out[8] = 0.0f;
        return;
    }
    const uint32_t tag = counters[2];
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.0f;
    for (int e = t; e < n; e += 256) {
        if (f2u(lg[(size_t)8 * n + e]) == tag) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += lg[(size_t)i * n + e];
            acc[8] += 1.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m, 64);   // fixed butterfly: deterministic
    }
    if ((t & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) part[t >> 6][i] = acc[i];
    }
    __syncthreads();
    if (t == 0) {
        float tot[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) tot[i] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
        const float cnt = tot[8];
#pragma unroll
        for (int i = 0; i < LIFT_NUM_REW; ++i) log_out[i] = tot[i] / cnt / c.max_episode_length_s;
        log_out[6] = tot[6];
        log_out[7] = tot[7];
        log_out[8] = tag == serial_now ? cnt : 0.0f;
        counters[1] = 0u;
    }
}

__global__ void lift_terms_kernel(lift_config c, int n, const float *obj_pos, const float *ee_pos, const float *root_state,
                                  const float *cmd, float *lifted, float *reach, float *goal, float *goal_fine, float *obj_pos_b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float op[3] = {obj_pos[3 * i], obj_pos[3 * i + 1], obj_pos[3 * i + 2]}, ep[3] = {ee_pos[3 * i], ee_pos[3 * i + 1], ee_pos[3 * i + 2]};
    float rs[7], cm[3], pb[3], l, r, g, gf;
    for (int k = 0; k < 7; ++k) rs[k] = root_state[13 * i + k];
    for (int k = 0; k < 3; ++k) cm[k] = cmd[7 * i + k];
    reference_terms(c, op, ep, rs, rs + 3, cm, l, r, g, gf, pb);
    reach[i] = r; lifted[i] = l; goal[i] = g; goal_fine[i] = gf;
    for (int k = 0; k < 3; ++k) obj_pos_b[3 * i + k] = pb[k];
}

struct DeviceGuardL {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuardL(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuardL() { if (switched) (void)hipSetDevice(prev); }
};

}  // namespace

struct rover_lift_sim {
    lift_config cfg;
    int n, env_id_offset, device;
    int lanes_per_env;      // 8; 16 (shadowed upper half-rows) is a measurement option
    float *state, *lg;
    unsigned *counters;     // [1] envs reset in the step under way (cleared by the log kernel)
    uint32_t log_serial;    // launch serial of the latest step (tags the log rows of the envs that reset in it); starts at 1
    bool log_deferred;      // rover_lift_set_log_deferred: rover_lift_step does not launch the log reduction
    int pipeline;           // eight lanes per env as two waves (arm one substep ahead of cube + managers): 1 on, 0 off, -1 = decide
                            // from the batch (on while the two waves of every eight envs all fit the chip at once)
    int simd_count;         // 4 x compute units of the device
    lift_config *cfg_dev;   // device copy of cfg (in the workspace), read by the tail of the step kernel
    bool cfg_dirty;         // host copy changed (rover_lift_set_seed): copy it over before the next launch
};

// The pipelined form doubles the waves (two per eight envs).  It wins while all of them are resident at once -- one wave per SIMD --
// and loses beyond (8192 envs on an MI355X: 2048 waves on 1024 SIMDs, 30.6 us against 22.0 us for the single-wave form).
static bool lift_pipelined(const rover_lift_sim *sim)
{
    if (sim->lanes_per_env != 8) return false;
    if (sim->pipeline >= 0) return sim->pipeline != 0;
    return 2 * ((sim->n + 7) / 8) <= (sim->simd_count > 0 ? sim->simd_count : 1024);
}
static void launch_step(rover_lift_sim *sim, hipStream_t st, const float *action, float *obs, float *reward, uint8_t *terminated,
                        uint8_t *truncated, float *log)
{
    if (sim->cfg_dirty) {       // rare (re-seeding): ordered on the launch stream, before the kernel that reads it
        (void)hipMemcpyAsync(sim->cfg_dev, &sim->cfg, sizeof(lift_config), hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);   // the source is the handle's own field: do not let a later set_seed race the copy
        sim->cfg_dirty = false;
    }
    const LiftHot hc = hot_of(sim->cfg);
    const uint32_t serial = ++sim->log_serial;
    const int lpe = sim->lanes_per_env, epw = 64 / lpe;
    const dim3 grid((sim->n + epw - 1) / epw);
#define LIFT_LAUNCH(L, P, THREADS)                                                                                                   \
    hipLaunchKernelGGL((lift_step_kernel<L, P>), grid, dim3(THREADS), 0, st, hc, sim->cfg_dev, sim->n, sim->env_id_offset, sim->state, \
                       action, obs, reward, terminated, truncated, sim->lg, sim->counters, serial)
    if (lpe == 16) LIFT_LAUNCH(16, false, 64);
    else if (lift_pipelined(sim)) LIFT_LAUNCH(8, true, 128);
    else LIFT_LAUNCH(8, false, 64);
#undef LIFT_LAUNCH
    if (!sim->log_deferred)
        hipLaunchKernelGGL(lift_log_kernel, dim3(1), dim3(256), 0, st, sim->cfg, sim->n, sim->lg, sim->counters, log, serial);
}

extern "C" {

int rover_lift_default_config(lift_config *c)
{
    if (!c) return rover_internal_fail(ROVER_ERR_INVALID, "cfg is NULL");
    memset(c, 0, sizeof(*c));
    c->sim_dt = 0.01f; c->decimation = 2; c->max_episode_length = 250; c->max_episode_length_s = 5.0f;    // manipulation_env_cfg.py:232-234
    c->action_scale = 0.5f; c->finger_open = 0.04f; c->finger_close = 0.0f;                                // joint_pos_env_cfg.py:35-43
    const float w[LIFT_NUM_REW] = {1.0f, 15.0f, 16.0f, 5.0f, 1.0e-3f, 1.0e-4f};                            // :120-144
    memcpy(c->rew_weight, w, sizeof(w));
    c->reach_std = 0.1f; c->goal_std = 0.3f; c->goal_fine_std = 0.05f; c->minimal_height = 0.06f;
    c->drop_height = -0.05f;                                                                               // :153
    c->cmd_lo[0] = 0.3f; c->cmd_hi[0] = 0.7f; c->cmd_lo[1] = 0.3f; c->cmd_hi[1] = 0.7f;                    // :170
    c->cmd_resample_time = 5.0f;
    c->obj_init[0] = 0.5f; c->obj_init[1] = 0.0f; c->obj_init[2] = 0.055f;                                 // joint_pos_env_cfg.py:51
    c->obj_range_lo[0] = -0.1f; c->obj_range_hi[0] = 0.1f; c->obj_range_lo[1] = -0.25f; c->obj_range_hi[1] = 0.25f;   // :185
    c->ee_offset_z = 0.1034f;                                                                              // joint_pos_env_cfg.py:78
    c->solver_iterations = 8;
    c->mu_table = 0.6f; c->mu_pad = 0.9f;
    return ROVER_OK;
}
size_t rover_lift_config_bytes(void) { return sizeof(lift_config); }
int rover_lift_state_words(void) { return LIFT_STATE_WORDS; }

int rover_lift_model_constants(float *out, int32_t cap)
{
    constexpr int KIND[7] = LF_KIND;
    constexpr float A[7] = LF_DH_A, D[7] = LF_DH_D, MASS[7] = LF_MASS, COM[7][3] = LF_COM, INERTIA[7][3] = LF_INERTIA;
    constexpr float Q_LO[7] = LF_Q_LO, Q_HI[7] = LF_Q_HI, QD_LIM[7] = LF_QD_LIM, EFFORT[7] = LF_EFFORT, QDEF[9] = LF_Q_DEFAULT;
    float t[160];
    int n = 0;
    for (int i = 0; i < 7; ++i) { t[n++] = (float)KIND[i]; t[n++] = A[i]; t[n++] = D[i]; t[n++] = MASS[i]; }
    for (int i = 0; i < 7; ++i) for (int k = 0; k < 3; ++k) t[n++] = COM[i][k];
    for (int i = 0; i < 7; ++i) for (int k = 0; k < 3; ++k) t[n++] = INERTIA[i][k];
    for (int i = 0; i < 7; ++i) { t[n++] = Q_LO[i]; t[n++] = Q_HI[i]; t[n++] = QD_LIM[i]; t[n++] = EFFORT[i]; }
    for (int i = 0; i < 9; ++i) t[n++] = QDEF[i];
    const float scalars[] = {K_GRAV, K_CUBE_HALF, K_CUBE_MASS, K_PAD_HALF_X, K_PAD_HALF_Z, K_FINGER_MASS, K_FINGER_KP, K_FINGER_KD,
                             K_FINGER_EFFORT, K_FINGER_VLIM, K_FINGER_TRAVEL, K_ARM_KP, K_ARM_KD, K_ARMATURE, K_BAUMGARTE, K_TORSION_R,
                             K_FLANGE_D, K_TABLE_MARGIN, K_PAD_MARGIN};
    for (unsigned i = 0; i < sizeof(scalars) / sizeof(scalars[0]); ++i) t[n++] = scalars[i];
    if (out) for (int i = 0; i < n && i < cap; ++i) out[i] = t[i];
    return n;
}

int rover_lift_create(const lift_config *cfg, int32_t num_envs, int32_t env_id_offset, int32_t device, rover_lift_sim **out)
{
    if (!cfg || !out) return rover_internal_fail(ROVER_ERR_INVALID, "cfg/out is NULL");
    if (num_envs <= 0 || env_id_offset < 0 || cfg->decimation < 0 || cfg->sim_dt <= 0.0f || cfg->solver_iterations < 0 ||
        cfg->max_episode_length <= 0)
        return rover_internal_fail(ROVER_ERR_INVALID, "invalid lift_config / num_envs");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return rover_internal_fail(ROVER_ERR_INVALID, "device ordinal out of range");
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    rover_lift_sim *s = new (std::nothrow) rover_lift_sim();
    if (!s) return rover_internal_fail(ROVER_ERR_INVALID, "out of host memory");
    s->cfg = *cfg; s->n = num_envs; s->env_id_offset = env_id_offset; s->device = device; s->state = nullptr; s->lg = nullptr;
    s->lanes_per_env = 8; s->counters = nullptr; s->cfg_dev = nullptr; s->cfg_dirty = false; s->log_serial = 0; s->log_deferred = false; s->pipeline = -1; s->simd_count = 4 * cus;
    *out = s;
    return ROVER_OK;
}
int rover_lift_destroy(rover_lift_sim *sim) { delete sim; return ROVER_OK; }
// per-env log contributions (9 x n floats, padded to 128 bytes) + two counters
static size_t lift_lg_bytes(const rover_lift_sim *sim) { return (((size_t)sim->n * 9 * sizeof(float)) + 127) & ~(size_t)127; }
// ... + 128 bytes of counters + a device copy of the configuration
size_t rover_lift_workspace_bytes(const rover_lift_sim *sim) { return sim ? lift_lg_bytes(sim) + 128 + ((sizeof(lift_config) + 127) & ~(size_t)127) : 0; }
int rover_lift_bind(rover_lift_sim *sim, float *state, void *workspace, size_t workspace_bytes)
{
    if (!sim || !state || !workspace) return rover_internal_fail(ROVER_ERR_INVALID, "NULL argument");
    if (workspace_bytes < rover_lift_workspace_bytes(sim)) return rover_internal_fail(ROVER_ERR_INVALID, "workspace too small");
    if (reinterpret_cast<uintptr_t>(workspace) & 127) return rover_internal_fail(ROVER_ERR_INVALID, "workspace must be 128-byte aligned");
    DeviceGuardL guard(sim->device);
    sim->state = state;
    sim->lg = static_cast<float *>(workspace);
    sim->counters = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + lift_lg_bytes(sim));
    sim->cfg_dev = reinterpret_cast<lift_config *>(static_cast<char *>(workspace) + lift_lg_bytes(sim) + 128);
    HIP_TRY(hipMemset(sim->counters, 0, 128));     // init-time, synchronous: the log kernel keeps them at zero afterwards
    HIP_TRY(hipMemset(sim->lg + (size_t)8 * sim->n, 0, (size_t)sim->n * sizeof(float)));   // no row carries a tag yet (serials start at 1)
    sim->log_serial = 0;
    HIP_TRY(hipMemcpy(sim->cfg_dev, &sim->cfg, sizeof(lift_config), hipMemcpyHostToDevice));
    sim->cfg_dirty = false;
    return ROVER_OK;
}
int rover_lift_set_seed(rover_lift_sim *sim, uint32_t seed_lo, uint32_t seed_hi)
{
    if (!sim) return rover_internal_fail(ROVER_ERR_INVALID, "sim is NULL");
    sim->cfg.seed_lo = seed_lo;
    sim->cfg.seed_hi = seed_hi;
    sim->cfg_dirty = true;
    return ROVER_OK;
}
// measurement hooks (tools/lift_time.py): lanes per env of the step kernel, 8 (default) or 16; the two-wave pipelined form of
// the eight-lane kernel (default) or the single-wave one
int rover_lift_debug_set_lanes(rover_lift_sim *sim, int lanes)
{
    if (!sim || (lanes != 8 && lanes != 16)) return ROVER_ERR_INVALID;
    sim->lanes_per_env = lanes;
    return ROVER_OK;
}
int rover_lift_debug_set_pipeline(rover_lift_sim *sim, int on)
{
    if (!sim) return ROVER_ERR_INVALID;
    sim->pipeline = on < 0 ? -1 : (on != 0);
    return ROVER_OK;
}
#ifdef LF_STAMP
int rover_lift_debug_set_stamps(void *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lift_stamps), &buf, sizeof(buf)) == hipSuccess ? ROVER_OK : ROVER_ERR_HIP;
}
#endif
int rover_lift_reset(rover_lift_sim *sim, float *obs, void *stream)
{
    if (!sim || !sim->state) return rover_internal_fail(ROVER_ERR_STATE, "rover_lift_bind has not been called");
    if (!obs) return rover_internal_fail(ROVER_ERR_INVALID, "obs is NULL");
    DeviceGuardL guard(sim->device);
    hipLaunchKernelGGL(lift_reset_kernel, dim3((sim->n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), sim->cfg, sim->n,
                       sim->env_id_offset, sim->state, obs);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}
int rover_lift_step(rover_lift_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated, uint8_t *truncated,
                    float *log, void *stream)
{
    if (!sim || !sim->state) return rover_internal_fail(ROVER_ERR_STATE, "rover_lift_bind has not been called");
    if (!action || !obs || !reward || !terminated || !truncated || !log) return rover_internal_fail(ROVER_ERR_INVALID, "NULL buffer");
    if ((reinterpret_cast<uintptr_t>(action) & 15) || (reinterpret_cast<uintptr_t>(obs) & 15))
        return rover_internal_fail(ROVER_ERR_INVALID, "action / obs must be 16-byte aligned");
    DeviceGuardL guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    launch_step(sim, st, action, obs, reward, terminated, truncated, log);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}
int rover_lift_set_log_deferred(rover_lift_sim *sim, int32_t deferred)
{
    if (!sim) return rover_internal_fail(ROVER_ERR_INVALID, "sim is NULL");
    sim->log_deferred = deferred != 0;
    return ROVER_OK;
}
int rover_lift_flush_log(rover_lift_sim *sim, float *log, void *stream)
{
    if (!sim || !sim->state) return rover_internal_fail(ROVER_ERR_STATE, "rover_lift_bind has not been called");
    if (!log) return rover_internal_fail(ROVER_ERR_INVALID, "log is NULL");
    DeviceGuardL guard(sim->device);
    hipLaunchKernelGGL(lift_log_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sim->cfg, sim->n, sim->lg, sim->counters,
                       log, sim->log_serial);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}
int rover_lift_profile_step(rover_lift_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated,
                            uint8_t *truncated, float *log, void *stream, float *ms_step_kernel, float *ms_event_overhead)
{
    if (!sim || !sim->state) return rover_internal_fail(ROVER_ERR_STATE, "rover_lift_bind has not been called");
    if (!action || !obs || !reward || !terminated || !truncated || !log || !ms_step_kernel || !ms_event_overhead)
        return rover_internal_fail(ROVER_ERR_INVALID, "NULL buffer");
    if ((reinterpret_cast<uintptr_t>(action) & 15) || (reinterpret_cast<uintptr_t>(obs) & 15))
        return rover_internal_fail(ROVER_ERR_INVALID, "action / obs must be 16-byte aligned");
    DeviceGuardL guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t ev[3];
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreate(&ev[i]));
    HIP_TRY(hipEventRecord(ev[0], st));
    launch_step(sim, st, action, obs, reward, terminated, truncated, log);
    HIP_TRY(hipEventRecord(ev[1], st));
    HIP_TRY(hipEventRecord(ev[2], st));              // empty pair: the fixed cost an event interval carries
    HIP_TRY(hipEventSynchronize(ev[2]));
    HIP_TRY(hipEventElapsedTime(ms_step_kernel, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(ms_event_overhead, ev[1], ev[2]));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventDestroy(ev[i]));
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}
int rover_lift_kernel_name(const rover_lift_sim *sim, char *step_kernel, size_t cap)
{
    if (!sim || !step_kernel || cap < 8) return rover_internal_fail(ROVER_ERR_INVALID, "bad argument");
    snprintf(step_kernel, cap, "lift_step_kernel<%d, %s>", sim->lanes_per_env, lift_pipelined(sim) ? "true" : "false");
    return ROVER_OK;
}
int rover_lift_terms(rover_lift_sim *sim, int32_t n, const float *obj_pos, const float *ee_pos, const float *root_state,
                     const float *cmd, float *lifted, float *reach, float *goal, float *goal_fine, float *obj_pos_b, void *stream)
{
    if (!sim || n <= 0 || !obj_pos || !ee_pos || !root_state || !cmd || !lifted || !reach || !goal || !goal_fine || !obj_pos_b)
        return rover_internal_fail(ROVER_ERR_INVALID, "bad argument");
    DeviceGuardL guard(sim->device);
    hipLaunchKernelGGL(lift_terms_kernel, dim3((n + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), sim->cfg, n, obj_pos,
                       ee_pos, root_state, cmd, lifted, reach, goal, goal_fine, obj_pos_b);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

}  // extern "C"
