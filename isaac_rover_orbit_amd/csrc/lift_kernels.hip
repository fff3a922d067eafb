// lift_kernels.hip -- FrankaCubeLift-v0 step()/reset() on gfx950 (BASELINE config 5) + its C ABI.
//
// One env per lane, SoA state (every state access is a coalesced wave access).  The model -- 7-DOF arm with full joint-space
// dynamics (recursive Newton-Euler, Cholesky-solved implicit PD), two-finger gripper, 6-DOF cube with table / finger-pad
// contacts (projected Gauss-Seidel), the task's MDP -- is defined in lift_model.h (plain C, also compiled into the CPU
// checker).  No MFMA here either: per-env 7x7 systems and contact rows, nothing batch-contractible; the path is
// latency-bound at N = 2048 (32 waves) by construction.  All arithmetic fp32, -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/rover_hip.h"
#include "../../include/rover_lift.h"
#include "lift_model.h"
#include "rover_internal.hpp"

namespace {

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, #expr ": %s", hipGetErrorString(_e));         \
    } while (0)

__global__ __launch_bounds__(64) void lift_reset_kernel(lift_config c, int n, int env_id_offset, float *__restrict__ state,
                                                        float *__restrict__ obs)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n) return;
    float S[LIFT_STATE_WORDS];
#pragma unroll
    for (int i = 0; i < LIFT_STATE_WORDS; ++i) S[i] = state[(size_t)i * n + e];
    lift_reset_one(&c, S, (uint32_t)(env_id_offset + e));
#pragma unroll
    for (int i = 0; i < LIFT_STATE_WORDS; ++i) state[(size_t)i * n + e] = S[i];
    float o[LIFT_OBS];
    lift_write_obs(&c, S, o);
#pragma unroll
    for (int i = 0; i < LIFT_OBS; ++i) obs[(size_t)e * LIFT_OBS + i] = o[i];
}

__global__ __launch_bounds__(64) void lift_step_kernel(lift_config c, int n, int env_id_offset, float *__restrict__ state,
                                                       const float *__restrict__ action, float *__restrict__ obs,
                                                       float *__restrict__ reward, uint8_t *__restrict__ terminated,
                                                       uint8_t *__restrict__ truncated, float *__restrict__ lg_out)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n) return;
    float S[LIFT_STATE_WORDS];
#pragma unroll
    for (int i = 0; i < LIFT_STATE_WORDS; ++i) S[i] = state[(size_t)i * n + e];
    float a[LIFT_ACT], o[LIFT_OBS], lg[10], r;
    uint8_t term, trunc;
#pragma unroll
    for (int i = 0; i < LIFT_ACT; ++i) a[i] = action[(size_t)e * LIFT_ACT + i];
    lift_step_one(&c, S, a, (uint32_t)(env_id_offset + e), o, &r, &term, &trunc, lg);
#pragma unroll
    for (int i = 0; i < LIFT_STATE_WORDS; ++i) state[(size_t)i * n + e] = S[i];
#pragma unroll
    for (int i = 0; i < LIFT_OBS; ++i) obs[(size_t)e * LIFT_OBS + i] = o[i];
    reward[e] = r;
    terminated[e] = term;
    truncated[e] = trunc;
#pragma unroll
    for (int i = 0; i < 10; ++i) lg_out[(size_t)i * n + e] = lg[i];
}

// deterministic reduction of the per-env log contributions (10 x n) in a fixed order: one workgroup, strided partials
__global__ __launch_bounds__(256) void lift_log_kernel(lift_config c, int n, const float *__restrict__ lg, float *__restrict__ log_out)
{
    __shared__ float part[10][256];
    const int t = threadIdx.x;
    for (int w = 0; w < 10; ++w) {
        float acc = 0.0f;
        for (int e = t; e < n; e += 256) acc += lg[(size_t)w * n + e];
        part[w][t] = acc;
    }
    __syncthreads();
    if (t < 10) {
        float s = 0.0f;
        for (int k = 0; k < 256; ++k) s += part[t][k];
        part[t][0] = s;
    }
    __syncthreads();
    if (t < 9) {
        const float cnt = part[8][0];
        if (t == 8) log_out[8] = cnt;
        else if (cnt > 0.0f) log_out[t] = t < LIFT_NUM_REW ? part[t][0] / cnt / c.max_episode_length_s : part[t][0];
    }
}

__global__ void lift_terms_kernel(lift_config c, int n, const float *obj_pos, const float *ee_pos, const float *root_state,
                                  const float *cmd, float *lifted, float *reach, float *goal, float *goal_fine, float *obj_pos_b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float rew[LIFT_NUM_REW], pb[3];
    uint8_t term[LIFT_NUM_TERM];
    const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float op[3] = {obj_pos[3 * i], obj_pos[3 * i + 1], obj_pos[3 * i + 2]}, ep[3] = {ee_pos[3 * i], ee_pos[3 * i + 1], ee_pos[3 * i + 2]};
    float rs[7], cm[3];
    for (int k = 0; k < 7; ++k) rs[k] = root_state[13 * i + k];
    for (int k = 0; k < 3; ++k) cm[k] = cmd[7 * i + k];
    lift_terms_one(&c, op, ep, rs, rs + 3, cm, zero8, zero8, zero9, 0, rew, term, pb);
    reach[i] = rew[0]; lifted[i] = rew[1]; goal[i] = rew[2]; goal_fine[i] = rew[3];
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
    float *state, *lg;
};

extern "C" {

int rover_lift_default_config(lift_config *cfg)
{
    if (!cfg) return rover_internal_fail(ROVER_ERR_INVALID, "cfg is NULL");
    memset(cfg, 0, sizeof(*cfg));
    lift_default_config(cfg);
    return ROVER_OK;
}
size_t rover_lift_config_bytes(void) { return sizeof(lift_config); }
int rover_lift_state_words(void) { return LIFT_STATE_WORDS; }

int rover_lift_create(const lift_config *cfg, int32_t num_envs, int32_t env_id_offset, int32_t device, rover_lift_sim **out)
{
    if (!cfg || !out) return rover_internal_fail(ROVER_ERR_INVALID, "cfg/out is NULL");
    if (num_envs <= 0 || env_id_offset < 0 || cfg->decimation < 0 || cfg->sim_dt <= 0.0f || cfg->solver_iterations < 0 ||
        cfg->max_episode_length <= 0)
        return rover_internal_fail(ROVER_ERR_INVALID, "invalid lift_config / num_envs");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return rover_internal_fail(ROVER_ERR_INVALID, "device ordinal out of range");
    rover_lift_sim *s = new (std::nothrow) rover_lift_sim();
    if (!s) return rover_internal_fail(ROVER_ERR_INVALID, "out of host memory");
    s->cfg = *cfg; s->n = num_envs; s->env_id_offset = env_id_offset; s->device = device; s->state = nullptr; s->lg = nullptr;
    *out = s;
    return ROVER_OK;
}
int rover_lift_destroy(rover_lift_sim *sim) { delete sim; return ROVER_OK; }
size_t rover_lift_workspace_bytes(const rover_lift_sim *sim) { return sim ? (size_t)sim->n * 10 * sizeof(float) : 0; }
int rover_lift_bind(rover_lift_sim *sim, float *state, void *workspace, size_t workspace_bytes)
{
    if (!sim || !state || !workspace) return rover_internal_fail(ROVER_ERR_INVALID, "NULL argument");
    if (workspace_bytes < rover_lift_workspace_bytes(sim)) return rover_internal_fail(ROVER_ERR_INVALID, "workspace too small");
    sim->state = state;
    sim->lg = static_cast<float *>(workspace);
    return ROVER_OK;
}
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
    DeviceGuardL guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lift_step_kernel, dim3((sim->n + 63) / 64), dim3(64), 0, st, sim->cfg, sim->n, sim->env_id_offset, sim->state,
                       action, obs, reward, terminated, truncated, sim->lg);
    hipLaunchKernelGGL(lift_log_kernel, dim3(1), dim3(256), 0, st, sim->cfg, sim->n, sim->lg, log);
    HIP_TRY(hipGetLastError());
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
