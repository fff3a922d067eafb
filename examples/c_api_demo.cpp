// c_api_demo.cpp -- the rover hot path driven from plain C++ through include/rover_hip.h: no Python, no torch.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_api_demo.cpp -o c_api_demo \
//         -Lisaac_rover_orbit_amd -lrover_hip -Wl,-rpath,$PWD/isaac_rover_orbit_amd
//   ./c_api_demo [num_envs] [steps]
//
// Flat 1024 x 1024 terrain, a constant "drive forward, turn slightly" action, prints the mean reward and a checksum of the
// last observation (tests/test_gpu_boundary.py compares it with the Python binding on the same configuration).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rover_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ROVER_OK_(x) do { int rc_ = (x); if (rc_ != ROVER_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, rover_last_error()); return 3; } } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256, steps = argc > 2 ? atoi(argv[2]) : 50;
    const int H = 1024, W = 1024, n_spawns = 2 * n;
    rover_config cfg;
    ROVER_OK_(rover_default_config(&cfg));
    const int obs_w = 4 + cfg.scan_nx * cfg.scan_ny;

    // shared terrain data: flat ground, no rocks, spawns on a diagonal inside the 20 m border
    std::vector<float> height((size_t)H * W, 0.0f), obstacle((size_t)H * W, 0.0f), spawns((size_t)n_spawns * 3);
    std::vector<uint8_t> mask((size_t)H * W, 0);
    for (int i = 0; i < n_spawns; ++i) {
        spawns[3 * i + 0] = 21.0f + 9.0f * (float)i / (float)n_spawns;
        spawns[3 * i + 1] = 30.0f - 9.0f * (float)i / (float)n_spawns;
        spawns[3 * i + 2] = 0.0f;
    }
    float *d_height, *d_obstacle, *d_spawns, *d_state, *d_obs, *d_rew, *d_act, *d_log;
    uint8_t *d_mask, *d_term, *d_trunc;
    void *d_ws;
    HIP_OK(hipMalloc(&d_height, height.size() * 4)); HIP_OK(hipMalloc(&d_obstacle, obstacle.size() * 4));
    HIP_OK(hipMalloc(&d_mask, mask.size())); HIP_OK(hipMalloc(&d_spawns, spawns.size() * 4));
    HIP_OK(hipMemcpy(d_height, height.data(), height.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_obstacle, obstacle.data(), obstacle.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_spawns, spawns.data(), spawns.size() * 4, hipMemcpyHostToDevice));

    rover_sim *sim = nullptr;
    ROVER_OK_(rover_create(&cfg, n, 0, 0, &sim));
    ROVER_OK_(rover_set_terrain(sim, d_height, d_obstacle, d_mask, H, W, 0.05f, 0.0f, 0.0f, d_spawns, n_spawns));
    const size_t ws = rover_workspace_bytes(sim);
    HIP_OK(hipMalloc(&d_state, (size_t)ROVER_STATE_WORDS * n * 4)); HIP_OK(hipMalloc(&d_ws, ws));
    HIP_OK(hipMemset(d_ws, 0, ws));
    std::vector<float> state((size_t)ROVER_STATE_WORDS * n, 0.0f);
    for (int e = 0; e < n; ++e) state[(size_t)ROVER_QUAT * n + e] = 1.0f;      // identity quaternion (w first)
    HIP_OK(hipMemcpy(d_state, state.data(), state.size() * 4, hipMemcpyHostToDevice));
    ROVER_OK_(rover_bind(sim, d_state, d_ws, ws));
    HIP_OK(hipMalloc(&d_obs, (size_t)n * obs_w * 4)); HIP_OK(hipMalloc(&d_rew, (size_t)n * 4));
    HIP_OK(hipMalloc(&d_act, (size_t)n * 2 * 4)); HIP_OK(hipMalloc(&d_log, ROVER_LOG_WORDS * 4));
    HIP_OK(hipMalloc(&d_term, n)); HIP_OK(hipMalloc(&d_trunc, n));
    HIP_OK(hipMemset(d_log, 0, ROVER_LOG_WORDS * 4));
    std::vector<float> act((size_t)n * 2);
    for (int e = 0; e < n; ++e) { act[2 * e] = 0.8f; act[2 * e + 1] = 0.1f; }
    HIP_OK(hipMemcpy(d_act, act.data(), act.size() * 4, hipMemcpyHostToDevice));

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    ROVER_OK_(rover_reset(sim, d_obs, st));
    double rew_sum = 0.0;
    std::vector<float> rew(n);
    for (int k = 0; k < steps; ++k) {
        ROVER_OK_(rover_step(sim, d_act, d_obs, d_rew, d_term, d_trunc, nullptr, d_log, st));
        HIP_OK(hipMemcpyAsync(rew.data(), d_rew, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (int e = 0; e < n; ++e) rew_sum += rew[e];
    }
    std::vector<float> obs((size_t)n * obs_w);
    HIP_OK(hipMemcpy(obs.data(), d_obs, obs.size() * 4, hipMemcpyDeviceToHost));
    uint32_t checksum = 2166136261u;                       // FNV-1a over the raw bits of the last observation
    for (float v : obs) { uint32_t b; memcpy(&b, &v, 4); for (int i = 0; i < 4; ++i) { checksum ^= (b >> (8 * i)) & 0xFF; checksum *= 16777619u; } }
    printf("%s | envs %d steps %d | mean reward %.9g | obs checksum %08x\n", rover_version(), n, steps, rew_sum / ((double)n * steps), checksum);
    rover_destroy(sim);
    return 0;
}
