/*
 * rover_policy.h -- C ABI of the fused policy / value network forward pass (librover_hip.so), SURVEY 8(f-3).
 *
 * Replaces, for inference, the torch modules of
 *     rover_envs/envs/navigation/learning/skrl/models.py
 *         HeightmapEncoder                     :24-36    (Linear + activation stack over the height scan)
 *         GaussianNeuralNetwork.compute        :89-102   (policy mean: encoder ++ proprioception -> MLP -> tanh)
 *         DeterministicNeuralNetwork.compute   :150-161  (value: same trunk, one output, no tanh)
 * built by rover_envs/learning/train/get_models.py:36-62 as encoder 961 -> 80 -> 60, MLP (4 + 60) -> 256 -> 160 ->
 * 128 -> {2, 1}, LeakyReLU(0.01).  The reference's slicing quirk is part of the contract: the encoder reads observation
 * columns [prop_dim - 1, obs_dim - 1) (models.py:95: `states[:, self.mlp_input_size - 1:-1]`), i.e. it starts ONE
 * column early (the heading term) and drops the last ray -- `enc_offset` / `enc_dim` below express it.
 *
 * One launch evaluates one network on a batch of observation rows: a 512-thread workgroup owns 16 rows, keeps them and
 * every activation in LDS and runs every layer on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered
 * fmaf chain), weights streamed from a fragment-ordered packed buffer.  Conventions as in rover_hip.h (plain C,
 * caller-owned device buffers, int return codes, rover_last_error()).
 */
#ifndef ROVER_POLICY_H
#define ROVER_POLICY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ROVER_POLICY_MAX_LAYERS = 8 };
enum { ROVER_ACT_NONE = 0, ROVER_ACT_LEAKY_RELU = 1, ROVER_ACT_TANH = 2 };

typedef struct rover_policy_layer {
    int32_t K, N;        /* in / out features of the Linear (torch weight shape (N, K)) */
    int32_t act;         /* ROVER_ACT_* applied after the bias */
    int32_t split_k;     /* 1: K is cut into 8 contiguous ranges of ceil(ceil(K / 16) / 8) * 16 inputs (one per wave) whose
                            partial sums are combined as ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)); 0: one fmaf
                            chain over k = 0 .. K-1.  Part of the numerics contract. */
    uint32_t w_off, b_off; /* offsets (in floats) of the packed weights / bias in the packed buffer; set by rover_policy_pack */
} rover_policy_layer;

typedef struct rover_policy_desc {
    int32_t obs_dim;     /* observation row width (965); rows are contiguous */
    int32_t prop_dim;    /* proprioceptive columns [0, prop_dim) fed to the MLP (4) */
    int32_t enc_offset;  /* first observation column the encoder reads (3 = prop_dim - 1, the reference's quirk) */
    int32_t enc_dim;     /* encoder input width (961); 0 = no encoder, the MLP reads columns [0, prop_dim) only */
    int32_t n_enc;       /* encoder layers (layers[0 .. n_enc)) */
    int32_t n_mlp;       /* MLP layers (layers[n_enc .. n_enc + n_mlp)); MLP input = prop_dim + encoder output */
    float leaky_slope;   /* 0.01 (torch.nn.LeakyReLU default) */
    rover_policy_layer layers[ROVER_POLICY_MAX_LAYERS];
} rover_policy_desc;

/* Fills `d` with the reference architecture (get_models.py:36-62): out_dim 2 + final tanh = policy mean,
 * out_dim 1 + no final activation = value. */
int rover_policy_default_desc(rover_policy_desc *d, int32_t out_dim, int32_t final_tanh);

/* Host-side packing (pure CPU, no GPU needed): weights[i] = torch `weight` of layer i, row-major (N, K); biases[i] (N,).
 * Sets w_off / b_off in `d` and writes rover_policy_packed_floats(d) floats to `packed` (host memory), which the caller
 * uploads once. */
size_t rover_policy_packed_floats(const rover_policy_desc *d);
int rover_policy_pack(rover_policy_desc *d, const float *const *weights, const float *const *biases, float *packed);

/* out[row, 0 .. N_last) = network(obs[row, :]) for row < n.  obs (n, obs_dim) and out (n, N_last) are device pointers,
 * `packed` the uploaded packed buffer (16-byte aligned), present `n_copies` (>= 1) times back to back: every workgroup
 * streams all weights in lock step with the others, and replicas (workgroup b reads replica b % n_copies) spread those
 * reads over the L2 channels.  Asynchronous on `stream`. */
int rover_policy_forward(const rover_policy_desc *d, const float *packed, int32_t n_copies, const float *obs, int32_t n,
                         float *out, void *stream);

/* Two networks on the same observation rows in one launch -- the policy mean and the value a PPO rollout step asks for
 * (skrl calls policy.act and value.act on the same states, rover_envs/utils/skrl_utils.py:114-135): the rows are fetched and
 * staged once, one launch boundary disappears.  Each network's arithmetic is exactly rover_policy_forward's.  Both descriptors
 * must be the reference architecture (get_models.py:36-62, rover_policy_default_desc); ROVER_ERR_UNSUPPORTED otherwise --
 * call rover_policy_forward twice then.  out_a (n, N_last of a), out_b (n, N_last of b). */
int rover_policy_forward_pair(const rover_policy_desc *da, const float *packed_a, const rover_policy_desc *db, const float *packed_b,
                              int32_t n_copies, const float *obs, int32_t n, float *out_a, float *out_b, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_POLICY_H */
