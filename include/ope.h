/*
 * ope.h -- C ABI of the MI355X-native off-policy MARL update engine ("ope").
 *
 * The reference (marlbenchmark/off-policy) is 100 % Python and has no FFI/plugin layer; the
 * drop-in boundary is the duck-typed Python surface of its replay buffers and trainers. This header declares
 * the C-ABI that sits UNDER our mirrors of those classes: plain pointers and sizes, no torch types, one
 * `extern "C"` shared library (libope.so, HIP/gfx950). Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is DEVICE memory unless the name ends in `_host`;
 *   - pointers are borrowed for the duration of the call; the caller (Python/torch) owns all memory;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls only enqueue work on it;
 *   - return value: 0 on success, negative OPE_E* on error; nothing throws, nothing allocates;
 *   - all data is float32 (the reference's buffers are np.float32, rec_buffer.py:120-141), indices int64/int32.
 */
#ifndef OPE_H_
#define OPE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPE_VERSION 1

#define OPE_OK 0
#define OPE_EINVAL (-1)  /* bad argument / unsupported dimension                      */
#define OPE_ELAUNCH (-2) /* kernel launch failed (hipGetLastError != hipSuccess)     */
#define OPE_ENOSPC (-3)  /* caller-provided workspace too small                       */
#define OPE_EHIP (-4)    /* a HIP runtime call (allocation, IPC) failed                */

int ope_version(void);
const char* ope_strerror(int code);
/* sizeof() of the structs of this header as the library was compiled, for bindings to check their mirror declarations against
 * (a ctypes / cgo / JNI struct that drifted would otherwise be read as garbage without an error). `name` is the type's name
 * ("ope_dims", "ope_rddpg_cfg", ...); -1 for an unknown name. */
int64_t ope_abi_sizeof(const char* name);

/* ------------------------------------------------------------------------------------------------
 * Episode dimensions (one policy): N agents, A actions (one-hot width), D obs, S centralized state, T steps.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ope_dims {
  int32_t n_agents;       /* N */
  int32_t act_dim;        /* A */
  int32_t obs_dim;        /* D */
  int32_t state_dim;      /* S */
  int32_t episode_length; /* T */
  int32_t layer_N;        /* args.layer_N (config.py:65): hidden blocks (Linear + ReLU + LayerNorm) behind fc1 in the agent network's MLP base
                           * (mlp.py:14-28). 0 or 1 = the reference default; 2 = a second block `rnn.mlp.fc2.1.*` (4 more tensors between
                           * fc2.0 and the GRU): recurrent QMIX / VDN with one shared policy only, anything else returns OPE_EINVAL */
  int32_t flags;          /* OPE_DIMS_* bits; 0 = the reference defaults */
} ope_dims;
/* args.use_feature_normalization = False (config.py:69, mlp.py:60-62, 77-78): the agent network has no input LayerNorm. The flat
 * parameter layout is UNCHANGED -- the two feature_norm slots stay where they are and must hold ones / zeros (the host mirrors do not
 * expose them as parameters; their gradient is written as zero) -- and the first-layer kernels take the rows as they are (mean 0,
 * 1/std 1): x * 1 + 0 is exact, so every other kernel and the weight-gradient identities run unchanged. Recurrent and (round 5) MLP
 * QMIX / VDN nets (ope_qmix_cfg with or without .mlp, ope_agent_forward, ope_agent_forward_mlp). */
#define OPE_DIMS_NO_FEATURE_NORM 1
/* args.use_ReLU = False (config.py:67, mlp.py:9-12): tanh instead of ReLU behind fc1 and fc2 of the agent network's MLP base (the
 * hyper-networks keep their ReLU, q_mixer.py). Recurrent and (round 5) MLP QMIX / VDN nets with one hidden block and an input width <= 384: the
 * register-resident trunk kernels (trunk_fwd3 / trunk_bwd3) carry the activation -- tanh as 2 / (1 + 2^(-2 log2(e) x)) - 1, its
 * derivative 1 - a^2 rebuilt from the saved normalised activation, 1/std and the row mean (kept in the slot of the unused ReLU mask). */
#define OPE_DIMS_TANH 2
/* Plain (non double-Q) targets take the maximum over the AVAILABLE actions only (avail_acts at t + 1), as the double-Q greedy choice
 * always does. The reference's Discrete QMIX / VDN does not (qmix.py:148: target_policy.get_actions(..., available_actions=None)); the
 * flag exists for MultiDiscrete action spaces on the accelerated path, where every (agent, sub-action) pair is presented to the kernels
 * as one agent whose availability mask is its sub-action's block of the stacked q head (QMixPolicy.py:76-93: per-head max). */
#define OPE_DIMS_MASK_TARGET_MAX 4

/* Seven per-episode fields, in the order of RecPolicyBuffer.sample_inds' return tuple
 * (offpolicy/utils/rec_buffer.py:192-240): obs, share_obs, acts, rewards, dones, dones_env, avail_acts -- plus the
 * transition buffers' valid_transition flag (offpolicy/utils/mlp_buffer.py:94,196), which the store copies in the same launch.
 * A NULL pointer = field not stored / not wanted. */
typedef struct ope_fields {
  float* obs;        /* [.., T+1, N, D]                                    */
  float* share_obs;  /* [.., T+1, S]      (use_same_share_obs)             */
  float* acts;       /* [.., T,   N, A]   one-hot float32                  */
  float* rewards;    /* [.., T,   N, 1]                                    */
  float* dones;      /* [.., T,   N, 1]                                    */
  float* dones_env;  /* [.., T,   1]                                       */
  float* avail_acts; /* [.., T+1, N, A]                                    */
  float* valid_transition; /* [.., T, N, 1]   MlpPolicyBuffer only; the training entry points ignore it */
} ope_fields;

/* ------------------------------------------------------------------------------------------------
 * Replay store: device-resident, EPISODE-major rings  store.f[e][t][agent][dim]  (the reference keeps
 * time-major numpy rings [T(+1), cap, N, dim], rec_buffer.py:120-141; sampling is by episode, so the device
 * layout makes every sampled episode one contiguous run per field).
 *
 * ope_store_insert  replaces RecPolicyBuffer.insert's ring write (rec_buffer.py:167-185).
 *   `staged` holds `n_insert` episodes already on the device in the layout insert() receives
 *   ([T(+1), n_insert, N, dim]; share_obs with the agent axis already dropped: [T+1, n_insert, S]);
 *   `slots[n_insert]` (device int64) are the ring slots (idx_range) computed by the host.
 * ope_store_gather  replaces RecPolicyBuffer.sample_inds' fancy-index + transpose (rec_buffer.py:206-238):
 *   out.f is written as [T(+1), N, B, dim] (agent-indexed fields) / [T(+1), B, dim] (share_obs, dones_env),
 *   i.e. exactly the memory the reference's `[N, T(+1), B, dim]` transpose-VIEW aliases, and at the same
 *   time the row-stacked `[T(+1), N*B, dim]` tensor QMix.train_policy_on_batch builds with torch.cat
 *   (qmix.py:108-109). Bit-exact copy; indices may repeat.
 * ---------------------------------------------------------------------------------------------- */
/* Device-resident prioritized replay bookkeeping: float64 sum / min segment trees + running max priority.
 * Replaces SumSegmentTree / MinSegmentTree (offpolicy/utils/segment_tree.py:18-165) and the tree traffic of
 * PrioritizedRecReplayBuffer.insert / sample / update_priorities (rec_buffer.py:262-324; mlp_buffer.py likewise), so the
 * prioritized configurations need no host round trip per update. capacity must be a power of two.
 *   ope_per_tree_set     leaves[idx[i]] = v_i ** alpha, v_i = priorities[i] (update_priorities; also raises the running max)
 *                        or, with priorities == NULL, the running max priority (insert); touched paths re-reduced.
 *                        Duplicate indices: the last one wins (numpy fancy assignment).
 *   ope_per_tree_sample  idx[i] = find_prefixsum_idx(mass01[i] * sum(leaves[0, filled-1))) with mass01 in [0,1) drawn by the
 *                        caller (the reference uses np.random.random); weights[i] = (p_i filled)^-beta / max_w (or NULL). */
int64_t ope_per_tree_bytes(int32_t capacity);
int ope_per_tree_init(void* trees, int32_t capacity, void* stream);
int ope_per_tree_set(void* trees, int32_t capacity, const int64_t* idx, const float* priorities, double alpha, int32_t n, void* stream);
int ope_per_tree_sample(const void* trees, int32_t capacity, int32_t filled, const double* mass01, double beta, int32_t n,
                        int64_t* idx_out, float* weights_out, void* stream);
/* The same with `filled` (device int32[1], clamped to [2, capacity]) and `beta` (device double[1]) read at run time: the launch
 * is identical from call to call, so a prioritized update can be captured in a HIP graph (MADDPG.make_graphed_step with PER: the
 * buffer refreshes `filled` on insert, the caller writes the annealed beta before a replay). */
int ope_per_tree_sample_dev(const void* trees, int32_t capacity, const int32_t* filled_dev, const double* mass01, const double* beta_dev,
                            int32_t n, int64_t* idx_out, float* weights_out, void* stream);

/* Reward normalisation (use_reward_normalization): statistics over the FILLED part of the reward ring
 *   episodes  (rec_buffer.py:209-222): nanmean / nanstd over steps whose previous step did not end the episode
 *             (dones_env[t-1] != 1; step 0 always counts), all agents;  pass the store's dones_env ring
 *   transitions (mlp_buffer.py:229-231): plain mean / std over rewards[:filled];  pass dones_env = NULL (T = 1)
 * stats_out (device float[4]) = {mean, population std, count, 0}; accumulation in double, fixed order.
 * ope_reward_normalize applies (r - mean) / std in place to a gathered reward block (rec_buffer.py:221-222). */
int64_t ope_reward_stats_scratch_bytes(void);
int ope_store_reward_stats(const ope_dims* dims, int32_t filled, const float* rewards, const float* dones_env, void* scratch,
                           float* stats_out, void* stream);
int ope_reward_normalize(float* rewards, int64_t n, const float* stats, void* stream);
/* Index checking: `slots` / `inds` are device data, so a bad value cannot be rejected on the host without a sync. Every
 * index is checked against [0, capacity) IN the kernel: rows of an out-of-range index are skipped (nothing is read or
 * written out of bounds) and *bad_index_flag (device int32, may be NULL, never cleared by the library) is set to 1; the
 * host mirrors read it lazily (RecPolicyBuffer.check_indices()). */
int ope_store_insert(const ope_dims* dims, int32_t capacity, const ope_fields* store, const ope_fields* staged,
                     const int64_t* slots, int32_t n_insert, int32_t* bad_index_flag, void* stream);
int ope_store_gather(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds,
                     int32_t batch, const ope_fields* out, int32_t* bad_index_flag, void* stream);
/* The same gather with the indices in HOST memory (what sample() draws with numpy): they travel inside the kernel-argument
 * block, so there is no index upload and no copy -> kernel dependency in front of the launch. batch <= 512; an index
 * outside [0, capacity) returns OPE_EINVAL (numpy's IndexError). */
int ope_store_gather_host_inds(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds_host,
                               int32_t batch, const ope_fields* out, void* stream);
/* sample(batch) entirely on the device: the gather draws its own indices -- uniform over [0, filled) with replacement, as
 * np.random.choice(filled, batch) in the reference's sample() (rec_buffer.py:86, mlp_buffer.py:74), from Philox4x32-10 keyed by
 * (seed, batch position, DEVICE int32 *counter (0 if NULL)). Same distribution, not numpy's stream. No index upload and no host
 * work per step; a captured HIP graph replays with fresh indices when the counter advances on the device (e.g.
 * ope_adam_cfg.step_counter). inds_out (device int64[batch], may be NULL) receives the drawn indices.
 * `filled_dev` (DEVICE int32, may be NULL): when given, the number of filled slots is read from it by the kernel at RUN time
 * (clamped to [1, capacity]) and `filled` is ignored -- a captured graph then keeps sampling from everything insert() has
 * written since the capture, as buffer.sample() does, instead of from the slots that were filled when it was captured. */
int ope_store_gather_sampled(const ope_dims* dims, int32_t capacity, int32_t filled, const int32_t* filled_dev,
                             const ope_fields* store, uint64_t seed, const int32_t* counter, int32_t batch, const ope_fields* out,
                             int64_t* inds_out, void* stream);
/* Per-dispatch timing of the gather (the roofline leg of bench.py): after ope_store_gather_profile(1) every gather launch
 * (after ope_store_gather_profile(N), N > 1: every N-th one -- an event pair costs the stream ~3.5 us, 1 % of a 0.34 ms step)
 * carries hipExtLaunchKernel start / stop events (up to 512 launches are kept); ope_store_gather_profile_read waits for them
 * and writes the kernel durations in milliseconds, oldest first, into HOST memory, returns how many (and clears the ring).
 * These time the dispatch itself -- what rocprofv3's kernel trace reports -- not the gap between two event markers. */
int ope_store_gather_profile(int32_t enable);
int ope_store_gather_profile_read(float* ms_out_host, int32_t max_n);
/* A/B knobs of the gather kernel (tools/bench_gather.py); negative or 0 = keep. Defaults are the measured best:
 * floats_per_block 2048 (contiguous floats of one episode a workgroup reads: whole time steps), xcd_run 8 (> 1: the B
 * workgroups writing one [t][agent][0..B) range share an XCD), unroll 8 (16-byte loads in flight per thread), nontemporal 0
 * (bit 0 loads, bit 1 stores), small_tiles 1 (short rows through LDS-transposing tiles), tile_floats 2048. Environment
 * twins: OPE_GATHER_FLOATS, OPE_GATHER_XCD, OPE_GATHER_UNROLL, OPE_GATHER_NT, OPE_GATHER_SMALL, OPE_GATHER_TILE (read once,
 * at the first call). */
void ope_set_gather_params(int floats_per_block, int xcd_run, int unroll, int nontemporal, int small_tiles, int tile_floats);
/* The same knobs PER CALL (two replay stores in one process need not share them; the setter above and the environment only provide the
 * process DEFAULTS): every field 0 = keep the default; floats_per_block / xcd_run / unroll / tile_floats as above; nontemporal = 1 + the bit
 * mask (1: none, 2: loads, 3: stores, 4: both); small_tiles = 1 on, 2 off. ope_store_gather_tuned = ope_store_gather (inds_dev: device
 * int64[batch], range-checked in the kernel, bad_index_flag as there) or ope_store_gather_host_inds (inds_host: host int64[batch], range-
 * checked here) -- exactly one of the two index sources non-NULL -- with `tune` (NULL = defaults). RecPolicyBuffer.sample_inds
 * (rec_buffer.py:192-240). */
typedef struct ope_gather_tune {
  int32_t floats_per_block, xcd_run, unroll, nontemporal, small_tiles, tile_floats;
} ope_gather_tune;
int ope_store_gather_tuned(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds_dev, const int64_t* inds_host,
                           int32_t batch, const ope_fields* out, int32_t* bad_index_flag, const ope_gather_tune* tune, void* stream);
/* Observations left in the store ("lazy batch"; SURVEY.md section 8(d): "fused into the first consumer, written = 0"). The sampled
 * episodes' observation rows are the largest field of a batch (3s5z: 39 of 48 MB) and the training step reads them exactly twice: the
 * trunk's first layer and its weight gradient. ope_store_gather_ref = ope_store_gather_tuned that (a) copies only the fields whose
 * `out` pointer is non-NULL (pass out->obs = NULL: every gather entry point skips NULL fields) and (b) writes the episode slots it
 * used to inds_out (DEVICE int64[batch]; the host-index form has them in its kernel arguments only). An ope_obs_ref then names those
 * rows for ope_qmix_loss_and_grad_ref: batch row (t, agent, b) = store_obs[inds[b]][t][agent][:]. The store must not be written
 * between the gather and the step (RecPolicyBuffer checks its insert count). Replaces rec_buffer.py:206-238 + qmix.py:108-109 for obs. */
typedef struct ope_obs_ref {
  const float* store_obs;   /* the store's obs ring [capacity][T+1][N][D] (ope_fields.obs of the store) */
  const int64_t* inds;      /* DEVICE int64[batch] episode slots, batch order */
  int32_t capacity;         /* slots outside [0, capacity) are clamped by the readers; the gather of the other fields raises bad_index_flag */
  int32_t reserved;
} ope_obs_ref;
int ope_store_gather_ref(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds_dev, const int64_t* inds_host,
                         int32_t batch, const ope_fields* out, int64_t* inds_out, int32_t* bad_index_flag, const ope_gather_tune* tune,
                         void* stream);

/* Bytes of one episode over all seven fields (SURVEY.md section 8(d) "episode bytes"). */
int64_t ope_episode_bytes(const ope_dims* dims);

/* ------------------------------------------------------------------------------------------------
 * Recurrent QMIX / VDN trainer  (QMix.train_policy_on_batch, offpolicy/algorithms/qmix/qmix.py:77-200).
 * Networks are the reference defaults: hidden 64, layer_N 1, feature-norm, 1-layer
 * GRU, hypernet_layers 2, mixer hidden 32, hypernet hidden 64. Anything else returns OPE_EINVAL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ope_qmix_cfg {
  ope_dims dims;
  int32_t batch;        /* B: episodes in this (local) batch                                   */
  int32_t vdn;          /* 1: VDN mixing (sum over agents, SURVEY A-2 fix) -- no mixer params  */
  int32_t use_double_q; /* args.use_double_q (config.py:144), default 1                         */
  int32_t use_huber;    /* args.use_huber_loss (config.py:120), default 0                       */
  int32_t use_per;      /* importance weights + priorities (qmix.py:168-181)                    */
  float gamma;          /* 0.99 */
  float huber_delta;    /* 10.0 */
  float per_nu;         /* 0.9  */
  float per_eps;        /* 1e-6 */
  int32_t mlp;          /* 1: non-recurrent agent nets on single transitions (M_QMix / M_VDN, mqmix.py:68-218):
                         *    dims.episode_length must be 1; batch.obs = [obs; next_obs], share_obs = [cent; next cent],
                         *    avail_acts = [avail; next avail]                                              */
  int32_t phase;        /* 0: the whole step. 1..3: one part of a step with SEVERAL POLICIES under one mixer (the
                         *    `for p_id in self.policy_ids` loops of qmix.py:100-150 / mqmix.py:95-178; policies may differ in
                         *    obs_dim / act_dim / agent count). Every policy p has its own cfg (dims.n_agents = its agents,
                         *    vdn = 1: no mixer block in its vector) and workspace; the mixing part has a joint cfg
                         *    (dims.n_agents = all agents; obs_dim / act_dim of policy 0) and workspace:
                         *      1  agent networks forward only: leaves "agent_q" / "agent_nq" [T][B][n_p] in the workspace
                         *         (ope_qmix_workspace_find); grad untouched
                         *      2  mixer (or VDN sum) + TD loss + mixer gradients only: reads "agent_q" / "agent_nq"
                         *         [T][B][N] the caller assembled from the policies' slices, leaves "d_agent_q" [T][B][N];
                         *         writes the mixer block and the tail of `grad`, leaves grad[0 .. agent block) untouched
                         *      3  agent networks backward only: reads "d_agent_q" [T][B][n_p] (the caller's slice of the
                         *         joint one) and the activations phase 1 left; writes grad[0 .. P_agent) and NO tail
                         *    The flat vectors of such a trainer are [agent_0 | agent_1 | ... | mixer | tail]; the joint call
                         *    gets theta / grad advanced by (mixer offset - agent_0 length) so that its mixer block lines up. */
  /* Per-trainer tuning / diagnostics (all 0 = defaults). These travel with the cfg of every call, so two trainers in one process
   * do not share them (the process-wide setters below and the environment variables only provide the DEFAULTS, read once). The
   * workspace size depends on mixer_path and time_chunks: use the same values for ope_qmix_workspace_bytes / _init / _find and the
   * step calls. */
  int32_t mixer_path;   /* forward mixer kernel: 0 = by shape; 1 = hyper-network weights resident in registers (mixer_fwd3; needs
                         *  state_dim <= 224, % 4 == 0, <= 8 agents, else falls to 2); 2 = weights streamed per 16-row workgroup
                         *  (mixer_fwd2); 3 = wide-state form: the first hyper-layers as one stream-K GEMM (ope_mixer_wide.hip;
                         *  what "by shape" picks for state_dim > 256, e.g. --use_global_all_local_state) */
  int32_t time_chunks;  /* recurrent nets: time chunks of the two-stream schedule (1 = whole episodes on one stream, the default) */
  int32_t scan_family;  /* GRU scan kernels: 4 = ope_gru4.hip, 1 = ope_gru1.hip, 0 = by row count                                  */
  int32_t scan_waves;   /* family 4: compute waves per row, 2 | 4, 0 = by row count                                                */
  int32_t debug;        /* 1: keep extra intermediates and per-wave phase stamps in the workspace ("q_all", "dbg")               */
  int32_t trunk_path;   /* forward trunk of the live + target agent nets: 0 = by shape; 3 = one trunk_fwd3 launch per net (weights in
                         *  registers, four waves per 16-row tile); 4 = both nets in one trunk_fwd4 launch (weights in LDS, one wave per
                         *  tile; recurrent nets whose input width is a multiple of 4 in (48, 64], (112, 128], (176, 192] or (240, 256] -- what "by shape"
                         *  picks from 2 048 rows on (the adjoint from 16 384); other widths run path 3)                    */
  int32_t chain_path;   /* the (t, b)-row chain between the GRU scan and its adjoint: 0 = by shape; 1 = four launches (head_fwd, mixer_fwd,
                         *  mixer_bwd, head_bwd); 2 = two launches (mixer_hyp: the mixers' first hyper-layers, a GEMM on the state alone, +
                         *  qchain: heads, second mixer stage, TD / loss, mixer and head adjoints; ope_chain.hip): whole steps (phase 0) of
                         *  recurrent nets with <= 16 agents and <= 32 actions, mixer_path != 3, time_chunks = 1 -- what "by shape" picks for
                         *  state_dim <= 512 (wider states: the stream-K GEMM of mixer_path 3 + the four launches); any other configuration
                         *  with chain_path = 2 returns OPE_EINVAL)                                  */
  int32_t hypernet_layers; /* args.hypernet_layers (config.py:146): 0 or 2 = the default two-layer hyper-networks; 1 = hyper_w1 / hyper_w2 as
                         *  single Linear layers from the state (q_mixer.py:39-44): 10 mixer tensors instead of 14 (ope_qmix_param_layout:
                         *  hyper_w1.{weight [N*32][S], bias}, hyper_w2.{weight [32][S], bias}, hyper_b1.*, hyper_b2.0.*, hyper_b2.2.*); runs on
                         *  the fused chain only (whole steps of recurrent nets; chain_path 1, mixer_path 3, mlp, phases: OPE_EINVAL)      */
  int32_t wgrad_path;   /* weight-gradient launch (grad_W = grad_out^T input of every Linear, qmix.py:190-191): 0 = by shape; 1 = one 64 x 64
                         *  tile per wave (wgrad_kernel + split_reduce, ope_wgrad.hip); 2 = register-blocked, up to four tiles that share an
                         *  operand per wave (wgrad2_kernel + w2_reduce, ope_wgrad2.hip): whole steps on one stream with 8-byte aligned rows of
                         *  at most 1 024 floats and the observations gathered -- what "by shape" picks then (process default OPE_WGRAD2 = 1 | 0,
                         *  read once); any other configuration with wgrad_path = 2 returns OPE_EINVAL. A table the register-blocked
                         *  form cannot plan (more than 40 units, 4-byte aligned operands) runs form 1 under "by shape" and fails
                         *  BEFORE the step's first launch under wgrad_path = 2                                       */
  int32_t live_rows;    /* rows of the padded batch that are computed: 0 = by shape; 1 = every row (T steps of every episode, as the
                         *  reference does); 2 = the LIVE rows only; 3 / 4 = as 2, and the plan region "live_plan" / "live_plan1" already
                         *  holds this batch's plan (ope_store_live_plan below: built ahead of the step). The reference pads every sampled episode to episode_length steps and
                         *  multiplies the Bellman error of every (t, b) with dones_env[t-1, b] = 1 by 1 - bad_transitions_mask = 0
                         *  (qmix.py:161-166), leaves it out of the loss normaliser (:184-186), the priorities (:177-181) and Q_tot's mean
                         *  (:198): such rows contribute exactly nothing. With live rows a small kernel finds, ON THE DEVICE at the start
                         *  of every step, each episode's length from the sampled dones_env (nothing is cached across steps or taken from the
                         *  host), ranks the episodes by it and packs the rows that matter; the agent trunk, the GRU scans, the mixer /
                         *  TD chain and every weight-gradient reduction then run on the packed rows (same per-row arithmetic; sums
                         *  over rows in a different order). Needs whole steps (phase 0, time_chunks 1) of one shared recurrent policy
                         *  on the fused chain (chain_path), trunk_fwd4 / trunk_bwd4 (trunk_path), the four-wave scans and wgrad2, no
                         *  debug outputs, batch <= 256, episode_length <= 1022 -- what "by shape" picks then (process default
                         *  OPE_LIVE_ROWS = 1 | 0, read once); any other configuration with live_rows = 2 returns OPE_EINVAL before
                         *  the first launch. Workspace region "live_plan" (int32): [0..2] = live agent rows, those with t < T, live
                         *  (t, b) rows of the last step; [8..15] as four int64 = their sums over all steps so far and the step count */
} ope_qmix_cfg;

/* Flat parameter vector: the reference's named_parameters() order (agent q-network, then mixer; qmix.py:67-72),
 * every tensor padded to a multiple of 4 floats so rows can be read as float4. The gradient vector has the
 * same layout followed by OPE_GRAD_TAIL floats: [loss_sum, mask_count, qtot_sum, 0].                        */
#define OPE_QMIX_NPARAM_AGENT 22      /* recurrent agent net */
#define OPE_QMIX_NPARAM_AGENT_2 26    /* ... with layer_N = 2 */
#define OPE_QMIX_NPARAM_AGENT_MLP 16  /* MLP agent net (no GRU, no rnn.norm) */
#define OPE_QMIX_NPARAM_MIXER 14
#define OPE_QMIX_NPARAM_MIXER_1 10    /* hypernet_layers = 1 */
#define OPE_GRAD_TAIL 4
/* Fills offsets[i]/sizes[i] (floats) for the 22 (layer_N = 2: 26) agent + 14 (hypernet_layers = 1: 10; vdn: 0) mixer tensors -- arrays
 * of at least 48 entries --; returns the padded total length. */
int64_t ope_qmix_param_layout(const ope_qmix_cfg* cfg, int64_t* offsets, int64_t* sizes);
/* Workspace (bytes) ope_qmix_loss_and_grad needs. */
int64_t ope_qmix_workspace_bytes(const ope_qmix_cfg* cfg);
/* One-time initialisation of a freshly allocated workspace (constant regions the kernels only read). Must be called
 * once per workspace buffer before the first ope_qmix_loss_and_grad on it. */
int ope_qmix_workspace_init(const ope_qmix_cfg* cfg, void* workspace, int64_t workspace_bytes, void* stream);
/* Process-wide DEFAULTS of the diagnostics above, for callers without a cfg of their own (the recurrent MADDPG entry points share
 * the scan kernels) and for tools: ope_set_debug(1) = ope_qmix_cfg.debug for every call; ope_set_scan_kernel(family, waves) =
 * scan_family / scan_waves; the environment variables OPE_GRU / OPE_GRU4_W / OPE_CHUNKS / OPE_MIXER_PERSIST set initial values,
 * read ONCE. A non-zero field of the cfg always wins. */
void ope_set_debug(int on);
void ope_set_scan_kernel(int family, int waves_per_row);
/* 1 (default; OPE_W2_FIN) | 0: where the register-blocked weight-gradient launch runs (wgrad2), fold the finalize step into its slab sum -- the
 * slabs of a LayerNorm-fed Linear then hold dW = C gamma + s (x) beta and the column partials of dgamma / dbeta, and ONE launch (w2_fin) sums
 * every slab straight into the flat gradient, with the loss tail, the zero ranges and the clip norm's partial sums of squares -- instead of
 * w2_reduce -> finalize. Same gradient to rounding (the identities are applied per workgroup instead of to the total). A/B runs and tests. */
void ope_set_w2_fin(int on);
/* In-process per-kernel timing (the per-kernel roofline table of bench.py, measured in the run): ope_kernel_profile(1, max_launches)
 * makes every kernel launch of the library carry hipExtLaunchKernel start / stop events (the dispatch's own duration -- what rocprofv3's
 * kernel trace reports), up to max_launches (<= 16384) launches; ope_kernel_profile_read waits for them, writes one line per distinct
 * kernel, in order of first launch, "demangled name \t calls \t total_ms \t min_ms \t max_ms \t flop \t bytes \n" into out[cap]
 * (NUL-terminated; flop / bytes = the ALGORITHMIC work of those launches as their launchers state it: GEMM-shaped FLOP = 2 x MACs with
 * LayerNorm / gates / elementwise excluded, bytes for the bandwidth-bound kernels; 0 where none is stated -- an 8th column " \t rows" follows
 * the bytes: 0, or for a launch on the LIVE rows of a step (ope_qmix_cfg.live_rows) which of the plan's counts scales the stated work, which is
 * that of the padded batch: 1 live agent rows / (T+1) N B, 2 those with t < T / T N B, 3 live (t, b) rows / T B), returns
 * the number of distinct kernels and clears the ring. ope_kernel_profile(0, 0) turns it off. Eager launches only (not while a HIP graph
 * is being captured). SURVEY.md section 8(d). */
int ope_kernel_profile(int32_t enable, int32_t max_launches);
int ope_kernel_profile_read(char* out, int32_t cap);
/* Launch log: the kernel variants the LAST ope_qmix_loss_and_grad call of this thread actually launched, separated by ';' in launch
 * order (e.g. "trunk_fwd4<16>;gru_fwd4<4>;head_fwd_mfma<1>;mixer_fwd3<14,1>;..."), written NUL-terminated into out[cap]; returns the number
 * of launches. Test hook: an explicit trunk_path / mixer_path that the shape does not allow returns OPE_EINVAL from the step (no silent
 * fall-back to another kernel), and tests that pin a kernel family assert here that it ran. No reference counterpart (the
 * reference's kernels are whatever ATen dispatches). */
int ope_last_launches(char* out, int32_t cap);

/* Named sub-buffers of the workspace, for tests/debugging: returns byte offset, writes element count. -1 if unknown. */
int64_t ope_qmix_workspace_find(const ope_qmix_cfg* cfg, const char* name, int64_t* n_floats);

/* Forward + backward of one batch (qmix.py:86-190 minus the optimizer):
 *   batch    -- ope_store_gather output layout
 *   theta / theta_tgt -- live / target flat parameters (read only)
 *   per_weights[B] -- importance weights or NULL
 *   grad     -- OUT flat gradient of the UN-normalised loss sum (divide by mask_count = grad[P+1]; done in
 *               ope_adam_step so that data-parallel ranks can all-reduce `grad` first) + the 4-float tail
 *   td_abs_stats[2*B] -- OUT per-episode [mean_t |err|, max_t |err|] (for priorities), or NULL
 * All intermediates live in `workspace`. */
int ope_qmix_loss_and_grad(const ope_qmix_cfg* cfg, const ope_fields* batch, const float* theta,
                           const float* theta_tgt, const float* per_weights, void* workspace,
                           int64_t workspace_bytes, float* grad, float* td_abs_stats, void* stream);
/* The same step with the observations read in place from the replay store (ope_obs_ref above; batch->obs is ignored and may be NULL):
 * same arithmetic on the same values, so the gradient is bit-identical to the gathered form. Only the configurations whose first-layer
 * kernels read rows through an index can do it -- ope_qmix_obs_ref_ok(cfg) = 1: recurrent nets, phase 0, obs_dim % 4 == 0 with
 * ceil(obs_dim / 16) in {4, 8, 12, 16}, state_dim % 4 == 0, batch <= 512, (T+1) N batch < 2^20 rows, and the LDS-resident trunk kernel
 * selected (trunk_path 4, or 0 with >= 2 048 rows) -- everything else returns OPE_EINVAL (the caller gathers obs instead). */
int ope_qmix_obs_ref_ok(const ope_qmix_cfg* cfg);
/* The live-row plan of a batch on its own (what ope_qmix_loss_and_grad builds at the start of a step that runs on live rows -- see
 * ope_qmix_cfg.live_rows and the reference lines cited there): from dones_env [T][B][1] (DEVICE, the gather's layout) into the
 * workspace region "live_plan" (int32): [0] live agent-network rows = N * sum_b len_b, [1] those with t < T, [2] live (t, b) rows,
 * [3] the longest len_b, where len_b = 2 + the last t with dones_env[t, b] != 1 (1 if there is none), then the ranking and the row
 * maps the kernels read. Also zero-fills the regions "err_abs" and "loss_part". ope_qmix_live_rows_ok(cfg) = 1 when a whole step of
 * this configuration would run on live rows "by shape" (the same test the step makes); OPE_EINVAL from the plan call when the
 * workspace of `cfg` holds no plan region (MLP nets, phases, batch > 256, episode_length > 1022). Diagnostics / tests / bench.py's
 * executed-row accounting; training never needs to call it. */
int ope_qmix_live_rows_ok(const ope_qmix_cfg* cfg);
/* The plan built AHEAD of the step, off its critical path (as a launch in front of the step it costs ~8 us of a 0.3 ms step: a chain of a
 * few dependent round trips on a nearly idle GPU). A step's plan depends on the store's dones_env of the sampled episodes only, so it can be
 * built as soon as the indices are drawn -- typically while the previous step is still running, on another stream:
 *   ope_qmix_live_target(cfg, workspace, bytes, which, &t)   where a plan for (cfg, workspace) goes: region "live_plan" (which = 0) or
 *                                                             "live_plan1" (which = 1) -- two, so that a plan can be written while the
 *                                                             step before still reads its own
 *   ope_store_live_plan(capacity, T, store.dones_env, inds_dev | inds_host, &t, stream)
 *                                                             the same computation as ope_qmix_live_plan, on the STORE's flags
 *                                                             ([capacity][T][1]) of the episode slots `inds` (exactly one of the two
 *                                                             pointers; host indices travel in the kernel arguments, batch <= 512)
 * and the step is told with ope_qmix_cfg.live_rows = 3 (region 0) / 4 (region 1): "live rows, and that region already holds THIS batch's
 * plan" -- the caller's promise that the batch is those slots' episodes, unmodified, and that the plan kernel is ordered before the step
 * (an event). The regions "err_abs" / "loss_part" are then cleared inside the step. off-policy_amd: RecPolicyBuffer.sample_inds(...,
 * live_for=trainer) launches it on the trainer's side stream and tags the batch; QMix.train_policy_on_batch checks the tag. */
typedef struct ope_live_target {
  int32_t* plan;        /* workspace region "live_plan" / "live_plan1" */
  float* err_abs;       /* workspace region "err_abs" [T*B] */
  float* loss_part;     /* workspace region "loss_part" */
  int32_t n_loss_part;
  int32_t n_agents, episode_length, batch;
  int32_t copy_live_only;   /* ope_store_gather_attach_live only: 1 = the copy of that launch writes, of the long-row fields (obs, share_obs: the
                             * episode-contiguous step path), only the time entries t < len_b (share_obs: t <= len_b) of each sampled episode -- the rows the live-row
                             * step reads; every later entry of the destination keeps whatever it held (those (t, b) carry a zero mask in the
                             * loss, qmix.py:161-166). The caller's promise: this batch is consumed by the live-row step of this plan only. */
} ope_live_target;
int ope_qmix_live_target(const ope_qmix_cfg* cfg, void* workspace, int64_t workspace_bytes, int32_t which, ope_live_target* out);
int ope_store_live_plan(int32_t capacity, int32_t episode_length, const float* store_dones_env, const int64_t* inds_dev,
                        const int64_t* inds_host, const ope_live_target* target, void* stream);
/* ... or inside the gather launch itself: the NEXT ope_store_gather* launched from the calling thread gets a few extra workgroups IN FRONT of
 * the copy's that build the plan into `target` from the store's flags through the launch's own indices (consumed by that one launch; ignored
 * -- no plan: do not promise one -- unless the store's episode_length, the batch size and the shape limits match the target). NULL cancels. */
int ope_store_gather_attach_live(const ope_live_target* target);
int ope_qmix_live_plan(const ope_qmix_cfg* cfg, const float* dones_env, void* workspace, int64_t workspace_bytes, void* stream);
/* A hook for work that wants to start in the MIDDLE of a step (a gather of the next batch on another stream, beside the latency-bound half of
 * this one): the next ope_qmix_loss_and_grad[_ref] launched from the calling thread records `event` (a hipEvent_t) on its stream directly in
 * front of launch `at` -- 1 the GRU scan, 2 the (t, b)-row chain, 3 the scan's adjoint, 4 the weight gradients, 5 their reduction -- or behind
 * its last launch when its path has no such point. Consumed by that one step; NULL cancels. The caller makes its other stream wait on the
 * event AFTER the step call has returned (hipStreamWaitEvent on an event not yet recorded is a no-op). off-policy_amd:
 * RecPolicyBuffer.sample_inds_ahead(after=...). */
int ope_qmix_signal_event(void* event, int32_t at);
int ope_qmix_loss_and_grad_ref(const ope_qmix_cfg* cfg, const ope_fields* batch, const ope_obs_ref* obs, const float* theta,
                               const float* theta_tgt, const float* per_weights, void* workspace, int64_t workspace_bytes,
                               float* grad, float* td_abs_stats, void* stream);


/* Forward only of the agent q-network on [L, R, D] observations with initial hidden h0 [R, 64] (NULL = zeros)
 * (AgentQFunction.forward, agent_q_function.py:34-67): q_out [L, R, A], h_out [L, R, 64] (un-normalised GRU
 * states; h_out[L-1] is h_final). Used by policy.get_actions / get_q_values. */
int64_t ope_agent_forward_workspace_bytes(const ope_dims* dims, int32_t seq_len, int32_t rows);
int ope_agent_forward(const ope_dims* dims, int32_t seq_len, int32_t rows, const float* obs, const float* h0,
                      const float* theta, void* workspace, int64_t workspace_bytes, float* q_out, float* h_out,
                      void* stream);

/* Same for the MLP agent q-network of the mqmix family (AgentQFunction.forward,
 * offpolicy/algorithms/mqmix/algorithm/agent_q_function.py:28-41): obs [rows, D] -> q_out [rows, A]. */
int64_t ope_agent_forward_mlp_workspace_bytes(const ope_dims* dims, int32_t rows);
int ope_agent_forward_mlp(const ope_dims* dims, int32_t rows, const float* obs, const float* theta, void* workspace,
                          int64_t workspace_bytes, float* q_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: clip_grad_norm_ + Adam + (optional) Polyak target update over the flat vectors
 * (qmix.py:190-193, torch.optim.Adam(lr, eps=opti_eps) qmix.py:71-72, soft_update util.py:123-134).
 *   g = grad * (1/grad[n+1]);  norm = ||g||_2;  g *= min(1, max_norm/(norm+1e-6));  Adam(g);  tgt = (1-tau) tgt + tau theta
 *   stats_out[4] = [loss, grad_norm (pre-clip), Q_tot mean, mask_count]   with loss = grad[n]/grad[n+1],
 *   Q_tot = grad[n+2]/qtot_denominator.
 * `scratch` must hold ope_adam_scratch_floats(n) floats.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ope_adam_cfg {
  float lr, beta1, beta2, eps, max_grad_norm, weight_decay;
  float tau;              /* Polyak factor; used only if do_polyak                        */
  int32_t do_polyak;      /* fuse soft_target_updates() into this call                    */
  int32_t step;           /* 1-based Adam step count (bias correction)                    */
  float qtot_denominator; /* T*B_global: Q_tot is a mean over ALL steps (qmix.py:198)     */
  int32_t tail_offset;    /* index of the 4-float tail inside `grad`; <= 0 means n. Lets a PREFIX of a parameter vector
                           * be optimised (n < full length) while the tail stays behind the full gradient: MADDPG's
                           * critic, whose q heads are unregistered upstream (SURVEY A-4) and therefore frozen.     */
  int32_t* step_counter;  /* optional DEVICE int32[2]: [0] = number of Adam steps taken so far, [1] = 0 (scratch ticket, zero
                           * between calls). When non-NULL, `step` is ignored: the update uses t = [0] + 1 and its last
                           * workgroup stores t back, so a captured HIP graph of the update replays with an advancing bias
                           * correction and without a separate "count += 1" launch.                                        */
  int32_t skip_begin, skip_end; /* elements [skip_begin, skip_end) are left untouched by Adam (no moment update, no step, no weight
                                 * decay) but still follow Polyak: tensors that never receive a gradient -- torch's Adam skips
                                 * them (the registered-but-unused fc_h block, SURVEY A-8). Only matters with weight_decay != 0:
                                 * with a zero gradient and zero moments the plain update is already a no-op. 0, 0 = none. */
  const float* sumsq_partials; /* optional DEVICE array of n_sumsq_partials floats whose sum is sum_i grad[i]^2 over the n    */
  int32_t n_sumsq_partials;    /* optimised elements, produced together with `grad` (ope_qmix_loss_and_grad leaves them in
                                * its workspace region "gsq_part", see ope_qmix_workspace_find; the fused MADDPG path leaves them
                                * in "gsq_critic" / "gsq_actor", see ope_ddpg_workspace_find). When given, the call skips
                                * its own norm pass over the gradient. ONLY valid if
                                * `grad` was not modified since (i.e. not on the all-reduced gradient of a multi-GPU run). */
} ope_adam_cfg;
int64_t ope_adam_scratch_floats(int64_t n);
int ope_adam_step(const ope_adam_cfg* cfg, int64_t n, float* theta, float* theta_tgt, float* adam_m, float* adam_v,
                  const float* grad, float* scratch, float* stats_out, void* stream);
/* soft_update (util.py:123-134) / hard_update (util.py:137-145) on flat vectors. tau=1 is a hard copy. */
int ope_polyak(int64_t n, const float* theta, float* theta_tgt, float tau, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MLP MADDPG / MATD3  (offpolicy/algorithms/maddpg/maddpg.py:90-249; matd3 = num_q 2 + target_gumbel 1).
 * One shared policy for all N agents. Actor: MLPBase(D) -> Linear(64, A). Critic: MLPBase(S + N*A) -> num_q x Linear(64, 1).
 * Flat vectors use the MLP agent layout (16 tensors; the head = last two); for the critic the head block is
 * [num_q][64] weights then [num_q] biases.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ope_ddpg_cfg {
  ope_dims dims;         /* episode_length unused */
  int32_t batch;         /* B transitions */
  int32_t num_q;         /* critic heads: 1 (MADDPG) or 2 (MATD3, min over heads in the target) */
  int32_t target_gumbel; /* 1: target actions by hard gumbel-softmax with caller-provided uniform noise (MATD3);
                            0: onehot_from_logits argmax (MADDPG)  -- MADDPGPolicy.py:94-105 */
  int32_t use_huber, use_per;
  float gamma, huber_delta, per_eps;
  uint64_t noise_seed;          /* != 0: when a call's uniform-noise argument is NULL the kernels draw it themselves -- Philox4x32-10 */
  const int32_t* noise_counter; /* keyed by this seed, counter = (row, column, DEVICE int32 *noise_counter (0 if NULL), stream: 0 =
                                 * target noise, 1 = actor noise). Same distribution as the reference's torch.rand, not the same
                                 * stream; a captured graph replays with fresh noise if the counter advances on the device (e.g.
                                 * the critic's ope_adam_cfg.step_counter). 0 = noise arguments are required as before.          */
  /* Multi-policy updates (share_policy = False: get_update_info, maddpg.py:40-80, loops over `policy_ids`, every policy with its own
   * actor, critic and buffer). All 0 / NULL = one policy for all agents. Otherwise dims.n_agents is the number of agents of the policy
   * being updated, n_total_agents the number of agents in the joint action (policy order) and agent_offset the update policy's first
   * agent in it. The batch then carries the update policy's fields, but `acts` of ALL agents: [n_total_agents][B][A] (same act_dim for
   * every policy on this path). */
  int32_t n_total_agents, agent_offset;
  const float* joint_next_acts; /* DEVICE [B][n_total_agents * A] joint target action, filled by one ope_ddpg_target_actions call per
                                 * policy; when non-NULL the critic call skips its own target-actor pass (theta_actor_tgt, next_obs and
                                 * target_noise_u are then unused). Required when n_total_agents > dims.n_agents.                  */
  int32_t continuous;           /* 1: Box action space (MADDPGPolicy.py:107-116): an action IS the actor's output (act_dim floats), the target
                                 * action = target actor output + target_noise_u (here ADDITIVE noise, already scaled: gaussian_noise(shape,
                                 * target_noise) of MATD3, util.py:217-218; NULL = none), the actor update differentiates straight through
                                 * (gumbel_noise_u unused, may be NULL). No availability masks. General kernel path only. 0 = discrete. */
  int32_t n_act_heads;          /* > 1: MULTI-DISCRETE action space (MADDPGPolicy.py:73-92, act.py:14-17): the act_dim outputs are n_act_heads
                                 * one-hot blocks of act_head_dims[i] entries (their sum = dims.act_dim); argmax / hard gumbel-softmax and its
                                 * straight-through adjoint act per block. No availability masks (upstream passes none). General kernel path
                                 * only; the flat layout is the single-head one (the heads' rows are consecutive). 0 or 1 = one head. */
  int32_t act_head_dims[6];
  /* Multi-policy updates whose policies have DIFFERENT action dimensions (maddpg.py:44-88 and 190-227 concatenate whatever widths the policies
   * have; MPE simple_speaker_listener). Then n_total_agents = agent_offset = 0 and the joint action is described in COLUMNS: joint_act_dim =
   * its width (sum over all agents of their policy's act_dim), joint_act_col = the first column of the update policy's first agent, joint_acts
   * = the buffer's joint action [B][joint_act_dim] (every agent's block side by side in policy order; batch->acts is then unused);
   * joint_next_acts (required) has the same width and ope_ddpg_target_actions writes a policy's agents at columns joint_act_col + a * act_dim.
   * General kernel path, caller-provided noise (noise_seed = 0). 0 / NULL = every policy has dims.act_dim actions. */
  int32_t joint_act_dim, joint_act_col;
  const float* joint_acts;
} ope_ddpg_cfg;

/* One sampled batch in MlpPolicyBuffer.sample_inds order/shapes (mlp_buffer.py:213-257), device pointers. */
typedef struct ope_mlp_batch {
  const float* obs;              /* [N][B][D] */
  const float* share_obs;        /* [B][S]    */
  const float* acts;             /* [N][B][A] */
  const float* rewards;          /* [N][B][1] */
  const float* next_obs;         /* [N][B][D] */
  const float* next_share_obs;   /* [B][S]    */
  const float* dones_env;        /* [B][1]    */
  const float* valid_transition; /* [N][B][1] */
  const float* avail_acts;       /* [N][B][A] or NULL */
  const float* next_avail_acts;  /* [N][B][A] or NULL */
} ope_mlp_batch;

/* which = 0 actor, 1 critic: offsets/sizes of its 16 tensors; returns the padded length. */
int64_t ope_ddpg_param_layout(const ope_ddpg_cfg* cfg, int32_t which, int64_t* offsets, int64_t* sizes);
int64_t ope_ddpg_workspace_bytes(const ope_ddpg_cfg* cfg);
int ope_ddpg_workspace_init(const ope_ddpg_cfg* cfg, void* workspace, int64_t workspace_bytes, void* stream);
int64_t ope_ddpg_workspace_find(const ope_ddpg_cfg* cfg, const char* name, int64_t* n_floats);
/* Regions "gsq_critic" / "gsq_actor" (present only on the fused small-network path; find returns < 0 otherwise): after a
 * critic / actor loss_and_grad call they hold n_floats = 2m per-workgroup sums of grad^2 -- the first m over the trunk
 * (MLPBase) elements, the last m over the head block -- for ope_adam_cfg.sumsq_partials. */
/* Critic update (maddpg.py:100-157 + get_update_info 38-81): grad = d(sum_k sum_b f(target - Q_k) w_b)/d theta_critic
 * + tail [loss_sum, B, sum Q, 0]; prio_out[B] = mean_k |err_k| + per_eps (or NULL).
 * target_noise_u [N*B][A]: uniform(0,1) noise for the MATD3 target gumbel (NULL unless target_gumbel). */
int ope_ddpg_critic_loss_and_grad(const ope_ddpg_cfg* cfg, const ope_mlp_batch* batch, const float* theta_actor_tgt,
                                  const float* theta_critic, const float* theta_critic_tgt, const float* target_noise_u,
                                  const float* per_weights, void* workspace, int64_t workspace_bytes, float* grad,
                                  float* prio_out, void* stream);
/* Actor update (maddpg.py:162-247): hard gumbel-softmax actions (noise gumbel_noise_u [N*B][A]) spliced into N stacked
 * copies of the joint action, -sum(Q_1 * valid) objective through the (frozen) critic; grad w.r.t. theta_actor + tail
 * [loss_sum, sum(valid), ...]. */
/* Target actions of ONE policy's agents (get_update_info, maddpg.py:56-74): its target actor on the next observations of its
 * dims.n_agents agents, onehot_from_logits or (target_gumbel) hard gumbel-softmax, written into columns (agent_offset + a) * A of
 * joint_next_acts [B][n_total_agents * A]. cfg / batch are that policy's (only next_obs and next_avail_acts are read). */
int ope_ddpg_target_actions(const ope_ddpg_cfg* cfg, const ope_mlp_batch* batch, const float* theta_actor_tgt,
                            const float* target_noise_u, void* workspace, int64_t workspace_bytes, float* joint_next_acts,
                            void* stream);
int ope_ddpg_actor_loss_and_grad(const ope_ddpg_cfg* cfg, const ope_mlp_batch* batch, const float* theta_actor,
                                 const float* theta_critic, const float* gumbel_noise_u, void* workspace,
                                 int64_t workspace_bytes, float* grad, void* stream);

/* The same two updates INCLUDING their optimiser step, one launch each (small networks in one process: ope_ddpg_update_ok(cfg) = 1 when
 * the fused tile path serves cfg and all its workgroups are co-resident). The launch's tail replaces the slab reduction, the norm pass
 * and ope_adam_step: behind a grid barrier every workgroup sums its share of the per-workgroup gradient slabs (same fixed order as the
 * separate reduction: the gradient is bit-identical), a second barrier publishes the partial sums of squares, then clip_grad_norm_ +
 * Adam (+ Polyak of the updated network's target) run on the same share. `grad` still receives the flat gradient + tail. opt->adam as
 * for ope_adam_step (sumsq_partials / tail_offset ignored); opt->n = elements optimised (a prefix: the critic's trunk, SURVEY A-4);
 * theta_critic / theta_actor are UPDATED in place. maddpg.py:100-157 + 192-249 with their optimizer.step() / soft_update calls. */
typedef struct ope_ddpg_opt {
  ope_adam_cfg adam;
  int64_t n;
  float* theta_tgt;     /* Polyak target of the updated network (adam.do_polyak), else NULL */
  float* adam_m;
  float* adam_v;
  float* stats_out;     /* [4] as ope_adam_step, or NULL */
} ope_ddpg_opt;
int ope_ddpg_update_ok(const ope_ddpg_cfg* cfg);
int ope_ddpg_critic_update(const ope_ddpg_cfg* cfg, const ope_mlp_batch* batch, const float* theta_actor_tgt, float* theta_critic,
                           const float* theta_critic_tgt, const float* target_noise_u, const float* per_weights, void* workspace,
                           int64_t workspace_bytes, float* grad, float* prio_out, const ope_ddpg_opt* opt, void* stream);
int ope_ddpg_actor_update(const ope_ddpg_cfg* cfg, const ope_mlp_batch* batch, float* theta_actor, const float* theta_critic,
                          const float* gumbel_noise_u, void* workspace, int64_t workspace_bytes, float* grad, const ope_ddpg_opt* opt,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Recurrent MADDPG / MATD3 update on sampled EPISODES (use_same_share_obs path, one shared policy).
 * Replaces, for discrete one-hot actions:
 *   R_MADDPG.get_update_info               offpolicy/algorithms/r_maddpg/r_maddpg.py:44-105
 *   R_MADDPG.shared_train_policy_on_batch  offpolicy/algorithms/r_maddpg/r_maddpg.py:114-331
 *   R_MADDPG_Actor / R_MADDPG_Critic       offpolicy/algorithms/r_maddpg/algorithm/r_actor_critic.py:7-129
 *   R_MADDPGPolicy.get_actions (target / gumbel branches)  .../algorithm/rMADDPGPolicy.py:61-131
 * Networks: RNNBase (feature LN -> fc1 -> fc2 -> GRU -> LN) + Linear head; flat vectors use the 22-tensor recurrent
 * agent layout (ope_qmix_param_layout order); the critic head block is [num_q][64] weights then [num_q] biases
 * (q_outs.k.weight / q_outs.k.bias, registered and trained -- unlike the MLP family).
 * The episode batch is the ope_fields block of ope_store_gather (time-major, rows agent*B + b); prev_act_inp = False.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ope_rddpg_cfg {
  ope_dims dims;         /* episode_length = T */
  int32_t batch;         /* B episodes */
  int32_t num_q;         /* 1 (R_MADDPG) or 2 (R_MATD3) */
  int32_t target_gumbel; /* 1: hard gumbel-softmax target actions (target_noise is not None, rMADDPGPolicy.py:108) */
  int32_t use_huber, use_per;
  float gamma, huber_delta;
  /* Multi-policy updates (share_policy = False: r_maddpg.py:44-105 loops over `policy_ids`, every policy with its own actor,
   * critic and buffer; scripts/train_mpe_rmaddpg.sh). All 0 / NULL = one policy for all agents. Otherwise dims.n_agents is the
   * number of agents of the policy being updated, n_total_agents the number of agents in the joint action (policy order), and
   * agent_offset the update policy's first agent in it. The batch then carries the update policy's obs / avail_acts / dones /
   * rewards / share_obs / dones_env, but `acts` of ALL agents: [T][n_total_agents][B][A] (same act_dim for every policy). */
  int32_t n_total_agents, agent_offset;
  const float* joint_next_acts; /* DEVICE [T][B][n_total_agents * A] joint target action, filled by one ope_rddpg_target_actions call per
                                 * policy; when non-NULL the critic call skips its own target-actor pass (theta_actor_tgt, batch obs and
                                 * target_noise_u are then unused). Required when n_total_agents > dims.n_agents.                    */
  const float* actor_row_weight; /* optional DEVICE [N][B]: the actor objective's mask (1 - shifted agent dones) of copy `rep`, episode b is
                                  * multiplied by weight[rep][b], in the loss, its normaliser and the gradient. NULL = 1. This is how
                                  * cent_train_policy_on_batch (r_maddpg.py:333-564: every agent has its OWN centralized observation) runs on
                                  * these entry points: the batch is laid out as N*B episodes -- episode (i, b) = episode b seen through
                                  * agent i's centralized observation, everything else repeated -- and copy `rep` of episode (i, b) counts
                                  * only for rep == i.                                                                               */
  int32_t continuous;            /* 1: Box action space (rMADDPGPolicy.py:121-129), as ope_ddpg_cfg.continuous: an action is the actor's output,
                                  * target_noise_u is ADDITIVE noise (NULL = none), gumbel_noise_u is unused, no availability masks. */
  int32_t n_act_heads;           /* > 1: multi-discrete action space, as ope_ddpg_cfg.n_act_heads / act_head_dims */
  int32_t act_head_dims[6];
  /* Multi-policy updates whose policies have DIFFERENT action dimensions (MPE simple_speaker_listener under scripts/train_mpe_rmaddpg.sh:
   * a 3-action speaker and a 5-action listener; r_maddpg.py:60-105 and 236-301 concatenate whatever widths the policies have). Then
   * n_total_agents = agent_offset = 0 and the joint action is described in COLUMNS: joint_act_dim = its width (sum over all agents of their
   * policy's act_dim), joint_act_col = the first column of the update policy's first agent; dims.act_dim / dims.n_agents stay the update
   * policy's own. `joint_acts` is the buffer's joint action [T][B][joint_act_dim] (the caller concatenates the policies' action blocks in
   * policy order; batch->acts is then unused), joint_next_acts (required) has the same width, and ope_rddpg_target_actions writes a policy's
   * agents at columns joint_act_col + a * act_dim. 0 / NULL = every policy has dims.act_dim actions (the fields above). */
  int32_t joint_act_dim, joint_act_col;
  const float* joint_acts;
} ope_rddpg_cfg;

/* which = 0 actor, 1 critic: offsets/sizes of its OPE_QMIX_NPARAM_AGENT tensors; returns the padded length. */
int64_t ope_rddpg_param_layout(const ope_rddpg_cfg* cfg, int32_t which, int64_t* offsets, int64_t* sizes);
int64_t ope_rddpg_workspace_bytes(const ope_rddpg_cfg* cfg);
int ope_rddpg_workspace_init(const ope_rddpg_cfg* cfg, void* workspace, int64_t workspace_bytes, void* stream);
int64_t ope_rddpg_workspace_find(const ope_rddpg_cfg* cfg, const char* name, int64_t* n_floats);
/* Critic update (r_maddpg.py:134-224): target actor scanned over the T+1 observations from a zero state, first action
 * dropped; target critic state follows the BUFFER sequence and branches one cell step per t on the target actions;
 * err_k = (Q_k - target)(1 - shifted dones_env). grad = d(sum_k sum f(err_k) w_b)/d theta_critic + tail
 * [loss_sum, sum(1 - shifted dones_env), sum Q_0, 0]. td_abs_stats [num_q][B][2] = per head, per episode
 * (mean_t |err|, max_t |err|) or NULL. target_noise_u [(T+1)*N*B][A] uniform noise (NULL unless target_gumbel). */
int ope_rddpg_critic_loss_and_grad(const ope_rddpg_cfg* cfg, const ope_fields* batch, const float* theta_actor_tgt,
                                   const float* theta_critic, const float* theta_critic_tgt, const float* target_noise_u,
                                   const float* per_weights, void* workspace, int64_t workspace_bytes, float* grad,
                                   float* td_abs_stats, void* stream);
/* Target actions of ONE policy's agents (get_update_info, r_maddpg.py:60-97): its target actor scanned over the T+1 observations
 * of its dims.n_agents agents, first action dropped, written into columns (agent_offset + a) * A of joint_next_acts
 * [T][B][n_total_agents * A]. cfg / batch are that policy's (only obs and avail_acts are read). */
int ope_rddpg_target_actions(const ope_rddpg_cfg* cfg, const ope_fields* batch, const float* theta_actor_tgt,
                             const float* target_noise_u, void* workspace, int64_t workspace_bytes, float* joint_next_acts,
                             void* stream);
/* Actor update (r_maddpg.py:236-327): actor scanned over obs[:-1], hard gumbel-softmax (noise gumbel_noise_u [T*N*B][A]),
 * actions spliced into N stacked copies of the joint action (multi-policy: copy `rep` replaces block agent_offset + rep); Q_t = head 0 of one critic cell step from the critic's
 * buffer-sequence state; loss = -sum(Q (1 - shifted agent dones)) / sum(1 - shifted agent dones). grad w.r.t.
 * theta_actor + tail [loss_sum, mask_count, sum Q, 0]. */
int ope_rddpg_actor_loss_and_grad(const ope_rddpg_cfg* cfg, const ope_fields* batch, const float* theta_actor,
                                  const float* theta_critic, const float* gumbel_noise_u, void* workspace,
                                  int64_t workspace_bytes, float* grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One-shot gradient all-reduce over xGMI (SURVEY.md section 8(b) `ope_allreduce_*`, 8(e)).
 * The reference has no distributed training (its only collective, `average_gradients`, offpolicy/utils/util.py:148-153,
 * is dead code); the data-parallel step exchanges ONE flat float32 vector [grads | loss_sum | mask_count | qtot_sum | 0]
 * of ~0.5 MB per update, far below the xGMI bandwidth-delay product, inside a 0.4 ms step -- so instead of a ring
 * (2(W-1) hops) every rank PUSHES its vector straight into a slot of every peer's exchange buffer (7 links in parallel,
 * one hop), raises per-chunk flags, waits for its peers' flags and sums the W slots in FIXED RANK ORDER, all in one
 * kernel launch on the caller's stream: results are bitwise identical on every rank and from run to run.
 *
 * Exchange buffer (one per rank, fine-grained device memory, shared with the peers of the node through HIP IPC):
 *   [2 parities][world][chunks] uint32 flags, then [2 parities][world][max_floats] float32 slots.
 * Exception to "the caller owns all memory": the buffer needs an uncached (fine-grained) allocation and an IPC handle,
 * which torch cannot provide, so the library allocates it (ope_allreduce_alloc / _free); handles travel between the
 * processes by whatever the host side has (torch.distributed.all_gather_object in off-policy_amd/dist.py).
 *   epoch   call counter, identical on all ranks, 1, 2, 3, ... (parity = epoch & 1 picks the half of the buffer);
 *   status  device int32, OR-ed with 1 if a peer's flag did not arrive within ctx.timeout_ms (default 10 s; ranks may be
 *           skewed by host work or lazy initialisation) -- a bounded spin: a dead peer cannot hang the GPU forever. The
 *           failure is also visible IN-BAND: every 1024-float chunk whose peers did not arrive is overwritten with NaN
 *           (never with a sum over stale slots), so the following clip norm / loss / parameters are NaN.
 * world == 1 degenerates to a copy through the own slot.
 * ope_allreduce_flat takes the epoch as a launch argument (not capturable in a HIP graph: a replay would repeat it).
 * ope_allreduce_flat_dev keeps it on the device instead: `epoch_state_dev` = uint32[2], zero-filled once by the caller and then
 * owned by the exchange ({epoch of the previous call, ticket}); every launch is identical, so the exchange can sit inside a
 * captured graph (the MADDPG graphed step at world > 1). Use ONE of the two entry points per context, the same on every rank.
 * ---------------------------------------------------------------------------------------------- */
#define OPE_AR_MAX_WORLD 16
#define OPE_AR_IPC_HANDLE_BYTES 64
typedef struct ope_allreduce_ctx {
  int32_t rank, world;
  int64_t max_floats;            /* capacity of one slot (floats), multiple of 1024                                  */
  void* peer[OPE_AR_MAX_WORLD];  /* peer[q] = rank q's exchange buffer as mapped in THIS process; peer[rank] = own     */
  int32_t timeout_ms;            /* 0 = default (10 000)                                                               */
} ope_allreduce_ctx;
int64_t ope_allreduce_buffer_bytes(int64_t max_floats, int32_t world);
int ope_allreduce_alloc(int64_t bytes, void** buf_out);                 /* zero-filled, fine-grained, on the current device */
int ope_allreduce_free(void* buf);
int ope_allreduce_ipc_export(void* buf, void* handle_host);            /* OPE_AR_IPC_HANDLE_BYTES bytes out                */
int ope_allreduce_ipc_import(const void* handle_host, void** mapped_out);
int ope_allreduce_ipc_close(void* mapped);
/* Peer access from the current device to `peer_device` (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess; already
 * enabled counts as success). Called for every peer before its buffer is imported. */
int ope_allreduce_enable_peer(int32_t peer_device);
int ope_allreduce_flat(const ope_allreduce_ctx* ctx_host, uint32_t epoch, float* flat, int64_t n, int32_t* status, void* stream);
int ope_allreduce_flat_dev(const ope_allreduce_ctx* ctx_host, uint32_t* epoch_state_dev, float* flat, int64_t n, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPE_H_ */
