// Argument blocks of the MADDPG / MATD3 kernels.
#pragma once
#include "ope_mixer.h"
#include "ope_rng.h"
#include "ope_wgrad.h"
#include "ope_workspace.h"

namespace ope {

// Blocks of a multi-discrete action vector (ope_ddpg_cfg.n_act_heads): n <= 1 = one block of all A entries
struct ActHeads { int n; int dim[6]; };
static inline ActHeads act_heads_of(int n, const int32_t* dims, int A) {
  ActHeads h;
  h.n = n > 1 ? n : 1;
  for (int i = 0; i < 6; ++i) h.dim[i] = (n > 1 && i < n) ? dims[i] : (i == 0 ? A : 0);
  return h;
}
static inline bool act_heads_ok(int n, const int32_t* dims, int A) {
  if (n <= 1) return n >= 0;
  if (n > 6) return false;
  int s = 0;
  for (int i = 0; i < n; ++i) { if (dims[i] < 1) return false; s += dims[i]; }
  return s == A;
}

struct CriticTdArgs {
  int B, K, K4;
  float gamma; int use_huber; float huber_delta; float per_eps;
  const float* q; const float* q_tgt;        // [B][K4]
  const float* rewards;                      // [N][B][1]: agent 0's slice is read
  const float* dones_env;                    // [B][1]
  const float* per_weights;                  // [B] or null
  float* dq;                                 // [B][K4]  d loss_sum / d Q_k
  float* prio_out;                           // [B] or null
  float* loss_part;                          // [tiles][4]
};

// Fused "critic input gradient restricted to the updating agent's action block" + straight-through gumbel adjoint.
struct ActGradArgs {
  int R, B, N, A, A4, S, Din;
  int a_off;                                 // the copies' own action block is agent a_off + rep of the joint action (0: the copies ARE all agents)
  int a_col;                                 // ... plus this many columns (joint actions of blocks of different widths: a_off = 0, a_col = first column)
  const float* dz1;                          // [R][64] adjoint of the fc1 pre-activation (zero where ReLU is off)
  const float* xhat1; const float* rstd1; const float* mu1;   // LN1 saves: relu(z1) = xhat1/rstd1 + mu1
  const float* mu0; const float* rstd0;      // input-LN saves
  const float* act;                          // [R][A] the actor's (straight-through) action = the row's own action block
  const float* y;                            // [R][A] soft sample
  const float* theta; int fc1_w, fc1_b, fn_w, fn_b;
  float* cvec;                               // [128] scratch: c_i = sum_k gamma_k W_ik ; cb_i = b_i + sum_k W_ik beta_k
  float* dlogits;                            // [R][A4]
  int identity;                              // continuous actions: d action / d actor output = 1 (no gumbel-softmax adjoint; `y` unused)
  ActHeads heads;                            // multi-discrete: the softmax adjoint acts per block (heads.n <= 1: one block of A)
};
int launch_action_grad(const ActGradArgs& a, hipStream_t st);

// shared with the recurrent family (ope_rddpg.hip)
// `rep_off`: copy `rep` replaces action block rep_off + rep (multi-policy updates: the update policy's agents sit at an offset)
int launch_build_cin(const float* cent, const float* acts, const float* repl, int T, int B, int N, int A, int S, int reps, float* out,
                     hipStream_t st, int rep_off = 0);
// nact_agents / a_off: the scatter target cent_nact holds nact_agents (default N) agents per row, this launch's agents start at a_off
int launch_action(const float* logits, const float* avail, NoiseSrc U, int rows, int B, int A, int N, int mode, int t_shift,
                  float* cent_nact, float* act_out, float* soft_out, hipStream_t st, int nact_agents = 0, int a_off = 0,
                  const ActHeads* heads = nullptr, int nact_stride = 0, int nact_col = 0);      // nact_stride > 0: row stride / first column in floats instead
int launch_build_cin_joint(const float* cent, const float* joint, const float* repl, int T, int B, int J, int S, int reps, int Ar, int col0,
                           float* out, hipStream_t st);

// Optimiser step in the tail of a tile launch (ope_ddpg_opt, ope.h): slab reduction + clip + Adam + Polyak behind two grid barriers.
struct TileOpt {
  int on;
  int n_opt;                       // elements Adam updates (prefix of the flat vector)
  int skip_begin, skip_end;        // the registered-but-unused fc_h block
  float* grad;                     // [P + 4] flat gradient + tail (also written by the separate reduction)
  float* theta; float* tgt; float* m; float* v;
  float lr, beta1, beta2, eps, max_norm, wd, tau, qden;
  int do_polyak, step;
  int* step_counter;               // device [count, ticket] or null
  int* sync;                       // device int[8] (8-byte aligned), zero at workspace_init: {error flag, -, critic arrivals (64-bit), actor arrivals (64-bit)}
  float* gsq;                      // [workgroups] partial sums of squares
  float* stats;                    // [4] or null
};
// fused small-network path (ope_ddpg_fused.hip): one launch per network update + one slab reduction
bool ddpg_fused_ok(int N, int A, int D, int S, int K);
bool ddpg_tile_opt_ok(int N, int A, int D, int S, int K, int B);    // the optimiser tail needs every workgroup of both launches co-resident
// workgroups of the slab reduction = half the floats of the "gsq_critic" / "gsq_actor" region ([trunk partials | head partials])
int ddpg_fused_gsq_blocks(int N, int A, int D, int S, int K, bool critic);
int64_t ddpg_fused_slab_floats(int N, int A, int D, int S, int K, int B);
int launch_ddpg_critic_fused(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor_tgt, const float* theta_critic,
                             const float* theta_critic_tgt, const float* U, const float* per_w, float* slabs, float* grad, float* prio_out,
                             float* gsq, hipStream_t st, const TileOpt* opt = nullptr);
int launch_ddpg_actor_fused(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor, const float* theta_critic,
                            const float* U, float* slabs, float* grad, float* gsq, hipStream_t st, const TileOpt* opt = nullptr);

}  // namespace ope
