// Argument blocks for the mixer / TD kernels.
#pragma once
#include "ope_agent.h"

namespace ope {

struct MixerFwdArgs {
  int TB, B, N, S;                       // TB = T*B rows; both nets evaluated in one launch
  const float* theta0; const float* theta1;
  MixerLayout L;
  const float* share;                    // [T+1][B][S] centralized observations
  const float* agent_q;                  // [TB][N] live chosen-action q
  const float* agent_nq;                 // [TB][N] target q at t+1
  float* qtot; float* nqtot;             // [TB]
  // saved for backward (live net)
  float* hw1; float* hw2; float* hb2;    // [TB][64] post-ReLU hyper hidden layers
  float* v1;                             // [TB][N*32] pre-abs w1
  float* hpre;                           // [TB][32]   pre-ELU hidden
  float* v2;                             // [TB][32]   pre-abs w2
  long long* dbg;                        // optional per-wave s_memtime stamps [waves][8] (profiling builds of the schedule)
  int k_stagger;                         // cooperative form: workgroups start their K loops at different chunks
  int path;                              // 0 auto, 1 resident weights (mixer_fwd3), 2 streamed weights (mixer_fwd2), 3 wide-state GEMM
  float* wide_slab;                      // wide-state path: stream-K partial sums of the first hyper-layers (wide_slab_floats())
};

// ---- wide-state path (ope_mixer_wide.hip): stream-K decomposition shared by the GEMM kernel, its consumer and the workspace plan ----
constexpr int kWideBM = 128, kWideBN = 224, kWideMaxWG = 256, kWideStageK = 32;
constexpr int kWideAutoS = 256;          // auto: states wider than this take the GEMM path
struct WidePlan { int nrb, ntiles, nst, units, nwg, maxseg; };
__host__ __device__ inline WidePlan wide_plan(int TB, int S) {
  WidePlan p;
  p.nrb = (TB + kWideBM - 1) / kWideBM;            // 128-row blocks of the T*B output rows
  p.ntiles = 2 * p.nrb;                            // tile = 2 * row block + net
  p.nst = (S + kWideStageK - 1) / kWideStageK;     // K stages per tile
  p.units = p.ntiles * p.nst;
  p.nwg = p.units < kWideMaxWG ? p.units : kWideMaxWG;
  const int upw = (p.units + p.nwg - 1) / p.nwg;   // most units a workgroup gets
  p.maxseg = (upw + p.nst - 1) / p.nst + 1;        // most tiles its contiguous range can touch
  return p;
}
// workgroup w covers units [wide_bound(w), wide_bound(w + 1))
__host__ __device__ inline int wide_bound(const WidePlan& p, int w) { return (int)((int64_t)w * p.units / p.nwg); }
// the workgroup that covers unit u
__host__ __device__ inline int wide_owner(const WidePlan& p, int u) { return (int)((((int64_t)u + 1) * p.nwg - 1) / p.units); }
inline int64_t wide_slab_floats(int TB, int S) {
  const WidePlan p = wide_plan(TB, S);
  return (int64_t)p.nwg * p.maxseg * kWideBM * kWideBN;
}
int launch_mixer_wide_gemm(const MixerFwdArgs& a, hipStream_t st);

struct TdArgs {
  int B, N;
  float gamma; int use_huber; float huber_delta;
  const float* rewards;                  // [T][N][B][1]
  const float* dones_env;                // [T][B][1]
  const float* per_weights;              // [B] or null
};

struct MixerBwdArgs {
  int TB, N;
  const float* theta; const float* thetaT;  // thetaT: w1bT [64][N*32] then w2bT [64][32]
  MixerLayout L;
  TdArgs td;
  const float* qtot; const float* nqtot; const float* agent_q;
  const float* hw1; const float* hw2; const float* hb2; const float* v1; const float* hpre; const float* v2;
  float* loss_part;                      // [tiles][4] = loss_sum, mask_count, qtot_sum, 0
  float* err_abs;                        // [TB]
  float* dqtot;                          // [TB][4] (dQ_tot, 0, 0, 0)
  float* d_agent_q;                      // [TB][N]
  float* d_b1; float* d_v2;              // [TB][32]
  float* d_v1;                           // [TB][N*32]
  float* d_hw1; float* d_hw2; float* d_hb2;  // [TB][64] adjoints of the hyper hidden PRE-activations
};

struct VdnArgs {
  int TB, N;
  TdArgs td;
  const float* agent_q; const float* agent_nq;
  float* loss_part; float* err_abs; float* d_agent_q;
};

// ---- fused (t, b)-row chain (ope_chain.hip) -----------------------------------------------------------------------------------------
// First hyper-network layers of both nets over the T*B rows: a GEMM on the centralized state alone (no dependence on the agent networks)
// One 16-feature output tile of a hyper-network layer that reads the state: rows [16 k, 16 k + 16) of a Linear's weight
struct HypTile {
  int w_off;       // theta offset of the tile's first weight row ([.][S] row-major)
  int b_off;       // theta offset of its 16 biases
  int dst;         // output array (HypFirstArgs::out index)
  int col;         // first column inside that array's rows
  int relu;        // ReLU on the output (the hidden layers of the two-layer hyper-networks and of hyper_b2)
};
constexpr int kHypMaxTiles = 48;               // one-layer hyper-networks with 16 agents: 2 * 16 + 8
// output arrays of the first-layer kernel, per net: post-ReLU hidden layers of hyper_w1 / hyper_w2 / hyper_b2 [TB][64], hyper_b1's output [TB][32],
// and (one-layer hyper-networks only) the pre-abs w1 [TB][N*32] and w2 [TB][32] themselves
enum { HYP_HW1 = 0, HYP_HW2 = 1, HYP_HB2 = 2, HYP_HB1 = 3, HYP_V1 = 4, HYP_V2 = 5, HYP_NOUT = 6 };
struct HypFirstArgs {
  int TB, B, S;
  const float* theta0; const float* theta1;   // live / target flat parameters
  const float* share;                         // [T+1][B][S]
  float* out[2][HYP_NOUT]; int ld[HYP_NOUT];  // [net][array], row length of each array ([0] live on s_t, [1] target on s_{t+1})
  int ntiles; HypTile tile[kHypMaxTiles];
  Transp4 side;                               // weight transposes carried as extra workgroups (side.total = 0: none)
  int main_blocks;                            // set by the launcher
  LivePlan lp;                                // lp.hdr != null: the packed (t, b) rows of the live plan (outputs in packed order)
};
// fills ntiles / tile[] / ld[] for a mixer layout (two-layer: 14 tiles; one-layer: 2 N + 8)
void hyp_tiles_for(const MixerLayout& L, int N, int S, HypFirstArgs* a);
// Agent q heads + chosen / target selection + second mixer stage of both nets + TD / loss + mixer adjoint + head adjoint, one launch
struct ChainArgs {
  int TB, B, N, T, A, NB;
  int vdn, double_q;
  int mask_target_max;                        // plain (non double-Q) targets: maximum over the available actions only (OPE_DIMS_MASK_TARGET_MAX)
  const float* theta0; const float* theta1;   // full flat vectors [agent | mixer] of the live / target nets
  AgentLayout AL; MixerLayout ML;
  const float* mixT;                          // live w1bT [64][N*32] then w2bT [64][32]
  const float* h0; const float* h1;           // GRU states [T+1][N*B][64] of the live / target net
  const float* acts; const float* avail;      // [T][N*B][A] one-hot, [T+1][N*B][A] or null
  const float* hw1[2]; const float* hw2[2]; const float* hb2[2]; const float* hb1[2];   // HypFirstArgs outputs
  const float* v1x[2]; const float* v2x[2];   // one-layer hyper-networks (ML.one_layer): pre-abs w1 [TB][N*32] / w2 [TB][32] straight from the state
  TdArgs td;
  // saved for the BPTT / trunk adjoint / weight-gradient kernels that follow
  float* xhat_o; float* rstd_o; int* act_idx;
  float* loss_part; float* err_abs; float* dqtot;
  float* d_v1; float* d_v2; float* d_b1; float* d_hw1; float* d_hw2; float* d_hb2;
  float* dh_out; float* dqoh;
  // test / debug outputs (null unless ope_qmix_cfg.debug)
  float* q_all; float* agent_q; float* agent_nq; float* qtot; float* nqtot; float* v1; float* v2; float* hpre; float* d_agent_q;
  long long* dbg;                             // optional per-wave s_memtime stamps [tile][wave][10] (tools/chain_phases.py)
  LivePlan lp;                                // lp.hdr != null: packed rows (LivePlan, ope_common.h) -- h0 / h1 and every output except err_abs
                                              // are indexed by packed rows, the batch fields (acts, avail, td.*) by the batch's own
};
bool qchain_shape_ok(int N, int A);
int launch_mixer_hyp(const HypFirstArgs& a, hipStream_t st);
int launch_qchain(const ChainArgs& a, hipStream_t st);

int launch_mixer_fwd(const MixerFwdArgs& a, hipStream_t st);
int launch_mixer_bwd(const MixerBwdArgs& a, hipStream_t st);
int launch_vdn(const VdnArgs& a, hipStream_t st);
int launch_td_stats(const float* err_abs, int T, int B, float* out, hipStream_t st);

}  // namespace ope
