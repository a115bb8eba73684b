// Argument blocks and device helpers shared by the agent-network kernels (forward and backward).
#pragma once
#include "ope_common.h"
#include "ope_wgrad.h"

namespace ope {

__device__ __forceinline__ int ope_round4_dev(int x) { return (x + 3) & ~3; }

// ---- argument blocks (passed by value to the kernels) -----------------------------------------------------
// "Replicated rows" input of the trunk (actor update of the MADDPG families, maddpg.py:207-227 / r_maddpg.py:291-301): the R rows
// are reps = N copies of T*B base rows x = [cent_obs | joint action], copy `rep` differing from its base row only in action block
// `rep`. The first layer is affine in x once the row's LayerNorm statistics are known, so it is evaluated ONCE per base row
// (u = W1 (gamma o (x - mean)), mean, M2 = sum (x - mean)^2) and corrected per copy with the 64 x A block of W1 gamma that the copy touches;
// the [R][D] input (463 MB at MMM2 / B = 128) is never built and the first layer's 7.4 GMAC shrink to 0.74 + 0.27.
struct RepIn {
  const float* u;      // [T*B][64]   W1 (gamma o (x_base - mean_base))
  const float* s12;    // [T*B][2]    mean_base, sum (x_base - mean_base)^2
  const float* acts;   // [T][NT][B][A] buffer actions of ALL agents (the blocks being replaced)
  const float* repl;   // [R][A]      replacement block of row r = (t*N + rep)*B + b: joint-action block a0 + rep
  const float* wblk;   // [64][NT*A]  (W1 gamma)[:, S:]
  const float* wsum;   // [64]        W1 gamma
  const float* cst;    // [64]        W1 beta + b1
  float* u_out; float* s12_out;    // MODE 2 (producer over the base rows) writes these
  int T, B, N, A, S;   // N = copies per base row
  int NT, a0;          // agents in the joint action, first agent the copies replace (NT = N, a0 = 0: every agent has its copy)
};

struct TrunkFwdArgs {
  RepIn rep;           // only read by the replicated-rows launches
  const float* x;      // [R][D] input rows
  int R, D;
  const float* theta;  // flat parameters of the net being evaluated
  AgentLayout L;
  float* gi;           // [R][192]  W_ih a2 + b_ih (recurrent nets)
  float* a2_out;       // [R][64]   trunk output (LN2 output) -- MLP nets: written instead of gi when non-null
  float* head_out;     // MLP nets, optional: [R][head_dim] = W_head a2 + b_head (L.q_w / L.q_b) fused into the same launch
  int head_dim;        //   (head_dim <= 16; the small MADDPG-family heads)
  long long* dbg;      // optional per-wave s_memtime stamps [waves][16] (ope_set_debug)
  // saved for backward (live net only)
  float* mu0;          // [R] input-LN mean
  float* rstd0;        // [R] input-LN 1/std
  float* xhat1;        // [R][64] normalised (pre-affine) LN1 input
  float* rstd1;        // [R]
  float* mu1;          // [R] mean of relu(fc1) (optional; lets a backward pass rebuild relu(z1) = xhat1/rstd1 + mu1)
  uint64_t* mask1;     // [R] ReLU mask of fc1 (bit f = z1[f] > 0)
  float* xhat2;
  float* rstd2;
  uint64_t* mask2;
  int tanh_act;        // OPE_DIMS_TANH: tanh behind fc1 / fc2 (trunk_fwd3 only); the mask slots then hold the row means of the activations
  int no_fn;           // OPE_DIMS_NO_FEATURE_NORM: rows enter fc1 as they are (statistics forced to mean 0, 1/std 1; theta holds gamma = 1, beta = 0)
  // observations left in the store (trunk_fwd4 only): x = the store's obs ring, local row r is batch row ref_row0 + r
  ObsRef ref;
  int ref_row0;
  LivePlan lp;         // lp.hdr != null (trunk_fwd4 pair only): R and every output are the plan's packed rows, x is read through lp.srcrow
};

struct GruFwdArgs {
  int nets;            // 1 or 2 (live [+ target]) processed in one launch
  int NB;              // rows per time step (agents * episodes)
  int L;               // sequence length (T+1)
  const float* theta0; const float* theta1;
  const float* gi0; const float* gi1;   // [L][NB][192]
  float* h0out; float* h1out;           // [L][NB][64]
  const float* hinit;                   // [NB][64] or null (zeros): initial state of net 0
  const float* hinit1;                  // same for net 1 (time-chunked scans continue from the previous chunk's last state)
  long long* dbg;                       // optional per-compute-wave phase cycle sums [rows][4][8] (ope_set_debug; gru4 only)
  int whh_off, bhh_off;
  float* rg; float* zg; float* ng; float* ghn;  // [L][NB][64] gate saves (live) or null
  int family, waves;                    // scan kernel family (4 | 1) / compute waves per row (4 | 2) asked for by the caller's cfg;
                                        // 0 = the process default (ope_set_scan_kernel / OPE_GRU, OPE_GRU4_W), then by row count
  int pair_cus;                         // packed rows: > 0 = CUs of the device, the chains are dealt out longest-beside-shortest (set by the launcher)
  LivePlan lp; int B, N;                // lp.hdr != null (gru4 only): packed rows -- row r = agent * B + j walks the lp.len[j] steps of the
                                        // episode ranked j; step t's gi / h / saves are at row N * cum[t] + agent * n[t] + j
};

struct HeadFwdArgs {
  int64_t R;           // rows = L * NB
  int NB, B, N, T, A;
  const float* theta0; const float* theta1;
  AgentLayout L;
  const float* h0; const float* h1;     // GRU states of live / target net [R][64]
  const float* acts;                    // [T][NB][A] one-hot
  const float* avail;                   // [T+1][NB][A] or null
  int double_q;
  int no_ln;                            // MLP nets: the head reads the trunk output directly (no rnn.norm)
  int target_mask_avail;                // plain (non double-Q) targets: mask unavailable actions (mqmix.py:167-170)
  float* agent_q;                       // [T][B][N]
  float* agent_nq;                      // [T][B][N]
  int* act_idx;                         // [T][NB]
  float* xhat_o; float* rstd_o;         // [R][64], [R] saved normalised GRU output (live)
  float* q_out;                         // MODE 1: [R][A]
  float* q_all;                         // optional debug output [R][A]
  int64_t r_begin;                      // first row handled by this launch (time-chunked launches); rows [r_begin, R)
  Transp4 side;                         // weight transposes carried as extra workgroups of this launch (side.total = 0: none)
  int main_blocks;                      // workgroups doing head work; blockIdx.x >= main_blocks run `side` (set by the launcher)
};

struct HeadBwdArgs {
  int64_t R;           // rows with a loss term = T * NB
  int NB, B, N, A;
  const float* theta;  // live
  AgentLayout L;
  const float* xhat_o; const float* rstd_o;
  const int* act_idx;
  int no_ln;                            // MLP nets: dh_out = dq * Wq[a] (adjoint of the trunk output)
  const float* d_agent_q;               // [T][B][N]
  float* dh_out;                        // [R][64]
  float* dqoh;                          // [R][A16] dq placed at the chosen action, zeros elsewhere (A16 = round4(A))
};

struct GruBwdArgs {
  int NB, T;           // backward over t = T-1 .. 0
  const float* theta; int whh_off;
  const float* h;      // [T+1][NB][64] forward states
  const float* rg; const float* zg; const float* ng; const float* ghn;
  const float* dh_out; // [T][NB][64]
  float* dgi;          // [T][NB][192] = (dr_pre, dz_pre, dn_pre)
  float* dghn;         // [T][NB][64]  = dn_pre * r
  int t_lo;            // this launch covers t = T-1 .. t_lo (time-chunked BPTT); 0 = to the start
  const float* dh_in;  // [NB][64] adjoint carried in from the later chunk, or null (zeros)
  float* dh_carry;     // [NB][64] adjoint w.r.t. h_{t_lo - 1} handed to the earlier chunk, or null
  long long* dbg;      // optional per-compute-wave phase cycle sums [NB][4][8] (ope_set_debug; gru4 only)
  int family, waves;   // as GruFwdArgs
  LivePlan lp; int B, N;   // as GruFwdArgs: row r = agent * B + j walks t = min(lp.len[j], T) - 1 .. 0 (gru4 only; t_lo = 0, no carries)
};

struct TrunkBwdArgs {
  int R;               // T * NB
  const float* theta;  // live
  const float* thetaT; // transposed copies: wihT [64][192] at 0, fc2T [64][64] at 64*192
  AgentLayout L;
  const float* dgi;    // [R][192]
  const float* da2_in; // MLP nets: adjoint of the trunk output given directly (dgi unused)
  const float* dout;   // MLP nets, optional instead of da2_in: adjoint of the Linear head's output [R][ldk]; the trunk-output
  int ldk, hdim;       //   adjoint da2 = dout W_head (L.q_w, [hdim][64]) is then formed in-kernel
  const float* xhat1; const float* rstd1; const uint64_t* mask1;
  const float* xhat2; const float* rstd2; const uint64_t* mask2;
  float* dz1; float* dz2;  // [R][64]
  int tanh_act;        // OPE_DIMS_TANH (trunk_bwd3 only): mask1 / mask2 hold the activations' row means (float in the low word), not ReLU bits
  const int* R_dev;    // non-null (trunk_bwd4 only): the rows of this launch, read on the device (<= R; the live plan's R1L)
};

// ---- second hidden block (layer_N = 2; ope_block.hip) ----------------------------------------------------------------------------
// forward: a3 = LN(ReLU(fc2.1 a2 + b)) and the GRU input projection gi = W_ih a3 + b_ih (the trunk kernels stop at a2 for such nets)
struct BlockFwdArgs {
  int R;
  const float* x;                 // [R][64] trunk output a2 (LayerNorm output of block fc2.0)
  const float* theta; AgentLayout L;
  float* gi;                      // [R][192]
  float* xhat3; float* rstd3; uint64_t* mask3;   // saved for the backward pass (live net) or null
};
// adjoint: dgi -> da3 = W_ih^T dgi -> LayerNorm / ReLU adjoint -> dz3 (pre-activation adjoint of fc2.1: its weight gradient) -> da2 = fc2.1^T dz3
struct BlockBwdArgs {
  int R;
  const float* dgi;               // [R][192]
  const float* theta; AgentLayout L;
  const float* wihT;              // [64][192]
  const float* fc2bT;             // [64][64]
  const float* xhat3; const float* rstd3; const uint64_t* mask3;
  float* dz3; float* da2;         // [R][64]
};
int launch_block_fwd(const BlockFwdArgs& a, hipStream_t st);
int launch_block_bwd(const BlockBwdArgs& a, hipStream_t st);

// allow4 = false: never the LDS-resident kernel (ope_trunk4.hip) -- the pair launcher's own fall-back, and tests that pin a family
int launch_trunk_fwd(const TrunkFwdArgs& a, bool save, hipStream_t st, bool allow4 = true);
bool trunk4_pair_can(int D, int64_t R, int path, bool tanh_act);      // would the pair launch run trunk_fwd4 on a gathered batch of this shape
int trunk4_pair_min_rows();      // rows from which "by shape" picks trunk_fwd4 for the live + target pair (ope_trunk4.hip)
int launch_trunk_fwd4_single(const TrunkFwdArgs& a, bool save, hipStream_t st);      // OPE_OK / error, or 1 = not this kernel's launch
// live (saving) + target trunk of the same input rows: one launch of trunk_fwd4 (ope_trunk4.hip) when the shape allows, else two launches
int launch_trunk_fwd_pair(const TrunkFwdArgs& live, const TrunkFwdArgs& tgt, int path, hipStream_t st);
int launch_trunk_fwd2(const TrunkFwdArgs& a, bool save, hipStream_t st);   // workgroup-cooperative form (ope_trunk2.hip)
int launch_trunk_fwd3(const TrunkFwdArgs& a, bool save, hipStream_t st);   // persistent, weights in registers (ope_trunk2.hip)
// Replicated-rows trunk (RepIn): `base` = TrunkFwdArgs over the T*B base rows (x = the base input [T*B][D]), `a` over the R = T*N*B
// copies (x unused). scratch: 64*NT*A + 128 floats. false from trunk_rep_ok = use the plain launch on a materialised input.
bool trunk_rep_ok(int D, int NT, int A, int copies);
int launch_trunk_fwd_rep(const TrunkFwdArgs& base, const TrunkFwdArgs& a, float* scratch, hipStream_t st);
// scans with at most this many rows use the latency-oriented four-waves-per-row kernels (ope_gru4.hip)
constexpr int kGru4MaxRows = 1024;
extern int g_scan_family;   // 0 auto | 1 | 4   (ope_set_scan_kernel / OPE_GRU)
extern int g_scan_waves;    // 0 auto | 2 | 4   (ope_set_scan_kernel / OPE_GRU4_W)
int launch_head_fwd_mfma(const HeadFwdArgs& a, hipStream_t st);   // ope_head.hip: 16 rows per wave, q on the matrix pipe
int launch_head_bwd_rows(const HeadBwdArgs& a, hipStream_t st);
int launch_gru_fwd(const GruFwdArgs& a, hipStream_t st);
int launch_gru_fwd1(const GruFwdArgs& a, hipStream_t st);   // one wave per row (ope_gru1.hip)
int launch_gru_fwd4(const GruFwdArgs& a, hipStream_t st);   // output-split over four waves per row (ope_gru4.hip)
int launch_head_fwd(const HeadFwdArgs& a, int mode, hipStream_t st);
int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st);
int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st);
int launch_gru_bwd1(const GruBwdArgs& a, hipStream_t st);
int launch_gru_bwd4(const GruBwdArgs& a, hipStream_t st);
int launch_trunk_bwd(const TrunkBwdArgs& a, hipStream_t st);
int launch_trunk_bwd3(const TrunkBwdArgs& a, hipStream_t st);   // persistent cooperative form (ope_trunk_bwd3.hip)
// recurrent trunk adjoint: trunk_bwd4 (ope_trunk_bwd4.hip: weights in LDS, a wave per tile) when the shape allows and `path` (ope_qmix_cfg.trunk_path) asks, else trunk_bwd3
int launch_trunk_bwd_path(const TrunkBwdArgs& a, int path, hipStream_t st);
bool trunk_bwd4_can(int64_t R, int path, bool tanh_act);      // would launch_trunk_bwd_path run trunk_bwd4 on R rows
int launch_transpose_weights(const float* theta, const AgentLayout& L, float* thetaT, hipStream_t st);
int launch_transpose(const float* src, int rows, int cols, float* dst, hipStream_t st);

// ---- device helpers ---------------------------------------------------------------------------------------
// 64->64 (x tiles) GEMM step of the transposed chain: acc[it] += W[16it + j][:] . act   (W row-major, ld floats)
template <int NT>
__device__ __forceinline__ void gemm64(const float* __restrict__ W, int ld, int j, int g, const f32x4 (&act)[4],
                                       f32x4 (&acc)[NT]) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    f32x4 w[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) w[it] = *reinterpret_cast<const f32x4*>(W + (int64_t)(16 * it + j) * ld + 16 * ft + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int it = 0; it < NT; ++it) acc[it] = mfma16(w[it][r], act[ft][r], acc[it]);
  }
}

// Same, RT row tiles sharing every weight fragment (8 MFMAs per float4 of weights at RT = 2).
template <int NT, int RT>
__device__ __forceinline__ void gemm64rt(const float* __restrict__ W, int ld, int j, int g, const f32x4 (&act)[RT][4],
                                         f32x4 (&acc)[RT][NT]) {
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    f32x4 w[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) w[it] = *reinterpret_cast<const f32x4*>(W + (int64_t)(16 * it + j) * ld + 16 * ft + 4 * g);
    // r outermost: consecutive MFMAs accumulate into different tiles (no back-to-back dependent pairs)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int it = 0; it < NT; ++it) acc[t][it] = mfma16(w[it][r], act[t][ft][r], acc[t][it]);
  }
}

// ReLU then LayerNorm over the 64 features of a row held as acc[it][r] = feature 16it+4g+r (4 lanes per row).
// Outputs: act = xhat*gamma+beta; if SAVE, acc is overwritten with xhat; rstd; mbits bit (4it+r) = (z > 0).
template <bool SAVE>
__device__ __forceinline__ void relu_ln64(f32x4 (&acc)[4], const float* __restrict__ gam, const float* __restrict__ bet,
                                          int g, f32x4 (&act)[4], float* rstd_out, uint32_t* mbits, float* mu_out = nullptr) {
  uint32_t mb = 0;
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = acc[it][r];
      if (z > 0.f) mb |= 1u << (4 * it + r);
      const float rl = fmaxf(z, 0.f);
      acc[it][r] = rl;
      s += rl;
    }
  const float mu = rowsum4(s) * (1.0f / OPE_H);
  float v = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = acc[it][r] - mu;
      v = fmaf(d, d, v);
    }
  const float rstd = 1.0f / sqrtf(rowsum4(v) * (1.0f / OPE_H) + OPE_LN_EPS);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 16 * it + 4 * g);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + 16 * it + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (acc[it][r] - mu) * rstd;
      act[it][r] = fmaf(xh, gm[r], bt[r]);
      if (SAVE) acc[it][r] = xh;
    }
  }
  *rstd_out = rstd;
  *mbits = mb;
  if (mu_out) *mu_out = mu;
}

// Merge the four lanes' 16-bit ReLU masks into one 64-bit row mask (bit f = feature f) and store with rstd.
__device__ __forceinline__ void store_mask_rstd(uint64_t* mask, float* rstd, int row, int g, uint32_t mbits, float rs) {
  // lane (j,g) holds bits for features 16it + 4g + r at position 4it + r
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t nib = (mbits >> (4 * it)) & 0xF;
    const int f0 = 16 * it + 4 * g;  // 0..60
    if (it < 2) lo |= nib << f0; else hi |= nib << (f0 - 32);
  }
  lo |= __shfl_xor((int)lo, 16, 64); hi |= __shfl_xor((int)hi, 16, 64);
  lo |= __shfl_xor((int)lo, 32, 64); hi |= __shfl_xor((int)hi, 32, 64);
  if (g == 0) {
    mask[row] = ((uint64_t)hi << 32) | lo;
    rstd[row] = rs;
  }
}

// LayerNorm of one 64-float row by a single thread (head kernels). gam/bet may live in LDS.
__device__ __forceinline__ void ln64_thread(const float* __restrict__ hrow, const float* gam, const float* bet,
                                            float (&y)[OPE_H], float* xhat_out, float* rstd_out) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < OPE_H; k += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(hrow + k);
    y[k] = v[0]; y[k + 1] = v[1]; y[k + 2] = v[2]; y[k + 3] = v[3];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mu = s * (1.0f / OPE_H);
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < OPE_H; ++k) {
    const float d = y[k] - mu;
    var = fmaf(d, d, var);
  }
  const float rstd = 1.0f / sqrtf(var * (1.0f / OPE_H) + OPE_LN_EPS);
#pragma unroll
  for (int k = 0; k < OPE_H; k += 4) {
    f32x4 xh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[r] = (y[k + r] - mu) * rstd;
      y[k + r] = fmaf(xh[r], gam[k + r], bet[k + r]);
    }
    if (xhat_out) *reinterpret_cast<f32x4*>(xhat_out + k) = xh;
  }
  if (rstd_out) *rstd_out = rstd;
}

}  // namespace ope
