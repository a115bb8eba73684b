// Recurrent MADDPG / MATD3 update on sampled episodes.
//   R_MADDPG.get_update_info / shared_train_policy_on_batch   offpolicy/algorithms/r_maddpg/r_maddpg.py:44-105, 114-331
//   R_MADDPG_Actor / R_MADDPG_Critic                          offpolicy/algorithms/r_maddpg/algorithm/r_actor_critic.py:7-129
//   RNNBase / RNNLayer                                        offpolicy/algorithms/utils/rnn.py:4-47
//   R_MADDPGPolicy.get_actions                                offpolicy/algorithms/r_maddpg/algorithm/rMADDPGPolicy.py:61-131
// Both networks are the QMIX agent network shape (trunk -> GRU -> LayerNorm -> Linear), so sequences run through the
// shared trunk / GRU-scan / wgrad / finalize kernels. What is specific to this family:
//   * the reference steps the (target) critic twice per time step: once along the buffer sequence (state update) and
//     once "sideways" on substituted actions whose state is thrown away. The sideways steps of all T time steps are
//     independent given the buffer-sequence states, so they run as ONE row-parallel GRU-cell kernel over T*B (critic
//     update) or T*N*B (actor update) rows instead of T sequential launches;
//   * dense heads (all outputs carry a gradient) and the TD / actor objectives on [T][B] grids with shifted done masks.
#include <string.h>

#include "ope_agent.h"
#include "ope_ddpg.h"

namespace ope {

// ---------------------------------------------------------------------------------------------------------
// One GRU cell step for R independent rows (nn.GRU gate order r,z,n):
//   gh = W_hh h_prev + b_hh ; r = s(gi_r+gh_r) ; z = s(gi_z+gh_z) ; n = tanh(gi_n + r gh_n) ; h' = (1-z) n + z h_prev
// Row r = (t, rep, b); its previous state is hprev[(t + prev_shift)*B + b] (zeros when t + prev_shift < 0):
//   prev_shift =  0  target critic: state AFTER consuming buffer step t   (r_maddpg.py:172-178)
//   prev_shift = -1  live critic in the actor update: state BEFORE buffer step t (r_maddpg.py:303-306)
// One wave = 16 rows; W_hh h_prev as 12 MFMA output tiles in the transposed-chain convention (ope_agent.h).
// ---------------------------------------------------------------------------------------------------------
struct CellFwdArgs {
  int R, B, reps, prev_shift;
  const float* gi;      // [R][192]
  const float* hprev;   // [T][B][64]
  const float* theta; int whh_off, bhh_off;
  float* hout;          // [R][64]
  float* rg; float* zg; float* ng; float* ghn;   // [R][64] saves or null
};

__global__ void __launch_bounds__(256) gru_cell_fwd_kernel(CellFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  if (row0 >= a.R) return;
  const int row = row0 + j;
  const bool ok = row < a.R;
  const int rr = ok ? row : a.R - 1;
  const int t = rr / (a.reps * a.B), b = rr % a.B;
  const int tp = t + a.prev_shift;
  f32x4 act[4];
  const float* hp = a.hprev + ((int64_t)(tp < 0 ? 0 : tp) * a.B + b) * OPE_H;
  const float keep = tp < 0 ? 0.f : 1.f;
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    act[ft] = *reinterpret_cast<const f32x4*>(hp + 16 * ft + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) act[ft][r] *= keep;
  }
  f32x4 acc[12];
#pragma unroll
  for (int it = 0; it < 12; ++it) acc[it] = *reinterpret_cast<const f32x4*>(a.theta + a.bhh_off + 16 * it + 4 * g);
  gemm64<12>(a.theta + a.whh_off, OPE_H, j, g, act, acc);
  const float* gir = a.gi + (int64_t)rr * 3 * OPE_H;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int f = 16 * it + 4 * g;
    const f32x4 g_r = *reinterpret_cast<const f32x4*>(gir + f);
    const f32x4 g_z = *reinterpret_cast<const f32x4*>(gir + OPE_H + f);
    const f32x4 g_n = *reinterpret_cast<const f32x4*>(gir + 2 * OPE_H + f);
    f32x4 vr, vz, vn, vg, vh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vr[r] = sigmoidf_(g_r[r] + acc[it][r]);
      vz[r] = sigmoidf_(g_z[r] + acc[4 + it][r]);
      vg[r] = acc[8 + it][r];
      vn[r] = tanhf(g_n[r] + vr[r] * vg[r]);
      vh[r] = (1.0f - vz[r]) * vn[r] + vz[r] * act[it][r];
    }
    if (ok) {
      const int64_t o = (int64_t)row * OPE_H + f;
      *reinterpret_cast<f32x4*>(a.hout + o) = vh;
      if (a.rg) {
        *reinterpret_cast<f32x4*>(a.rg + o) = vr;
        *reinterpret_cast<f32x4*>(a.zg + o) = vz;
        *reinterpret_cast<f32x4*>(a.ng + o) = vn;
        *reinterpret_cast<f32x4*>(a.ghn + o) = vg;
      }
    }
  }
}

// Adjoint of the cell step w.r.t. its input projection gi only (the previous state and W_hh carry no gradient in the
// actor update: critic parameters are frozen and the buffer-sequence state is a constant, r_maddpg.py:239-306).
//   dn = dh (1-z) ; dn_pre = dn (1-n^2) ; dz_pre = dh (h_prev - n) z (1-z) ; dr_pre = dn_pre ghn r (1-r)
struct CellBwdArgs {
  int R, B, reps, prev_shift;
  const float* dh; const float* hprev;
  const float* rg; const float* zg; const float* ng; const float* ghn;
  float* dgi;           // [R][192]
};

__global__ void __launch_bounds__(256) gru_cell_bwd_kernel(CellBwdArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.R * 16) return;
  const int row = (int)(i >> 4), f = (int)(i & 15) * 4;
  const int t = row / (a.reps * a.B), b = row % a.B;
  const int tp = t + a.prev_shift;
  const int64_t o = (int64_t)row * OPE_H + f;
  const f32x4 dh = *reinterpret_cast<const f32x4*>(a.dh + o);
  const f32x4 vr = *reinterpret_cast<const f32x4*>(a.rg + o);
  const f32x4 vz = *reinterpret_cast<const f32x4*>(a.zg + o);
  const f32x4 vn = *reinterpret_cast<const f32x4*>(a.ng + o);
  const f32x4 vg = *reinterpret_cast<const f32x4*>(a.ghn + o);
  f32x4 hp = {0.f, 0.f, 0.f, 0.f};
  if (tp >= 0) hp = *reinterpret_cast<const f32x4*>(a.hprev + ((int64_t)tp * a.B + b) * OPE_H + f);
  f32x4 dr, dz, dn;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float dnp = dh[r] * (1.0f - vz[r]) * (1.0f - vn[r] * vn[r]);
    dn[r] = dnp;
    dz[r] = dh[r] * (hp[r] - vn[r]) * vz[r] * (1.0f - vz[r]);
    dr[r] = dnp * vg[r] * vr[r] * (1.0f - vr[r]);
  }
  float* d = a.dgi + (int64_t)row * 3 * OPE_H + f;
  *reinterpret_cast<f32x4*>(d) = dr;
  *reinterpret_cast<f32x4*>(d + OPE_H) = dz;
  *reinterpret_cast<f32x4*>(d + 2 * OPE_H) = dn;
}

// ---------------------------------------------------------------------------------------------------------
// Adjoint of  out = Linear_K(LayerNorm(h))  w.r.t. h for a dense output gradient (every output carries one).
//   dy = sum_k dout[k] W[k][f] ; dyh = dy gamma ; dh = rstd (dyh - mean dyh - xhat mean(dyh xhat))
// 16 rows per wave in the trunk kernels' lane convention: lane (j, g) holds features 16c + 4g + 0..3 (c = 0..3) of row 16w + j, so a
// row is four 16-byte loads / stores per lane, the two row means are 16 local values + two VALU lane swaps (rowsum4), and the head
// weights / gamma sit in LDS. (The first form -- one wave per row, lane = feature, K dependent broadcast loads and 12 ds_bpermute
// steps per row -- ran at 2.2 TB/s of its own traffic: 60 us for 231 k rows.)
// ---------------------------------------------------------------------------------------------------------
constexpr int kHbMaxK = 64;
__global__ void __launch_bounds__(256) head_bwd_dense_kernel(const float* __restrict__ dout, int ldk, int K, const float* __restrict__ Wq,
                                                              const float* __restrict__ gamma, const float* __restrict__ xhat,
                                                              const float* __restrict__ rstd, int rows, float* __restrict__ dh) {
  __shared__ __attribute__((aligned(16))) float s_w[(kHbMaxK + 1) * OPE_H];      // [K][64] head weights, then gamma
  for (int e = threadIdx.x; e < K * OPE_H; e += 256) s_w[e] = Wq[e];
  if (threadIdx.x < OPE_H) s_w[K * OPE_H + threadIdx.x] = gamma[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int64_t row = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + j;
  const bool live = row < rows;
  const int64_t rr = live ? row : 0;
  f32x4 xh[4], dy[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    xh[c] = *reinterpret_cast<const f32x4*>(xhat + rr * OPE_H + 16 * c + 4 * g);
    dy[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float rs = rstd[rr];
  const float* drow = dout + rr * ldk;
#pragma unroll 2
  for (int k = 0; k < K; ++k) {
    const float d = drow[k];
#pragma unroll
    for (int c = 0; c < 4; ++c) dy[c] += d * *reinterpret_cast<const f32x4*>(s_w + k * OPE_H + 16 * c + 4 * g);
  }
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dy[c] *= *reinterpret_cast<const f32x4*>(s_w + K * OPE_H + 16 * c + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) { m1 += dy[c][r]; m2 = fmaf(dy[c][r], xh[c][r], m2); }
  }
  m1 = rowsum4(m1) * (1.0f / OPE_H);
  m2 = rowsum4(m2) * (1.0f / OPE_H);
  if (live) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = rs * (dy[c][r] - m1 - xh[c][r] * m2);
      *reinterpret_cast<f32x4*>(dh + row * OPE_H + 16 * c + 4 * g) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// out = Linear_K(LayerNorm(h)) for every row (rnn.norm + the actor's action head / the critic's q heads, r_actor_critic.py:50-56,
// 108-118), optionally saving the normalised input and 1/std for the adjoint above. Same 16-rows-per-wave layout as
// head_bwd_dense_kernel: statistics = 16 local values + two lane swaps, every lane forms its 16-feature partial of each output,
// rowsum4 completes it, lane g stores outputs k = g (mod 4). (The thread-per-row head_fwd_kernel<1> it replaces here walked a
// 64-long dependent chain per thread: 24 us for 231 k rows x 18 outputs.)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_fwd_dense_kernel(const float* __restrict__ h, int K, const float* __restrict__ Wq,
                                                              const float* __restrict__ bq, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int64_t rows, float* __restrict__ out,
                                                              float* __restrict__ xhat_o, float* __restrict__ rstd_o) {
  __shared__ __attribute__((aligned(16))) float s_w[(kHbMaxK + 2) * OPE_H];      // [K][64] head weights, gamma, beta
  for (int e = threadIdx.x; e < K * OPE_H; e += 256) s_w[e] = Wq[e];
  if (threadIdx.x < OPE_H) { s_w[K * OPE_H + threadIdx.x] = gamma[threadIdx.x]; s_w[(K + 1) * OPE_H + threadIdx.x] = beta[threadIdx.x]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int64_t row = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + j;
  const bool live = row < rows;
  const int64_t rr = live ? row : 0;
  f32x4 x[4];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    x[c] = *reinterpret_cast<const f32x4*>(h + rr * OPE_H + 16 * c + 4 * g);
    s += (x[c][0] + x[c][1]) + (x[c][2] + x[c][3]);
  }
  const float mu = rowsum4(s) * (1.0f / OPE_H);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[c][r] -= mu; q = fmaf(x[c][r], x[c][r], q); }
  const float rs = 1.0f / sqrtf(rowsum4(q) * (1.0f / OPE_H) + OPE_LN_EPS);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    x[c] *= rs;
    if (xhat_o && live) *reinterpret_cast<f32x4*>(xhat_o + row * OPE_H + 16 * c + 4 * g) = x[c];
    const f32x4 ga = *reinterpret_cast<const f32x4*>(s_w + K * OPE_H + 16 * c + 4 * g);
    const f32x4 be = *reinterpret_cast<const f32x4*>(s_w + (K + 1) * OPE_H + 16 * c + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[c][r] = fmaf(x[c][r], ga[r], be[r]);
  }
  if (rstd_o && live && g == 0) rstd_o[row] = rs;
  for (int k = 0; k < K; ++k) {
    float p = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(s_w + k * OPE_H + 16 * c + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) p = fmaf(x[c][r], w[r], p);
    }
    p = rowsum4(p) + bq[k];
    if (live && (k & 3) == g) out[row * K + k] = p;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Critic TD on the [T][B] grid (r_maddpg.py:134-224):
//   keep = 1 - dones_env[t-1] (1 at t = 0) ; nextQ = (1 - dones_env[t]) min_k Q'_k ; target = r + gamma nextQ
//   err_k = (Q_k - target) keep ; loss_sum = sum_k w_b f(err_k) ; dQ_k = w_b f'(err_k) keep ; mask_count = sum keep
// ---------------------------------------------------------------------------------------------------------
struct RTdArgs {
  int T, B, N, K;
  float gamma; int use_huber; float huber_delta;
  const float* q; const float* nq;          // [T*B][K]
  const float* rewards;                     // [T][N][B][1]: agent 0's slice (r_maddpg.py:134)
  const float* dones_env;                   // [T][B][1]
  const float* per_weights;                 // [B] or null
  float* dq;                                // [T*B][K]
  float* err_abs;                           // [K][T*B]
  float* loss_part;                         // [tiles][4]
};

__global__ void __launch_bounds__(256) rcritic_td_kernel(RTdArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int TB = a.T * a.B;
  const bool valid = i < TB;
  const int ii = valid ? i : 0;
  const int t = ii / a.B, b = ii - t * a.B;
  const float keep = t == 0 ? 1.0f : 1.0f - a.dones_env[ii - a.B];
  float qn = a.nq[(int64_t)ii * a.K];
  for (int k = 1; k < a.K; ++k) qn = fminf(qn, a.nq[(int64_t)ii * a.K + k]);
  const float target = (a.rewards[((int64_t)t * a.N) * a.B + b] + a.gamma * ((1.0f - a.dones_env[ii]) * qn)) * keep;
  const float w = a.per_weights ? a.per_weights[b] : 1.0f;
  float ls = 0.f, qs = 0.f;
  for (int k = 0; k < a.K; ++k) {
    const float q = a.q[(int64_t)ii * a.K + k];
    const float e = q * keep - target;
    float fe, dfe;
    if (a.use_huber) {
      const float ae = fabsf(e), dl = a.huber_delta;
      if (ae <= dl) { fe = e * e * 0.5f; dfe = e; } else { fe = dl * (ae - dl * 0.5f); dfe = dl * (e > 0.f ? 1.f : -1.f); }
    } else {
      fe = e * e;
      dfe = 2.0f * e;
    }
    ls += w * fe;
    if (k == 0) qs = q;
    if (valid) {
      a.dq[(int64_t)i * a.K + k] = w * dfe * keep;
      a.err_abs[(int64_t)k * TB + i] = fabsf(e);
    }
  }
  float cs = keep;
  if (!valid) { ls = 0.f; qs = 0.f; cs = 0.f; }
  for (int o = 1; o < 16; o <<= 1) {
    ls += __shfl_xor(ls, o, 64);
    cs += __shfl_xor(cs, o, 64);
    qs += __shfl_xor(qs, o, 64);
  }
  if ((threadIdx.x & 15) == 0 && (i >> 4) < ((TB + 15) >> 4)) {
    float* lp = a.loss_part + (i >> 4) * 4;
    lp[0] = ls; lp[1] = cs; lp[2] = qs; lp[3] = 0.f;
  }
}

// Actor objective (r_maddpg.py:309-313): rows (t, agent i, b); keep = 1 - dones[t-1][i][b] (1 at t = 0);
//   loss_sum = -sum Q_0 keep ; mask_count = sum keep ; dQ_0 = -keep
__global__ void __launch_bounds__(256) ractor_obj_kernel(const float* __restrict__ q, int K, const float* __restrict__ dones, int rows,
                                                          int NB, const float* __restrict__ row_weight, float* __restrict__ dq,
                                                          float* __restrict__ loss_part) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = r < rows;
  const int rr = ok ? r : 0;
  float keep = !ok ? 0.f : (rr < NB ? 1.0f : 1.0f - dones[rr - NB]);   // dones [T][N][B][1] has the row order of q
  if (row_weight) keep *= row_weight[rr % NB];                          // ope_rddpg_cfg.actor_row_weight [N][B] (row within a step)
  const float q0 = ok ? q[(int64_t)rr * K] : 0.f;
  if (ok) {
    dq[(int64_t)r * K] = -keep;
    for (int k = 1; k < K; ++k) dq[(int64_t)r * K + k] = 0.f;
  }
  float ls = -q0 * keep, cs = keep, qs = q0 * keep;
  for (int o = 1; o < 16; o <<= 1) {
    ls += __shfl_xor(ls, o, 64);
    cs += __shfl_xor(cs, o, 64);
    qs += __shfl_xor(qs, o, 64);
  }
  if ((threadIdx.x & 15) == 0 && (r >> 4) < ((rows + 15) >> 4)) {
    float* lp = loss_part + (r >> 4) * 4;
    lp[0] = ls; lp[1] = cs; lp[2] = qs; lp[3] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
static int launch1d(int64_t n) { return ope_cdiv(n, 256); }
#define OPE_L(call)                                            \
  do {                                                         \
    call;                                                      \
    if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;   \
  } while (0)

// forward saves of one recurrent net over a row set
struct SaveSet {
  int64_t mu0, rstd0, xhat1, rstd1, mu1, mask1, xhat2, rstd2, mask2, gi, h, rg, zg, ng, ghn, xhat_o, rstd_o;
};
#define OPE_ADD_SET(W, s, pre, R)                                                                                          \
  do {                                                                                                                     \
    s.mu0 = W.add(pre "mu0", R); s.rstd0 = W.add(pre "rstd0", R); s.xhat1 = W.add(pre "xhat1", (R) * OPE_H);               \
    s.rstd1 = W.add(pre "rstd1", R); s.mask1 = W.add(pre "mask1", 2 * (R)); s.xhat2 = W.add(pre "xhat2", (R) * OPE_H);     \
    s.rstd2 = W.add(pre "rstd2", R); s.mask2 = W.add(pre "mask2", 2 * (R)); s.gi = W.add(pre "gi", (R) * 3 * OPE_H);       \
    s.h = W.add(pre "h", (R) * OPE_H); s.rg = W.add(pre "rg", (R) * OPE_H); s.zg = W.add(pre "zg", (R) * OPE_H);           \
    s.ng = W.add(pre "ng", (R) * OPE_H); s.ghn = W.add(pre "ghn", (R) * OPE_H); s.xhat_o = W.add(pre "xhat_o", (R) * OPE_H); \
    s.rstd_o = W.add(pre "rstd_o", R); s.mu1 = W.add(pre "mu1", R);                                                        \
  } while (0)

struct RPlan {
  int T, N, A, D, S, B, K, A4, Din, NB;
  int NT, a0;                   // agents in the joint action / first agent of this policy (multi-policy: N < NT)
  int J, c0;                    // width of the joint action / first column of this policy's first agent in it
  bool hetero;                  // joint action given as [T][B][J] (policies of different action dimensions: ope_rddpg_cfg.joint_act_dim)
  int64_t TB, Ra, Ra1;          // critic rows, actor rows (T steps), target-actor rows (T+1 steps)
  AgentLayout AL, CL;
  int raw_size, ns_c, ns_a;
  int P1, s1, P2, s2, P3, s3, WHH, shh, E, sq;
  Workspace ws;
  SaveSet SA, SC;               // actor saves (Ra rows) / critic saves (max(TB, Ra) rows)
  int64_t gi_t, h_t, lg_t, cnact, xin, xin_n, gi_n, h_n, nq, q, dq, err_abs, loss_part, lnz, lno, thetaT, raw, rsum,
      dh_out, dgi, dghn, dz1, dz2, lga, ysoft, actout, xin_a, h_b, cvec, dlg, c_gi, c_h, rep_u, rep_s12, rep_scratch;
  bool rep;
};

static int rddpg_cfg_ok(const ope_rddpg_cfg* c) {
  if (!c) return 0;
  const ope_dims& d = c->dims;
  if (d.n_agents < 1 || d.n_agents > 64 || d.act_dim < 1 || d.act_dim > 64 || d.obs_dim < 1 || d.obs_dim > 512 || d.state_dim < 1) return 0;
  if (d.layer_N > 1 || d.flags) return 0;      // the non-default network shapes (a second hidden block, no input LayerNorm) exist for the Q-learning nets only
  if (d.state_dim + d.n_agents * d.act_dim > 1024 || d.episode_length < 1) return 0;
  if (c->n_total_agents != 0 && (c->n_total_agents < d.n_agents || c->agent_offset < 0 || c->agent_offset + d.n_agents > c->n_total_agents ||
                                 d.state_dim + c->n_total_agents * d.act_dim > 1024 || c->n_total_agents > 64)) return 0;
  if (c->joint_act_dim != 0 && (c->n_total_agents != 0 || c->joint_act_col < 0 || c->joint_act_col + d.n_agents * d.act_dim > c->joint_act_dim ||
                                d.state_dim + c->joint_act_dim > 1024 || c->actor_row_weight))
    return 0;
  if (c->batch < 1 || c->num_q < 1 || c->num_q > 4) return 0;
  if ((int64_t)(d.episode_length + 1) * d.n_agents * c->batch > (int64_t)1 << 24) return 0;
  if ((c->continuous != 0 && c->continuous != 1) || (c->continuous && c->target_gumbel)) return 0;
  if (!act_heads_ok(c->n_act_heads, c->act_head_dims, d.act_dim) || (c->n_act_heads > 1 && c->continuous)) return 0;
  return 1;
}

static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void rddpg_plan(const ope_rddpg_cfg* c, RPlan* p) {
  const ope_dims& d = c->dims;
  p->T = d.episode_length; p->N = d.n_agents; p->A = d.act_dim; p->D = d.obs_dim; p->S = d.state_dim; p->B = c->batch; p->K = c->num_q;
  p->NT = c->n_total_agents > 0 ? c->n_total_agents : p->N; p->a0 = c->n_total_agents > 0 ? c->agent_offset : 0;
  p->hetero = c->joint_act_dim > 0;
  p->J = p->hetero ? c->joint_act_dim : p->NT * p->A; p->c0 = p->hetero ? c->joint_act_col : p->a0 * p->A;
  p->A4 = ope_round4(p->A); p->Din = p->S + p->J; p->NB = p->N * p->B;
  p->TB = (int64_t)p->T * p->B; p->Ra = (int64_t)p->T * p->NB; p->Ra1 = (int64_t)(p->T + 1) * p->NB;
  p->AL = ope_agent_layout(p->D, p->A, 0);
  p->CL = ope_agent_layout(p->Din, p->K, 0);
  const int Dmax = p->D > p->Din ? p->D : p->Din;
  const int Hmax = p->A > p->K ? p->A : p->K;
  int o = 0;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  p->P1 = take(OPE_H * Dmax); p->s1 = take(OPE_H); p->P2 = take(OPE_H * OPE_H); p->s2 = take(OPE_H);
  p->P3 = take(3 * OPE_H * OPE_H); p->s3 = take(3 * OPE_H); p->WHH = take(3 * OPE_H * OPE_H); p->shh = take(3 * OPE_H);
  p->E = take(Hmax * OPE_H); p->sq = take(ope_round4(Hmax));
  p->raw_size = o;
  p->ns_c = clampi(ope_cdiv(p->TB, 300), 1, 128);
  p->ns_a = clampi(ope_cdiv(p->Ra, 300), 1, 128);
  Workspace& W = p->ws;
  const int64_t TB = p->TB, Ra = p->Ra, Ra1 = p->Ra1;
  const int64_t Rc = TB > Ra ? TB : Ra;
  OPE_ADD_SET(W, p->SA, "a_", Ra);
  OPE_ADD_SET(W, p->SC, "c_", Rc);
  // critic update
  p->gi_t = W.add("gi_t", Ra1 * 3 * OPE_H);      // target actor (T+1 steps); reused for the target critic (TB rows)
  p->h_t = W.add("h_t", Ra1 * OPE_H);
  p->lg_t = W.add("logits_n", Ra1 * p->A);
  p->cnact = W.add("cent_nact", TB * p->J);
  p->xin = W.add("xin", TB * p->Din); p->xin_n = W.add("xin_n", TB * p->Din);
  p->c_gi = W.add("tc_gi", TB * 3 * OPE_H); p->c_h = W.add("tc_h", TB * OPE_H);
  p->gi_n = W.add("gi_n", TB * 3 * OPE_H); p->h_n = W.add("h_n", TB * OPE_H);
  p->nq = W.add("next_q", TB * p->K);
  p->q = W.add("q", Rc * p->K); p->dq = W.add("dq", Rc * p->K);
  p->err_abs = W.add("err_abs", TB * p->K);
  p->loss_part = W.add("loss_part", (int64_t)ope_cdiv(Rc, 16) * 4);
  p->lnz = W.add("ln_zero", Rc); p->lno = W.add("ln_one", Rc);
  p->thetaT = W.add("thetaT", OPE_H * 3 * OPE_H + OPE_H * OPE_H);
  const int nsmax = p->ns_c > p->ns_a ? p->ns_c : p->ns_a;
  p->raw = W.add("raw", (int64_t)nsmax * o); p->rsum = W.add("rsum", o + 4);
  p->dh_out = W.add("dh_out", Rc * OPE_H); p->dgi = W.add("dgi", Rc * 3 * OPE_H); p->dghn = W.add("dghn", Rc * OPE_H);
  p->dz1 = W.add("dz1", Rc * OPE_H); p->dz2 = W.add("dz2", Rc * OPE_H);
  // actor update
  p->lga = W.add("logits", Ra * p->A); p->ysoft = W.add("y_soft", Ra * p->A); p->actout = W.add("act_out", Ra * p->A);
  p->rep = !p->hetero && trunk_rep_ok(p->Din, p->NT, p->A, p->N);
  p->xin_a = W.add("xin_a", p->rep ? 4 : Ra * p->Din); p->h_b = W.add("h_branch", Ra * OPE_H);
  p->rep_u = W.add("rep_u", p->rep ? p->TB * OPE_H : 4); p->rep_s12 = W.add("rep_s12", p->rep ? 2 * p->TB : 4);
  p->rep_scratch = W.add("rep_scratch", p->rep ? (int64_t)OPE_H * p->NT * p->A + 2 * OPE_H : 4);
  p->cvec = W.add("fc1_colsums", 2 * OPE_H); p->dlg = W.add("dlogits", Ra * p->A4);
}

static void set_trunk_saves(TrunkFwdArgs& tf, float* W, const SaveSet& s) {
  tf.mu0 = W + s.mu0; tf.rstd0 = W + s.rstd0; tf.xhat1 = W + s.xhat1; tf.rstd1 = W + s.rstd1; tf.mask1 = (uint64_t*)(W + s.mask1);
  tf.mu1 = W + s.mu1;
  tf.xhat2 = W + s.xhat2; tf.rstd2 = W + s.rstd2; tf.mask2 = (uint64_t*)(W + s.mask2);
}

// trunk (feature LN -> fc1 -> fc2 -> W_ih projection) over `rows` independent rows
static int rtrunk(const float* x, int64_t rows, int Dw, const float* theta, const AgentLayout& L, float* gi, float* W, const SaveSet* save,
                  hipStream_t st) {
  TrunkFwdArgs tf;
  memset(&tf, 0, sizeof(tf));
  tf.x = x; tf.R = (int)rows; tf.D = Dw; tf.theta = theta; tf.L = L; tf.gi = gi;
  if (save) set_trunk_saves(tf, W, *save);
  return launch_trunk_fwd(tf, save != nullptr, st);
}

// GRU scan from a zero state over `steps` steps of `rps` rows
static int rscan(const float* gi, int rps, int steps, const float* theta, const AgentLayout& L, float* h, float* W, const SaveSet* save,
                 hipStream_t st) {
  GruFwdArgs gf;
  memset(&gf, 0, sizeof(gf));
  gf.nets = 1; gf.NB = rps; gf.L = steps; gf.theta0 = theta; gf.theta1 = theta; gf.gi0 = gi; gf.gi1 = gi; gf.h0out = h; gf.h1out = h;
  gf.whh_off = L.whh; gf.bhh_off = L.bhh;
  if (save) { gf.rg = W + save->rg; gf.zg = W + save->zg; gf.ng = W + save->ng; gf.ghn = W + save->ghn; }
  return launch_gru_fwd(gf, st);
}

// LayerNorm + Linear head on every row: out [rows][Hout]
static int rhead(const float* h, int64_t rows, int Hout, const float* theta, const AgentLayout& L, float* out, float* W, const SaveSet* save,
                 hipStream_t st) {
  if (Hout > kHbMaxK) return OPE_EINVAL;
  kprof_work(2.0 * rows * (double)OPE_H * Hout);
  OPE_L(OPE_LAUNCH(head_fwd_dense_kernel, dim3(ope_cdiv(rows, 64)), dim3(256), 0, st, h, Hout, theta + L.q_w, theta + L.q_b,
                           theta + L.lno_w, theta + L.lno_b, rows, out, save ? W + save->xhat_o : nullptr, save ? W + save->rstd_o : nullptr));
  return OPE_OK;
}

static int rcell(const RPlan& p, const float* gi, const float* hprev, int64_t rows, int reps, int prev_shift, const float* theta,
                 const AgentLayout& L, float* hout, float* W, const SaveSet* save, hipStream_t st) {
  CellFwdArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.R = (int)rows; ca.B = p.B; ca.reps = reps; ca.prev_shift = prev_shift; ca.gi = gi; ca.hprev = hprev; ca.theta = theta;
  ca.whh_off = L.whh; ca.bhh_off = L.bhh; ca.hout = hout;
  if (save) { ca.rg = W + save->rg; ca.zg = W + save->zg; ca.ng = W + save->ng; ca.ghn = W + save->ghn; }
  kprof_work(2.0 * rows * 3.0 * OPE_H * OPE_H);
  OPE_L(OPE_LAUNCH(gru_cell_fwd_kernel, dim3(ope_cdiv(rows, 64)), dim3(256), 0, st, ca));
  return OPE_OK;
}

static int rtranspose(const RPlan& p, const float* theta, const AgentLayout& L, float* W, hipStream_t st) {
  Transp4 tr;
  memset(&tr, 0, sizeof(tr));
  tr.src[0] = theta + L.wih; tr.dst[0] = W + p.thetaT; tr.rows[0] = 3 * OPE_H; tr.cols[0] = OPE_H; tr.begin[0] = 0;
  tr.src[1] = theta + L.fc2_w; tr.dst[1] = W + p.thetaT + OPE_H * 3 * OPE_H; tr.rows[1] = OPE_H; tr.cols[1] = OPE_H;
  tr.begin[1] = 3 * OPE_H * OPE_H;
  tr.n = 2; tr.total = 4 * OPE_H * OPE_H;
  return launch_transpose4(tr, st);
}

// Backward of one recurrent net (trunk -> GRU scan -> LN -> dense head) over `steps` steps of `rps` rows, given
// d(head output) [rows][ldk]; writes the flat gradient (+tail from the plan's loss partials).
static int rnn_backward(const RPlan& p, float* W, const SaveSet& s, const float* x, int rps, int steps, int Dw, int Hout, int ldk,
                        const float* dout, const float* theta, const AgentLayout& L, int nsplit, int n_loss_tiles, float* grad,
                        hipStream_t st) {
  int rc;
  const int64_t rows = (int64_t)rps * steps;
  if (Hout > kHbMaxK) return OPE_EINVAL;
  kprof_work(2.0 * rows * (double)OPE_H * Hout);
  OPE_L(OPE_LAUNCH(head_bwd_dense_kernel, dim3(ope_cdiv(rows, 64)), dim3(256), 0, st, dout, ldk, Hout, theta + L.q_w,
                           theta + L.lno_w, W + s.xhat_o, W + s.rstd_o, (int)rows, W + p.dh_out));
  GruBwdArgs gb;
  memset(&gb, 0, sizeof(gb));
  gb.NB = rps; gb.T = steps; gb.theta = theta; gb.whh_off = L.whh; gb.h = W + s.h;
  gb.rg = W + s.rg; gb.zg = W + s.zg; gb.ng = W + s.ng; gb.ghn = W + s.ghn; gb.dh_out = W + p.dh_out; gb.dgi = W + p.dgi; gb.dghn = W + p.dghn;
  if ((rc = launch_gru_bwd(gb, st))) return rc;
  if ((rc = rtranspose(p, theta, L, W, st))) return rc;
  TrunkBwdArgs tb;
  memset(&tb, 0, sizeof(tb));
  tb.R = (int)rows; tb.theta = theta; tb.thetaT = W + p.thetaT; tb.L = L; tb.dgi = W + p.dgi;
  tb.xhat1 = W + s.xhat1; tb.rstd1 = W + s.rstd1; tb.mask1 = (const uint64_t*)(W + s.mask1);
  tb.xhat2 = W + s.xhat2; tb.rstd2 = W + s.rstd2; tb.mask2 = (const uint64_t*)(W + s.mask2);
  tb.dz1 = W + p.dz1; tb.dz2 = W + p.dz2;
  if ((rc = launch_trunk_bwd(tb, st))) return rc;

  WgTable wt;
  memset(&wt, 0, sizeof(wt));
  int n = 0;
  const int K1 = (int)rows;
  auto prob = [&](const float* A_, int lda, int M, const float* B_, int ldb, int N_, int out_off, int ldc, int s_off) -> WgProb& {
    WgProb& q = wt.p[n++];
    q.A = A_; q.lda = lda; q.M = M; q.B = B_; q.ldb = ldb; q.N = N_; q.K = K1; q.b_shift = 0; q.ln_mu = W + p.lnz; q.ln_rstd = W + p.lno;
    q.out_off = out_off; q.ldc = ldc; q.s_off = s_off; q.nsplit = nsplit; q.raw_base = p.raw; q.raw_stride = p.raw_size;
    return q;
  };
  {
    WgProb& q = prob(W + p.dz1, OPE_H, OPE_H, x, Dw, Dw, p.P1, Dw, p.s1);
    q.ln_mu = W + s.mu0; q.ln_rstd = W + s.rstd0; q.ln_on = 1;
  }
  prob(W + p.dz2, OPE_H, OPE_H, W + s.xhat1, OPE_H, OPE_H, p.P2, OPE_H, p.s2);
  prob(W + p.dgi, 3 * OPE_H, 3 * OPE_H, W + s.xhat2, OPE_H, OPE_H, p.P3, OPE_H, p.s3);
  prob(W + p.dgi, 3 * OPE_H, 2 * OPE_H, W + s.h, OPE_H, OPE_H, p.WHH, OPE_H, p.shh).b_shift = rps;   // h_{t-1}
  prob(W + p.dghn, OPE_H, OPE_H, W + s.h, OPE_H, OPE_H, p.WHH + 2 * OPE_H * OPE_H, OPE_H, p.shh + 2 * OPE_H).b_shift = rps;
  prob(dout, ldk, Hout, W + s.xhat_o, OPE_H, OPE_H, p.E, OPE_H, p.sq);
  wt.n = n;
  if ((rc = wg_finish(&wt))) return rc;
  if ((rc = launch_wgrad(wt, W, st))) return rc;
  SplitRed sr;
  sr.raw0 = W + p.raw; sr.n0 = p.raw_size; sr.ns0 = wg_slabs(wt, nsplit); sr.raw1 = W + p.raw; sr.n1 = 0; sr.ns1 = 0; sr.rsum = W + p.rsum;
  if ((rc = launch_split_reduce(sr, st))) return rc;

  FinTable ft;
  memset(&ft, 0, sizeof(ft));
  int k = 0;
  auto seg = [&](int begin, int size, int kind, int src, int src_s, int M, int K_, int w, int gamma, int beta) {
    FinSeg& g = ft.seg[k++];
    g.begin = begin; g.size = size; g.kind = kind; g.src = src; g.src_s = src_s; g.M = M; g.K = K_; g.w = w; g.gamma = gamma; g.beta = beta;
  };
  seg(L.fn_w, Dw, FIN_LNLIN_G, p.P1, p.s1, OPE_H, Dw, L.fc1_w, 0, 0);
  seg(L.fn_b, Dw, FIN_LNLIN_B, p.P1, p.s1, OPE_H, Dw, L.fc1_w, 0, 0);
  seg(L.fc1_w, OPE_H * Dw, FIN_LNLIN_W, p.P1, p.s1, OPE_H, Dw, L.fc1_w, L.fn_w, L.fn_b);
  seg(L.fc1_b, OPE_H, FIN_COPY, p.s1, 0, 0, 0, 0, 0, 0);
  seg(L.ln1_w, OPE_H, FIN_LNLIN_G, p.P2, p.s2, OPE_H, OPE_H, L.fc2_w, 0, 0);
  seg(L.ln1_b, OPE_H, FIN_LNLIN_B, p.P2, p.s2, OPE_H, OPE_H, L.fc2_w, 0, 0);
  seg(L.fch_w, 0, FIN_ZERO, 0, 0, 0, 0, 0, 0, 0);   // fc_h.*: registered, never used (mlp.py:21-23)
  seg(L.fc2_w, OPE_H * OPE_H, FIN_LNLIN_W, p.P2, p.s2, OPE_H, OPE_H, L.fc2_w, L.ln1_w, L.ln1_b);
  seg(L.fc2_b, OPE_H, FIN_COPY, p.s2, 0, 0, 0, 0, 0, 0);
  seg(L.ln2_w, OPE_H, FIN_LNLIN_G, p.P3, p.s3, 3 * OPE_H, OPE_H, L.wih, 0, 0);
  seg(L.ln2_b, OPE_H, FIN_LNLIN_B, p.P3, p.s3, 3 * OPE_H, OPE_H, L.wih, 0, 0);
  seg(L.wih, 3 * OPE_H * OPE_H, FIN_LNLIN_W, p.P3, p.s3, 3 * OPE_H, OPE_H, L.wih, L.ln2_w, L.ln2_b);
  seg(L.whh, 3 * OPE_H * OPE_H, FIN_COPY, p.WHH, 0, 0, 0, 0, 0, 0);
  seg(L.bih, 3 * OPE_H, FIN_COPY, p.s3, 0, 0, 0, 0, 0, 0);
  seg(L.bhh, 3 * OPE_H, FIN_COPY, p.shh, 0, 0, 0, 0, 0, 0);
  seg(L.lno_w, OPE_H, FIN_LNLIN_G, p.E, p.sq, Hout, OPE_H, L.q_w, 0, 0);
  seg(L.lno_b, OPE_H, FIN_LNLIN_B, p.E, p.sq, Hout, OPE_H, L.q_w, 0, 0);
  seg(L.q_w, Hout * OPE_H, FIN_LNLIN_W, p.E, p.sq, Hout, OPE_H, L.q_w, L.lno_w, L.lno_b);
  seg(L.q_b, Hout, FIN_COPY, p.sq, 0, 0, 0, 0, 0, 0);
  seg(L.end, OPE_GRAD_TAIL, FIN_TAIL, 0, 0, 0, 0, 0, 0, 0);
  ft.n = k;
  ft.total = L.end + OPE_GRAD_TAIL;
  return launch_finalize(ft, W + p.rsum, theta, W + p.loss_part, n_loss_tiles, grad, st);
}

}  // namespace ope

using namespace ope;

extern "C" int64_t ope_rddpg_param_layout(const ope_rddpg_cfg* cfg, int32_t which, int64_t* offsets, int64_t* sizes) {
  if (!rddpg_cfg_ok(cfg) || which < 0 || which > 1) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  const AgentLayout& L = which == 0 ? p.AL : p.CL;
  const int Dw = which == 0 ? p.D : p.Din, Ho = which == 0 ? p.A : p.K;
  const int off[OPE_QMIX_NPARAM_AGENT] = {L.fn_w, L.fn_b, L.fc1_w, L.fc1_b, L.ln1_w, L.ln1_b, L.fch_w, L.fch_b, L.lnh_w, L.lnh_b, L.fc2_w,
                                          L.fc2_b, L.ln2_w, L.ln2_b, L.wih, L.whh, L.bih, L.bhh, L.lno_w, L.lno_b, L.q_w, L.q_b};
  const int siz[OPE_QMIX_NPARAM_AGENT] = {Dw, Dw, OPE_H * Dw, OPE_H, OPE_H, OPE_H, OPE_H * OPE_H, OPE_H, OPE_H, OPE_H, OPE_H * OPE_H,
                                          OPE_H, OPE_H, OPE_H, 3 * OPE_H * OPE_H, 3 * OPE_H * OPE_H, 3 * OPE_H, 3 * OPE_H, OPE_H, OPE_H,
                                          Ho * OPE_H, Ho};
  for (int i = 0; i < OPE_QMIX_NPARAM_AGENT; ++i) {
    if (offsets) offsets[i] = off[i];
    if (sizes) sizes[i] = siz[i];
  }
  return L.end;
}

extern "C" int64_t ope_rddpg_workspace_bytes(const ope_rddpg_cfg* cfg) {
  if (!rddpg_cfg_ok(cfg)) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  return p.ws.total * (int64_t)sizeof(float);
}

extern "C" int64_t ope_rddpg_workspace_find(const ope_rddpg_cfg* cfg, const char* name, int64_t* n_floats) {
  if (!rddpg_cfg_ok(cfg) || !name) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  const int64_t off = p.ws.find(name, n_floats);
  return off < 0 ? -1 : off * (int64_t)sizeof(float);
}

extern "C" int ope_rddpg_workspace_init(const ope_rddpg_cfg* cfg, void* workspace, int64_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!rddpg_cfg_ok(cfg) || !workspace) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  float* W = (float*)workspace;
  const int64_t Rc = p.TB > p.Ra ? p.TB : p.Ra;
  int rc;
  if ((rc = launch_fill(W + p.lnz, Rc, 0.f, (hipStream_t)stream))) return rc;
  return launch_fill(W + p.lno, Rc, 1.f, (hipStream_t)stream);
}

extern "C" int ope_rddpg_critic_loss_and_grad(const ope_rddpg_cfg* cfg, const ope_fields* bt, const float* theta_actor_tgt,
                                              const float* theta_critic, const float* theta_critic_tgt, const float* target_noise_u,
                                              const float* per_weights, void* workspace, int64_t workspace_bytes, float* grad,
                                              float* td_abs_stats, void* stream) {
  (void)hipGetLastError();
  if (!rddpg_cfg_ok(cfg) || !bt || (!theta_actor_tgt && !cfg->joint_next_acts) || !theta_critic || !theta_critic_tgt || !workspace || !grad)
    return OPE_EINVAL;
  if ((!bt->obs && !cfg->joint_next_acts) || !bt->share_obs || (!bt->acts && !cfg->joint_acts) || !bt->rewards || !bt->dones_env) return OPE_EINVAL;
  if (cfg->joint_act_dim > 0 && (!cfg->joint_next_acts || !cfg->joint_acts)) return OPE_EINVAL;
  if (cfg->target_gumbel && !target_noise_u && !cfg->joint_next_acts) return OPE_EINVAL;
  if (cfg->use_per && !per_weights) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;
  // target actor over all T+1 observations from a zero state; drop the first action (r_maddpg.py:79-96). Multi-policy updates
  // bring the joint target action of ALL policies' target actors (ope_rddpg_target_actions per policy) instead.
  const float* nact = cfg->joint_next_acts;
  const ActHeads hd = act_heads_of(cfg->n_act_heads, cfg->act_head_dims, p.A);
  if (!nact) {
    if (p.NT != p.N) return OPE_EINVAL;       // a policy's own actor cannot produce the other policies' target actions
    if ((rc = rtrunk(bt->obs, p.Ra1, p.D, theta_actor_tgt, p.AL, W + p.gi_t, W, nullptr, st))) return rc;
    if ((rc = rscan(W + p.gi_t, p.NB, p.T + 1, theta_actor_tgt, p.AL, W + p.h_t, W, nullptr, st))) return rc;
    if ((rc = rhead(W + p.h_t, p.Ra1, p.A, theta_actor_tgt, p.AL, W + p.lg_t, W, nullptr, st))) return rc;
    if ((rc = launch_action(W + p.lg_t, bt->avail_acts, NoiseSrc{target_noise_u, 0, nullptr, 0}, (int)p.Ra1, p.B, p.A, p.N, cfg->continuous ? 2 : (cfg->target_gumbel ? 1 : 0), 1,
                            W + p.cnact, nullptr, nullptr, st, 0, 0, &hd))) return rc;
    nact = W + p.cnact;
  }
  // critic inputs: buffer sequence [cent_obs[t] | acts[t]] and branch rows [cent_obs[t+1] | target actions]
  if ((rc = launch_build_cin(bt->share_obs, p.hetero ? cfg->joint_acts : bt->acts, nullptr, p.T, p.B, p.hetero ? 1 : p.NT, p.hetero ? p.J : p.A, p.S, 1,
                             W + p.xin, st)))
    return rc;
  if ((rc = launch_build_cin(bt->share_obs + (int64_t)p.B * p.S, nact, nullptr, p.T, p.B, 1, p.J, p.S, 1, W + p.xin_n, st)))
    return rc;
  // live critic over the buffer sequence (saved for backward)
  if ((rc = rtrunk(W + p.xin, p.TB, p.Din, theta_critic, p.CL, W + p.SC.gi, W, &p.SC, st))) return rc;
  if ((rc = rscan(W + p.SC.gi, p.B, p.T, theta_critic, p.CL, W + p.SC.h, W, &p.SC, st))) return rc;
  if ((rc = rhead(W + p.SC.h, p.TB, p.K, theta_critic, p.CL, W + p.q, W, &p.SC, st))) return rc;
  // target critic: buffer-sequence states, then one sideways cell step per (t, b) on the target actions
  if ((rc = rtrunk(W + p.xin, p.TB, p.Din, theta_critic_tgt, p.CL, W + p.c_gi, W, nullptr, st))) return rc;
  if ((rc = rscan(W + p.c_gi, p.B, p.T, theta_critic_tgt, p.CL, W + p.c_h, W, nullptr, st))) return rc;
  if ((rc = rtrunk(W + p.xin_n, p.TB, p.Din, theta_critic_tgt, p.CL, W + p.gi_n, W, nullptr, st))) return rc;
  if ((rc = rcell(p, W + p.gi_n, W + p.c_h, p.TB, 1, 0, theta_critic_tgt, p.CL, W + p.h_n, W, nullptr, st))) return rc;
  if ((rc = rhead(W + p.h_n, p.TB, p.K, theta_critic_tgt, p.CL, W + p.nq, W, nullptr, st))) return rc;
  RTdArgs td;
  td.T = p.T; td.B = p.B; td.N = p.N; td.K = p.K; td.gamma = cfg->gamma; td.use_huber = cfg->use_huber; td.huber_delta = cfg->huber_delta;
  td.q = W + p.q; td.nq = W + p.nq; td.rewards = bt->rewards; td.dones_env = bt->dones_env;
  td.per_weights = cfg->use_per ? per_weights : nullptr; td.dq = W + p.dq; td.err_abs = W + p.err_abs; td.loss_part = W + p.loss_part;
  OPE_L(OPE_LAUNCH(rcritic_td_kernel, dim3(launch1d(p.TB)), dim3(256), 0, st, td));
  if (td_abs_stats)
    for (int k = 0; k < p.K; ++k)
      if ((rc = launch_td_stats(W + p.err_abs + (int64_t)k * p.TB, p.T, p.B, td_abs_stats + (int64_t)k * 2 * p.B, st))) return rc;
  return rnn_backward(p, W, p.SC, W + p.xin, p.B, p.T, p.Din, p.K, p.K, W + p.dq, theta_critic, p.CL, p.ns_c, ope_cdiv(p.TB, 16), grad, st);
}

extern "C" int ope_rddpg_target_actions(const ope_rddpg_cfg* cfg, const ope_fields* bt, const float* theta_actor_tgt,
                                        const float* target_noise_u, void* workspace, int64_t workspace_bytes, float* joint_next_acts,
                                        void* stream) {
  (void)hipGetLastError();
  if (!rddpg_cfg_ok(cfg) || !bt || !bt->obs || !theta_actor_tgt || !workspace || !joint_next_acts) return OPE_EINVAL;
  if (cfg->target_gumbel && !target_noise_u) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;
  const ActHeads hd = act_heads_of(cfg->n_act_heads, cfg->act_head_dims, p.A);
  if ((rc = rtrunk(bt->obs, p.Ra1, p.D, theta_actor_tgt, p.AL, W + p.gi_t, W, nullptr, st))) return rc;
  if ((rc = rscan(W + p.gi_t, p.NB, p.T + 1, theta_actor_tgt, p.AL, W + p.h_t, W, nullptr, st))) return rc;
  if ((rc = rhead(W + p.h_t, p.Ra1, p.A, theta_actor_tgt, p.AL, W + p.lg_t, W, nullptr, st))) return rc;
  return launch_action(W + p.lg_t, bt->avail_acts, NoiseSrc{target_noise_u, 0, nullptr, 0}, (int)p.Ra1, p.B, p.A, p.N, cfg->continuous ? 2 : (cfg->target_gumbel ? 1 : 0), 1,
                       joint_next_acts, nullptr, nullptr, st, p.NT, p.a0, &hd, p.hetero ? p.J : 0, p.c0);
}

extern "C" int ope_rddpg_actor_loss_and_grad(const ope_rddpg_cfg* cfg, const ope_fields* bt, const float* theta_actor,
                                             const float* theta_critic, const float* gumbel_noise_u, void* workspace,
                                             int64_t workspace_bytes, float* grad, void* stream) {
  (void)hipGetLastError();
  if (!rddpg_cfg_ok(cfg) || !bt || !theta_actor || !theta_critic || (!cfg->continuous && !gumbel_noise_u) || !workspace || !grad) return OPE_EINVAL;
  if (!bt->obs || !bt->share_obs || (!bt->acts && !cfg->joint_acts) || !bt->dones) return OPE_EINVAL;
  if (cfg->joint_act_dim > 0 && !cfg->joint_acts) return OPE_EINVAL;
  RPlan p;
  rddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;
  const int Ra = (int)p.Ra;
  const ActHeads hd = act_heads_of(cfg->n_act_heads, cfg->act_head_dims, p.A);
  // actor over obs[:-1] from a zero state, straight-through hard gumbel-softmax sample (r_maddpg.py:277-280)
  if ((rc = rtrunk(bt->obs, p.Ra, p.D, theta_actor, p.AL, W + p.SA.gi, W, &p.SA, st))) return rc;
  if ((rc = rscan(W + p.SA.gi, p.NB, p.T, theta_actor, p.AL, W + p.SA.h, W, &p.SA, st))) return rc;
  if ((rc = rhead(W + p.SA.h, p.Ra, p.A, theta_actor, p.AL, W + p.lga, W, &p.SA, st))) return rc;
  if ((rc = launch_action(W + p.lga, bt->avail_acts, NoiseSrc{cfg->continuous ? nullptr : gumbel_noise_u, 0, nullptr, 0}, Ra, p.B, p.A, p.N, cfg->continuous ? 2 : 1, 0, nullptr,
                          W + p.actout, W + p.ysoft, st, 0, 0, &hd)))
    return rc;
  // critic state along the buffer sequence (identical for the N stacked copies)
  if ((rc = launch_build_cin(bt->share_obs, p.hetero ? cfg->joint_acts : bt->acts, nullptr, p.T, p.B, p.hetero ? 1 : p.NT, p.hetero ? p.J : p.A, p.S, 1,
                             W + p.xin, st)))
    return rc;
  if ((rc = rtrunk(W + p.xin, p.TB, p.Din, theta_critic, p.CL, W + p.c_gi, W, nullptr, st))) return rc;
  if ((rc = rscan(W + p.c_gi, p.B, p.T, theta_critic, p.CL, W + p.c_h, W, nullptr, st))) return rc;
  // sideways cell step on the spliced actions for all (t, agent copy, b) rows at once
  if (p.rep) {
    // the N copies of a base row differ in one action block: first layer once per base row + a per-copy correction, the
    // [T*N*B][Din] input is never built (RepIn, ope_agent.h)
    TrunkFwdArgs tb0, tr;
    memset(&tb0, 0, sizeof(tb0));
    tb0.x = W + p.xin; tb0.R = (int)p.TB; tb0.D = p.Din; tb0.theta = theta_critic; tb0.L = p.CL;
    tb0.rep.u_out = W + p.rep_u; tb0.rep.s12_out = W + p.rep_s12;
    memset(&tr, 0, sizeof(tr));
    tr.R = Ra; tr.D = p.Din; tr.theta = theta_critic; tr.L = p.CL; tr.gi = W + p.SC.gi;
    set_trunk_saves(tr, W, p.SC);
    tr.rep.u = W + p.rep_u; tr.rep.s12 = W + p.rep_s12; tr.rep.acts = bt->acts; tr.rep.repl = W + p.actout;
    tr.rep.T = p.T; tr.rep.B = p.B; tr.rep.N = p.N; tr.rep.A = p.A; tr.rep.S = p.S; tr.rep.NT = p.NT; tr.rep.a0 = p.a0;
    if ((rc = launch_trunk_fwd_rep(tb0, tr, W + p.rep_scratch, st))) return rc;
  } else {
    if (p.hetero) {
      if ((rc = launch_build_cin_joint(bt->share_obs, cfg->joint_acts, W + p.actout, p.T, p.B, p.J, p.S, p.N, p.A, p.c0, W + p.xin_a, st))) return rc;
    } else if ((rc = launch_build_cin(bt->share_obs, bt->acts, W + p.actout, p.T, p.B, p.NT, p.A, p.S, p.N, W + p.xin_a, st, p.a0))) {
      return rc;
    }
    if ((rc = rtrunk(W + p.xin_a, p.Ra, p.Din, theta_critic, p.CL, W + p.SC.gi, W, &p.SC, st))) return rc;
  }
  if ((rc = rcell(p, W + p.SC.gi, W + p.c_h, p.Ra, p.N, -1, theta_critic, p.CL, W + p.h_b, W, &p.SC, st))) return rc;
  if ((rc = rhead(W + p.h_b, p.Ra, p.K, theta_critic, p.CL, W + p.q, W, &p.SC, st))) return rc;
  OPE_L(OPE_LAUNCH(ractor_obj_kernel, dim3(launch1d(Ra)), dim3(256), 0, st, W + p.q, p.K, bt->dones, Ra, p.NB, cfg->actor_row_weight,
                           W + p.dq, W + p.loss_part));
  // critic adjoint down to its input (parameters frozen), through the gumbel-softmax into the actor logits
  kprof_work(2.0 * Ra * (double)OPE_H * p.K);
  OPE_L(OPE_LAUNCH(head_bwd_dense_kernel, dim3(ope_cdiv(Ra, 64)), dim3(256), 0, st, W + p.dq, p.K, p.K, theta_critic + p.CL.q_w,
                           theta_critic + p.CL.lno_w, W + p.SC.xhat_o, W + p.SC.rstd_o, Ra, W + p.dh_out));
  CellBwdArgs cb;
  cb.R = Ra; cb.B = p.B; cb.reps = p.N; cb.prev_shift = -1; cb.dh = W + p.dh_out; cb.hprev = W + p.c_h;
  cb.rg = W + p.SC.rg; cb.zg = W + p.SC.zg; cb.ng = W + p.SC.ng; cb.ghn = W + p.SC.ghn; cb.dgi = W + p.dgi;
  kprof_work(2.0 * Ra * 3.0 * OPE_H * OPE_H);
  OPE_L(OPE_LAUNCH(gru_cell_bwd_kernel, dim3(launch1d((int64_t)Ra * 16)), dim3(256), 0, st, cb));
  if ((rc = rtranspose(p, theta_critic, p.CL, W, st))) return rc;
  TrunkBwdArgs tb;
  memset(&tb, 0, sizeof(tb));
  tb.R = Ra; tb.theta = theta_critic; tb.thetaT = W + p.thetaT; tb.L = p.CL; tb.dgi = W + p.dgi;
  tb.xhat1 = W + p.SC.xhat1; tb.rstd1 = W + p.SC.rstd1; tb.mask1 = (const uint64_t*)(W + p.SC.mask1);
  tb.xhat2 = W + p.SC.xhat2; tb.rstd2 = W + p.SC.rstd2; tb.mask2 = (const uint64_t*)(W + p.SC.mask2);
  tb.dz1 = W + p.dz1; tb.dz2 = W + p.dz2;
  if ((rc = launch_trunk_bwd(tb, st))) return rc;
  ActGradArgs ag;
  ag.R = Ra; ag.B = p.B; ag.N = p.N; ag.A = p.A; ag.A4 = p.A4; ag.S = p.S; ag.Din = p.Din; ag.a_off = p.hetero ? 0 : p.a0; ag.a_col = p.hetero ? p.c0 : 0; ag.dz1 = W + p.dz1;
  ag.xhat1 = W + p.SC.xhat1; ag.rstd1 = W + p.SC.rstd1; ag.mu1 = W + p.SC.mu1; ag.mu0 = W + p.SC.mu0; ag.rstd0 = W + p.SC.rstd0;
  ag.act = W + p.actout; ag.y = W + p.ysoft; ag.theta = theta_critic; ag.fc1_w = p.CL.fc1_w; ag.fc1_b = p.CL.fc1_b; ag.fn_w = p.CL.fn_w;
  ag.fn_b = p.CL.fn_b; ag.cvec = W + p.cvec; ag.dlogits = W + p.dlg; ag.identity = cfg->continuous ? 1 : 0; ag.heads = hd;
  if ((rc = launch_action_grad(ag, st))) return rc;
  // actor BPTT and gradients
  return rnn_backward(p, W, p.SA, bt->obs, p.NB, p.T, p.D, p.A, p.A4, W + p.dlg, theta_actor, p.AL, p.ns_a, ope_cdiv(Ra, 16), grad, st);
}
