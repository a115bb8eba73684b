// MLP MADDPG / MATD3 update (centralised critic over [cent_obs || all agents' actions], gumbel-softmax actor).
//   MADDPG.get_update_info / shared_train_policy_on_batch   offpolicy/algorithms/maddpg/maddpg.py:38-81, 90-249
//   MADDPG_Actor / MADDPG_Critic                            offpolicy/algorithms/maddpg/algorithm/actor_critic.py:7-87
//   MADDPGPolicy.get_actions (target / gumbel paths)         offpolicy/algorithms/maddpg/algorithm/MADDPGPolicy.py:63-119
//   onehot_from_logits / gumbel_softmax                     offpolicy/utils/util.py:156-214
// Both networks are MLPBase trunks + a Linear head, so the heavy lifting is the shared trunk_fwd / trunk_bwd / wgrad /
// finalize kernels in "mlp" mode; this file adds the small row-parallel pieces around them and the two C-ABI steps.
#include <stdlib.h>
#include <string.h>

#include "ope_ddpg.h"

namespace ope {

// ---------------------------------------------------------------------------------------------------------
// rows of the critic input:  out[r] = [ cent[t][b] (S) | joint action (N*A) ],  r = (t*reps + rep)*B + b
//   joint action block a = acts[t][a][b]  unless  (repl != null and a == rep): then repl[r]   (reps == N in that case)
// (maddpg.py:128 for the critic update, 207-227 for the actor update: "mask * actor + (1 - mask) * buffer";
//  r_maddpg.py:162, 291-301 for the sequence form; the MLP family is T = 1)
// ---------------------------------------------------------------------------------------------------------
// A workgroup builds 16 consecutive output rows. Their (t, rep, b) decode -- integer divisions -- is done once per row by 16
// threads and shared through LDS; the copy itself runs over (row, column pair) with float reciprocals on small integers and, when
// every width is even (rows then start on 8-byte boundaries), 8-byte accesses; consecutive threads write consecutive addresses.
// (The first form was one wave per row with 4-byte accesses and an integer division per action element: 2 TB/s.)
template <int VEC>
__global__ void __launch_bounds__(256) build_cin_kernel(const float* __restrict__ cent, const float* __restrict__ acts,
                                                         const float* __restrict__ repl, int T, int B, int N, int A, int S, int reps,
                                                         int rep_off, float* __restrict__ out) {
  constexpr int kRows = 16;
  __shared__ int64_t s_cent[kRows], s_act[kRows], s_repl[kRows];
  __shared__ int s_rep[kRows];
  const int Din = S + N * A;
  const int64_t rows = (int64_t)T * reps * B, r0 = (int64_t)blockIdx.x * kRows;
  if (threadIdx.x < kRows) {
    const int64_t r = r0 + threadIdx.x < rows ? r0 + threadIdx.x : rows - 1;
    const int t = (int)(r / (reps * B));
    const int rem = (int)(r - (int64_t)t * (reps * B));
    const int rep = rem / B, b = rem - rep * B;
    s_cent[threadIdx.x] = ((int64_t)t * B + b) * S;
    s_act[threadIdx.x] = ((int64_t)t * N * B + b) * A;     // + a * B * A + j
    s_repl[threadIdx.x] = r * A;
    s_rep[threadIdx.x] = repl ? rep_off + rep : -1;
  }
  __syncthreads();
  const int W = Din / VEC;                    // column groups per row
  const float invW = 1.0f / (float)W, invA = 1.0f / (float)A;
  const int nrow = (int)(rows - r0 < kRows ? rows - r0 : kRows);
  for (int e = threadIdx.x; e < nrow * W; e += 256) {
    const int lr = (int)(((float)e + 0.5f) * invW), c = (e - lr * W) * VEC;
    float v[VEC];
    if (c < S) {                              // S % VEC == 0: a group never straddles the two parts
      const float* p = cent + s_cent[lr] + c;
      if (VEC == 2) { const f32x2 t2 = *reinterpret_cast<const f32x2*>(p); v[0] = t2[0]; v[VEC - 1] = t2[1]; } else { v[0] = p[0]; }
    } else {
      const int k = c - S, a = (int)(((float)k + 0.5f) * invA), j = k - a * A;      // A % VEC == 0: a group stays inside one agent's block
      const float* p = (a == s_rep[lr]) ? repl + s_repl[lr] + j : acts + s_act[lr] + (int64_t)a * B * A + j;
      if (VEC == 2) { const f32x2 t2 = *reinterpret_cast<const f32x2*>(p); v[0] = t2[0]; v[VEC - 1] = t2[1]; } else { v[0] = p[0]; }
    }
    float* o = out + (r0 + lr) * Din + c;
    if (VEC == 2) *reinterpret_cast<f32x2*>(o) = f32x2{v[0], v[VEC - 1]}; else o[0] = v[0];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Target actions. mode 0: onehot_from_logits (util.py:156-175): unavailable -> -1e10, one-hot of (== max), ties give
// several ones. mode 1 (MATD3 target smoothing, MADDPGPolicy.py:94-95): hard gumbel-softmax with the caller's
// uniform noise U: g = -log(-log(U+1e-20)+1e-20); y = softmax(logits+g masked); value (y_hard - y) + y.
// Output is scattered into the joint next action cent_nact[b][a*A + j] (maddpg.py:67-74).
// With `soft_out` non-null (actor update, mode 1 only) the soft sample y is kept for the backward pass and the
// straight-through value goes to act_out[row][A] instead.
// ---------------------------------------------------------------------------------------------------------
// Rows are (t, agent, b) with t < rows / (N*B) (t = 0 only for the MLP family); the joint action of step t goes to
// cent_nact[t - t_shift] and steps t < t_shift are dropped (the recurrent trainer discards the first target action,
// r_maddpg.py:88).
// Blocks stage `rpb` rows through LDS so that all global traffic is coalesced and the gumbel transform (two logs per
// element) is evaluated once; one thread then owns one row in LDS.
__global__ void __launch_bounds__(256) action_kernel(const float* __restrict__ logits, const float* __restrict__ avail,
                                                      NoiseSrc U, int rows, int B, int A, int N, int mode, int t_shift,
                                                      int rpb, float* __restrict__ cent_nact, float* __restrict__ act_out,
                                                      float* __restrict__ soft_out, int nact_stride, int nact_col, ActHeads hd) {
  extern __shared__ float sm[];
  const int pitch = A | 1;
  float* val = sm;                      // [rpb][pitch] masked (noisy) logits, overwritten by the output values
  float* soft = sm + rpb * pitch;       // [rpb][pitch] soft sample (mode 1 with soft_out only)
  const int r0 = blockIdx.x * rpb;
  const int nrows = min(rpb, rows - r0);
  const int64_t base = (int64_t)r0 * A;
  const float invA = 1.0f / (float)A;      // index math on small integers: float reciprocals instead of integer divisions
  for (int e = threadIdx.x; e < nrows * A; e += blockDim.x) {
    const int rr = (int)(((float)e + 0.5f) * invA), j = e - rr * A;
    float v = logits[base + e];
    if (mode == 1) v += -logf(-logf(U.at(r0 + rr, A, j) + 1e-20f) + 1e-20f);
    if (mode == 2 && U.u) v += U.u[base + e];      // continuous actions: the actor output (+ the caller's additive noise) is the action
    if (mode != 2 && avail && avail[base + e] == 0.f) v = -1e10f;
    val[rr * pitch + j] = v;
  }
  __syncthreads();
  if (mode != 2 && (int)threadIdx.x < nrows) {
    float* vr0 = val + threadIdx.x * pitch;
    float* sr0 = soft + threadIdx.x * pitch;
    int j0 = 0;
    for (int h = 0; h < hd.n; ++h) {        // one block, or the blocks of a multi-discrete action (each its own argmax / softmax)
      const int Ah = hd.dim[h];
      float* vr = vr0 + j0;
      float* sr = sr0 + j0;
      j0 += Ah;
      float mx = -3.0e38f;
      for (int j = 0; j < Ah; ++j) mx = fmaxf(mx, vr[j]);
      if (mode == 0) {
        for (int j = 0; j < Ah; ++j) vr[j] = (vr[j] == mx) ? 1.f : 0.f;
      } else {
        // one exp and one division per element (the row was evaluated three times before): e_j -> y_j = e_j / den in `sr`
        float den = 0.f;
        for (int j = 0; j < Ah; ++j) { const float ej = expf(vr[j] - mx); sr[j] = ej; den += ej; }
        // the one-hot is taken on the softmax output y (onehot_from_logits(y)): y == max(y)
        float ymax = 0.f;
        for (int j = 0; j < Ah; ++j) { const float y = sr[j] / den; sr[j] = y; ymax = fmaxf(ymax, y); }
        for (int j = 0; j < Ah; ++j) {
          const float y = sr[j];
          const float hard = (y == ymax) ? 1.f : 0.f;
          vr[j] = (hard - y) + y;
        }
      }
    }
  }
  __syncthreads();
  const float invNB = 1.0f / (float)(N * B), invB = 1.0f / (float)B;
  for (int e = threadIdx.x; e < nrows * A; e += blockDim.x) {
    const int rr = (int)(((float)e + 0.5f) * invA), j = e - rr * A;
    const float out = val[rr * pitch + j];
    if (act_out) act_out[base + e] = out;
    if (soft_out && mode != 2) soft_out[base + e] = soft[rr * pitch + j];
    if (cent_nact) {
      const int r = r0 + rr;
      int t = (int)(((float)r + 0.5f) * invNB);        // exact for r < 2^24; corrected below for larger row counts
      t += (r - t * (N * B) >= N * B) - (r - t * (N * B) < 0);
      if (t >= t_shift) {
        const int rem = r - t * (N * B);
        const int a = (int)(((float)rem + 0.5f) * invB), b = rem - a * B;
        cent_nact[((int64_t)(t - t_shift) * B + b) * nact_stride + nact_col + a * A + j] = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Critic TD (maddpg.py:112-157): target = r + gamma (1 - done) min_k Q'_k ; err_k = target - Q_k ;
//   loss = sum_k mean_b f(err_k) [* w_b] ; dQ_k = -f'(err_k) w_b (un-normalised; mask_count = B) ; priority = mean_k |err_k| + eps
// One thread per transition; per-16-row loss partials in the [tile][4] format finalize expects.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) critic_td_kernel(CriticTdArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < a.B;
  const int bb = valid ? b : 0;
  float qn = a.q_tgt[(int64_t)bb * a.K4];
  for (int k = 1; k < a.K; ++k) qn = fminf(qn, a.q_tgt[(int64_t)bb * a.K4 + k]);
  const float rew = a.rewards[bb];   // agent 0's reward: rewards[0][b] (maddpg.py:106)
  const float den = a.dones_env[bb];
  const float target = rew + a.gamma * (1.0f - den) * qn;
  const float w = a.per_weights ? a.per_weights[bb] : 1.0f;
  float ls = 0.f, pr = 0.f, qs = 0.f;
  for (int k = 0; k < a.K; ++k) {
    const float q = a.q[(int64_t)bb * a.K4 + k];
    const float e = target - q;
    float fe, dfe;
    if (a.use_huber) {
      const float ae = fabsf(e), dl = a.huber_delta;
      if (ae <= dl) { fe = e * e * 0.5f; dfe = e; } else { fe = dl * (ae - dl * 0.5f); dfe = dl * (e > 0.f ? 1.f : -1.f); }
    } else {
      fe = e * e;
      dfe = 2.0f * e;
    }
    ls += w * fe;
    pr += fabsf(e);
    qs += q;
    if (valid) a.dq[(int64_t)b * a.K4 + k] = -dfe * w;   // d loss_sum / d Q_k  (e = target - Q)
  }
  if (valid) {
    for (int k = a.K; k < a.K4; ++k) a.dq[(int64_t)b * a.K4 + k] = 0.f;
    if (a.prio_out) a.prio_out[b] = pr / (float)a.K + a.per_eps;
  }
  if (!valid) { ls = 0.f; qs = 0.f; }
  float cs = valid ? 1.f : 0.f;
  for (int o = 1; o < 16; o <<= 1) {
    ls += __shfl_xor(ls, o, 64);
    cs += __shfl_xor(cs, o, 64);
    qs += __shfl_xor(qs, o, 64);
  }
  if ((threadIdx.x & 15) == 0 && (b >> 4) < ((a.B + 15) >> 4)) {
    float* lp = a.loss_part + (b >> 4) * 4;
    lp[0] = ls; lp[1] = cs; lp[2] = qs; lp[3] = 0.f;
  }
}

// Actor objective (maddpg.py:229-232): loss = -sum(Q_1 * valid) / sum(valid)  ->  dQ_1 = -valid (un-normalised)
__global__ void __launch_bounds__(256) actor_obj_kernel(const float* __restrict__ q, int K4, const float* __restrict__ valid_tr,
                                                         int rows, float* __restrict__ dq, float* __restrict__ loss_part) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = r < rows;
  const float v = ok ? valid_tr[r] : 0.f;
  const float q1 = ok ? q[(int64_t)r * K4] : 0.f;
  if (ok) {
    dq[(int64_t)r * K4] = -v;
    for (int k = 1; k < K4; ++k) dq[(int64_t)r * K4 + k] = 0.f;
  }
  float ls = -q1 * v, cs = v, qs = q1 * v;
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
// Actor update, critic side, fused: the actor only needs d(loss)/d(action block of "its" agent) of the critic input,
// so the full input gradient dx [R][Din] is never formed. With dy_k = gamma_k sum_i W_ik dz_i (k over ALL inputs) the
// input-LayerNorm adjoint needs two row scalars that collapse to 64-long dot products:
//   m1 = mean_k dy_k      = (1/D) sum_i dz_i c_i ,          c_i = sum_k gamma_k W_ik
//   m2 = mean_k dy_k xh_k = (1/D) sum_i dz_i (z1_i - cb_i), cb_i = b_i + sum_k W_ik beta_k ,  z1_i = xhat1_i/rstd1 + mu1
// (z1 is only needed where the ReLU is on; elsewhere dz_i = 0). Then for the A columns k of the agent's action block
//   dx_k = rstd0 (dy_k - m1 - xh_k m2) ,   dlogit_j = y_j (dx_j - sum_m dx_m y_m)        (util.py:210-213)
// One thread per row (t, agent copy, b); weight reads are wave-uniform (rows of a wave share the agent copy).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) fc1_colsum_kernel(ActGradArgs a) {
  const int i = blockIdx.x, lane = threadIdx.x;
  float c = 0.f, e = 0.f;
  for (int k = lane; k < a.Din; k += 64) {
    const float w = a.theta[a.fc1_w + (int64_t)i * a.Din + k];
    c = fmaf(w, a.theta[a.fn_w + k], c);
    e = fmaf(w, a.theta[a.fn_b + k], e);
  }
  for (int o = 32; o > 0; o >>= 1) {
    c += __shfl_xor(c, o, 64);
    e += __shfl_xor(e, o, 64);
  }
  if (lane == 0) {
    a.cvec[i] = c;
    a.cvec[OPE_H + i] = e + a.theta[a.fc1_b + i];
  }
}

__global__ void __launch_bounds__(256) action_grad_kernel(ActGradArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.R) return;
  const int rep = a.a_off + (r / a.B) % a.N;
  const float rs1 = a.rstd1[r], m1u = a.mu1[r], rs0 = a.rstd0[r], mu0 = a.mu0[r];
  const float inv_rs1 = 1.0f / rs1;
  float dz[OPE_H];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int i = 0; i < OPE_H; i += 4) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(a.dz1 + (int64_t)r * OPE_H + i);
    const f32x4 xh = *reinterpret_cast<const f32x4*>(a.xhat1 + (int64_t)r * OPE_H + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dz[i + q] = d[q];
      m1 = fmaf(d[q], a.cvec[i + q], m1);
      m2 = fmaf(d[q], fmaf(xh[q], inv_rs1, m1u) - a.cvec[OPE_H + i + q], m2);
    }
  }
  const float invD = 1.0f / (float)a.Din;
  m1 *= invD;
  m2 *= invD;
  const int col0 = a.S + a.a_col + rep * a.A;
  const float* W = a.theta + a.fc1_w + col0;
  float* out = a.dlogits + (int64_t)r * a.A4;
  float dot = 0.f;
  for (int j = 0; j < a.A; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < OPE_H; ++i) s = fmaf(W[(int64_t)i * a.Din + j], dz[i], s);
    const float dy = s * a.theta[a.fn_w + col0 + j];
    const float xh = (a.act[(int64_t)r * a.A + j] - mu0) * rs0;
    const float dx = rs0 * (dy - m1 - xh * m2);
    if (!a.identity) dot = fmaf(dx, a.y[(int64_t)r * a.A + j], dot);
    out[j] = dx;
  }
  if (a.heads.n > 1) {      // multi-discrete: the softmax adjoint of every block on its own
    int j0 = 0;
    for (int h = 0; h < a.heads.n; ++h) {
      const int Ah = a.heads.dim[h];
      float dh = 0.f;
      for (int j = j0; j < j0 + Ah; ++j) dh = fmaf(out[j], a.y[(int64_t)r * a.A + j], dh);
      for (int j = j0; j < j0 + Ah; ++j) out[j] = a.y[(int64_t)r * a.A + j] * (out[j] - dh);
      j0 += Ah;
    }
    for (int j = a.A; j < a.A4; ++j) out[j] = 0.f;
    return;
  }
  for (int j = 0; j < a.A4; ++j) out[j] = j < a.A ? (a.identity ? out[j] : a.y[(int64_t)r * a.A + j] * (out[j] - dot)) : 0.f;
}

// Same computation on the matrix pipe, for many rows: a wave owns 16 consecutive rows that share the agent copy (B % 16 == 0),
// dy[row][action] = sum_i dz[row][i] W[i][col0 + action] is one or two 16x16 MFMA tiles over K = 64 (transposed-chain lane
// convention: lane (j, g) holds hidden units / actions 16*tile + 4g + r of row j); the two row scalars are lane-local dot
// products + a 4-lane sum. 230 k rows (MMM2, B = 128): 370 us thread-per-row -> this form.
__global__ void __launch_bounds__(256) action_grad_mfma_kernel(ActGradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  if (row0 >= a.R) return;
  const int row = row0 + j;
  const bool valid = row < a.R;
  const int64_t rr = valid ? row : a.R - 1;
  const int rep = a.a_off + (row0 / a.B) % a.N;          // uniform over the tile
  const int col0 = a.S + a.a_col + rep * a.A;
  const float rs1 = a.rstd1[rr], mu1 = a.mu1[rr], rs0 = a.rstd0[rr], mu0 = a.mu0[rr];
  const float inv_rs1 = 1.0f / rs1;
  f32x4 dz[4];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    dz[ft] = *reinterpret_cast<const f32x4*>(a.dz1 + rr * OPE_H + 16 * ft + 4 * g);
    const f32x4 xh = *reinterpret_cast<const f32x4*>(a.xhat1 + rr * OPE_H + 16 * ft + 4 * g);
    const f32x4 cv = *reinterpret_cast<const f32x4*>(a.cvec + 16 * ft + 4 * g);
    const f32x4 cb = *reinterpret_cast<const f32x4*>(a.cvec + OPE_H + 16 * ft + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m1 = fmaf(dz[ft][r], cv[r], m1);
      m2 = fmaf(dz[ft][r], fmaf(xh[r], inv_rs1, mu1) - cb[r], m2);
    }
  }
  const float invD = 1.0f / (float)a.Din;
  m1 = rowsum4(m1) * invD;
  m2 = rowsum4(m2) * invD;
  const int ntile = (a.A + 15) >> 4;           // 1 or 2
  f32x4 dy[2];
  dy[0] = dy[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* W = a.theta + a.fc1_w;
  for (int u = 0; u < ntile; ++u) {
    const int o = 16 * u + j;                   // output (action) index supplied by this lane as the A-operand row
    const int oc = col0 + (o < a.A ? o : a.A - 1);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      f32x4 wv;
#pragma unroll
      for (int r = 0; r < 4; ++r) wv[r] = W[(int64_t)(16 * ft + 4 * g + r) * a.Din + oc];
#pragma unroll
      for (int r = 0; r < 4; ++r) dy[u] = mfma16(wv[r], dz[ft][r], dy[u]);
    }
  }
  // this lane now holds dy for row j, actions 16u + 4g + r
  float dx[2][4], yv[2][4];
  float dot = 0.f;
  for (int u = 0; u < ntile; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * u + 4 * g + r;
      const bool ok = k < a.A;
      const int kc = ok ? k : 0;
      const float y = a.y[rr * a.A + kc];
      const float xh = (a.act[rr * a.A + kc] - mu0) * rs0;
      const float v = rs0 * (dy[u][r] * a.theta[a.fn_w + col0 + kc] - m1 - xh * m2);
      dx[u][r] = ok ? v : 0.f;
      yv[u][r] = ok ? y : 0.f;
      dot = fmaf(dx[u][r], yv[u][r], dot);
    }
  dot = rowsum4(dot);
  if (!valid) return;
  for (int u = 0; u < ntile; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * u + 4 * g + r;
      if (k < a.A4) a.dlogits[(int64_t)row * a.A4 + k] = k < a.A ? yv[u][r] * (dx[u][r] - dot) : 0.f;
    }
}

// Same computation, one wave per row: lane i owns hidden unit i, the 2 + A dot products over the 64 units are wave
// reductions. Used when there are too few rows for the thread-per-row form to fill the machine.
__global__ void __launch_bounds__(256) action_grad_wave_kernel(ActGradArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const int rep = a.a_off + (r / a.B) % a.N;
  const float rs1 = a.rstd1[r], rs0 = a.rstd0[r], mu0 = a.mu0[r];
  const float dz = a.dz1[(int64_t)r * OPE_H + lane];
  const float z1 = a.xhat1[(int64_t)r * OPE_H + lane] / rs1 + a.mu1[r];
  float m1 = dz * a.cvec[lane], m2 = dz * (z1 - a.cvec[OPE_H + lane]);
  for (int o = 32; o > 0; o >>= 1) {
    m1 += __shfl_xor(m1, o, 64);
    m2 += __shfl_xor(m2, o, 64);
  }
  const float invD = 1.0f / (float)a.Din;
  m1 *= invD;
  m2 *= invD;
  const int col0 = a.S + a.a_col + rep * a.A;
  const float* W = a.theta + a.fc1_w + (int64_t)lane * a.Din + col0;
  float mine = 0.f, dot = 0.f;   // lane j (< A) keeps dx_j
  for (int j = 0; j < a.A; ++j) {
    float s = W[j] * dz;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float dy = s * a.theta[a.fn_w + col0 + j];
    const float xh = (a.act[(int64_t)r * a.A + j] - mu0) * rs0;
    const float dx = rs0 * (dy - m1 - xh * m2);
    dot = fmaf(dx, a.y[(int64_t)r * a.A + j], dot);
    if (lane == j) mine = dx;
  }
  if (lane < a.A4) a.dlogits[(int64_t)r * a.A4 + lane] = lane < a.A ? a.y[(int64_t)r * a.A + lane] * (mine - dot) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
static int launch1d(int64_t n) { return ope_cdiv(n, 256); }
#define OPE_L(call)                                            \
  do {                                                         \
    call;                                                      \
    if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;   \
  } while (0)

// The same rows over a joint action of blocks of DIFFERENT widths (policies with different action dimensions, r_maddpg.py:236-301):
//   out[r] = [ cent[t][b] (S) | joint[t][b] (J) ],  r = (t*reps + rep)*B + b,  columns [col0 + rep*Ar, + Ar) of the joint part <- repl[r]
// One thread per output element (small environments; the equal-width form above is the fast one).
__global__ void __launch_bounds__(256) build_cin_joint_kernel(const float* __restrict__ cent, const float* __restrict__ joint,
                                                               const float* __restrict__ repl, int T, int B, int J, int S, int reps, int Ar,
                                                               int col0, float* __restrict__ out) {
  const int Din = S + J;
  const int64_t total = (int64_t)T * reps * B * Din;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / Din;
  const int c = (int)(e - r * Din);
  const int t = (int)(r / ((int64_t)reps * B));
  const int rem = (int)(r - (int64_t)t * reps * B);
  const int rep = rem / B, b = rem - rep * B;
  float v;
  if (c < S) {
    v = cent[((int64_t)t * B + b) * S + c];
  } else {
    const int k = c - S, lo = col0 + rep * Ar;
    v = (repl && k >= lo && k < lo + Ar) ? repl[r * Ar + (k - lo)] : joint[((int64_t)t * B + b) * J + k];
  }
  out[e] = v;
}
int launch_build_cin_joint(const float* cent, const float* joint, const float* repl, int T, int B, int J, int S, int reps, int Ar, int col0,
                           float* out, hipStream_t st) {
  const int64_t total = (int64_t)T * reps * B * (S + J);
  OPE_L(OPE_LAUNCH(build_cin_joint_kernel, dim3(ope_cdiv(total, 256)), dim3(256), 0, st, cent, joint, repl, T, B, J, S, reps, Ar, col0, out));
  return OPE_OK;
}
int launch_build_cin(const float* cent, const float* acts, const float* repl, int T, int B, int N, int A, int S, int reps, float* out,
                     hipStream_t st, int rep_off) {
  const int64_t rows = (int64_t)T * reps * B;
  const bool even = !(S & 1) && !(A & 1) && !((uintptr_t)cent & 7) && !((uintptr_t)acts & 7) && !((uintptr_t)out & 7) && !((uintptr_t)repl & 7);
  if (even)
    OPE_L(OPE_LAUNCH(build_cin_kernel<2>, dim3(ope_cdiv(rows, 16)), dim3(256), 0, st, cent, acts, repl, T, B, N, A, S, reps, rep_off, out));
  else
    OPE_L(OPE_LAUNCH(build_cin_kernel<1>, dim3(ope_cdiv(rows, 16)), dim3(256), 0, st, cent, acts, repl, T, B, N, A, S, reps, rep_off, out));
  return OPE_OK;
}
int launch_action(const float* logits, const float* avail, NoiseSrc U, int rows, int B, int A, int N, int mode, int t_shift,
                  float* cent_nact, float* act_out, float* soft_out, hipStream_t st, int nact_agents, int a_off, const ActHeads* heads,
                  int nact_stride, int nact_col) {
  const ActHeads hd = heads ? *heads : act_heads_of(1, nullptr, A);
  if (hd.n > 1) avail = nullptr;      // multi-discrete: upstream passes no availability masks (MADDPGPolicy.py:73-92)
  const int pitch = A | 1;
  const int rpb = pitch <= 31 ? 256 : 64;
  const size_t lds = (size_t)2 * rpb * pitch * sizeof(float);
  OPE_L(OPE_LAUNCH(action_kernel, dim3(ope_cdiv(rows, rpb)), dim3(256), lds, st, logits, avail, U, rows, B, A, N, mode, t_shift,
                           rpb, cent_nact, act_out, soft_out, nact_stride > 0 ? nact_stride : (nact_agents > 0 ? nact_agents : N) * A,
                           nact_stride > 0 ? nact_col : a_off * A, hd));
  return OPE_OK;
}
int launch_action_grad(const ActGradArgs& a, hipStream_t st) {
  OPE_L(OPE_LAUNCH(fc1_colsum_kernel, dim3(OPE_H), dim3(64), 0, st, a));
  // OPE_ACTGRAD = wave | mfma | thread forces a form (tests); default by size
  const char* f = getenv("OPE_ACTGRAD");
  const bool mfma_ok = a.B % 16 == 0 && a.A <= 32;
  int form = f ? (f[0] == 'w' ? 0 : (f[0] == 'm' && mfma_ok ? 1 : 2)) : (a.R <= 16384 ? 0 : (mfma_ok ? 1 : 2));
  if (a.identity || a.heads.n > 1) form = 2;      // continuous / multi-discrete actions: the thread-per-row form carries those adjoints
  if (form == 0)   // few rows (MLP family): one wave per row, lane = hidden unit -- a 64-long serial chain per thread otherwise
    OPE_L(OPE_LAUNCH(action_grad_wave_kernel, dim3(ope_cdiv(a.R, 4)), dim3(256), 0, st, a));
  else if (form == 1) {
    kprof_work(2.0 * a.R * (2.0 * OPE_H * OPE_H + (double)OPE_H * a.A));     // fc2^T dz2, the 64-long dot products of the LN adjoint, the A action columns of fc1
    OPE_L(OPE_LAUNCH(action_grad_mfma_kernel, dim3(ope_cdiv(ope_cdiv(a.R, 16), 4)), dim3(256), 0, st, a));
  } else
    OPE_L(OPE_LAUNCH(action_grad_kernel, dim3(launch1d(a.R)), dim3(256), 0, st, a));
  return OPE_OK;
}

struct DdpgPlan {
  int N, A, D, S, B, K, K4, A4, Din, Ra;
  int NT, a0;               // agents in the joint action / the update policy's first agent in it (multi-policy; NT = N, a0 = 0 otherwise)
  int J, c0;                // width of the joint action / first column of the update policy's first agent
  bool hetero;              // joint action given in columns (policies of different action dimensions: ope_ddpg_cfg.joint_act_dim)
  AgentLayout AL, CL;       // actor (D -> A), critic (Din -> K)
  Workspace ws;
  int ns_c, ns_a;
  int raw_size_c, raw_size_a;
  int P1, s1, P2, s2, E, sq;    // raw slab offsets (same recipe for actor and critic, sized by the larger)
  int64_t xin_t, xin, a2n, lgn, cnact, a2t, qt, a2c, qc, dq, da2, dz1, dz2, mu0, rstd0, xhat1, rstd1, mask1, xhat2, rstd2, mask2,
      thetaT, raw, rsum, loss_part, lnz, lno, xin_a, a2a, lga, ysoft, actout, mu1, cvec, dlg, err, fused_slabs, gsq_critic, gsq_actor, opt_sync;
  bool fused;
};

static int ddpg_cfg_ok(const ope_ddpg_cfg* c) {
  if (!c) return 0;
  const ope_dims& d = c->dims;
  if (d.n_agents < 1 || d.n_agents > 64 || d.act_dim < 1 || d.act_dim > 64 || d.obs_dim < 1 || d.obs_dim > 512 || d.state_dim < 1) return 0;
  if (c->continuous != 0 && c->continuous != 1) return 0;
  if (c->continuous && c->target_gumbel) return 0;      // continuous actions: the target noise is additive (target_noise_u), there is no gumbel
  if (!act_heads_ok(c->n_act_heads, c->act_head_dims, d.act_dim) || (c->n_act_heads > 1 && c->continuous)) return 0;
  if (d.layer_N > 1 || d.flags) return 0;      // the non-default network shapes (a second hidden block, no input LayerNorm) exist for the Q-learning nets only
  const int nt = c->n_total_agents > 0 ? c->n_total_agents : d.n_agents;
  if (c->n_total_agents < 0 || c->agent_offset < 0 || c->agent_offset + d.n_agents > nt || nt > 64) return 0;
  if (c->n_total_agents <= 0 && c->agent_offset != 0) return 0;
  if (d.state_dim + nt * d.act_dim > 512) return 0;
  if (c->joint_act_dim != 0 && (c->n_total_agents != 0 || c->joint_act_col < 0 || c->joint_act_col + d.n_agents * d.act_dim > c->joint_act_dim ||
                                d.state_dim + c->joint_act_dim > 512 || c->noise_seed))
    return 0;
  if (c->batch < 1 || c->num_q < 1 || c->num_q > 4) return 0;
  return 1;
}

static void ddpg_plan(const ope_ddpg_cfg* c, DdpgPlan* p) {
  const ope_dims& d = c->dims;
  p->N = d.n_agents; p->A = d.act_dim; p->D = d.obs_dim; p->S = d.state_dim; p->B = c->batch; p->K = c->num_q;
  p->NT = c->n_total_agents > 0 ? c->n_total_agents : p->N; p->a0 = c->agent_offset;
  p->hetero = c->joint_act_dim > 0;
  p->J = p->hetero ? c->joint_act_dim : p->NT * p->A; p->c0 = p->hetero ? c->joint_act_col : p->a0 * p->A;
  p->K4 = ope_round4(p->K); p->A4 = ope_round4(p->A); p->Din = p->S + p->J; p->Ra = p->N * p->B;
  p->AL = ope_agent_layout_mlp(p->D, p->A, 0);
  p->CL = ope_agent_layout_mlp(p->Din, p->K, 0);
  const int Dmax = p->D > p->Din ? p->D : p->Din;
  const int Hmax = p->A > p->K ? p->A : p->K;
  int o = 0;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  p->P1 = take(OPE_H * Dmax); p->s1 = take(OPE_H); p->P2 = take(OPE_H * OPE_H); p->s2 = take(OPE_H);
  p->E = take(Hmax * OPE_H); p->sq = take(ope_round4(Hmax));
  p->raw_size_c = p->raw_size_a = o;
  auto splits = [](int64_t rows) { int s = ope_cdiv(rows, 32);   // short K per wave: these launches are latency-bound
    s = s < 1 ? 1 : (s > 64 ? 64 : s); return s >= 4 ? (s / 4) * 4 : s; };
  p->ns_c = splits(p->B);
  p->ns_a = splits(p->Ra);
  Workspace& W = p->ws;
  const int64_t B = p->B, Ra = p->Ra, R = Ra;   // save buffers sized for the larger (actor-side) row count
  p->xin_t = W.add("xin_t", B * p->Din); p->xin = W.add("xin", B * p->Din);
  p->a2n = W.add("a2n", Ra * OPE_H); p->lgn = W.add("logits_n", Ra * p->A); p->cnact = W.add("cent_nact", B * p->J);
  p->a2t = W.add("a2t", B * OPE_H); p->qt = W.add("q_tgt", B * p->K4);
  p->a2c = W.add("a2c", R * OPE_H); p->qc = W.add("q", R * p->K4);
  p->dq = W.add("dq", R * p->K4); p->da2 = W.add("da2", R * OPE_H); p->dz1 = W.add("dz1", R * OPE_H); p->dz2 = W.add("dz2", R * OPE_H);
  p->mu0 = W.add("mu0", R); p->rstd0 = W.add("rstd0", R); p->xhat1 = W.add("xhat1", R * OPE_H); p->rstd1 = W.add("rstd1", R);
  p->mask1 = W.add("mask1", 2 * R); p->xhat2 = W.add("xhat2", R * OPE_H); p->rstd2 = W.add("rstd2", R); p->mask2 = W.add("mask2", 2 * R);
  p->thetaT = W.add("thetaT", OPE_H * 3 * OPE_H + OPE_H * OPE_H);
  const int nsmax = p->ns_c > p->ns_a ? p->ns_c : p->ns_a;
  p->raw = W.add("raw", (int64_t)nsmax * o); p->rsum = W.add("rsum", o + 4);
  p->loss_part = W.add("loss_part", (int64_t)ope_cdiv(R, 16) * 4);
  p->lnz = W.add("ln_zero", R); p->lno = W.add("ln_one", R);
  p->xin_a = W.add("xin_a", Ra * p->Din); p->a2a = W.add("a2a", Ra * OPE_H); p->lga = W.add("logits", Ra * p->A);
  p->ysoft = W.add("y_soft", Ra * p->A); p->actout = W.add("act_out", Ra * p->A);
  p->mu1 = W.add("mu1", R); p->cvec = W.add("fc1_colsums", 2 * OPE_H); p->dlg = W.add("dlogits", Ra * p->A4);
  // second set of trunk saves for the actor's own backward (the critic pass of the actor step reuses the first set)
  p->err = W.add("saves2", Ra * (2 * OPE_H + 8));
  p->fused = ddpg_fused_ok(p->N, p->A, p->D, p->S, p->K) && p->NT == p->N && !p->hetero && !c->continuous && c->n_act_heads <= 1;   // (the tile kernels are the one-shared-policy, discrete-action form)
  p->fused_slabs = W.add("fused_slabs", p->fused ? ddpg_fused_slab_floats(p->N, p->A, p->D, p->S, p->K, p->B) + 64 : 4);   // + debug stamps
  if (p->fused) {    // per-workgroup sums of squares of the gradient the slab reduction wrote (only the fused path produces them)
    p->gsq_critic = W.add("gsq_critic", 2 * ddpg_fused_gsq_blocks(p->N, p->A, p->D, p->S, p->K, true));
    p->gsq_actor = W.add("gsq_actor", 2 * ddpg_fused_gsq_blocks(p->N, p->A, p->D, p->S, p->K, false));
    p->opt_sync = W.add("opt_sync", 8);      // grid-barrier state of the optimiser tails (ints; zero between launches)
  }
}

// trunk forward in mlp mode on `rows` rows of width Dw; saves go to the plan's first save set unless `alt` is given
// (+ the net's small Linear head fused into the same launch: head_out [rows][head_dim])
static int trunk_mlp(const DdpgPlan& p, float* W, const float* x, int rows, int Dw, const float* theta, const AgentLayout& L,
                     float* a2_out, bool save, float* alt, float* head_out, int head_dim, hipStream_t st) {
  TrunkFwdArgs tf;
  memset(&tf, 0, sizeof(tf));
  tf.x = x; tf.R = rows; tf.D = Dw; tf.theta = theta; tf.L = L; tf.a2_out = a2_out; tf.head_out = head_out; tf.head_dim = head_dim;
  if (save) {
    if (!alt) {
      tf.mu0 = W + p.mu0; tf.rstd0 = W + p.rstd0; tf.xhat1 = W + p.xhat1; tf.rstd1 = W + p.rstd1; tf.mask1 = (uint64_t*)(W + p.mask1);
      tf.mu1 = W + p.mu1;
      tf.xhat2 = W + p.xhat2; tf.rstd2 = W + p.rstd2; tf.mask2 = (uint64_t*)(W + p.mask2);
    } else {   // packed alternate save set: [mu0 R][rstd0 R][rstd1 R][rstd2 R][mask1 2R][mask2 2R][xhat1 64R][xhat2 64R]
      const int64_t R = p.Ra;
      tf.mu0 = alt; tf.rstd0 = alt + R; tf.rstd1 = alt + 2 * R; tf.rstd2 = alt + 3 * R;
      tf.mask1 = (uint64_t*)(alt + 4 * R); tf.mask2 = (uint64_t*)(alt + 6 * R); tf.xhat1 = alt + 8 * R; tf.xhat2 = alt + 8 * R + R * OPE_H;
    }
  }
  return launch_trunk_fwd(tf, save, st);
}

// backward of one MLP net given d(head output) [rows][ldk]: da2 -> trunk_bwd -> weight gradients -> flat grad (+tail)
static int mlp_backward(const DdpgPlan& p, float* W, const float* x, int rows, int Dw, int Hout, int ldk, const float* dout,
                        const float* theta, const AgentLayout& L, const float* saves_alt, int nsplit, int n_loss_tiles,
                        float* grad, hipStream_t st) {
  int rc;
  if ((rc = launch_transpose(theta + L.fc2_w, OPE_H, OPE_H, W + p.thetaT + OPE_H * 3 * OPE_H, st))) return rc;
  const float *mu0, *rstd0, *xhat1, *rstd1, *xhat2, *rstd2;
  const uint64_t *mask1, *mask2;
  if (!saves_alt) {
    mu0 = W + p.mu0; rstd0 = W + p.rstd0; xhat1 = W + p.xhat1; rstd1 = W + p.rstd1; mask1 = (const uint64_t*)(W + p.mask1);
    xhat2 = W + p.xhat2; rstd2 = W + p.rstd2; mask2 = (const uint64_t*)(W + p.mask2);
  } else {
    const int64_t R = p.Ra;
    mu0 = saves_alt; rstd0 = saves_alt + R; rstd1 = saves_alt + 2 * R; rstd2 = saves_alt + 3 * R;
    mask1 = (const uint64_t*)(saves_alt + 4 * R); mask2 = (const uint64_t*)(saves_alt + 6 * R);
    xhat1 = saves_alt + 8 * R; xhat2 = saves_alt + 8 * R + R * OPE_H;
  }
  TrunkBwdArgs tb;
  memset(&tb, 0, sizeof(tb));
  tb.R = rows; tb.theta = theta; tb.thetaT = W + p.thetaT; tb.L = L; tb.dout = dout; tb.ldk = ldk; tb.hdim = Hout;   // head adjoint fused
  tb.xhat1 = xhat1; tb.rstd1 = rstd1; tb.mask1 = mask1; tb.xhat2 = xhat2; tb.rstd2 = rstd2; tb.mask2 = mask2;
  tb.dz1 = W + p.dz1; tb.dz2 = W + p.dz2;
  if ((rc = launch_trunk_bwd(tb, st))) return rc;
  if (!grad) return OPE_OK;   // input-gradient-only pass (critic inside the actor update)
  WgTable wt;
  memset(&wt, 0, sizeof(wt));
  int n = 0;
  auto prob = [&](const float* A_, int lda, int M, const float* B_, int ldb, int N_, int out_off, int ldc, int s_off) -> WgProb& {
    WgProb& q = wt.p[n++];
    q.A = A_; q.lda = lda; q.M = M; q.B = B_; q.ldb = ldb; q.N = N_; q.K = rows; q.b_shift = 0; q.ln_mu = W + p.lnz; q.ln_rstd = W + p.lno;
    q.out_off = out_off; q.ldc = ldc; q.s_off = s_off; q.nsplit = nsplit; q.raw_base = p.raw; q.raw_stride = p.raw_size_c;
    return q;
  };
  {
    WgProb& q = prob(W + p.dz1, OPE_H, OPE_H, x, Dw, Dw, p.P1, Dw, p.s1);
    q.ln_mu = mu0; q.ln_rstd = rstd0; q.ln_on = 1;
  }
  prob(W + p.dz2, OPE_H, OPE_H, xhat1, OPE_H, OPE_H, p.P2, OPE_H, p.s2);
  prob(dout, ldk, Hout, xhat2, OPE_H, OPE_H, p.E, OPE_H, p.sq);
  wt.n = n;
  if ((rc = wg_finish(&wt))) return rc;
  if ((rc = launch_wgrad(wt, W, st))) return rc;
  SplitRed sr;
  sr.raw0 = W + p.raw; sr.n0 = p.raw_size_c; sr.ns0 = wg_slabs(wt, nsplit); sr.raw1 = W + p.raw; sr.n1 = 0; sr.ns1 = 0; sr.rsum = W + p.rsum;
  if ((rc = launch_split_reduce(sr, st))) return rc;
  FinTable ft;
  memset(&ft, 0, sizeof(ft));
  int k = 0;
  auto seg = [&](int begin, int size, int kind, int src, int src_s, int M, int K_, int w, int gamma, int beta) {
    FinSeg& s = ft.seg[k++];
    s.begin = begin; s.size = size; s.kind = kind; s.src = src; s.src_s = src_s; s.M = M; s.K = K_; s.w = w; s.gamma = gamma; s.beta = beta;
  };
  seg(L.fn_w, Dw, FIN_LNLIN_G, p.P1, p.s1, OPE_H, Dw, L.fc1_w, 0, 0);
  seg(L.fn_b, Dw, FIN_LNLIN_B, p.P1, p.s1, OPE_H, Dw, L.fc1_w, 0, 0);
  seg(L.fc1_w, OPE_H * Dw, FIN_LNLIN_W, p.P1, p.s1, OPE_H, Dw, L.fc1_w, L.fn_w, L.fn_b);
  seg(L.fc1_b, OPE_H, FIN_COPY, p.s1, 0, 0, 0, 0, 0, 0);
  seg(L.ln1_w, OPE_H, FIN_LNLIN_G, p.P2, p.s2, OPE_H, OPE_H, L.fc2_w, 0, 0);
  seg(L.ln1_b, OPE_H, FIN_LNLIN_B, p.P2, p.s2, OPE_H, OPE_H, L.fc2_w, 0, 0);
  seg(L.fch_w, 0, FIN_ZERO, 0, 0, 0, 0, 0, 0, 0);
  seg(L.fc2_w, OPE_H * OPE_H, FIN_LNLIN_W, p.P2, p.s2, OPE_H, OPE_H, L.fc2_w, L.ln1_w, L.ln1_b);
  seg(L.fc2_b, OPE_H, FIN_COPY, p.s2, 0, 0, 0, 0, 0, 0);
  seg(L.ln2_w, OPE_H, FIN_LNLIN_G, p.E, p.sq, Hout, OPE_H, L.q_w, 0, 0);
  seg(L.ln2_b, OPE_H, FIN_LNLIN_B, p.E, p.sq, Hout, OPE_H, L.q_w, 0, 0);
  seg(L.q_w, Hout * OPE_H, FIN_LNLIN_W, p.E, p.sq, Hout, OPE_H, L.q_w, L.ln2_w, L.ln2_b);
  seg(L.q_b, Hout, FIN_COPY, p.sq, 0, 0, 0, 0, 0, 0);
  seg(L.end, OPE_GRAD_TAIL, FIN_TAIL, 0, 0, 0, 0, 0, 0, 0);
  ft.n = k;
  ft.total = L.end + OPE_GRAD_TAIL;
  return launch_finalize(ft, W + p.rsum, theta, W + p.loss_part, n_loss_tiles, grad, st);
}

}  // namespace ope

using namespace ope;

extern "C" int64_t ope_ddpg_param_layout(const ope_ddpg_cfg* cfg, int32_t which, int64_t* offsets, int64_t* sizes) {
  if (!ddpg_cfg_ok(cfg) || which < 0 || which > 1) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  const AgentLayout& L = which == 0 ? p.AL : p.CL;
  const int Dw = which == 0 ? p.D : p.Din, Ho = which == 0 ? p.A : p.K;
  const int off[16] = {L.fn_w, L.fn_b, L.fc1_w, L.fc1_b, L.ln1_w, L.ln1_b, L.fch_w, L.fch_b, L.lnh_w, L.lnh_b, L.fc2_w, L.fc2_b, L.ln2_w, L.ln2_b, L.q_w, L.q_b};
  const int siz[16] = {Dw, Dw, OPE_H * Dw, OPE_H, OPE_H, OPE_H, OPE_H * OPE_H, OPE_H, OPE_H, OPE_H, OPE_H * OPE_H, OPE_H, OPE_H, OPE_H, Ho * OPE_H, Ho};
  for (int i = 0; i < 16; ++i) {
    if (offsets) offsets[i] = off[i];
    if (sizes) sizes[i] = siz[i];
  }
  return L.end;
}

extern "C" int64_t ope_ddpg_workspace_bytes(const ope_ddpg_cfg* cfg) {
  if (!ddpg_cfg_ok(cfg)) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  return p.ws.total * (int64_t)sizeof(float);
}

extern "C" int64_t ope_ddpg_workspace_find(const ope_ddpg_cfg* cfg, const char* name, int64_t* n_floats) {
  if (!ddpg_cfg_ok(cfg) || !name) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  const int64_t off = p.ws.find(name, n_floats);
  return off < 0 ? -1 : off * (int64_t)sizeof(float);
}

extern "C" int ope_ddpg_workspace_init(const ope_ddpg_cfg* cfg, void* workspace, int64_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!ddpg_cfg_ok(cfg) || !workspace) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  float* W = (float*)workspace;
  int rc;
  if ((rc = launch_fill(W + p.lnz, p.Ra, 0.f, (hipStream_t)stream))) return rc;
  if (p.fused && (rc = launch_fill(W + p.opt_sync, 8, 0.f, (hipStream_t)stream))) return rc;
  return launch_fill(W + p.lno, p.Ra, 1.f, (hipStream_t)stream);
}

// ope_ddpg_opt -> the tile launch's argument block
static int tile_opt_from(const ope_ddpg_opt* o, float* theta, int* sync, TileOpt* t) {
  if (!o || !theta || !o->adam_m || !o->adam_v || o->n < 4 || (o->n & 3) || (o->adam.do_polyak && !o->theta_tgt)) return OPE_EINVAL;
  if (!o->adam.step_counter && o->adam.step < 1) return OPE_EINVAL;
  memset(t, 0, sizeof(*t));
  t->n_opt = (int)o->n; t->theta = theta; t->tgt = o->theta_tgt; t->m = o->adam_m; t->v = o->adam_v;
  t->lr = o->adam.lr; t->beta1 = o->adam.beta1; t->beta2 = o->adam.beta2; t->eps = o->adam.eps; t->max_norm = o->adam.max_grad_norm;
  t->wd = o->adam.weight_decay; t->tau = o->adam.tau; t->qden = o->adam.qtot_denominator != 0.f ? o->adam.qtot_denominator : 1.f;
  t->do_polyak = o->adam.do_polyak; t->step = o->adam.step; t->step_counter = o->adam.step_counter; t->sync = sync; t->stats = o->stats_out;
  return OPE_OK;
}

extern "C" int ope_ddpg_update_ok(const ope_ddpg_cfg* cfg) {
  if (!ddpg_cfg_ok(cfg)) return 0;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  return p.fused && ddpg_tile_opt_ok(p.N, p.A, p.D, p.S, p.K, p.B) ? 1 : 0;
}

extern "C" int ope_ddpg_critic_update(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor_tgt, float* theta_critic,
                                      const float* theta_critic_tgt, const float* target_noise_u, const float* per_weights, void* workspace,
                                      int64_t workspace_bytes, float* grad, float* prio_out, const ope_ddpg_opt* opt, void* stream) {
  (void)hipGetLastError();
  if (!ddpg_cfg_ok(cfg) || !bt || !theta_critic || !theta_critic_tgt || !theta_actor_tgt || !workspace || !grad || cfg->joint_next_acts) return OPE_EINVAL;
  if (!bt->next_obs || !bt->share_obs || !bt->acts || !bt->rewards || !bt->next_share_obs || !bt->dones_env) return OPE_EINVAL;
  if (cfg->target_gumbel && !target_noise_u && !cfg->noise_seed) return OPE_EINVAL;
  if (cfg->use_per && !per_weights) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  if (!p.fused) return OPE_EINVAL;
  float* W = (float*)workspace;
  TileOpt t;
  int rc = tile_opt_from(opt, theta_critic, reinterpret_cast<int*>(W + p.opt_sync), &t);
  if (rc) return rc;
  return launch_ddpg_critic_fused(cfg, bt, theta_actor_tgt, theta_critic, theta_critic_tgt, target_noise_u, per_weights, W + p.fused_slabs,
                                  grad, prio_out, W + p.gsq_critic, (hipStream_t)stream, &t);
}

extern "C" int ope_ddpg_actor_update(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, float* theta_actor, const float* theta_critic,
                                     const float* gumbel_noise_u, void* workspace, int64_t workspace_bytes, float* grad,
                                     const ope_ddpg_opt* opt, void* stream) {
  (void)hipGetLastError();
  if (!ddpg_cfg_ok(cfg) || !bt || !theta_actor || !theta_critic || (!gumbel_noise_u && !cfg->noise_seed) || !workspace || !grad) return OPE_EINVAL;
  if (!bt->obs || !bt->share_obs || (!bt->acts && !cfg->joint_acts) || !bt->valid_transition) return OPE_EINVAL;
  if (cfg->joint_act_dim > 0 && !cfg->joint_acts) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  if (!p.fused) return OPE_EINVAL;
  float* W = (float*)workspace;
  TileOpt t;
  int rc = tile_opt_from(opt, theta_actor, reinterpret_cast<int*>(W + p.opt_sync), &t);
  if (rc) return rc;
  return launch_ddpg_actor_fused(cfg, bt, theta_actor, theta_critic, gumbel_noise_u, W + p.fused_slabs, grad, W + p.gsq_actor,
                                 (hipStream_t)stream, &t);
}

extern "C" int ope_ddpg_critic_loss_and_grad(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor_tgt,
                                             const float* theta_critic, const float* theta_critic_tgt, const float* target_noise_u,
                                             const float* per_weights, void* workspace, int64_t workspace_bytes, float* grad,
                                             float* prio_out, void* stream) {
  (void)hipGetLastError();
  if (!ddpg_cfg_ok(cfg) || !bt || !theta_critic || !theta_critic_tgt || !workspace || !grad) return OPE_EINVAL;
  const bool joint = cfg->joint_next_acts != nullptr;
  if (!joint && (!theta_actor_tgt || !bt->next_obs)) return OPE_EINVAL;
  if (!joint && cfg->n_total_agents > cfg->dims.n_agents) return OPE_EINVAL;   // other policies' target actions must come from the caller
  if (!bt->share_obs || (!bt->acts && !cfg->joint_acts) || !bt->rewards || !bt->next_share_obs || !bt->dones_env) return OPE_EINVAL;
  if (cfg->joint_act_dim > 0 && (!joint || !cfg->joint_acts)) return OPE_EINVAL;
  if (!joint && cfg->target_gumbel && !target_noise_u && !cfg->noise_seed) return OPE_EINVAL;
  if (cfg->use_per && !per_weights) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;
  if (p.fused)   // small networks: the whole critic update in one launch + a slab reduction (ope_ddpg_fused.hip)
    return launch_ddpg_critic_fused(cfg, bt, theta_actor_tgt, theta_critic, theta_critic_tgt, target_noise_u, per_weights,
                                    W + p.fused_slabs, grad, prio_out, W + p.gsq_critic, st);
  // target actor on the next observations -> joint next action (multi-policy: the caller collected it, one
  // ope_ddpg_target_actions per policy)
  const ActHeads hd = act_heads_of(cfg->n_act_heads, cfg->act_head_dims, p.A);
  if (!joint) {
    if ((rc = trunk_mlp(p, W, bt->next_obs, p.Ra, p.D, theta_actor_tgt, p.AL, W + p.a2n, false, nullptr, W + p.lgn, p.A, st))) return rc;
    if ((rc = launch_action(W + p.lgn, bt->next_avail_acts, NoiseSrc{target_noise_u, cfg->noise_seed, cfg->noise_counter, 0}, p.Ra, p.B, p.A, p.N, cfg->continuous ? 2 : (cfg->target_gumbel ? 1 : 0), 0, W + p.cnact,
                            nullptr, nullptr, st, 0, 0, &hd))) return rc;
  }
  const float* cnact = joint ? cfg->joint_next_acts : W + p.cnact;
  // critic inputs
  if ((rc = launch_build_cin(bt->next_share_obs, cnact, nullptr, 1, p.B, 1, p.J, p.S, 1, W + p.xin_t, st))) return rc;
  if ((rc = launch_build_cin(bt->share_obs, p.hetero ? cfg->joint_acts : bt->acts, nullptr, 1, p.B, p.hetero ? 1 : p.NT, p.hetero ? p.J : p.A, p.S, 1,
                             W + p.xin, st)))
    return rc;
  // target critic, live critic
  if ((rc = trunk_mlp(p, W, W + p.xin_t, p.B, p.Din, theta_critic_tgt, p.CL, W + p.a2t, false, nullptr, W + p.qt, p.K, st))) return rc;   // [B][K]
  if ((rc = trunk_mlp(p, W, W + p.xin, p.B, p.Din, theta_critic, p.CL, W + p.a2c, true, nullptr, W + p.qc, p.K, st))) return rc;
  CriticTdArgs td;
  td.B = p.B; td.K = p.K; td.K4 = p.K; td.gamma = cfg->gamma; td.use_huber = cfg->use_huber; td.huber_delta = cfg->huber_delta;
  td.per_eps = cfg->per_eps; td.q = W + p.qc; td.q_tgt = W + p.qt; td.rewards = bt->rewards; td.dones_env = bt->dones_env;
  td.per_weights = cfg->use_per ? per_weights : nullptr; td.dq = W + p.dq; td.prio_out = prio_out; td.loss_part = W + p.loss_part;
  OPE_L(OPE_LAUNCH(critic_td_kernel, dim3(launch1d(p.B)), dim3(256), 0, st, td));
  return mlp_backward(p, W, W + p.xin, p.B, p.Din, p.K, p.K, W + p.dq, theta_critic, p.CL, nullptr, p.ns_c, ope_cdiv(p.B, 16), grad, st);
}

extern "C" int ope_ddpg_target_actions(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor_tgt,
                                       const float* target_noise_u, void* workspace, int64_t workspace_bytes, float* joint_next_acts,
                                       void* stream) {
  (void)hipGetLastError();
  if (!ddpg_cfg_ok(cfg) || !bt || !bt->next_obs || !theta_actor_tgt || !workspace || !joint_next_acts) return OPE_EINVAL;
  if (cfg->target_gumbel && !target_noise_u && !cfg->noise_seed) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;
  const ActHeads hd = act_heads_of(cfg->n_act_heads, cfg->act_head_dims, p.A);
  if ((rc = trunk_mlp(p, W, bt->next_obs, p.Ra, p.D, theta_actor_tgt, p.AL, W + p.a2n, false, nullptr, W + p.lgn, p.A, st))) return rc;
  // (device-drawn noise: one Philox stream per policy, so that two policies' target noise is not the same numbers)
  return launch_action(W + p.lgn, bt->next_avail_acts, NoiseSrc{target_noise_u, cfg->noise_seed, cfg->noise_counter, 16 + p.a0}, p.Ra, p.B, p.A,
                       p.N, cfg->continuous ? 2 : (cfg->target_gumbel ? 1 : 0), 0, joint_next_acts, nullptr, nullptr, st, p.NT, p.a0, &hd,
                       p.hetero ? p.J : 0, p.c0);
}

extern "C" int ope_ddpg_actor_loss_and_grad(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor,
                                            const float* theta_critic, const float* gumbel_noise_u, void* workspace,
                                            int64_t workspace_bytes, float* grad, void* stream) {
  (void)hipGetLastError();
  if (!ddpg_cfg_ok(cfg) || !bt || !theta_actor || !theta_critic || (!cfg->continuous && !gumbel_noise_u && !cfg->noise_seed) || !workspace || !grad) return OPE_EINVAL;
  if (!bt->obs || !bt->share_obs || (!bt->acts && !cfg->joint_acts) || !bt->valid_transition) return OPE_EINVAL;
  if (cfg->joint_act_dim > 0 && !cfg->joint_acts) return OPE_EINVAL;
  DdpgPlan p;
  ddpg_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;
  if (p.fused) return launch_ddpg_actor_fused(cfg, bt, theta_actor, theta_critic, gumbel_noise_u, W + p.fused_slabs, grad, W + p.gsq_actor, st);
  float* saves2 = W + p.err;
  const ActHeads hd = act_heads_of(cfg->n_act_heads, cfg->act_head_dims, p.A);
  // actor forward (saves -> alternate set) and straight-through hard gumbel sample
  if ((rc = trunk_mlp(p, W, bt->obs, p.Ra, p.D, theta_actor, p.AL, W + p.a2a, true, saves2, W + p.lga, p.A, st))) return rc;
  if ((rc = launch_action(W + p.lga, bt->avail_acts, NoiseSrc{cfg->continuous ? nullptr : gumbel_noise_u, cfg->noise_seed, cfg->noise_counter, 1}, p.Ra, p.B, p.A, p.N, cfg->continuous ? 2 : 1, 0, nullptr, W + p.actout, W + p.ysoft, st, 0, 0, &hd)))
    return rc;
  // N stacked copies of the joint action, copy i carrying the actor's action for agent i
  // (multi-policy: copy i of the joint action of ALL agents, block agent_offset + i replaced)
  if (p.hetero) {
    if ((rc = launch_build_cin_joint(bt->share_obs, cfg->joint_acts, W + p.actout, 1, p.B, p.J, p.S, p.N, p.A, p.c0, W + p.xin_a, st))) return rc;
  } else if ((rc = launch_build_cin(bt->share_obs, bt->acts, W + p.actout, 1, p.B, p.NT, p.A, p.S, p.N, W + p.xin_a, st, p.a0))) {
    return rc;
  }
  // critic (parameters frozen) on the stacked input; only head 0 enters the objective
  if ((rc = trunk_mlp(p, W, W + p.xin_a, p.Ra, p.Din, theta_critic, p.CL, W + p.a2c, true, nullptr, W + p.qc, p.K, st))) return rc;
  OPE_L(OPE_LAUNCH(actor_obj_kernel, dim3(launch1d(p.Ra)), dim3(256), 0, st, W + p.qc, p.K, bt->valid_transition, p.Ra,
                           W + p.dq, W + p.loss_part));
  // critic backward down to its input, then through the gumbel-softmax into the actor logits
  if ((rc = mlp_backward(p, W, W + p.xin_a, p.Ra, p.Din, p.K, p.K, W + p.dq, theta_critic, p.CL, nullptr, p.ns_a, 0, nullptr, st))) return rc;
  ActGradArgs ag;
  ag.R = p.Ra; ag.B = p.B; ag.N = p.N; ag.A = p.A; ag.A4 = p.A4; ag.S = p.S; ag.Din = p.Din; ag.a_off = p.hetero ? 0 : p.a0; ag.a_col = p.hetero ? p.c0 : 0; ag.dz1 = W + p.dz1;
  ag.xhat1 = W + p.xhat1; ag.rstd1 = W + p.rstd1; ag.mu1 = W + p.mu1; ag.mu0 = W + p.mu0; ag.rstd0 = W + p.rstd0;
  ag.act = W + p.actout; ag.y = W + p.ysoft; ag.theta = theta_critic; ag.fc1_w = p.CL.fc1_w; ag.fc1_b = p.CL.fc1_b; ag.fn_w = p.CL.fn_w;
  ag.fn_b = p.CL.fn_b; ag.cvec = W + p.cvec; ag.dlogits = W + p.dlg; ag.identity = cfg->continuous ? 1 : 0; ag.heads = hd;
  if ((rc = launch_action_grad(ag, st))) return rc;
  // actor backward and gradients
  return mlp_backward(p, W, bt->obs, p.Ra, p.D, p.A, p.A4, W + p.dlg, theta_actor, p.AL, saves2, p.ns_a, ope_cdiv(p.Ra, 16), grad, st);
}
