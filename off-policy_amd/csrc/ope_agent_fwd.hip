// Agent q-network forward kernels (AgentQFunction.forward: agent_q_function.py:34-67 -> RNNBase.forward rnn.py:33-47
// -> MLPLayer.forward mlp.py:25-29, RNNLayer.forward rnn.py:19-23, ACTLayer.forward act.py:21-37).
//
//   trunk_fwd : per data row   LN_D -> fc1+ReLU+LN -> fc2+ReLU+LN -> gi = W_ih a2 + b_ih        (f32 MFMA chain)
//   gru_fwd   : per (agent,episode) row, serial over t: h_t = GRU(gi_t, h_{t-1})                (ope_gru4.hip / ope_gru1.hip)
//   head_fwd  : per data row   LN(h_t) -> q = W_q y + b_q, chosen-action q, masked greedy argmax, target q at greedy
#include <stdlib.h>
#include <string.h>

#include "ope_agent.h"

namespace ope {

// ---------------------------------------------------------------------------------------------------------
// trunk_fwd. One wave = RT x 16 data rows through the whole MLP trunk; weights stream from L2 as the MFMA A operand
// and every weight fragment is reused by the RT row tiles. The K loop of the first (widest) layer is branch-free
// (clamped loads, masked gamma/beta) and register double-buffered: the loads of chunk c+1 are in flight while the
// 16*RT MFMAs of chunk c issue.
// ---------------------------------------------------------------------------------------------------------
template <int VEC, int RT>
struct TrunkChunk {
  f32x4 w[4];       // fc1 rows 16it+j, features k..k+3
  f32x4 gam, bet;   // feature-norm affine (zeroed beyond D)
  f32x4 x[RT];
};

template <int VEC, int RT>
__device__ __forceinline__ void trunk_fetch(TrunkChunk<VEC, RT>& ch, const float* __restrict__ th, const AgentLayout& L,
                                            const float* const (&xrow)[RT], int j, int k, int D) {
#pragma unroll
  for (int it = 0; it < 4; ++it) ch.w[it] = load4c<VEC>(th + L.fc1_w + (int64_t)(16 * it + j) * D, k, D);
  ch.gam = load4c<VEC>(th + L.fn_w, k, D);
  ch.bet = load4c<VEC>(th + L.fn_b, k, D);
#pragma unroll
  for (int t = 0; t < RT; ++t) ch.x[t] = load4c<VEC>(xrow[t], k, D);
}

template <int VEC, int RT, bool SAVE>
__global__ void __launch_bounds__(256) trunk_fwd_kernel(TrunkFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * RT);
  if (row0 >= a.R) return;
  const int D = a.D;
  const int KC = (D + 15) >> 4;
  const float* __restrict__ th = a.theta;
  int row[RT];
  bool valid[RT];
  const float* xrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    row[t] = row0 + 16 * t + j;
    valid[t] = row[t] < a.R;
    xrow[t] = a.x + (int64_t)min(row[t], a.R - 1) * D;
  }

  // ---- input LayerNorm statistics: one pass, shifted by the row's first element (stable for mean != 0) ----
  float mu[RT], rstd[RT];
  {
    float s1[RT], s2[RT], sh[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { s1[t] = 0.f; s2[t] = 0.f; sh[t] = xrow[t][0]; }
    for (int c = 0; c < KC; ++c) {
      const int k = 16 * c + 4 * g;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const f32x4 v = load4c<VEC>(xrow[t], k, D);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = (k + r < D) ? v[r] - sh[t] : 0.f;
          s1[t] += d;
          s2[t] = fmaf(d, d, s2[t]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float m = rowsum4(s1[t]) / (float)D;
      const float var = fmaxf(rowsum4(s2[t]) / (float)D - m * m, 0.f);
      mu[t] = a.no_fn ? 0.f : sh[t] + m;
      rstd[t] = a.no_fn ? 1.0f : 1.0f / sqrtf(var + OPE_LN_EPS);
      if (SAVE && valid[t] && g == 0) {
        a.mu0[row[t]] = mu[t];
        a.rstd0[row[t]] = rstd[t];
      }
    }
  }

  // ---- fc1: z1 = W1 (xhat*gamma+beta) + b1, K loop double-buffered ----
  f32x4 acc[RT][4];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[t][it] = *reinterpret_cast<const f32x4*>(th + a.L.fc1_b + 16 * it + 4 * g);
  {
    // two named chunk buffers, loop unrolled by two: no register copies, so the compiler cannot fold the prefetch away.
    // A chunk index past KC-1 is harmless: loads are clamped and gamma/beta are masked to zero -> its MFMAs add 0.
    TrunkChunk<VEC, RT> bufA, bufB;
    auto compute = [&](const TrunkChunk<VEC, RT>& ch, int c) {
      const int k = 16 * c + 4 * g;
      const f32x4 gam = mask4(ch.gam, k, D), bet = mask4(ch.bet, k, D);
      f32x4 xn[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) xn[t][r] = fmaf((ch.x[t][r] - mu[t]) * rstd[t], gam[r], bet[r]);  // exactly 0 beyond D
      // r outermost: consecutive MFMAs hit different accumulators (a dependent 16x16x4 pair costs 40 cycles, not 32)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int it = 0; it < 4; ++it) acc[t][it] = mfma16(ch.w[it][r], xn[t][r], acc[t][it]);
    };
    trunk_fetch<VEC, RT>(bufA, th, a.L, xrow, j, 4 * g, D);
    for (int c = 0; c < KC; c += 2) {
      trunk_fetch<VEC, RT>(bufB, th, a.L, xrow, j, 16 * (c + 1) + 4 * g, D);
      __builtin_amdgcn_sched_barrier(0);   // pin: loads of one buffer issue before the MFMAs of the other
      compute(bufA, c);
      __builtin_amdgcn_sched_barrier(0);
      trunk_fetch<VEC, RT>(bufA, th, a.L, xrow, j, 16 * (c + 2) + 4 * g, D);
      __builtin_amdgcn_sched_barrier(0);
      compute(bufB, c + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- ReLU + LN, fc2, ReLU + LN ----
  f32x4 act[RT][4];
  uint32_t mbits[RT];
  float rs[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    float mu1v;
    relu_ln64<SAVE>(acc[t], th + a.L.ln1_w, th + a.L.ln1_b, g, act[t], &rs[t], &mbits[t], &mu1v);
    if (SAVE && valid[t]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.xhat1 + (int64_t)row[t] * OPE_H + 16 * it + 4 * g) = acc[t][it];
      store_mask_rstd(a.mask1, a.rstd1, row[t], g, mbits[t], rs[t]);
      if (a.mu1 && g == 0) a.mu1[row[t]] = mu1v;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[t][it] = *reinterpret_cast<const f32x4*>(th + a.L.fc2_b + 16 * it + 4 * g);
  }
  gemm64rt<4, RT>(th + a.L.fc2_w, OPE_H, j, g, act, acc);
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    relu_ln64<SAVE>(acc[t], th + a.L.ln2_w, th + a.L.ln2_b, g, act[t], &rs[t], &mbits[t]);
    if (SAVE && valid[t]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.xhat2 + (int64_t)row[t] * OPE_H + 16 * it + 4 * g) = acc[t][it];
      store_mask_rstd(a.mask2, a.rstd2, row[t], g, mbits[t], rs[t]);
    }
  }

  if (a.a2_out) {   // MLP nets stop here: the trunk output feeds the head directly
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.a2_out + (int64_t)row[t] * OPE_H + 16 * it + 4 * g) = act[t][it];
      }
    return;
  }
  // ---- gi = W_ih a2 + b_ih : 192 outputs, 4 tiles at a time ----
#pragma unroll
  for (int grp = 0; grp < 3; ++grp) {
    f32x4 o[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int it = 0; it < 4; ++it) o[t][it] = *reinterpret_cast<const f32x4*>(th + a.L.bih + 64 * grp + 16 * it + 4 * g);
    gemm64rt<4, RT>(th + a.L.wih + (int64_t)(64 * grp) * OPE_H, OPE_H, j, g, act, o);
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t]) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
          *reinterpret_cast<f32x4*>(a.gi + (int64_t)row[t] * (3 * OPE_H) + 64 * grp + 16 * it + 4 * g) = o[t][it];
      }
  }
}

template <int VEC, bool SAVE>
static int launch_trunk_vec(const TrunkFwdArgs& a, hipStream_t st) {
  // two row tiles per wave once there is enough work to fill the chip that way (OPE_TRUNK_RT=1|2 overrides, for A/B runs)
  static const int forced = getenv("OPE_TRUNK_RT") ? atoi(getenv("OPE_TRUNK_RT")) : 0;
  const bool two = forced ? forced == 2 : a.R >= 2 * 16 * 4 * 256;
  kprof_work(2.0 * a.R * ((double)a.D * OPE_H + OPE_H * OPE_H + (a.gi ? 3.0 * OPE_H * OPE_H : 0.0) + (a.head_out ? (double)OPE_H * a.head_dim : 0.0)));
  if (two) {
    OPE_LAUNCH((trunk_fwd_kernel<VEC, 2, SAVE>), dim3(ope_cdiv(ope_cdiv(a.R, 32), 4)), dim3(256), 0, st, a);
  } else {
    OPE_LAUNCH((trunk_fwd_kernel<VEC, 1, SAVE>), dim3(ope_cdiv(ope_cdiv(a.R, 16), 4)), dim3(256), 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("trunk_fwd1", VEC, two ? 2 : 1);
  return OPE_OK;
}

int launch_trunk_fwd(const TrunkFwdArgs& a, bool save, hipStream_t st, bool allow4) {
  if (a.R < 1 || a.D < 1) return OPE_EINVAL;
  if (allow4) {     // many rows of a width the LDS-resident kernel is built for (ope_trunk4.hip); 1 = not its launch
    const int rc4 = launch_trunk_fwd4_single(a, save, st);
    if (rc4 != 1) return rc4;
  }
  // default (OPE_TRUNK2 unset or 3): persistent workgroups with the weights in registers; 2: the non-persistent cooperative
  // form; 0: the one-wave-per-row-tile form below (A/B runs)
  static const int v2 = getenv("OPE_TRUNK2") ? atoi(getenv("OPE_TRUNK2")) : 3;
  if (a.head_out && (!a.a2_out || a.head_dim < 1 || a.head_dim > 16)) return OPE_EINVAL;
  // inputs wider than 384: the first-layer fragments alone are 128 VGPRs and the kernel spills ~100 registers, so those stay
  // on the re-streaming cooperative form. Up to 384 (96 VGPRs of fragments, fc2 / W_ih fragments re-fetched per tile) the
  // persistent kernel wins since its biases and LayerNorm parameters moved to LDS: MMM2 (D = 370) QMIX 1 308 -> 1 383 steps/s.
  static const int maxd3 = getenv("OPE_TRUNK3_MAXD") ? atoi(getenv("OPE_TRUNK3_MAXD")) : 384;
  if (a.tanh_act) return (a.D <= 384 && !a.head_out) ? launch_trunk_fwd3(a, save, st) : OPE_EINVAL;      // only trunk_fwd3 carries the tanh activation
  if (v2 == 3 && a.D <= maxd3 && a.D <= 512) return launch_trunk_fwd3(a, save, st);
  if (v2 && a.D <= 512) return launch_trunk_fwd2(a, save, st);
  if (a.head_out) {   // one-wave form: the head is a second launch
    TrunkFwdArgs b = a;
    b.head_out = nullptr;
    int rc = launch_trunk_fwd(b, save, st);
    if (rc) return rc;
    HeadFwdArgs hf;
    memset(&hf, 0, sizeof(hf));
    hf.R = a.R; hf.NB = a.R; hf.B = a.R; hf.N = 1; hf.T = 1; hf.A = a.head_dim; hf.theta0 = a.theta; hf.theta1 = a.theta; hf.L = a.L;
    hf.h0 = a.a2_out; hf.q_out = a.head_out; hf.no_ln = 1;
    return launch_head_fwd(hf, 1, st);
  }
  const int vec = ope_vec_of(a.D);
  if (save) {
    if (vec == 4) return launch_trunk_vec<4, true>(a, st);
    if (vec == 2) return launch_trunk_vec<2, true>(a, st);
    return launch_trunk_vec<1, true>(a, st);
  }
  if (vec == 4) return launch_trunk_vec<4, false>(a, st);
  if (vec == 2) return launch_trunk_vec<2, false>(a, st);
  return launch_trunk_vec<1, false>(a, st);
}

// ---------------------------------------------------------------------------------------------------------
// gru_fwd: h_t = GRU(gi_t, h_{t-1}) over the T+1 entries of every (agent, episode) row. Two kernel families, chosen by
// the number of rows in the launch (OPE_GRU = 4 | 1 forces one): ope_gru4.hip splits a row over 2 or 4 compute waves
// (latency-bound launches: QMIX at B = 32 has 512 rows), ope_gru1.hip runs one compute wave per row (more rows than
// SIMDs: the MATD3 actor at B = 128 has 1 280 per net).
// ---------------------------------------------------------------------------------------------------------
static int env_choice(const char* name, int a, int b) {
  const char* v = getenv(name);
  const int x = v ? atoi(v) : 0;
  return (x == a || x == b) ? x : 0;
}
int g_scan_family = env_choice("OPE_GRU", 1, 4);
int g_scan_waves = env_choice("OPE_GRU4_W", 2, 4);

int launch_gru_fwd(const GruFwdArgs& a, hipStream_t st) {
  if (a.nets < 1 || a.nets > 2 || a.NB < 1 || a.L < 1) return OPE_EINVAL;
  const int64_t rows = (int64_t)a.nets * a.NB;
  const int want = (a.family == 1 || a.family == 4) ? a.family : g_scan_family;
  const int kind = want ? want : (rows <= kGru4MaxRows ? 4 : 1);
  return kind == 4 ? launch_gru_fwd4(a, st) : launch_gru_fwd1(a, st);
}

// ---------------------------------------------------------------------------------------------------------
// head_fwd. One thread per data row (t, agent, b).  y = LN(h_t); q_a = W_q[a].y + b_q[a].
//   live  : q of the action taken (t < T) -> agent_q[t][b][agent]; greedy = argmax over available actions
//           (unavailable -> -1e10, first max wins; QMixPolicy.actions_from_q, QMixPolicy.py:102-174, util.py:297-302)
//   target: q_tgt at the live greedy action (double-Q, qmix.py:138-146) or plain max (qmix.py:148), for t >= 1
//           -> agent_nq[t-1][b][agent]
// Head weights of both nets sit in LDS (broadcast reads). xhat of the live net is kept for the backward pass.
// ---------------------------------------------------------------------------------------------------------
template <int MODE>  // 0: training heads (live+target), 1: plain q output for ope_agent_forward
__global__ void __launch_bounds__(256) head_fwd_kernel(HeadFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int A = a.A;
  // LDS: [Wq live A*64][bq live A][gam 64][bet 64] then the same for the target net
  const int per = A * OPE_H + ope_round4_dev(A) + 2 * OPE_H;
  const int nets = (MODE == 0) ? 2 : 1;
  for (int i = threadIdx.x; i < per * nets; i += blockDim.x) {
    const int net = i / per, o = i - net * per;
    const float* th = net == 0 ? a.theta0 : a.theta1;
    float v;
    if (o < A * OPE_H) v = th[a.L.q_w + o];
    else if (o < A * OPE_H + ope_round4_dev(A)) v = (o - A * OPE_H < A) ? th[a.L.q_b + (o - A * OPE_H)] : 0.f;
    else if (a.no_ln) v = 0.f;
    else if (o < A * OPE_H + ope_round4_dev(A) + OPE_H) v = th[a.L.lno_w + (o - A * OPE_H - ope_round4_dev(A))];
    else v = th[a.L.lno_b + (o - A * OPE_H - ope_round4_dev(A) - OPE_H)];
    sm[i] = v;
  }
  __syncthreads();
  const int64_t r = a.r_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.R) return;
  const int NB = a.NB;
  const int t = (int)(r / NB);
  const int rowi = (int)(r - (int64_t)t * NB);  // agent*B + b
  const int agent = rowi / a.B, b = rowi - agent * a.B;

  float y[OPE_H];
  if (a.no_ln) {
#pragma unroll
    for (int k = 0; k < OPE_H; k += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.h0 + r * OPE_H + k);
      y[k] = v[0]; y[k + 1] = v[1]; y[k + 2] = v[2]; y[k + 3] = v[3];
    }
  } else {
    ln64_thread(a.h0 + r * OPE_H, sm + A * OPE_H + ope_round4_dev(A), sm + A * OPE_H + ope_round4_dev(A) + OPE_H, y,
                a.xhat_o ? a.xhat_o + r * OPE_H : nullptr, a.rstd_o ? a.rstd_o + r : nullptr);
  }

  if (MODE == 1) {
    for (int k = 0; k < A; ++k) {
      float q = sm[A * OPE_H + k];
#pragma unroll
      for (int f = 0; f < OPE_H; ++f) q = fmaf(sm[k * OPE_H + f], y[f], q);
      a.q_out[r * A + k] = q;
    }
    return;
  }

  // chosen action = first max of the one-hot row (QMixPolicy.q_values_from_actions, QMixPolicy.py:69-93)
  int chosen = 0;
  if (t < a.T) {
    const float* ac = a.acts + r * A;
    float best = ac[0];
    for (int k = 1; k < A; ++k) {
      const float v = ac[k];
      if (v > best) { best = v; chosen = k; }
    }
    a.act_idx[r] = chosen;
  }
  const float* av = a.avail ? a.avail + r * A : nullptr;
  float qc = 0.f, best = 0.f;
  int greedy = 0;
  for (int k = 0; k < A; ++k) {
    float q = sm[A * OPE_H + k];
#pragma unroll
    for (int f = 0; f < OPE_H; ++f) q = fmaf(sm[k * OPE_H + f], y[f], q);
    if (k == chosen) qc = q;
    const float qm = (av && av[k] == 0.f) ? -1e10f : q;
    if (k == 0 || qm > best) { best = qm; greedy = k; }
  }
  if (t < a.T) a.agent_q[((int64_t)t * a.B + b) * a.N + agent] = qc;
  if (a.q_all) {
    for (int k = 0; k < A; ++k) {  // debug/test output of the full live q row
      float q = sm[A * OPE_H + k];
#pragma unroll
      for (int f = 0; f < OPE_H; ++f) q = fmaf(sm[k * OPE_H + f], y[f], q);
      a.q_all[r * A + k] = q;
    }
  }
  if (t >= 1) {
    const float* s1 = sm + per;
    if (a.no_ln) {
#pragma unroll
      for (int k = 0; k < OPE_H; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.h1 + r * OPE_H + k);
        y[k] = v[0]; y[k + 1] = v[1]; y[k + 2] = v[2]; y[k + 3] = v[3];
      }
    } else {
      ln64_thread(a.h1 + r * OPE_H, s1 + A * OPE_H + ope_round4_dev(A), s1 + A * OPE_H + ope_round4_dev(A) + OPE_H, y, nullptr, nullptr);
    }
    float tq;
    if (a.double_q) {
      tq = s1[A * OPE_H + greedy];
#pragma unroll
      for (int f = 0; f < OPE_H; ++f) tq = fmaf(s1[greedy * OPE_H + f], y[f], tq);
    } else {
      tq = 0.f;
      for (int k = 0; k < A; ++k) {
        float q = s1[A * OPE_H + k];
#pragma unroll
        for (int f = 0; f < OPE_H; ++f) q = fmaf(s1[k * OPE_H + f], y[f], q);
        if (a.target_mask_avail && av && av[k] == 0.f) q = -1e10f;
        if (k == 0 || q > tq) tq = q;
      }
    }
    a.agent_nq[((int64_t)(t - 1) * a.B + b) * a.N + agent] = tq;
  }
}

int launch_head_fwd(const HeadFwdArgs& a, int mode, hipStream_t st) {
  if (a.R < 1 || a.A < 1 || a.A > 1024) return OPE_EINVAL;
  const int per = a.A * OPE_H + ope_round4(a.A) + 2 * OPE_H;
  const int nets = mode == 0 ? 2 : 1;
  const size_t lds = (size_t)per * nets * sizeof(float);
  if (lds > 64 * 1024) return OPE_EINVAL;
  if (a.r_begin < 0 || a.r_begin >= a.R) return OPE_EINVAL;
  if (mode == 0 && a.A <= 32) return launch_head_fwd_mfma(a, st);   // ope_head.hip; wider action spaces: thread per row below
  const int blocks = ope_cdiv(a.R - a.r_begin, 256);
  note_launch("head_fwd_rows", mode);
  if (mode == 0)
    OPE_LAUNCH(head_fwd_kernel<0>, dim3(blocks), dim3(256), lds, st, a);
  else
    OPE_LAUNCH(head_fwd_kernel<1>, dim3(blocks), dim3(256), lds, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

}  // namespace ope
