// Device helpers shared by the (t, agent, b)-row and (t, b)-row kernels: the q head on the MFMA row-tile layout (ope_head.hip), the mixer's
// hyper-network addressing, the TD / loss arithmetic of one row (ope_mixer.hip) -- and the fused chain kernel that does all of them in one
// launch (ope_chain.hip).
#pragma once
#include "ope_mixer.h"

namespace ope {

constexpr float kNegInf = -3.0e38f;

// LayerNorm (or copy, no_ln) of this lane's 16 features of row `hrow`; optionally saves xhat / rstd (g == 0 stores rstd)
__device__ __forceinline__ void ln_row16(const float* __restrict__ hrow, const float* __restrict__ th, int lno_w, int lno_b, bool no_ln,
                                         int g, f32x4 (&y)[4], float* xhat_out, float* rstd_out) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    y[c] = *reinterpret_cast<const f32x4*>(hrow + 16 * c + 4 * g);
    s += (y[c][0] + y[c][1]) + (y[c][2] + y[c][3]);
  }
  if (no_ln) return;
  const float mu = rowsum4(s) * (1.0f / OPE_H);
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = y[c][r] - mu;
      v = fmaf(d, d, v);
    }
  const float rstd = 1.0f / sqrtf(rowsum4(v) * (1.0f / OPE_H) + OPE_LN_EPS);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(th + lno_w + 16 * c + 4 * g);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(th + lno_b + 16 * c + 4 * g);
    f32x4 xh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[r] = (y[c][r] - mu) * rstd;
      y[c][r] = fmaf(xh[r], gm[r], bt[r]);
    }
    if (xhat_out) *reinterpret_cast<f32x4*>(xhat_out + 16 * c + 4 * g) = xh;
  }
  if (rstd_out && g == 0) *rstd_out = rstd;
}

// q[16 it + 4g + r] of row j for it < NT: bias + W_q y on the matrix pipe
template <int NT>
__device__ __forceinline__ void q_tiles(const float* __restrict__ th, const AgentLayout& L, int A, int j, int g, const f32x4 (&y)[4],
                                        f32x4 (&q)[NT]) {
#pragma unroll
  for (int it = 0; it < NT; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) q[it][r] = (16 * it + 4 * g + r < A) ? th[L.q_b + 16 * it + 4 * g + r] : 0.f;
    const int m = 16 * it + j;                       // weight row of the A operand held by this lane
    const float* __restrict__ wrow = th + L.q_w + (int64_t)(m < A ? m : A - 1) * OPE_H + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 w = *reinterpret_cast<const f32x4*>(wrow + 16 * c);
      if (m >= A) w = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) q[it] = mfma16(w[r], y[c][r], q[it]);
    }
  }
}

// (value, index) maximum over the 4 lanes of a row: greater value wins, equal values -> lower index (first max)
__device__ __forceinline__ void row_argmax4(float& v, int& k) {
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float ov = __shfl_xor(v, off, 64);
    const int ok = __shfl_xor(k, off, 64);
    if (ov > v || (ov == v && ok < k)) { v = ov; k = ok; }
  }
}


__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// first hyper-layers: tiles 0-3 hyper_w1.0, 4-7 hyper_w2.0, 8-11 hyper_b2.0, 12-13 hyper_b1
__device__ __forceinline__ const float* stageA_row(const float* th, const MixerLayout& L, int S, int it, int i) {
  if (it < 4) return th + L.w1a_w + (int64_t)(16 * it + i) * S;
  if (it < 8) return th + L.w2a_w + (int64_t)(16 * (it - 4) + i) * S;
  if (it < 12) return th + L.b2a_w + (int64_t)(16 * (it - 8) + i) * S;
  return th + L.b1_w + (int64_t)(16 * (it - 12) + i) * S;
}
__device__ __forceinline__ const float* stageA_bias(const float* th, const MixerLayout& L, int it) {
  if (it < 4) return th + L.w1a_b + 16 * it;
  if (it < 8) return th + L.w2a_b + 16 * (it - 4);
  if (it < 12) return th + L.b2a_b + 16 * (it - 8);
  return th + L.b1_b + 16 * (it - 12);
}


// ---------------------------------------------------------------------------------------------------------
// TD target, masked error, loss terms and dQ_tot for one (t,b) row (qmix.py:158-176). The loss is NOT divided by
// the mask count here: gradients are those of the un-normalised sum (ope.h), so data-parallel ranks can add them.
// ---------------------------------------------------------------------------------------------------------
struct TdOut { float err, keep, lossel, dq; };
__device__ __forceinline__ TdOut td_row(const TdArgs& d, int t, int b, float qtot, float nqtot) {
  TdOut o;
  const float bad = t == 0 ? 0.f : d.dones_env[(int64_t)(t - 1) * d.B + b];
  o.keep = 1.0f - bad;
  const float rew = d.rewards[((int64_t)t * d.N + 0) * d.B + b];   // agents share the reward: agent 0 (qmix.py:159)
  const float den = d.dones_env[(int64_t)t * d.B + b];
  const float target = rew + (1.0f - den) * d.gamma * nqtot;
  const float e = (qtot - target) * o.keep;
  o.err = e;
  const float wgt = d.per_weights ? d.per_weights[b] : 1.0f;
  float fe, dfe;
  if (d.use_huber) {
    const float ae = fabsf(e), dl = d.huber_delta;
    if (ae <= dl) { fe = e * e * 0.5f; dfe = e; }
    else { fe = dl * (ae - dl * 0.5f); dfe = dl * sgn(e); }
  } else {
    fe = e * e;
    dfe = 2.0f * e;
  }
  o.lossel = wgt * fe;
  o.dq = wgt * dfe * o.keep;
  return o;
}

// sum over the 16 rows (lanes j) of a wave-tile; result valid in every lane
__device__ __forceinline__ float tilesum16(float x) {
  x += __shfl_xor(x, 1, 64);
  x += __shfl_xor(x, 2, 64);
  x += __shfl_xor(x, 4, 64);
  x += __shfl_xor(x, 8, 64);
  return x;
}


}  // namespace ope
