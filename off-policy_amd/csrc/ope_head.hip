// Q-head kernels of the agent network on the MFMA row-tile layout (16 data rows per wave, 4 lanes per row).
//
// Replaces (reference): ACTLayer.forward (offpolicy/algorithms/utils/act.py:21-37) on top of RNNBase's output LayerNorm
// (utils/rnn.py:33-47), QMixPolicy.q_values_from_actions (QMixPolicy.py:69-93), the double-Q / plain target selection
// (qmix.py:138-148, QMixPolicy.actions_from_q QMixPolicy.py:102-174, avail_choose util.py:297-302), and their autograd.
//
// The first version ran one THREAD per data row (LayerNorm of 64 values, 14 dot products of 64, the target row again):
// ~1 500 dependent instructions per wave and only 600 waves for 38 656 rows, i.e. less than one wave per SIMD doing a
// long serial chain (23 us forward, 13 us backward at 3s5z). Here a wave owns 16 rows in the lane convention of the
// trunk kernels -- lane (j, g) = (lane & 15, lane >> 4) holds features 16c + 4g + r (c, r < 4) of row j -- so that
//   * a row is four 16-byte loads per lane and its LayerNorm statistics two 4-lane sums,
//   * q = W_q y + b_q is 16 f32 MFMAs per 16-action tile (W_q rows as the A operand, zero beyond A), after which lane
//     (j, g) holds q[16 it + 4g + r] of row j: the per-row argmax / one-hot pick / max are a local scan over <= 8
//     candidates and two 4-lane exchanges with "greater value, then lower index" as the order (first max wins, as in
//     torch.max and the reference's loops),
//   * 2 416 waves (3s5z) instead of 604.
#include <stdlib.h>

#include "ope_rowops.h"

namespace ope {
namespace {

template <int NT>
__global__ void __launch_bounds__(256) head_fwd_mfma_kernel(HeadFwdArgs a) {
  if ((int)blockIdx.x >= a.main_blocks) {   // passengers: weight transposes for the backward kernels of this step
    transpose4_element(a.side, ((int)blockIdx.x - a.main_blocks) * 256 + (int)threadIdx.x);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int A = a.A;
  const int64_t r_raw = a.r_begin + ((int64_t)blockIdx.x * 4 + wave) * 16 + j;
  const bool valid = r_raw < a.R;
  const int64_t r = valid ? r_raw : a.R - 1;
  const int t = (int)(r / a.NB);
  const int rowi = (int)(r - (int64_t)t * a.NB);  // agent*B + b
  const int agent = rowi / a.B, b = rowi - agent * a.B;
  const bool first = valid && g == 0;             // the lane that stores a row's scalars

  // ---- live net: q of every action
  f32x4 y[4], q[NT];
  ln_row16(a.h0 + r * OPE_H, a.theta0, a.L.lno_w, a.L.lno_b, a.no_ln != 0, g, y,
           (a.xhat_o && valid) ? a.xhat_o + r * OPE_H : nullptr, (a.rstd_o && valid) ? a.rstd_o + r : nullptr);
  q_tiles<NT>(a.theta0, a.L, A, j, g, y, q);
  if (a.q_all && valid) {
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (16 * it + 4 * g + rr < A) a.q_all[r * A + 16 * it + 4 * g + rr] = q[it][rr];
  }

  // ---- chosen action = first max of the one-hot row, its q; greedy = first max over the available actions
  float cv = kNegInf, gv = kNegInf;
  int chosen = 1 << 30, greedy = 1 << 30;
  float avl[NT][4];
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int k = 16 * it + 4 * g + rr;
      avl[it][rr] = 1.f;
      if (k < A) {
        if (t < a.T) {
          const float v = a.acts[r * A + k];
          if (v > cv) { cv = v; chosen = k; }     // ascending k within the lane: strict > keeps the first
        }
        if (a.avail) avl[it][rr] = a.avail[r * A + k];
        const float qm = (avl[it][rr] == 0.f) ? -1e10f : q[it][rr];
        if (qm > gv) { gv = qm; greedy = k; }
      }
    }
  row_argmax4(gv, greedy);
  if (t < a.T) {
    row_argmax4(cv, chosen);
    float qc = 0.f;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) qc += (16 * it + 4 * g + rr == chosen) ? q[it][rr] : 0.f;
    qc = rowsum4(qc);
    if (first) {
      a.act_idx[r] = chosen;
      a.agent_q[((int64_t)t * a.B + b) * a.N + agent] = qc;
    }
  }

  // ---- target net at the same row: q at the live net's greedy action (double Q) or the (masked) maximum. Evaluated for
  // every row (the MFMAs need all lanes of the wave; rows of one tile may straddle t = 0 / t = 1), stored for t >= 1.
  {
    f32x4 q1[NT];
    ln_row16(a.h1 + r * OPE_H, a.theta1, a.L.lno_w, a.L.lno_b, a.no_ln != 0, g, y, nullptr, nullptr);
    q_tiles<NT>(a.theta1, a.L, A, j, g, y, q1);
    float tq;
    if (a.double_q) {
      tq = 0.f;
#pragma unroll
      for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) tq += (16 * it + 4 * g + rr == greedy) ? q1[it][rr] : 0.f;
      tq = rowsum4(tq);
    } else {
      tq = kNegInf;
#pragma unroll
      for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (16 * it + 4 * g + rr < A) {
            const float v = (a.target_mask_avail && avl[it][rr] == 0.f) ? -1e10f : q1[it][rr];
            tq = fmaxf(tq, v);
          }
      tq = fmaxf(tq, __shfl_xor(tq, 16, 64));
      tq = fmaxf(tq, __shfl_xor(tq, 32, 64));
    }
    if (first && t >= 1) a.agent_nq[((int64_t)(t - 1) * a.B + b) * a.N + agent] = tq;
  }
}

// d agent_q -> dq at the chosen action (one-hot row for the head's weight gradient) -> LayerNorm adjoint -> dh_out
__global__ void __launch_bounds__(256) head_bwd_rows_kernel(HeadBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t r = ((int64_t)blockIdx.x * 4 + wave) * 16 + j;
  if (r >= a.R) return;                            // (whole rows drop out together: the 4-lane sums below stay complete)
  const int t = (int)(r / a.NB);
  const int rowi = (int)(r - (int64_t)t * a.NB);
  const int agent = rowi / a.B, b = rowi - agent * a.B;
  const float dq = a.d_agent_q[((int64_t)t * a.B + b) * a.N + agent];
  const int act = a.act_idx[r];
  const int A4 = ope_round4_dev(a.A);
  for (int k0 = 4 * g; k0 < A4; k0 += 16)
    *reinterpret_cast<f32x4*>(a.dqoh + r * A4 + k0) =
        f32x4{k0 == act ? dq : 0.f, k0 + 1 == act ? dq : 0.f, k0 + 2 == act ? dq : 0.f, k0 + 3 == act ? dq : 0.f};
  const float* __restrict__ wq = a.theta + a.L.q_w + (int64_t)act * OPE_H + 4 * g;
  f32x4 d[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(wq + 16 * c);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) d[c][rr] = dq * w[rr];
  }
  if (!a.no_ln) {
    const float rstd = a.rstd_o[r];
    f32x4 xh[4];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      xh[c] = *reinterpret_cast<const f32x4*>(a.xhat_o + r * OPE_H + 16 * c + 4 * g);
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.theta + a.L.lno_w + 16 * c + 4 * g);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        d[c][rr] *= gm[rr];
        m1 += d[c][rr];
        m2 = fmaf(d[c][rr], xh[c][rr], m2);
      }
    }
    m1 = rowsum4(m1) * (1.0f / OPE_H);
    m2 = rowsum4(m2) * (1.0f / OPE_H);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) d[c][rr] = rstd * (d[c][rr] - m1 - xh[c][rr] * m2);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(a.dh_out + r * OPE_H + 16 * c + 4 * g) = d[c];
}

}  // namespace

// training heads (live + target) for up to 32 actions; wider action spaces stay on the thread-per-row kernel
int launch_head_fwd_mfma(const HeadFwdArgs& a0, hipStream_t st) {
  HeadFwdArgs a = a0;
  a.main_blocks = (int)ope_cdiv(a.R - a.r_begin, 64);
  const int blocks = a.main_blocks + (a.side.total > 0 ? ope_cdiv(a.side.total, 256) : 0);
  kprof_work(2.0 * 2.0 * (double)(a.R - a.r_begin) * OPE_H * a.A);           // live + target q of every action
  if (a.A <= 16)
    OPE_LAUNCH(head_fwd_mfma_kernel<1>, dim3(blocks), dim3(256), 0, st, a);
  else
    OPE_LAUNCH(head_fwd_mfma_kernel<2>, dim3(blocks), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("head_fwd_mfma", a.A <= 16 ? 1 : 2);
  return OPE_OK;
}

int launch_head_bwd_rows(const HeadBwdArgs& a, hipStream_t st) {
  kprof_work(0.0, (double)a.R * (2.0 * OPE_H + ope_round4(a.A) + 2.0) * 4.0);   // row-wise adjoint: reads xhat, writes dh_out + the one-hot dq row
  OPE_LAUNCH(head_bwd_rows_kernel, dim3((int)ope_cdiv(a.R, 64)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("head_bwd_rows");
  return OPE_OK;
}

}  // namespace ope
