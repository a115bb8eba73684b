// Second hidden block of the agent network's MLP base (args.layer_N = 2) and the GRU input projection behind it.
//
// Replaces (reference): the second iteration of `for i in range(self._layer_N): x = self.fc2[i](x)` in MLPLayer.forward
// (offpolicy/algorithms/utils/mlp.py:25-29; fc2 = layer_N clones of Sequential(Linear(64, 64), ReLU, LayerNorm(64)), mlp.py:21-23) followed by
// the input half of nn.GRU (utils/rnn.py:19-23), and their autograd (loss.backward(), qmix.py:191).
//
// The trunk kernels (ope_trunk2.hip, ope_agent_fwd.hip) are specialised to ONE block behind fc1 and end either in the GRU projection or --
// for the non-recurrent nets -- at the block's output a2. For layer_N = 2 they run in that second mode and these two kernels continue:
//   block_fwd   a3 = LN(ReLU(W_b a2 + b_b));  gi = W_ih a3 + b_ih          (saves xhat3, 1/std, the ReLU mask for the adjoint)
//   block_bwd   da3 = W_ih^T dgi;  dz3 = ReLU' . LN'(da3);  da2 = W_b^T dz3  (dz3 feeds the weight-gradient launch, da2 the trunk adjoint)
// A non-default network shape: written for correctness in the lane convention of the other row kernels (a wave = 16 rows, weights as MFMA
// A operands straight from L2), not tuned.
#include "ope_agent.h"

namespace ope {
namespace {

__global__ void __launch_bounds__(256) block_fwd_kernel(BlockFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile * 16 >= a.R) return;                         // (whole waves; no barrier in this kernel)
  const int row_raw = tile * 16 + j;
  const bool valid = row_raw < a.R;
  const int row = valid ? row_raw : a.R - 1;
  const float* __restrict__ th = a.theta;
  const AgentLayout& L = a.L;
  f32x4 act[4], acc[4];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    act[ft] = *reinterpret_cast<const f32x4*>(a.x + (int64_t)row * OPE_H + 16 * ft + 4 * g);
    acc[ft] = *reinterpret_cast<const f32x4*>(th + L.fc2b_b + 16 * ft + 4 * g);
  }
  gemm64<4>(th + L.fc2b_w, OPE_H, j, g, act, acc);
  float rstd;
  uint32_t mbits;
  if (a.xhat3) {
    relu_ln64<true>(acc, th + L.ln2b_w, th + L.ln2b_b, g, act, &rstd, &mbits);
    // the four lanes' 16-bit ReLU masks -> one 64-bit row mask (bit f = feature f), as store_mask_rstd does
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint32_t nib = (mbits >> (4 * it)) & 0xF;
      const int f0 = 16 * it + 4 * g;
      if (it < 2) lo |= nib << f0; else hi |= nib << (f0 - 32);
    }
    lo |= __shfl_xor((int)lo, 16, 64); hi |= __shfl_xor((int)hi, 16, 64);
    lo |= __shfl_xor((int)lo, 32, 64); hi |= __shfl_xor((int)hi, 32, 64);
    if (valid) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.xhat3 + (int64_t)row * OPE_H + 16 * it + 4 * g) = acc[it];
      if (g == 0) {
        a.mask3[row] = ((uint64_t)hi << 32) | lo;
        a.rstd3[row] = rstd;
      }
    }
  } else {
    relu_ln64<false>(acc, th + L.ln2b_w, th + L.ln2b_b, g, act, &rstd, &mbits);
  }
  // gi = W_ih a3 + b_ih: twelve 16-gate tiles in three groups of four
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    f32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) o[it] = *reinterpret_cast<const f32x4*>(th + L.bih + 64 * u + 16 * it + 4 * g);
    gemm64<4>(th + L.wih + (int64_t)(64 * u) * OPE_H, OPE_H, j, g, act, o);
    if (valid) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.gi + (int64_t)row * (3 * OPE_H) + 64 * u + 16 * it + 4 * g) = o[it];
    }
  }
}

__global__ void __launch_bounds__(256) block_bwd_kernel(BlockBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile * 16 >= a.R) return;
  const int row_raw = tile * 16 + j;
  const bool valid = row_raw < a.R;
  const int row = valid ? row_raw : a.R - 1;
  const float* __restrict__ th = a.theta;
  // da3[i] = sum_k W_ih[k][i] dgi[k]: A operand = rows of W_ih^T ([64][192]), B operand = this row's 192 gate adjoints
  f32x4 d[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) d[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const f32x4 dg = *reinterpret_cast<const f32x4*>(a.dgi + (int64_t)row * (3 * OPE_H) + 16 * c + 4 * g);
    f32x4 w[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) w[it] = *reinterpret_cast<const f32x4*>(a.wihT + (int64_t)(16 * it + j) * (3 * OPE_H) + 16 * c + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int it = 0; it < 4; ++it) d[it] = mfma16(w[it][r], dg[r], d[it]);
  }
  // LayerNorm + ReLU adjoint over the row's 64 features (d[it][r] = feature 16 it + 4 g + r):
  //   d *= gamma;  m1 = mean(d), m2 = mean(d xhat);  d = relu_bit ? rstd (d - m1 - xhat m2) : 0
  const float rstd = a.rstd3[row];
  const uint64_t mask = a.mask3[row];
  f32x4 xh[4];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    xh[it] = *reinterpret_cast<const f32x4*>(a.xhat3 + (int64_t)row * OPE_H + 16 * it + 4 * g);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(th + a.L.ln2b_w + 16 * it + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[it][r] *= gm[r];
      m1 += d[it][r];
      m2 = fmaf(d[it][r], xh[it][r], m2);
    }
  }
  m1 = rowsum4(m1) * (1.0f / OPE_H);
  m2 = rowsum4(m2) * (1.0f / OPE_H);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool on = (mask >> (16 * it + 4 * g + r)) & 1;
      d[it][r] = on ? rstd * (d[it][r] - m1 - xh[it][r] * m2) : 0.f;
    }
    if (valid) *reinterpret_cast<f32x4*>(a.dz3 + (int64_t)row * OPE_H + 16 * it + 4 * g) = d[it];
  }
  // da2[i] = sum_k W_b[k][i] dz3[k]: A operand = rows of W_b^T
  f32x4 e[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) e[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm64<4>(a.fc2bT, OPE_H, j, g, d, e);
  if (valid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.da2 + (int64_t)row * OPE_H + 16 * it + 4 * g) = e[it];
  }
}

}  // namespace

int launch_block_fwd(const BlockFwdArgs& a, hipStream_t st) {
  if (a.R < 1 || !a.x || !a.gi || a.L.layer_N != 2 || (a.xhat3 && (!a.rstd3 || !a.mask3))) return OPE_EINVAL;
  kprof_work(2.0 * a.R * (double)(OPE_H * OPE_H + 3 * OPE_H * OPE_H));
  OPE_LAUNCH(block_fwd_kernel, dim3(ope_cdiv(ope_cdiv(a.R, 16), 4)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("block_fwd", a.xhat3 ? 1 : 0);
  return OPE_OK;
}

int launch_block_bwd(const BlockBwdArgs& a, hipStream_t st) {
  if (a.R < 1 || !a.dgi || !a.wihT || !a.fc2bT || !a.xhat3 || !a.rstd3 || !a.mask3 || !a.dz3 || !a.da2 || a.L.layer_N != 2) return OPE_EINVAL;
  kprof_work(2.0 * a.R * (double)(3 * OPE_H * OPE_H + OPE_H * OPE_H));
  OPE_LAUNCH(block_bwd_kernel, dim3(ope_cdiv(ope_cdiv(a.R, 16), 4)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("block_bwd");
  return OPE_OK;
}

}  // namespace ope
