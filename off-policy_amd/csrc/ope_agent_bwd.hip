// Agent q-network backward kernels. The reference gets these from torch autograd over AgentQFunction
// (loss.backward(), qmix.py:191); here every op's adjoint is written out:
//
//   head_bwd  : d agent_q -> dq at the chosen action -> LN backward -> dh_out[t]             (thread per row)
//   gru_bwd   : BPTT over t = T-1..0 (ope_gru4.hip / ope_gru1.hip)
//   trunk_bwd : dgi -> W_ih^T -> LN2 bwd -> ReLU -> fc2^T -> LN1 bwd -> ReLU -> dz1           (f32 MFMA chain)
//
// Weight gradients are K-reductions over all rows and are done by ope_wgrad.hip from the per-row adjoints
// stored here (dz1, dz2, dgi, dghn, dqoh).
#include <stdlib.h>

#include "ope_agent.h"

namespace ope {

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_bwd_kernel(HeadBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [Wq A*64][gamma 64]
  const int A = a.A;
  for (int i = threadIdx.x; i < A * OPE_H + OPE_H; i += blockDim.x)
    sm[i] = (i < A * OPE_H) ? a.theta[a.L.q_w + i] : (a.no_ln ? 1.0f : a.theta[a.L.lno_w + (i - A * OPE_H)]);
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.R) return;
  const int t = (int)(r / a.NB);
  const int rowi = (int)(r - (int64_t)t * a.NB);
  const int agent = rowi / a.B, b = rowi - agent * a.B;
  const float dq = a.d_agent_q[((int64_t)t * a.B + b) * a.N + agent];
  const int act = a.act_idx[r];
  const float* wq = sm + act * OPE_H;
  const float* gam = sm + A * OPE_H;
  const int A4 = ope_round4_dev(A);
  for (int k = 0; k < A4; ++k) a.dqoh[r * A4 + k] = (k == act) ? dq : 0.f;
  if (a.no_ln) {   // no LayerNorm between trunk and head: dh_out = dq * Wq[act]
#pragma unroll
    for (int k = 0; k < OPE_H; k += 4)
      *reinterpret_cast<f32x4*>(a.dh_out + r * OPE_H + k) = f32x4{dq * wq[k], dq * wq[k + 1], dq * wq[k + 2], dq * wq[k + 3]};
    return;
  }
  const float rstd = a.rstd_o[r];
  float xh[OPE_H], dyh[OPE_H];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int k = 0; k < OPE_H; k += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a.xhat_o + r * OPE_H + k);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      xh[k + q] = x[q];
      const float d = dq * wq[k + q] * gam[k + q];
      dyh[k + q] = d;
      m1 += d;
      m2 = fmaf(d, x[q], m2);
    }
  }
  m1 *= (1.0f / OPE_H);
  m2 *= (1.0f / OPE_H);
#pragma unroll
  for (int k = 0; k < OPE_H; k += 4) {
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = rstd * (dyh[k + q] - m1 - xh[k + q] * m2);
    *reinterpret_cast<f32x4*>(a.dh_out + r * OPE_H + k) = o;
  }
}

int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  static const int rows16 = getenv("OPE_HEAD") ? atoi(getenv("OPE_HEAD")) : 1;   // 0: thread-per-row kernel (A/B runs)
  if (rows16) return launch_head_bwd_rows(a, st);
  const size_t lds = (size_t)(a.A * OPE_H + OPE_H) * sizeof(float);
  if (lds > 64 * 1024) return OPE_EINVAL;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(ope_cdiv(a.R, 256)), dim3(256), lds, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// gru_bwd: BPTT over t = T-1 .. t_lo; kernels in ope_gru4.hip / ope_gru1.hip, chosen like launch_gru_fwd.
// ---------------------------------------------------------------------------------------------------------
int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st) {
  if (a.NB < 1 || a.T < 1 || a.t_lo < 0 || a.t_lo >= a.T) return OPE_EINVAL;
  const int kind = g_scan_family ? g_scan_family : (a.NB <= kGru4MaxRows ? 4 : 1);
  return kind == 4 ? launch_gru_bwd4(a, st) : launch_gru_bwd1(a, st);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm + ReLU adjoint on the transposed-chain layout: given dy (w.r.t. LN output), xhat, rstd, gamma, mask:
//   dyh = dy*gamma ; dr = rstd (dyh - mean(dyh) - xhat mean(dyh xhat)) ; dz = dr * (z > 0)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ln_relu_bwd64(f32x4 (&d)[4], const float* __restrict__ xhat_row, float rstd,
                                              const float* __restrict__ gam, uint64_t mask, int g) {
  f32x4 xh[4];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    xh[it] = *reinterpret_cast<const f32x4*>(xhat_row + 16 * it + 4 * g);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 16 * it + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = d[it][r] * gm[r];
      d[it][r] = v;
      m1 += v;
      m2 = fmaf(v, xh[it][r], m2);
    }
  }
  m1 = rowsum4(m1) * (1.0f / OPE_H);
  m2 = rowsum4(m2) * (1.0f / OPE_H);
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * it + 4 * g + r;
      const float v = rstd * (d[it][r] - m1 - xh[it][r] * m2);
      d[it][r] = ((mask >> f) & 1ull) ? v : 0.f;
    }
}

__global__ void __launch_bounds__(256) trunk_bwd_kernel(TrunkBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  if (row0 >= a.R) return;
  const int row = row0 + j;
  const bool valid = row < a.R;
  const int64_t rr = valid ? row : row0;
  const float* wihT = a.thetaT;                       // [64][192]
  const float* fc2T = a.thetaT + OPE_H * 3 * OPE_H;   // [64][64]

  // da2 = W_ih^T dgi   (K = 192), or given directly for MLP nets
  f32x4 d[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) d[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.da2_in) {
#pragma unroll
    for (int it = 0; it < 4; ++it) d[it] = *reinterpret_cast<const f32x4*>(a.da2_in + rr * OPE_H + 16 * it + 4 * g);
  } else if (a.dout) {   // da2 = dout W_head, a handful of FMAs per feature (small heads)
    for (int k = 0; k < a.hdim; ++k) {
      const float dk = a.dout[rr * a.ldk + k];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.theta + a.L.q_w + (int64_t)k * OPE_H + 16 * it + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[it][r] = fmaf(dk, w[r], d[it][r]);
      }
    }
  }
  if (a.dgi) {
    // the whole dgi row slice of this lane (12 x 16 B) and the LayerNorm operands are requested up front: at ~2 waves per
    // SIMD the kernel is bound by load latency, so it is paid once instead of once per 16-column chunk
    const float* grow = a.dgi + rr * (3 * OPE_H);
    f32x4 bv[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) bv[c] = *reinterpret_cast<const f32x4*>(grow + 16 * c + 4 * g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      if (!valid) bv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wihT + (int64_t)(16 * it + j) * (3 * OPE_H) + 16 * c + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[it] = mfma16(w[r], bv[c][r], d[it]);
      }
    }
  }
  ln_relu_bwd64(d, a.xhat2 + rr * OPE_H, a.rstd2[rr], a.theta + a.L.ln2_w, a.mask2[rr], g);
  if (valid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.dz2 + (int64_t)row * OPE_H + 16 * it + 4 * g) = d[it];
  }
  // da1 = fc2^T dz2
  f32x4 e[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) e[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm64<4>(fc2T, OPE_H, j, g, d, e);
  ln_relu_bwd64(e, a.xhat1 + rr * OPE_H, a.rstd1[rr], a.theta + a.L.ln1_w, a.mask1[rr], g);
  if (valid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(a.dz1 + (int64_t)row * OPE_H + 16 * it + 4 * g) = e[it];
  }
}

int launch_trunk_bwd(const TrunkBwdArgs& a, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  static const int v3 = getenv("OPE_TRUNKB3") ? atoi(getenv("OPE_TRUNKB3")) : 1;
  if (v3) return launch_trunk_bwd3(a, st);
  hipLaunchKernelGGL(trunk_bwd_kernel, dim3(ope_cdiv(ope_cdiv(a.R, 16), 4)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Transposed copies of the matrices the backward chains read column-wise: dst[c][r] = src[r][c].
// ---------------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int c = i / rows, r = i - c * rows;  // dst index i = c*rows + r  (coalesced writes)
  dst[i] = src[(int64_t)r * cols + c];
}

int launch_transpose(const float* src, int rows, int cols, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(ope_cdiv((int64_t)rows * cols, 256)), dim3(256), 0, st, src, rows, cols, dst);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int launch_transpose_weights(const float* theta, const AgentLayout& L, float* thetaT, hipStream_t st) {
  int rc = launch_transpose(theta + L.wih, 3 * OPE_H, OPE_H, thetaT, st);
  if (rc) return rc;
  return launch_transpose(theta + L.fc2_w, OPE_H, OPE_H, thetaT + OPE_H * 3 * OPE_H, st);
}

}  // namespace ope
