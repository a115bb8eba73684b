// Agent q-network backward kernels. The reference gets these from torch autograd over AgentQFunction
// (loss.backward(), qmix.py:191); here every op's adjoint is written out:
//
//   head_bwd  : d agent_q -> dq at the chosen action -> LN backward -> dh_out[t]             (thread per row)
//   gru_bwd   : BPTT over t = T-1..0 (one wave per (agent,episode) row, W_hh^T columns in VGPRs)
//   trunk_bwd : dgi -> W_ih^T -> LN2 bwd -> ReLU -> fc2^T -> LN1 bwd -> ReLU -> dz1           (f32 MFMA chain)
//
// Weight gradients are K-reductions over all rows and are done by ope_wgrad.hip from the per-row adjoints
// stored here (dz1, dz2, dgi, dghn, dqoh).
#include "ope_agent.h"

namespace ope {

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_bwd_kernel(HeadBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [Wq A*64][gamma 64]
  const int A = a.A;
  for (int i = threadIdx.x; i < A * OPE_H + OPE_H; i += blockDim.x)
    sm[i] = (i < A * OPE_H) ? a.theta[a.L.q_w + i] : a.theta[a.L.lno_w + (i - A * OPE_H)];
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
  const int A4 = ope_round4_dev(A);
  for (int k = 0; k < A4; ++k) a.dqoh[r * A4 + k] = (k == act) ? dq : 0.f;
}

int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  const size_t lds = (size_t)(a.A * OPE_H + OPE_H) * sizeof(float);
  if (lds > 64 * 1024) return OPE_EINVAL;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(ope_cdiv(a.R, 256)), dim3(256), lds, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// gru_bwd. Lane k owns hidden feature k and keeps COLUMN k of W_hh (192 floats as 96 float2) in VGPRs:
//   dh_{t-1}[k] = dh_t[k] z[k] + sum_i ( W_hr[i][k] dr_pre[i] + W_hz[i][k] dz_pre[i] + W_hn[i][k] dghn[i] )
// The three 64-vectors go through the wave's private LDS slot and come back as broadcast ds_read_b128; the
// 192-long reduction runs as packed FMAs on 12 independent partial sums.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gru_bwd_kernel(GruBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float ds[4][3 * OPE_H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.NB) return;
  f32x2 wr[OPE_H / 2], wz[OPE_H / 2], wn[OPE_H / 2];
  {
    const float* w = a.theta + a.whh_off;
#pragma unroll
    for (int i = 0; i < OPE_H / 2; ++i) {
      wr[i] = f32x2{w[(int64_t)(2 * i) * OPE_H + lane], w[(int64_t)(2 * i + 1) * OPE_H + lane]};
      wz[i] = f32x2{w[(int64_t)(OPE_H + 2 * i) * OPE_H + lane], w[(int64_t)(OPE_H + 2 * i + 1) * OPE_H + lane]};
      wn[i] = f32x2{w[(int64_t)(2 * OPE_H + 2 * i) * OPE_H + lane], w[(int64_t)(2 * OPE_H + 2 * i + 1) * OPE_H + lane]};
    }
  }
  float dh = 0.f;
  const int64_t NB = a.NB;
  float* my = ds[wave];
  // software prefetch of the per-step saved activations
  int64_t o = ((int64_t)(a.T - 1) * NB + row) * OPE_H + lane;
  float r = a.rg[o], z = a.zg[o], n = a.ng[o], gn = a.ghn[o], dho = a.dh_out[o];
  float hp = a.T - 1 > 0 ? a.h[o - NB * OPE_H] : 0.f;
  for (int t = a.T - 1; t >= 0; --t) {
    float r2 = 0.f, z2 = 0.f, n2 = 0.f, gn2 = 0.f, dho2 = 0.f, hp2 = 0.f;
    const int64_t o2 = o - NB * OPE_H;
    if (t > 0) {
      r2 = a.rg[o2]; z2 = a.zg[o2]; n2 = a.ng[o2]; gn2 = a.ghn[o2]; dho2 = a.dh_out[o2];
      hp2 = t - 1 > 0 ? a.h[o2 - NB * OPE_H] : 0.f;
    }
    const float dht = dh + dho;
    const float dn = dht * (1.0f - z);
    const float dzg = dht * (hp - n);
    const float dn_pre = dn * (1.0f - n * n);
    const float dz_pre = dzg * z * (1.0f - z);
    const float dr_pre = dn_pre * gn * r * (1.0f - r);
    const float dgn = dn_pre * r;
    my[lane] = dr_pre;
    my[OPE_H + lane] = dz_pre;
    my[2 * OPE_H + lane] = dgn;
    __builtin_amdgcn_wave_barrier();
    float* gout = a.dgi + ((int64_t)t * NB + row) * (3 * OPE_H) + lane;
    gout[0] = dr_pre;
    gout[OPE_H] = dz_pre;
    gout[2 * OPE_H] = dn_pre;
    a.dghn[o] = dgn;
    f32x2 c0 = {dht * z, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c3 = {0.f, 0.f}, c4 = {0.f, 0.f}, c5 = {0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f32x4 rv[4], zv[4], nv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rv[q] = *reinterpret_cast<const f32x4*>(my + 16 * blk + 4 * q);
        zv[q] = *reinterpret_cast<const f32x4*>(my + OPE_H + 16 * blk + 4 * q);
        nv[q] = *reinterpret_cast<const f32x4*>(my + 2 * OPE_H + 16 * blk + 4 * q);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i2 = 8 * blk + 2 * q;
        c0 = __builtin_elementwise_fma(wr[i2], f32x2{rv[q][0], rv[q][1]}, c0);
        c1 = __builtin_elementwise_fma(wr[i2 + 1], f32x2{rv[q][2], rv[q][3]}, c1);
        c2 = __builtin_elementwise_fma(wz[i2], f32x2{zv[q][0], zv[q][1]}, c2);
        c3 = __builtin_elementwise_fma(wz[i2 + 1], f32x2{zv[q][2], zv[q][3]}, c3);
        c4 = __builtin_elementwise_fma(wn[i2], f32x2{nv[q][0], nv[q][1]}, c4);
        c5 = __builtin_elementwise_fma(wn[i2 + 1], f32x2{nv[q][2], nv[q][3]}, c5);
      }
    }
    __builtin_amdgcn_wave_barrier();
    dh = ((c0[0] + c0[1]) + (c1[0] + c1[1])) + ((c2[0] + c2[1]) + (c3[0] + c3[1])) + ((c4[0] + c4[1]) + (c5[0] + c5[1]));
    r = r2; z = z2; n = n2; gn = gn2; dho = dho2; hp = hp2; o = o2;
  }
}

int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st) {
  if (a.NB < 1 || a.T < 1) return OPE_EINVAL;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(ope_cdiv(a.NB, 4)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
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

  // da2 = W_ih^T dgi   (K = 192)
  f32x4 d[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) d[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* grow = a.dgi + rr * (3 * OPE_H);
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    f32x4 bv = *reinterpret_cast<const f32x4*>(grow + 16 * c + 4 * g);
    if (!valid) bv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wihT + (int64_t)(16 * it + j) * (3 * OPE_H) + 16 * c + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) d[it] = mfma16(w[r], bv[r], d[it]);
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
