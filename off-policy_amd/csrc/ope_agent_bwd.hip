// Agent q-network backward kernels. The reference gets these from torch autograd over AgentQFunction
// (loss.backward(), qmix.py:191); here every op's adjoint is written out:
//
//   head_bwd  : d agent_q -> dq at the chosen action -> LN backward -> dh_out[t]             (thread per row)
//   gru_bwd   : BPTT over t = T-1..0 (one wave per (agent,episode) row, W_hh^T columns in VGPRs)
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
  const size_t lds = (size_t)(a.A * OPE_H + OPE_H) * sizeof(float);
  if (lds > 64 * 1024) return OPE_EINVAL;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(ope_cdiv(a.R, 256)), dim3(256), lds, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// gru_bwd: BPTT over t = T-1 .. 0. Same decomposition as gru_fwd: a row is split over WPR waves, wave q owning the
// slice i in [q*64/WPR, (q+1)*64/WPR) of the reduction
//   dh_{t-1}[k] = dh_t[k] z[k] + sum_i ( W_hr[i][k] dr_pre[i] + W_hz[i][k] dz_pre[i] + W_hn[i][k] dghn[i] )
// with lane k holding its 3 x 64/WPR weights (column k of W_hh) in VGPRs. Every wave evaluates the gate adjoints
// redundantly (bit-identical), broadcasts its i-slice of them through a private LDS slot, and the WPR partial sums
// meet behind one LDS barrier per step. Wave 0 only loads (the saved activations, one 8-step chunk ahead, via
// compiler-invisible asm loads), the last wave only stores (dgi, dghn).
// ---------------------------------------------------------------------------------------------------------
constexpr int kGruChunkB = 8;

template <int WPR>
__global__ void __launch_bounds__(256) gru_bwd_kernel(GruBwdArgs a) {
  constexpr int RPW = 4 / WPR;
  constexpr int IW = OPE_H / WPR;
  constexpr int C = kGruChunkB;
  __shared__ __attribute__((aligned(16))) float ds[4][3][OPE_H];
  __shared__ __attribute__((aligned(16))) float part[2][RPW][WPR][OPE_H];
  __shared__ __attribute__((aligned(16))) float sav[RPW][2][C][6][OPE_H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = wave / WPR, q = wave % WPR;
  const int row_raw = blockIdx.x * RPW + rl;
  const bool active = row_raw < a.NB;
  const int row = active ? row_raw : a.NB - 1;
  const bool loader = (q == 0), storer = (q == WPR - 1) && active;

  f32x2 wr[IW / 2], wz[IW / 2], wn[IW / 2];
  {
    const float* w = a.theta + a.whh_off + (int64_t)(IW * q) * OPE_H + lane;
#pragma unroll
    for (int i = 0; i < IW / 2; ++i) {
      wr[i] = f32x2{w[(int64_t)(2 * i) * OPE_H], w[(int64_t)(2 * i + 1) * OPE_H]};
      wz[i] = f32x2{w[(int64_t)(OPE_H + 2 * i) * OPE_H], w[(int64_t)(OPE_H + 2 * i + 1) * OPE_H]};
      wn[i] = f32x2{w[(int64_t)(2 * OPE_H + 2 * i) * OPE_H], w[(int64_t)(2 * OPE_H + 2 * i + 1) * OPE_H]};
    }
  }
  const int64_t NB = a.NB;
  float preA[C][3], preB[C][3];   // loader: r,z,n | ghn,dh_out,h_prev of the next chunk
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int s2 = 0; s2 < C; ++s2) {
      const int t = max(a.T - 1 - (c * C + s2), a.t_lo);
      const int64_t o = ((int64_t)t * NB + row) * OPE_H + lane;
      gload_async(preA[s2][0], a.rg + o);
      gload_async(preA[s2][1], a.zg + o);
      gload_async(preA[s2][2], a.ng + o);
      gload_async(preB[s2][0], a.ghn + o);
      gload_async(preB[s2][1], a.dh_out + o);
      gload_async(preB[s2][2], a.h + (t > 0 ? o - NB * OPE_H : o));
    }
  };
  auto publish_chunk = [&](int c, int buf) {
    OPE_GWAIT24(preA);
    asm volatile("" : "+v"(preB[0][0]), "+v"(preB[0][1]), "+v"(preB[0][2]), "+v"(preB[1][0]), "+v"(preB[1][1]), "+v"(preB[1][2]),
                 "+v"(preB[2][0]), "+v"(preB[2][1]), "+v"(preB[2][2]), "+v"(preB[3][0]), "+v"(preB[3][1]), "+v"(preB[3][2]),
                 "+v"(preB[4][0]), "+v"(preB[4][1]), "+v"(preB[4][2]), "+v"(preB[5][0]), "+v"(preB[5][1]), "+v"(preB[5][2]),
                 "+v"(preB[6][0]), "+v"(preB[6][1]), "+v"(preB[6][2]), "+v"(preB[7][0]), "+v"(preB[7][1]), "+v"(preB[7][2]));
#pragma unroll
    for (int s2 = 0; s2 < C; ++s2) {
      const int t = a.T - 1 - (c * C + s2);
      sav[rl][buf][s2][0][lane] = preA[s2][0];
      sav[rl][buf][s2][1][lane] = preA[s2][1];
      sav[rl][buf][s2][2][lane] = preA[s2][2];
      sav[rl][buf][s2][3][lane] = preB[s2][0];
      sav[rl][buf][s2][4][lane] = preB[s2][1];
      sav[rl][buf][s2][5][lane] = t > 0 ? preB[s2][2] : 0.f;   // h_{-1} = 0
    }
  };
  if (loader) {
    load_chunk(0);
    publish_chunk(0, 0);
  }
  lds_barrier();

  float dh = a.dh_in ? a.dh_in[(int64_t)row * OPE_H + lane] : 0.f;
  float(*myds)[OPE_H] = ds[wave];
  const int nsteps = a.T - a.t_lo;
  const int nchunks = (nsteps + C - 1) / C;
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (loader && c + 1 < nchunks) load_chunk(c + 1);
    const int ns = min(C, nsteps - c * C);
    for (int s2 = 0; s2 < ns; ++s2) {
      const int t = a.T - 1 - (c * C + s2);
      const float r = sav[rl][buf][s2][0][lane], z = sav[rl][buf][s2][1][lane], n = sav[rl][buf][s2][2][lane];
      const float gn = sav[rl][buf][s2][3][lane], dho = sav[rl][buf][s2][4][lane], hp = sav[rl][buf][s2][5][lane];
      const float dht = dh + dho;
      const float dn = dht * (1.0f - z);
      const float dzg = dht * (hp - n);
      const float dn_pre = dn * (1.0f - n * n);
      const float dz_pre = dzg * z * (1.0f - z);
      const float dr_pre = dn_pre * gn * r * (1.0f - r);
      const float dgn = dn_pre * r;
      myds[0][lane] = dr_pre;
      myds[1][lane] = dz_pre;
      myds[2][lane] = dgn;
      __builtin_amdgcn_wave_barrier();
      if (storer) {
        const int64_t o = ((int64_t)t * NB + row) * OPE_H + lane;
        float* gout = a.dgi + ((int64_t)t * NB + row) * (3 * OPE_H) + lane;
        gout[0] = dr_pre;
        gout[OPE_H] = dz_pre;
        gout[2 * OPE_H] = dn_pre;
        a.dghn[o] = dgn;
      }
      f32x2 c0 = {0.f, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c3 = {0.f, 0.f}, c4 = {0.f, 0.f}, c5 = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < IW / 4; ++v) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(&myds[0][IW * q + 4 * v]);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(&myds[1][IW * q + 4 * v]);
        const f32x4 nv = *reinterpret_cast<const f32x4*>(&myds[2][IW * q + 4 * v]);
        c0 = __builtin_elementwise_fma(wr[2 * v], f32x2{rv[0], rv[1]}, c0);
        c1 = __builtin_elementwise_fma(wr[2 * v + 1], f32x2{rv[2], rv[3]}, c1);
        c2 = __builtin_elementwise_fma(wz[2 * v], f32x2{zv[0], zv[1]}, c2);
        c3 = __builtin_elementwise_fma(wz[2 * v + 1], f32x2{zv[2], zv[3]}, c3);
        c4 = __builtin_elementwise_fma(wn[2 * v], f32x2{nv[0], nv[1]}, c4);
        c5 = __builtin_elementwise_fma(wn[2 * v + 1], f32x2{nv[2], nv[3]}, c5);
      }
      float(*pp)[OPE_H] = part[t & 1][rl];
      pp[q][lane] = ((c0[0] + c0[1]) + (c1[0] + c1[1])) + ((c2[0] + c2[1]) + (c3[0] + c3[1])) + ((c4[0] + c4[1]) + (c5[0] + c5[1]));
      lds_barrier();
      float acc = dht * z;
#pragma unroll
      for (int w2 = 0; w2 < WPR; ++w2) acc += pp[w2][lane];
      dh = acc;
    }
    // Hand the next chunk over. Its first use is at the top of the next step, BEFORE that step's barrier, so the
    // publish needs a barrier of its own (one per 8 steps); without it the readers race the loader whenever its
    // prefetch lands late, e.g. under memory contention from a kernel running concurrently on another stream.
    if (c + 1 < nchunks) {
      if (loader) publish_chunk(c + 1, buf ^ 1);
      lds_barrier();
    }
  }
  if (storer && a.dh_carry) a.dh_carry[(int64_t)row * OPE_H + lane] = dh;
}

int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st) {
  if (a.NB < 1 || a.T < 1 || a.t_lo < 0 || a.t_lo >= a.T) return OPE_EINVAL;
  if ((int64_t)a.NB * 4 <= 1280) {
    hipLaunchKernelGGL(gru_bwd_kernel<4>, dim3(a.NB), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(gru_bwd_kernel<2>, dim3(ope_cdiv(a.NB, 2)), dim3(256), 0, st, a);
  }
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
