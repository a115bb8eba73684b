// trunk_bwd, persistent workgroup-cooperative form.
// Adjoint of the trunk:  dgi -> da2 = W_ih^T dgi -> LN2/ReLU adjoint -> dz2 -> da1 = fc2^T dz2 -> LN1/ReLU adjoint -> dz1
// (the weight gradients are taken from dz1 / dz2 / dgi by the batched wgrad launch).
// Same decomposition as trunk_fwd3 (ope_trunk2.hip): the 64 features of each layer are split over the 4 waves of a
// workgroup, every wave keeps ITS rows of W_ih^T (12 fragments) and fc2^T (4 fragments) in registers for the life of the
// kernel and the grid walks over 16-row tiles; per tile the dgi rows are staged once in LDS with full-line loads, the
// two LayerNorm adjoints exchange their row sums through LDS (one barrier each).
#include <stdlib.h>

#include "ope_agent.h"

namespace ope {

constexpr int kDgPitch = 3 * OPE_H + 4;
constexpr int kDzPitch = OPE_H + 4;

__global__ void __launch_bounds__(256, 2) trunk_bwd3_kernel(TrunkBwdArgs a) {
  constexpr int TR = 16;
  __shared__ __attribute__((aligned(16))) float dg[TR * kDgPitch];
  __shared__ __attribute__((aligned(16))) float dzb[TR * kDzPitch];
  __shared__ __attribute__((aligned(8))) float stat2[2][2 * 4 * TR];   // one exchange buffer per LayerNorm: no WAR hazard across tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const float* __restrict__ th = a.theta;
  const float* __restrict__ wihT = a.thetaT;                       // [64][192]
  const float* __restrict__ fc2T = a.thetaT + OPE_H * 3 * OPE_H;   // [64][64]
  const int ntiles = (a.R + TR - 1) / TR;
  const bool recurrent = a.dgi != nullptr;
  const int fo = 16 * wave + 4 * g;   // this lane's 4 features

  f32x4 wA[12], wB[4];
  if (recurrent) {
#pragma unroll
    for (int c = 0; c < 12; ++c) wA[c] = *reinterpret_cast<const f32x4*>(wihT + (int64_t)(16 * wave + j) * (3 * OPE_H) + 16 * c + 4 * g);
  }
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) wB[ft] = *reinterpret_cast<const f32x4*>(fc2T + (int64_t)(16 * wave + j) * OPE_H + 16 * ft + 4 * g);
  const f32x4 gm2 = *reinterpret_cast<const f32x4*>(th + a.L.ln2_w + fo), gm1 = *reinterpret_cast<const f32x4*>(th + a.L.ln1_w + fo);

  // LayerNorm + ReLU adjoint for this lane's 4 features of row j; the row means over all 64 features meet through LDS
  // (xh, rs, bits: the row's saved normalised values, 1/std and this wave's 16 ReLU bits, loaded at the top of the tile)
  auto ln_relu_bwd = [&](f32x4& d, const f32x4& gm, const f32x4& xh, float rs, uint32_t bits, float mu, float* stat) {
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[r] *= gm[r];
      m1 += d[r];
      m2 = fmaf(d[r], xh[r], m2);
    }
    m1 = rowsum4(m1);
    m2 = rowsum4(m2);
    if (g == 0) *reinterpret_cast<f32x2*>(stat + 2 * (wave * TR + j)) = f32x2{m1, m2};
    lds_barrier();
    const f32x2 p0 = *reinterpret_cast<const f32x2*>(stat + 2 * j), p1 = *reinterpret_cast<const f32x2*>(stat + 2 * (TR + j));
    const f32x2 p2 = *reinterpret_cast<const f32x2*>(stat + 2 * (2 * TR + j)), p3 = *reinterpret_cast<const f32x2*>(stat + 2 * (3 * TR + j));
    m1 = ((p0[0] + p1[0]) + (p2[0] + p3[0])) * (1.0f / OPE_H);
    m2 = ((p0[1] + p1[1]) + (p2[1] + p3[1])) * (1.0f / OPE_H);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = rs * (d[r] - m1 - xh[r] * m2);
      if (a.tanh_act) {      // a = tanh(z) = xhat / rstd + mean (what the LayerNorm normalised); d tanh = 1 - a^2
        const float act = fmaf(xh[r], 1.0f / rs, mu);
        d[r] = v * (1.0f - act * act);
      } else {
        d[r] = ((bits >> (4 * g + r)) & 1u) ? v : 0.f;
      }
    }
  };

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    const int row = row0 + j;
    const bool valid = row < a.R;
    const int64_t rr = valid ? row : a.R - 1;
    // every global load of the tile is issued here, so that the tile pays ONE memory round trip (the saved activations
    // used to be fetched inside the two LayerNorm adjoints: three dependent round trips per tile)
    const f32x4 xh2 = *reinterpret_cast<const f32x4*>(a.xhat2 + rr * OPE_H + fo);
    const f32x4 xh1 = *reinterpret_cast<const f32x4*>(a.xhat1 + rr * OPE_H + fo);
    const float rs2 = a.rstd2[rr], rs1 = a.rstd1[rr];
    const uint32_t bits2 = reinterpret_cast<const uint16_t*>(a.mask2 + rr)[wave];   // this wave's 16 ReLU bits of the row
    const uint32_t bits1 = reinterpret_cast<const uint16_t*>(a.mask1 + rr)[wave];
    const float mu2 = reinterpret_cast<const float*>(a.mask2 + rr)[0], mu1 = reinterpret_cast<const float*>(a.mask1 + rr)[0];   // (OPE_DIMS_TANH: the slots hold row means)
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    if (recurrent) {
      // stage the tile's dgi rows (16 x 192 floats, contiguous in memory) in LDS: 3 x 16-byte pieces per thread
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = threadIdx.x + 256 * q;          // 0 .. 767 float4 pieces
        const int r = e / 48, c4 = e - r * 48;
        const int64_t src = (int64_t)(row0 + r < a.R ? row0 + r : a.R - 1);
        *reinterpret_cast<f32x4*>(dg + r * kDgPitch + 4 * c4) = *reinterpret_cast<const f32x4*>(a.dgi + src * (3 * OPE_H) + 4 * c4);
      }
      lds_barrier();
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(dg + j * kDgPitch + 16 * c + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) d = mfma16(wA[c][r], bv[r], d);
      }
    } else if (a.da2_in) {
      d = *reinterpret_cast<const f32x4*>(a.da2_in + rr * OPE_H + fo);
    } else {   // da2 = dout W_head (small heads)
      for (int k = 0; k < a.hdim; ++k) {
        const float dk = a.dout[rr * a.ldk + k];
        const f32x4 w = *reinterpret_cast<const f32x4*>(th + a.L.q_w + (int64_t)k * OPE_H + fo);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = fmaf(dk, w[r], d[r]);
      }
    }
    ln_relu_bwd(d, gm2, xh2, rs2, bits2, mu2, stat2[0]);
    if (valid) *reinterpret_cast<f32x4*>(a.dz2 + (int64_t)row * OPE_H + fo) = d;
    *reinterpret_cast<f32x4*>(dzb + j * kDzPitch + fo) = d;
    lds_barrier();
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(dzb + j * kDzPitch + 16 * ft + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) e = mfma16(wB[ft][r], bv[r], e);
    }
    ln_relu_bwd(e, gm1, xh1, rs1, bits1, mu1, stat2[1]);   // its barrier also orders the dzb reads before the next tile's writes
    if (valid) *reinterpret_cast<f32x4*>(a.dz1 + (int64_t)row * OPE_H + fo) = e;
  }
}

int launch_trunk_bwd3(const TrunkBwdArgs& a, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  const int ntiles = ope_cdiv(a.R, 16);
  const int blocks = ntiles < 512 ? ntiles : 512;
  kprof_work(2.0 * a.R * ((a.dgi ? 3.0 * OPE_H * OPE_H : 0.0) + OPE_H * OPE_H + (a.dout ? (double)a.hdim * OPE_H : 0.0)));
  OPE_LAUNCH(trunk_bwd3_kernel, dim3(blocks), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("trunk_bwd3");
  return OPE_OK;
}

}  // namespace ope
