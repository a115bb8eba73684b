// trunk_bwd4: adjoint of the recurrent agent trunk, weights resident in LDS, one wave per 16-row tile.
//   dgi -> da2 = W_ih^T dgi -> LN2 / ReLU adjoint -> dz2 -> da1 = fc2^T dz2 -> LN1 / ReLU adjoint -> dz1
// autograd of MLPLayer.forward + the GRU input projection (offpolicy/algorithms/utils/mlp.py:25-29, rnn.py:19-23) as taken by
// loss.backward() in QMix.train_policy_on_batch (offpolicy/algorithms/qmix/qmix.py:190-193); the weight gradients are taken from
// dz1 / dz2 / dgi by the batched wgrad launch.
//
// Same idea as trunk_fwd4 (ope_trunk4.hip) for the mirror image: trunk_bwd3 keeps W_ih^T / fc2^T fragments in registers, four waves
// share a tile and meet behind four workgroup barriers per tile (24.8 us at 3s5z batch 32, 33.8 us at MMM2). Here a
// CU stages W_ih^T (64 x 192) and fc2^T (64 x 64) once into XOR-swizzled LDS (64 KB) and every WAVE walks whole 16-row tiles alone:
// the dgi rows arrive in the MFMA B layout straight from global memory (lane (j, g): 16-byte piece 4 c + g of row j, requested one
// tile ahead, behind the first product), an adjoint fragment (lane (j, g): features 16 it + 4 g .. + 3 of row j) IS the next
// product's B operand, and the LayerNorm adjoints' row sums are 16 local values + two lane swaps. No barrier after the prologue;
// the two waves of a SIMD draw their tiles from one LDS counter. Measured: 23.5 us at 3s5z, 30.4 us at MMM2 (-5 % / -10 %): the launch moves
// 69 MB (dgi in, the saved activations in, dz1 / dz2 out) and has 2.3 tiles per SIMD at 3s5z, so it sits near its memory time, not its MFMA time.
#include <stdlib.h>

#include "ope_agent.h"

namespace ope {

namespace {

constexpr int kG = 3 * OPE_H;                                                     // 192 gate rows
__device__ __forceinline__ int bslot(int row, int p) { return p ^ ((row & 7) << 1); }   // swizzled 16-byte slot of logical slot p in `row`

}  // namespace

__global__ void __launch_bounds__(512, 2) trunk_bwd4_kernel(TrunkBwdArgs a) {
  constexpr int NW = 8, NT = 64 * NW;      // eight waves (two per SIMD); twelve (157 registers fit three per SIMD) measured the same: 23.8 vs 23.5 us
  __shared__ __attribute__((aligned(16))) float sm[OPE_H * kG + OPE_H * OPE_H + 2 * OPE_H + 16];
  float* const WAs = sm;                       // W_ih^T  [64][192], swizzled
  float* const WBs = WAs + OPE_H * kG;         // fc2^T   [64][64],  swizzled
  float* const gms = WBs + OPE_H * OPE_H;      // ln2 weight, ln1 weight
  int* const ctr = reinterpret_cast<int*>(gms + 2 * OPE_H);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int R = a.R_dev ? min(__builtin_amdgcn_readfirstlane(*a.R_dev), a.R) : a.R;      // (the live plan's packed rows with a loss term)
  const int ntiles = (R + 15) >> 4;
  const int nslots = 4 * (int)gridDim.x, slot0 = 4 * (int)blockIdx.x + (wave & 3);
  auto grab = [&]() -> int {
    int k = 0;
    if (lane == 0) k = __hip_atomic_fetch_add(&ctr[wave & 3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    k = __builtin_amdgcn_readfirstlane(k);
    const int64_t t = (int64_t)slot0 + (int64_t)nslots * k;
    return t < ntiles ? (int)t : ntiles;
  };
  // every global access of the tile loop is (uniform base) + (32-bit byte offset): R * 768 B < 4 GB is checked by the launcher
  const char* __restrict__ dgb = reinterpret_cast<const char*>(a.dgi);
  f32x4 dgv[12];
  auto request = [&](int tile) {
    const int row = tile * 16 + j;
    const uint32_t o = (uint32_t)(row < R ? row : R - 1) * (uint32_t)(4 * kG) + 16u * g;
#pragma unroll
    for (int c = 0; c < 12; ++c) dgv[c] = *reinterpret_cast<const f32x4*>(dgb + (o + 64u * c));
  };
  // first tile assigned statically (slot, round = wave / 4): its rows are requested before the weights are staged
  int tile;
  {
    const int64_t t = (int64_t)slot0 + (int64_t)nslots * (wave >> 2);
    tile = t < ntiles ? (int)t : ntiles;
  }
  if (tile < ntiles) request(tile);
  // ---- prologue: W_ih^T and fc2^T (the transposed copies the backward pass keeps: thetaT) -> swizzled LDS ----
  {
    constexpr int PA = OPE_H * (kG / 4), PB = OPE_H * (OPE_H / 4);     // 16-byte pieces: 3 072 and 1 024
    constexpr int NA = (PA + NT - 1) / NT, NB = (PB + NT - 1) / NT;    // per thread (a partly used last round when NT does not divide)
    const float* __restrict__ wihT = a.thetaT;
    const float* __restrict__ fc2T = a.thetaT + OPE_H * kG;
    f32x4 pa[NA], pb[NB];
#pragma unroll
    for (int u = 0; u < NA; ++u) pa[u] = *reinterpret_cast<const f32x4*>(wihT + 4 * min(tid + NT * u, PA - 1));
#pragma unroll
    for (int u = 0; u < NB; ++u) pb[u] = *reinterpret_cast<const f32x4*>(fc2T + 4 * min(tid + NT * u, PB - 1));
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int i = tid + NT * u, row = i / (kG / 4), p = i - row * (kG / 4);
      if (i < PA) *reinterpret_cast<f32x4*>(WAs + row * kG + 4 * bslot(row, p)) = pa[u];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = tid + NT * u, row = i >> 4, p = i & 15;
      if (i < PB) *reinterpret_cast<f32x4*>(WBs + row * OPE_H + 4 * bslot(row, p)) = pb[u];
    }
  }
  if (tid < OPE_H) {
    gms[tid] = a.theta[a.L.ln2_w + tid];
    gms[OPE_H + tid] = a.theta[a.L.ln1_w + tid];
  }
  if (tid < 4) ctr[tid] = NW / 4;        // the first tile of each of a SIMD's waves is assigned statically
  __syncthreads();

  // fragment addresses (see trunk_fwd4): bslot(j, 4 c + g) = 4 ((c & 3) ^ m) + (g ^ 2 (j & 1)) + 16 (c >> 2), m = (j >> 1) & 3
  uint32_t woA[4], woB[4];
  {
    const int m = (j >> 1) & 3, gl = g ^ (2 * (j & 1));
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      woA[cl] = (uint32_t)j * (4u * kG) + 64u * (cl ^ m) + 16u * gl;
      woB[cl] = (uint32_t)j * (4u * OPE_H) + 64u * (cl ^ m) + 16u * gl;
    }
  }
  const char* const WAb = reinterpret_cast<const char*>(WAs);
  const char* const WBb = reinterpret_cast<const char*>(WBs);
  const char* const xh2b = reinterpret_cast<const char*>(a.xhat2);
  const char* const xh1b = reinterpret_cast<const char*>(a.xhat1);

  // LayerNorm + ReLU adjoint of row j over its 64 features, d[it][r] = feature 16 it + 4 g + r (the formula of trunk_bwd3):
  //   d *= gamma;  m1 = mean(d), m2 = mean(d xhat);  d = relu_bit ? rstd (d - m1 - xhat m2) : 0
  auto ln_relu_bwd = [&](f32x4 (&d)[4], const float* gm, const f32x4 (&xh)[4], float rs, uint64_t mask) {
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gm + 16 * it + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[it][r] *= gv[r];
        m1 += d[it][r];
        m2 = fmaf(d[it][r], xh[it][r], m2);
      }
    }
    m1 = rowsum4(m1) * (1.0f / OPE_H);
    m2 = rowsum4(m2) * (1.0f / OPE_H);
    const uint32_t lo = (uint32_t)mask >> (4 * g), hi = (uint32_t)(mask >> 32) >> (4 * g);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint32_t nib = ((it < 2 ? lo : hi) >> (16 * (it & 1))) & 15u;      // bits of features 16 it + 4 g .. + 3
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = rs * (d[it][r] - m1 - xh[it][r] * m2);
        d[it][r] = ((nib >> r) & 1u) ? v : 0.f;
      }
    }
  };

  while (tile < ntiles) {
    asm volatile("" ::: "memory");       // LDS is read-only after the prologue: keep hipcc from hoisting (and spilling) the gamma reads
    const int row = tile * 16 + j;
    const bool valid = row < R;
    const uint32_t rr = (uint32_t)(valid ? row : R - 1);
    // saved activations of the tile: requested now, needed after the first product (192 MFMAs later)
    const uint32_t xo = rr * (4u * OPE_H) + 16u * g;
    f32x4 xh2[4], xh1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) xh2[it] = *reinterpret_cast<const f32x4*>(xh2b + (xo + 64u * it));
#pragma unroll
    for (int it = 0; it < 4; ++it) xh1[it] = *reinterpret_cast<const f32x4*>(xh1b + (xo + 64u * it));
    const float rs2 = a.rstd2[rr], rs1 = a.rstd1[rr];
    const uint64_t mk2 = a.mask2[rr], mk1 = a.mask1[rr];
    // ---- da2 = W_ih^T dgi: 4 output tiles x 12 chunks of the 192 gate rows ----
    f32x4 d[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) d[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      if ((c & 1) == 0) __builtin_amdgcn_sched_barrier(0);      // weight reads at most two chunks ahead of their MFMAs
      f32x4 wv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) wv[it] = *reinterpret_cast<const f32x4*>(WAb + (woA[c & 3] + (uint32_t)(16 * it * 4 * kG + 256 * (c >> 2))));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int it = 0; it < 4; ++it) d[it] = mfma16(wv[it][r], dgv[c][r], d[it]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // The dgi registers are free: the next tile's rows are requested now and land behind the rest of this tile. UNCONDITIONALLY (past
    // the last tile the rows clamp to R - 1, a wasted but valid read): behind a branch hipcc's waitcnt pass merges the two paths and
    // waits for this tile's rstd / mask loads with vmcnt(1) -- i.e. for eleven of the twelve loads just issued.
    const int next = grab();
    request(next);

    ln_relu_bwd(d, gms, xh2, rs2, mk2);
    if (valid) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.dz2) + (xo + 64u * it)) = d[it];
    }
    // ---- da1 = fc2^T dz2 ----
    f32x4 e[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) e[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      f32x4 wv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) wv[it] = *reinterpret_cast<const f32x4*>(WBb + (woB[ft] + (uint32_t)(16 * it * 4 * OPE_H)));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int it = 0; it < 4; ++it) e[it] = mfma16(wv[it][r], d[ft][r], e[it]);
    }
    ln_relu_bwd(e, gms + OPE_H, xh1, rs1, mk1);
    if (valid) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.dz1) + (xo + 64u * it)) = e[it];
    }
    tile = next;
  }
}

// The recurrent trunk's adjoint with enough rows to give every SIMD a few tiles: this kernel; anything else: trunk_bwd3.
// path (ope_qmix_cfg.trunk_path): 0 by shape, 3 trunk_bwd3, 4 this kernel whenever it can run the shape. Process default of
// "by shape": OPE_TRUNK_BWD4 = 1 | 0 (read once).
int launch_trunk_bwd_path(const TrunkBwdArgs& a, int path, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  static const int on = getenv("OPE_TRUNK_BWD4") ? atoi(getenv("OPE_TRUNK_BWD4")) : 1;
  const bool can = a.dgi && a.xhat1 && a.xhat2 && a.rstd1 && a.rstd2 && a.mask1 && a.mask2 && a.dz1 && a.dz2 && a.thetaT &&
                   (int64_t)a.R * (4 * kG) < ((int64_t)1 << 32);
  if (path == 4 && (!can || a.tanh_act)) return OPE_EINVAL;      // explicit request that cannot run: no silent fall-back
  if (a.tanh_act || !(can && (path == 4 || (path == 0 && on && a.R >= 16 * 1024)))) return a.R_dev ? OPE_EINVAL : launch_trunk_bwd3(a, st);
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  kprof_work(2.0 * a.R * ((a.dgi ? 3.0 * OPE_H * OPE_H : 0.0) + OPE_H * OPE_H + (a.dout ? (double)a.hdim * OPE_H : 0.0)));
  if (a.R_dev) kprof_rows(2);
  OPE_LAUNCH(trunk_bwd4_kernel, dim3(cus), dim3(512), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(a.R_dev ? "trunk_bwd4_live" : "trunk_bwd4");
  return OPE_OK;
}

bool trunk_bwd4_can(int64_t R, int path, bool tanh_act) {
  static const int on = getenv("OPE_TRUNK_BWD4") ? atoi(getenv("OPE_TRUNK_BWD4")) : 1;
  return !tanh_act && R * (4 * kG) < ((int64_t)1 << 32) && (path == 4 || (path == 0 && on && R >= 16 * 1024));
}

}  // namespace ope
