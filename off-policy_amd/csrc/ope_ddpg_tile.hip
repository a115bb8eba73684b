// Fused MLP MADDPG / MATD3 update for small networks (MPE-sized: input widths <= 128, <= 16 actions): the whole critic
// update -- target actor, target action, target critic, live critic, TD error, critic backward, weight gradients -- is
// ONE launch, and so is the whole actor update (actor, hard gumbel-softmax, critic on the substituted joint action, critic
// input adjoint, straight-through adjoint, actor backward, weight gradients). A second small launch sums the per-workgroup
// gradient slabs in a fixed order.
//
// Replaces (reference): MADDPG.shared_train_policy_on_batch, offpolicy/algorithms/maddpg/maddpg.py:90-249 (+ get_update_info
// 38-81, MADDPG_Actor / MADDPG_Critic actor_critic.py:7-87, gumbel_softmax / onehot_from_logits util.py:156-214). The
// general path (ope_ddpg.hip: ~14 launches of 4-8 us per network update through the shared trunk / wgrad / finalize
// kernels) stays for wide inputs.
//
// Shape of the work: 256-768 data rows through 64-wide layers, ~90 MFLOP per update -- nothing here is bound by a
// throughput roof, only by how long ONE row's chain of dependent layers takes and by launch count. So:
//   * a WORKGROUP (4 waves) owns a tile of 16 data rows and walks it through the entire update; between layers the tile's
//     activations live in LDS ([16][64 + 4] floats), never in HBM;
//   * every layer is f32 MFMA 16x16x4 (exact fp32 FMA chains): wave w produces output features 16w..16w+15 of all 16 rows
//     with the weight rows as the A operand (straight from L2 into registers, 16-byte loads in the weights' own [out][in]
//     orientation) and the rows' activations as the B operand (one ds_read_b128 per 16 features). 16 MFMAs per wave per
//     64x64 layer instead of the 64 FMAs + 64 broadcasts per ROW of a VALU mat-vec;
//   * LayerNorm / ReLU / bias / the TD error / the gumbel-softmax are applied on the fragment a lane holds anyway as MFMA
//     operand (lane (j, g) = features 16c + 4g + 0..3 of row j): row statistics are 16 local values + two cross-lane steps,
//     computed redundantly by the 4 waves instead of being exchanged -- one LDS barrier per layer;
//   * weight gradients are MFMAs over the tile's 16 rows (A = adjoint^T, B = layer input, both from LDS) written straight
//     into the workgroup's slab, which has the flat-gradient layout; bias / LayerNorm gradients are DPP row sums.
// No atomics: bitwise deterministic.
#include <stdlib.h>
#include <string.h>

#include "ope_ddpg.h"
#include "ope_rng.h"

namespace ope {
namespace {

constexpr int kRT = 16;        // data rows per tile
constexpr int kHS = 68;        // LDS row stride of 64-wide activations (4 x odd: conflict-free operand reads)
constexpr int kHO = 16;        // max head outputs
constexpr float kEps = OPE_LN_EPS;

// Small vectors of one net staged in LDS (float offsets)
enum { V_FNW = 0, V_FNB = 128, V_B1 = 256, V_G1 = 320, V_BE1 = 384, V_B2 = 448, V_G2 = 512, V_BE2 = 576, V_HW = 640,
       V_HB = V_HW + kHO * OPE_H, V_SIZE = V_HB + kHO };

// Flat-parameter offsets of an MLP net, derived from (K0, Hout) on the device: the same arithmetic as ope_agent_layout_mlp
// (checked against it on the host at launch) -- a handful of scalar adds instead of 23 kernel-argument words per net.
struct TL { int fn_w, fn_b, fc1_w, fc1_b, ln1_w, ln1_b, fc2_w, fc2_b, ln2_w, ln2_b, q_w, q_b, end; };
__host__ __device__ __forceinline__ TL tl_of(int K0, int Hout) {
  TL L;
  const int k4 = (K0 + 3) & ~3;
  L.fn_w = 0; L.fn_b = k4; L.fc1_w = 2 * k4; L.fc1_b = L.fc1_w + OPE_H * K0; L.ln1_w = L.fc1_b + OPE_H; L.ln1_b = L.ln1_w + OPE_H;
  L.fc2_w = L.ln1_b + OPE_H + OPE_H * OPE_H + 3 * OPE_H;      // behind the registered-but-unused fc_h block
  L.fc2_b = L.fc2_w + OPE_H * OPE_H; L.ln2_w = L.fc2_b + OPE_H; L.ln2_b = L.ln2_w + OPE_H;
  L.q_w = L.ln2_b + OPE_H; L.q_b = L.q_w + Hout * OPE_H; L.end = L.q_b + ((Hout + 3) & ~3);
  return L;
}
struct TNet {
  const float* th;       // flat parameters
  int K0, Hout, nc0;     // input width, head outputs, 16-feature chunks of an input row (the kernel's template bucket)
  int vec;               // LDS offset of the staged vectors
  int xs;                // LDS row stride of input-wide tiles (4 x odd >= 16 nc0)
};
struct TBuf { float *x0, *xn, *r1, *a1, *r2; };    // LDS: raw input, LN0 output (stride xs); relu(fc1), LN1 output, relu(fc2) (stride kHS)

// A wave's weight fragments, loaded from L2 into registers one phase ahead of their use (nothing in a tile's chain of layers is
// long enough to hide an exposed L2 round trip): rows 16w + j of fc1 / fc2 for the forward products, columns for the adjoints.
template <int NC> struct WFwd { f32x4 A1[NC]; f32x4 A2[4]; };
template <int NC> struct WBwd { f32x4 AT[4]; f32x4 AT1[(NC + 3) / 4][4]; };

// All-reduce over the 4 lanes (j, g = 0..3) that share data row j, on gfx950's VALU lane swaps instead of two ds_bpermute round
// trips through the LDS pipe (the LayerNorm statistics sit on every layer's critical path): v_permlane16_swap pairs row g with
// row g ^ 1, v_permlane32_swap half with half; with both operands = v the two results are the pair's values in the SAME order in
// both lanes, so the sum is bitwise identical across the 4 lanes.
__device__ __forceinline__ float xg_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float xg_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum4(const f32x4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ f32x4 pick(const f32x4 (&x)[4], int w) {     // x[w], w uniform: selects, not a scratch round trip
  f32x4 m;
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = w == 0 ? x[0][r] : (w == 1 ? x[1][r] : (w == 2 ? x[2][r] : x[3][r]));
  return m;
}

// The small vectors of a net: global -> registers (issue) and registers -> LDS (after the loads of everything else needed first
// have been issued too: a load-then-store loop pays one memory round trip per iteration).
struct VecRegs { float fw, fb, s1, s2, hw[kHO * OPE_H / 256], hb; };
__device__ __forceinline__ void vecs_load(const TNet& n, VecRegs& r) {
  const TL L = tl_of(n.K0, n.Hout);
  const float* th = n.th;
  const int tid = threadIdx.x;
  // unconditional loads from clamped addresses (no exec-masked branches in front of the loads that follow); the stores select
  const int k0 = tid < n.K0 ? tid : 0, k3 = tid < 3 * OPE_H ? tid : 0, kh = tid < n.Hout ? tid : 0;
  r.fw = th[L.fn_w + k0]; r.fb = th[L.fn_b + k0];
  // [fc1_b | ln1_w | ln1_b] and [fc2_b | ln2_w | ln2_b] are each 192 consecutive floats of the flat layout
  r.s1 = th[L.fc1_b + k3]; r.s2 = th[L.fc2_b + k3];
#pragma unroll
  for (int u = 0; u < kHO * OPE_H / 256; ++u) {
    const int e = tid + 256 * u;
    r.hw[u] = th[L.q_w + (e < n.Hout * OPE_H ? e : 0)];
  }
  r.hb = th[L.q_b + kh];
}
__device__ __forceinline__ void vecs_store(const TNet& n, const VecRegs& r, float* lds) {
  float* v = lds + n.vec;
  const int tid = threadIdx.x;
  if (tid < n.K0) { v[V_FNW + tid] = r.fw; v[V_FNB + tid] = r.fb; }
  if (tid < 3 * OPE_H) { v[V_B1 + tid] = r.s1; v[V_B2 + tid] = r.s2; }
#pragma unroll
  for (int u = 0; u < kHO * OPE_H / 256; ++u) {
    const int e = tid + 256 * u;
    if (e < n.Hout * OPE_H) v[V_HW + e] = r.hw[u];
  }
  if (tid < n.Hout) v[V_HB + tid] = r.hb;
}

template <int NC>
__device__ __forceinline__ void load_wfwd(const TNet& n, int wave, int j, int g, WFwd<NC>& w) {
  const TL L = tl_of(n.K0, n.Hout);
  const float* p = n.th + L.fc1_w + (16 * wave + j) * n.K0;
  const bool al = (n.K0 & 3) == 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {       // branch-free: clamped addresses, then zero what lies beyond K0
    const int k = 16 * c + 4 * g;
    w.A1[c] = mask4(al ? load4c<4>(p, k, n.K0) : load4c<1>(p, k, n.K0), k, n.K0);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) w.A2[c] = *reinterpret_cast<const f32x4*>(n.th + L.fc2_w + (16 * wave + j) * OPE_H + 16 * c + 4 * g);
}
// Column `col` of a [64][ld] matrix as A-operand fragments of a product over its 64 rows: A[c][r] = W[16c + 4g + r][col]
__device__ __forceinline__ void load_wcol(const float* __restrict__ w, int ld, int col, bool on, int g, f32x4 (&A)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = w[(16 * c + 4 * g + r) * ld + (on ? col : 0)];      // clamped address + select: no branch around the load
      A[c][r] = on ? t : 0.f;
    }
}
template <int NC>
__device__ __forceinline__ void load_wbwd(const TNet& n, int wave, int j, int g, WBwd<NC>& w) {
  const TL L = tl_of(n.K0, n.Hout);
  load_wcol(n.th + L.fc2_w, OPE_H, 16 * wave + j, true, g, w.AT);
#pragma unroll
  for (int u = 0; u < (NC + 3) / 4; ++u) {
    const int col = 16 * (wave + 4 * u) + j;
    load_wcol(n.th + L.fc1_w, n.K0, col, col < n.K0, g, w.AT1[u]);
  }
}

// Row j's features as operand fragments: f[c] = buf[j][16c + 4g .. + 3]
template <int NC>
__device__ __forceinline__ void load_frag(const float* buf, int stride, int j, int g, f32x4 (&f)[NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) f[c] = *reinterpret_cast<const f32x4*>(buf + j * stride + 16 * c + 4 * g);
}
// nn.LayerNorm statistics over the K valid features of the row (two-pass); f -> normalised values (0 beyond K)
template <int NC>
__device__ __forceinline__ float ln_frag(f32x4 (&f)[NC], int K, int g) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += sum4(mask4(f[c], 16 * c + 4 * g, K));
  const float invK = 1.0f / (float)K;
  const float mu = xg_sum(s) * invK;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = (16 * c + 4 * g + r < K) ? f[c][r] - mu : 0.f;
      f[c][r] = d;
      q = fmaf(d, d, q);
    }
  const float rstd = __builtin_amdgcn_rsqf(xg_sum(q) * invK + kEps);     // v_rsq_f32 (1 ulp) instead of an IEEE sqrt + divide chain
#pragma unroll
  for (int c = 0; c < NC; ++c) f[c] *= rstd;
  return rstd;
}
template <int NC>
__device__ __forceinline__ void affine_frag(f32x4 (&f)[NC], int K, const float* gam, const float* bet, int g) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = 16 * c + 4 * g;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + k), be = *reinterpret_cast<const f32x4*>(bet + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) f[c][r] = (k + r < K) ? fmaf(f[c][r], ga[r], be[r]) : 0.f;
  }
}
template <int NC>
__device__ __forceinline__ f32x4 mm_frag(const f32x4 (&A)[NC], const f32x4 (&B)[NC]) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains: a dependent MFMA waits 40 cycles, an independent one 32
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      acc0 = mfma16(A[c][r], B[c][r], acc0);
      acc1 = mfma16(A[c][r + 1], B[c][r + 1], acc1);
    }
  return acc0 + acc1;
}

// ---- forward of a tile --------------------------------------------------------------------------------------------------
// b.x0 holds the raw input rows (zero beyond K0 up to 16 NC). Returns the head outputs i = 4g + r of row j (0 for i >= Hout).
// SAVE: also leave LN0 / LN1 outputs in b.xn / b.a1 (operands of the weight-gradient products).
template <bool SAVE, int NC>
__device__ __forceinline__ f32x4 tile_forward(const TNet& n, const float* lds, const TBuf& b, int wave, int lane,
                                              const WFwd<NC>& w) {
  const int j = lane & 15, g = lane >> 4;
  const float* v = lds + n.vec;
  const int fo = 16 * wave + 4 * g;     // first of the 4 output features this lane produces
  // feature norm + fc1
  f32x4 f[NC];
  load_frag<NC>(b.x0, n.xs, j, g, f);
  ln_frag<NC>(f, n.K0, g);
  affine_frag<NC>(f, n.K0, v + V_FNW, v + V_FNB, g);
  if (SAVE) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if ((c & 3) == wave) *reinterpret_cast<f32x4*>(b.xn + j * n.xs + 16 * c + 4 * g) = f[c];
  }
  f32x4 z = mm_frag<NC>(w.A1, f) + *reinterpret_cast<const f32x4*>(v + V_B1 + fo);
#pragma unroll
  for (int r = 0; r < 4; ++r) z[r] = fmaxf(z[r], 0.f);
  *reinterpret_cast<f32x4*>(b.r1 + j * kHS + fo) = z;
  lds_barrier();
  // LN1 + fc2
  f32x4 h[4];
  load_frag<4>(b.r1, kHS, j, g, h);
  ln_frag<4>(h, OPE_H, g);
  affine_frag<4>(h, OPE_H, v + V_G1, v + V_BE1, g);
  if (SAVE) *reinterpret_cast<f32x4*>(b.a1 + j * kHS + fo) = pick(h, wave);
  z = mm_frag<4>(w.A2, h) + *reinterpret_cast<const f32x4*>(v + V_B2 + fo);
#pragma unroll
  for (int r = 0; r < 4; ++r) z[r] = fmaxf(z[r], 0.f);
  *reinterpret_cast<f32x4*>(b.r2 + j * kHS + fo) = z;
  lds_barrier();
  // LN2 + head. Every wave holds the whole normalised row anyway (the statistics are computed redundantly), so each forms the
  // complete head output itself: 16 MFMAs on an idle matrix pipe instead of 4 + an LDS exchange behind a third barrier.
  load_frag<4>(b.r2, kHS, j, g, h);
  ln_frag<4>(h, OPE_H, g);
  affine_frag<4>(h, OPE_H, v + V_G2, v + V_BE2, g);
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 Ah = {0.f, 0.f, 0.f, 0.f};
    if (j < n.Hout) Ah = *reinterpret_cast<const f32x4*>(v + V_HW + j * OPE_H + 16 * c + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      p0 = mfma16(Ah[r], h[c][r], p0);
      p1 = mfma16(Ah[r + 1], h[c][r + 1], p1);
    }
  }
  f32x4 out = p0 + p1;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r] = (4 * g + r < n.Hout) ? out[r] + v[V_HB + 4 * g + r] : 0.f;
  return out;
}

// ---- backward of a tile -------------------------------------------------------------------------------------------------
// dhead: d loss / d head output i = 4g + r of row j (0 for i >= Hout and for dead rows). ACC: write (first) / add (!first) the
// parameter gradients of this tile into `slab` (flat layout); needs b.xn / b.a1. DX: return the adjoint of the RAW input in dx.
// LDS scratch: dz, d1 [16][kHS], da [16][xs].
__device__ __forceinline__ void slab_put(float* p, float v, bool first) { *p = first ? v : *p + v; }

// adjoint of a 64-wide LayerNorm output -> adjoint of the pre-ReLU linear output (in place); LN parameter gradients on the way
template <bool ACC>
__device__ __forceinline__ void ln_backward(f32x4 (&d)[4], const f32x4 (&xhat)[4], const f32x4 (&raw)[4], float rstd, const float* gam,
                                            float* slab_g, float* slab_b, bool first, int wave, int j, int g) {
  if (ACC) {
    const f32x4 dm = pick(d, wave), xm = pick(xhat, wave);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = row16_sum(dm[r] * xm[r]), sb = row16_sum(dm[r]);
      if (j == 0) { slab_put(slab_g + 16 * wave + 4 * g + r, sg, first); slab_put(slab_b + 16 * wave + 4 * g + r, sb, first); }
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    d[c] *= *reinterpret_cast<const f32x4*>(gam + 16 * c + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1 += d[c][r]; s2 = fmaf(d[c][r], xhat[c][r], s2); }
  }
  const float m1 = xg_sum(s1) * (1.0f / OPE_H), m2 = xg_sum(s2) * (1.0f / OPE_H);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[c][r] = raw[c][r] > 0.f ? rstd * (d[c][r] - m1 - xhat[c][r] * m2) : 0.f;
}

template <bool ACC, bool DX, int NC>
__device__ __forceinline__ void tile_backward(const TNet& n, const float* lds, const TBuf& b, float* dz, float* d1, float* da, f32x4 dhead,
                                              float* slab, bool first, int wave, int lane, const WBwd<NC>& w, f32x4 (&dx)[NC]) {
  const TL L = tl_of(n.K0, n.Hout);
  const int j = lane & 15, g = lane >> 4;
  const float* v = lds + n.vec;
  const int fo = 16 * wave + 4 * g;
  // ---- head + LN2
  f32x4 raw[4], xh[4], dA[4];
  load_frag<4>(b.r2, kHS, j, g, raw);
#pragma unroll
  for (int c = 0; c < 4; ++c) { xh[c] = raw[c]; dA[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float rstd = ln_frag<4>(xh, OPE_H, g);
  f32x4 a2m = pick(xh, wave);           // this wave's 16 features of the head's input (LN2 output)
  {
    const f32x4 ga = *reinterpret_cast<const f32x4*>(v + V_G2 + fo), be = *reinterpret_cast<const f32x4*>(v + V_BE2 + fo);
#pragma unroll
    for (int r = 0; r < 4; ++r) a2m[r] = fmaf(a2m[r], ga[r], be[r]);
  }
#pragma unroll 1
  for (int k = 0; k < n.Hout; ++k) {
    const int kr = k & 3;
    const float src = kr == 0 ? dhead[0] : (kr == 1 ? dhead[1] : (kr == 2 ? dhead[2] : dhead[3]));
    const float dk = __shfl(src, j + 16 * (k >> 2), 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) dA[c] += dk * *reinterpret_cast<const f32x4*>(v + V_HW + k * OPE_H + 16 * c + 4 * g);
    if (ACC) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = row16_sum(dk * a2m[r]);
        if (j == 0) slab_put(slab + L.q_w + k * OPE_H + fo + r, s, first);
      }
    }
  }
  if (ACC && wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = row16_sum(dhead[r]);
      if (j == 0) {
        if (4 * g + r < n.Hout) slab_put(slab + L.q_b + 4 * g + r, s, first);
        else if (4 * g + r < ((n.Hout + 3) & ~3)) slab[L.q_b + 4 * g + r] = 0.f;      // alignment padding of the flat layout
      }
    }
  }
  ln_backward<ACC>(dA, xh, raw, rstd, v + V_G2, slab + L.ln2_w, slab + L.ln2_b, first, wave, j, g);     // dA = dz2
  // ---- fc2
  if (ACC) *reinterpret_cast<f32x4*>(dz + j * kHS + fo) = pick(dA, wave);
  f32x4 o = mm_frag<4>(w.AT, dA);                                    // da1[j][16w + 4g + r]
  *reinterpret_cast<f32x4*>(d1 + j * kHS + fo) = o;
  lds_barrier();
  if (ACC) {
    const f32x4 m = pick(dA, wave);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = row16_sum(m[r]);
      if (j == 0) slab_put(slab + L.fc2_b + fo + r, s, first);
    }
    float az[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) az[r] = dz[(4 * g + r) * kHS + 16 * wave + j];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma16(az[r], b.a1[(4 * g + r) * kHS + 16 * t + j], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) slab_put(slab + L.fc2_w + (fo + r) * OPE_H + 16 * t + j, acc[r], first);
    }
  }
  // ---- LN1
  load_frag<4>(d1, kHS, j, g, dA);
  load_frag<4>(b.r1, kHS, j, g, raw);
#pragma unroll
  for (int c = 0; c < 4; ++c) xh[c] = raw[c];
  rstd = ln_frag<4>(xh, OPE_H, g);
  ln_backward<ACC>(dA, xh, raw, rstd, v + V_G1, slab + L.ln1_w, slab + L.ln1_b, first, wave, j, g);     // dA = dz1
  // ---- fc1
  if (ACC) {
    lds_barrier();                     // every wave has read dz2 out of `dz`
    *reinterpret_cast<f32x4*>(dz + j * kHS + fo) = pick(dA, wave);
  }
#pragma unroll
  for (int u = 0; u < (NC + 3) / 4; ++u) {
    const int t = wave + 4 * u;
    if (t < NC) {
      o = mm_frag<4>(w.AT1[u], dA);                                  // d xn[j][16t + 4g + r]
      *reinterpret_cast<f32x4*>(da + j * n.xs + 16 * t + 4 * g) = o;
    }
  }
  lds_barrier();
  if (ACC) {
    const f32x4 m = pick(dA, wave);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = row16_sum(m[r]);
      if (j == 0) slab_put(slab + L.fc1_b + fo + r, s, first);
    }
    float az[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) az[r] = dz[(4 * g + r) * kHS + 16 * wave + j];
#pragma unroll
    for (int t = 0; t < NC; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma16(az[r], b.xn[(4 * g + r) * n.xs + 16 * t + j], acc);
      if (16 * t + j < n.K0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) slab_put(slab + L.fc1_w + (fo + r) * n.K0 + 16 * t + j, acc[r], first);
      }
    }
  }
  // ---- feature norm
  f32x4 dn[NC], xh0[NC];
  load_frag<NC>(da, n.xs, j, g, dn);
  load_frag<NC>(b.x0, n.xs, j, g, xh0);
  const float rstd0 = ln_frag<NC>(xh0, n.K0, g);
  if (ACC) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if ((c & 3) == wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sg = row16_sum(dn[c][r] * xh0[c][r]), sb = row16_sum(dn[c][r]);
          const int k = 16 * c + 4 * g + r;
          if (j == 0) {
            if (k < n.K0) { slab_put(slab + L.fn_w + k, sg, first); slab_put(slab + L.fn_b + k, sb, first); }
            else if (k < ((n.K0 + 3) & ~3)) { slab[L.fn_w + k] = 0.f; slab[L.fn_b + k] = 0.f; }   // alignment padding
          }
        }
      }
  }
  if (DX) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = 16 * c + 4 * g;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(v + V_FNW + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hh = (k + r < n.K0) ? dn[c][r] * ga[r] : 0.f;
        dn[c][r] = hh;
        s1 += hh;
        s2 = fmaf(hh, xh0[c][r], s2);
      }
    }
    const float invK = 1.0f / (float)n.K0;
    const float m1 = xg_sum(s1) * invK, m2 = xg_sum(s2) * invK;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) dx[c][r] = (16 * c + 4 * g + r < n.K0) ? rstd0 * (dn[c][r] - m1 - xh0[c][r] * m2) : 0.f;
  }
}

// Target / sampled action of row j from the logits a lane holds (i = 4g + r). mode 0: one-hot of (masked logit == max), ties give
// several ones (util.py:156-175); mode 1: hard gumbel-softmax, straight-through value, `y` = the soft sample (util.py:178-214).
__device__ __forceinline__ f32x4 select_action_frag(f32x4 logit, int A, int g, const float (&avail)[4], const float (&u)[4], int mode,
                                                    f32x4& y) {
  f32x4 vv;
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool on = 4 * g + r < A;
    float t = logit[r];
    if (mode == 1) t += -logf(-logf(u[r] + 1e-20f) + 1e-20f);
    if (avail[r] == 0.f) t = -1e10f;
    vv[r] = on ? t : -3.0e38f;
    mx = fmaxf(mx, vv[r]);
  }
  mx = xg_max(mx);
  f32x4 out;
  y = f32x4{0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (4 * g + r < A && vv[r] == mx) ? 1.f : 0.f;
    return out;
  }
  float den = 0.f;
  f32x4 e;
#pragma unroll
  for (int r = 0; r < 4; ++r) { e[r] = (4 * g + r < A) ? expf(vv[r] - mx) : 0.f; den += e[r]; }
  den = xg_sum(den);
  float ymax = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { y[r] = e[r] / den; ymax = fmaxf(ymax, y[r]); }
  ymax = xg_max(ymax);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool on = 4 * g + r < A;
    const float hard = (on && y[r] == ymax) ? 1.f : 0.f;
    out[r] = on ? (hard - y[r]) + y[r] : 0.f;
  }
  return out;
}

struct TileArgs {
  TNet n0, n1, n2;                // critic update: actor target, critic target, critic live; actor update: actor, critic
  ope_mlp_batch bt;
  NoiseSrc noise;
  int noisy;
  const float* per_w;
  float* prio_out;
  float* slabs;
  int N, A, D, S, B, K, Din, use_huber, tiles;
  int64_t slab_stride;
  int tail;                       // slab offset of [loss_sum, count, q_sum, 0]
  float gamma, huber_delta, per_eps;
  long long* dbg;                 // optional: s_memtime stamps of workgroup 0 / thread 0 (tools/ddpg_phases.py)
  TileOpt opt;                    // optimiser step in the launch's tail (opt.on)
  // LDS offsets (floats)
  int o_xa, o_xna, o_xt, o_xl, o_xnl, o_r1, o_r2, o_r1s, o_a1s, o_r2s, o_r1c, o_r2c, o_dz, o_d1, o_da;
};

// A [16][16 NB] input tile, global -> registers -> LDS in two steps (all loads of a staging phase are issued before the first
// store). Element e = tid + 256u is (row e / wpad, column e % wpad), wpad = 16 NB.
// rows_*: rows [row0, row0 + 16) x [0, W) of a dense [rows][W] matrix, zero for dead rows and columns >= W.
template <int NB>
__device__ __forceinline__ void rows_load(float (&v)[NB], const float* __restrict__ src, int64_t row0, int64_t rows, int W) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int e = threadIdx.x + 256 * u, jj = e / (16 * NB), k = e % (16 * NB);
    const bool ok = k < W && row0 + jj < rows;
    const float t = src[ok ? (row0 + jj) * W + k : 0];      // clamped address + select: no branch around the load
    v[u] = ok ? t : 0.f;
  }
}
template <int NB>
__device__ __forceinline__ void rows_store(const float (&v)[NB], float* dst, int xs) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int e = threadIdx.x + 256 * u, jj = e / (16 * NB), k = e % (16 * NB);
    dst[jj * xs + k] = v[u];
  }
}
// critic input [cent_obs | joint buffer action] of the transitions of rows r0 .. r0 + 15 (row r -> transition r % B; the critic
// update passes rows < B, the actor update its (agent, transition) rows)
template <int NB>
__device__ __forceinline__ void cin_load(float (&v)[NB], const ope_mlp_batch& bt, int64_t r0, int64_t R, int B, int S, int A, int Din) {
  const float invB = 1.0f / (float)B, invA = 1.0f / (float)A;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int e = threadIdx.x + 256 * u, jj = e / (16 * NB), p = e % (16 * NB);
    const int64_t r = r0 + jj;
    const bool ok = r < R && p < Din;
    const int64_t rc = ok ? r : 0;
    const int ag = (int)(((float)rc + 0.5f) * invB), b = (int)(rc - (int64_t)ag * B);
    const int q = p >= S ? p - S : 0, a3 = (int)(((float)q + 0.5f) * invA), i = q - a3 * A;
    const float* src = p < S ? bt.share_obs + (int64_t)b * S + p : bt.acts + ((int64_t)a3 * B + b) * A + i;
    const float t = *(ok ? src : bt.share_obs);
    v[u] = ok ? t : 0.f;
  }
}

// ---- critic update ------------------------------------------------------------------------------------------------------
// ---- optimiser step in the tail of a tile launch (TileOpt) ----------------------------------------------------------------
// Grid barrier over the launch's workgroups (all co-resident: at most one per CU, checked on the host). One 64-bit arrivals counter
// that only ever grows (zero at workspace_init): the ticket a workgroup draws tells it which barrier it is in (ticket / nwg -- every
// barrier takes exactly nwg arrivals), and it waits until the counter has reached that barrier's end. No reset, no generation word,
// nothing to reorder.
// Memory: the eight XCDs' L2s are not coherent with each other. What crosses workgroups here is (a) the gradient slabs, written with
// plain stores by the tile code: one agent-scope RELEASE fence (L2 write-back) by one wave after the workgroup's s_barrier (whose
// s_waitcnt has seen every wave's stores acknowledged by the L2) -- and read with agent-scope relaxed atomic loads, which do not hit
// stale L2 lines; (b) a few floats (partial sums of squares, the tail) published and read with agent-scope relaxed atomics. So the
// polling loop and the readers need no ACQUIRE fence: an L2 invalidate per poll, by 48 workgroups at once, made the first version of
// this tail cost 17 - 42 us (measured) instead of saving the two launches.
// MEASURED (MI355X, config 3, batch 256): with all of that the tail still costs 11 us behind the critic's 16 workgroups and 23 us behind
// the actor's 48, against 9.3 us for the slab-reduction + adam launches it replaces (step 0.1076 vs 0.0894 ms). Every phase that
// crosses workgroups is a round trip through memory (~2 - 3 us on this 8-XCD part: arrive, poll, slab reads, partial sums), and there are
// five of them in a row; a kernel boundary costs less. So the trainer keeps the separate launches by default (MADDPG.update_in_launch,
// OPE_DDPG_OPT_TAIL=1 to try this form); the path stays tested (tests/test_gpu_ddpg.py).
// The spin is bounded: a workgroup that never sees the counter advance raises sync[2] and goes on (wrong numbers and a flag the host
// can read, not a hung GPU).
__device__ __forceinline__ void grid_barrier(unsigned long long* count, int* err, int nwg, bool release) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned long long ticket = __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (ticket / (unsigned long long)nwg + 1ull) * (unsigned long long)nwg;
    int spins = 0;
    while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { *err = 1; break; }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ipow_d(double b, int t) {
  double r = 1.0;
  while (t > 0) { if (t & 1) r *= b; b *= b; t >>= 1; }
  return r;
}
// slabs -> flat gradient (the arithmetic of ddpg_tile_reduce_kernel: eight interleaved partial sums, combined pairwise), sum of squares
// of the optimised elements, clip coefficient, Adam, Polyak: ope_optim.hip's adam_kernel on this workgroup's share of the elements.
// `kind` 0 / 1 = critic / actor launch (their own barrier counters: the two launches have different workgroup counts).
__device__ __forceinline__ void tile_opt_tail(const TileOpt& o, const float* __restrict__ slabs, int64_t stride, int P, int kind, float* red) {
  const int skip_begin = o.skip_begin, skip_end = o.skip_end;
  const int ns = gridDim.x, nthreads = ns * 256, gt = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* count = reinterpret_cast<unsigned long long*>(o.sync) + 1 + kind;      // sync: [error flag, -, critic counter, actor counter]
  grid_barrier(count, o.sync, ns, true);      // every slab is complete and written back
  const int P4 = (P >> 2) + 1;                // float4 chunks of [gradient | tail]
  float sq = 0.f;
  for (int c = gt; c < P4; c += nthreads) {
    const int e = 4 * c;
    const bool skip = e >= skip_begin && e < skip_end;
    f32x4 p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ns; s0 += 8) {
      f32x4 t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float* src = slabs + (int64_t)(s0 + q < ns ? s0 + q : 0) * stride + (skip ? 0 : e);
#pragma unroll
        for (int r = 0; r < 4; ++r) t[q][r] = ld_agent(src + r);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (s0 + q < ns) p[q] += t[q];
    }
    f32x4 v = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    if (skip) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e >= P) {
      v[3] = 0.f;
      st_agent(o.gsq + ns, v[0]); st_agent(o.gsq + ns + 1, v[1]); st_agent(o.gsq + ns + 2, v[2]);      // the tail, for every workgroup
    }
    *reinterpret_cast<f32x4*>(o.grad + e) = v;
    if (e < o.n_opt) sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d, 64);
  if (lane == 0) red[wave] = sq;
  __syncthreads();
  if (threadIdx.x == 0) st_agent(o.gsq + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]));
  grid_barrier(count, o.sync, ns, false);     // the partial sums of squares and the tail are published
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int q = 0; q < ns; ++q) tot += ld_agent(o.gsq + q);      // the same order in every workgroup
    const int t = o.step_counter ? o.step_counter[0] + 1 : o.step;
    red[4] = (float)((double)o.lr / (1.0 - ipow_d((double)o.beta1, t)));
    red[5] = (float)(1.0 / sqrt(1.0 - ipow_d((double)o.beta2, t)));
    reinterpret_cast<int*>(red)[6] = t;
    red[7] = tot;
    red[8] = ld_agent(o.gsq + ns); red[9] = ld_agent(o.gsq + ns + 1); red[10] = ld_agent(o.gsq + ns + 2);
  }
  __syncthreads();
  const float lr_t = red[4], inv_sqrt_bc2 = red[5], tot = red[7];
  const float cnt = red[9], inv = 1.0f / cnt;
  const float norm = sqrtf(tot) * inv;
  const float coef = fminf(1.0f, o.max_norm / (norm + 1e-6f));
  const float scale = coef * inv;
  if (gt == 0 && o.stats) {
    o.stats[0] = red[8] * inv; o.stats[1] = norm; o.stats[2] = red[10] / o.qden; o.stats[3] = cnt;
  }
  for (int c = gt; 4 * c < o.n_opt; c += nthreads) {
    const int e = 4 * c;
    const bool skip = e >= skip_begin && e < skip_end;
    const f32x4 th4 = *reinterpret_cast<const f32x4*>(o.theta + e);
    f32x4 out = th4;
    if (!skip) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(o.grad + e), m4 = *reinterpret_cast<const f32x4*>(o.m + e),
                  v4 = *reinterpret_cast<const f32x4*>(o.v + e);
      f32x4 mo, vo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float gi = g4[q] * scale;
        if (o.wd != 0.f) gi = fmaf(o.wd, th4[q], gi);
        mo[q] = o.beta1 * m4[q] + (1.0f - o.beta1) * gi;
        vo[q] = o.beta2 * v4[q] + (1.0f - o.beta2) * gi * gi;
        const float den = sqrtf(vo[q]) * inv_sqrt_bc2 + o.eps;
        out[q] = th4[q] - lr_t * (mo[q] / den);
      }
      *reinterpret_cast<f32x4*>(o.m + e) = mo;
      *reinterpret_cast<f32x4*>(o.v + e) = vo;
      *reinterpret_cast<f32x4*>(o.theta + e) = out;
    }
    if (o.do_polyak) {
      f32x4 tg4 = *reinterpret_cast<const f32x4*>(o.tgt + e);
#pragma unroll
      for (int q = 0; q < 4; ++q) tg4[q] = tg4[q] * (1.0f - o.tau) + out[q] * o.tau;
      *reinterpret_cast<f32x4*>(o.tgt + e) = tg4;
    }
  }
  // device step counter: the last workgroup to finish publishes t (as adam_kernel does)
  if (o.step_counter && threadIdx.x == 0) {
    if (atomicAdd(&o.step_counter[1], 1) == ns - 1) {
      o.step_counter[1] = 0;
      o.step_counter[0] = reinterpret_cast<int*>(red)[6];
    }
  }
}

// ONE: the launch has one workgroup per tile (tiles <= kMaxTilesPerLaunch: config 3's 16 / 48) -- no tile loop, so nothing is hoisted out of one
// and kept alive across the whole chain (round 6: the looped form of <2, 5> carried 449 spilled SGPRs -- Philox keys, null-pointer masks, row
// masks of the NEXT tile's staging -- through 14.6 k instructions; `first` is a constant here)
template <int NCA, int NCC, bool ONE>
__global__ void __launch_bounds__(256, 1) ddpg_critic_tile_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave: uniform (SGPR)
  const int j = lane & 15, g = lane >> 4;
  const int N = a.N, A = a.A, S = a.S, B = a.B, K = a.K;
  const bool stamp = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
#define OPE_STAMP(i) if (stamp) a.dbg[i] = (long long)__builtin_amdgcn_s_memtime();
  OPE_STAMP(0)
  WFwd<NCA> wa;
  WFwd<NCC> wt, wl;
  WBwd<NCC> wb;
  float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
  float ls = 0.f, cs = 0.f, qs = 0.f;
  bool first = true;
  const int xsa = a.n0.xs, xsc = a.n1.xs;
  constexpr int kAB = 4;                // agents whose inputs are staged in one batch of loads
  // Staging. Inputs: every agent's next observation, and [cent_obs | joint action] of both critics (the target's action block is
  // filled in agent by agent). ALL global loads of the phase are issued before the first LDS store, without branches around
  // them; the weight fragments are requested behind them and land (in order) while the first forward passes run.
  float ia[kAB][NCA], it[NCC], il[NCC];
  auto in_load = [&](int t) {
    const int b0 = t * kRT;
#pragma unroll
    for (int q = 0; q < kAB; ++q) {
      const int64_t ag = q < N ? q : 0;
      rows_load<NCA>(ia[q], a.bt.next_obs, ag * B + b0, ag * B + B, a.D);
    }
    rows_load<NCC>(it, a.bt.next_share_obs, b0, B, S);
    cin_load<NCC>(il, a.bt, b0, B, B, S, A, a.Din);
  };
  auto in_store = [&]() {
#pragma unroll
    for (int q = 0; q < kAB; ++q)
      if (q < N) rows_store<NCA>(ia[q], lds + a.o_xa + q * kRT * xsa, xsa);
    rows_store<NCC>(it, lds + a.o_xt, xsc);
    rows_store<NCC>(il, lds + a.o_xl, xsc);
  };
  int tile = blockIdx.x;                // the launch has at most `tiles` workgroups
  in_load(tile);
  {
    VecRegs v0, v1, v2;
    vecs_load(a.n0, v0); vecs_load(a.n1, v1); vecs_load(a.n2, v2);
    load_wfwd<NCA>(a.n0, wave, j, g, wa);
    OPE_STAMP(20)
    in_store();
    vecs_store(a.n0, v0, lds); vecs_store(a.n1, v1, lds); vecs_store(a.n2, v2, lds);
    OPE_STAMP(22)
  }
  load_wfwd<NCC>(a.n1, wave, j, g, wt);
  load_wfwd<NCC>(a.n2, wave, j, g, wl);
  load_wbwd<NCC>(a.n2, wave, j, g, wb);
  OPE_STAMP(23)
  for (;;) {
    const int b0 = tile * kRT, b = b0 + j;
    const bool live = b < B;
    const int bb = live ? b : 0;
    for (int ag0 = kAB; ag0 < N; ag0 += kAB) {      // more agents than one batch holds
      float ib[kAB][NCA];
#pragma unroll
      for (int q = 0; q < kAB; ++q) {
        const int64_t ag = ag0 + q < N ? ag0 + q : 0;
        rows_load<NCA>(ib[q], a.bt.next_obs, ag * B + b0, ag * B + B, a.D);
      }
#pragma unroll
      for (int q = 0; q < kAB; ++q)
        if (ag0 + q < N) rows_store<NCA>(ib[q], lds + a.o_xa + (ag0 + q) * kRT * xsa, xsa);
    }
    const float rew = a.bt.rewards[bb], den = a.bt.dones_env[bb], wgt = a.per_w ? a.per_w[bb] : 1.0f;
    OPE_STAMP(1)
    for (int ag = 0; ag < N; ++ag) {
      const int64_t row = (int64_t)ag * B + bb;
      float av[4] = {1.f, 1.f, 1.f, 1.f}, uu[4] = {0.5f, 0.5f, 0.5f, 0.5f};
      if (a.bt.next_avail_acts) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < A) av[r] = a.bt.next_avail_acts[row * A + 4 * g + r];
      }
      if (a.noisy) a.noise.at4(row, A, g, uu);
      lds_barrier();                    // inputs staged (first agent) / head partials of the previous agent consumed
      const TBuf tb{lds + a.o_xa + ag * kRT * xsa, nullptr, lds + a.o_r1, nullptr, lds + a.o_r2};
      const f32x4 logit = tile_forward<false, NCA>(a.n0, lds, tb, wave, lane, wa);
      f32x4 y;
      const f32x4 act = select_action_frag(logit, A, g, av, uu, a.noisy ? 1 : 0, y);
      if (wave == (ag & 3)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < A) lds[a.o_xt + j * xsc + S + ag * A + 4 * g + r] = live ? act[r] : 0.f;
      }
      OPE_STAMP(8 + ag)
    }
    lds_barrier();
    OPE_STAMP(2)
    const TBuf tt{lds + a.o_xt, nullptr, lds + a.o_r1, nullptr, lds + a.o_r2};
    const f32x4 qt = tile_forward<false, NCC>(a.n1, lds, tt, wave, lane, wt);
    float qn = 3.0e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < K) qn = fminf(qn, qt[r]);
    qn = -xg_max(-qn);
    OPE_STAMP(3)
    const TBuf tl{lds + a.o_xl, lds + a.o_xnl, lds + a.o_r1s, lds + a.o_a1s, lds + a.o_r2s};
    const f32x4 q = tile_forward<true, NCC>(a.n2, lds, tl, wave, lane, wl);
    OPE_STAMP(4)
    // TD error (maddpg.py:112-157): target = r + gamma (1 - done) min_k Q'_k ; e_k = target - Q_k
    const float target = rew + a.gamma * (1.0f - den) * qn;
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    float fl = 0.f, fa = 0.f, fq = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (live && 4 * g + r < K) {
        const float e = target - q[r];
        float fe, dfe;
        if (a.use_huber) {
          const float x = fabsf(e), dl = a.huber_delta;
          if (x <= dl) { fe = e * e * 0.5f; dfe = e; } else { fe = dl * (x - dl * 0.5f); dfe = dl * (e > 0.f ? 1.f : -1.f); }
        } else {
          fe = e * e;
          dfe = 2.0f * e;
        }
        dq[r] = -dfe * wgt;
        fl += wgt * fe; fa += fabsf(e); fq += q[r];
      }
    fl = xg_sum(fl); fa = xg_sum(fa); fq = xg_sum(fq);
    if (a.prio_out && wave == 0 && g == 0 && live) a.prio_out[b] = fa / (float)K + a.per_eps;
    ls += row16_sum(fl); cs += row16_sum(live ? 1.f : 0.f); qs += row16_sum(fq);
    OPE_STAMP(5)
    f32x4 dxu[NCC];
    tile_backward<true, false, NCC>(a.n2, lds, tl, lds + a.o_dz, lds + a.o_d1, lds + a.o_da, dq, slab, first, wave, lane, wb, dxu);
    OPE_STAMP(6)
    if (ONE) break;
    first = false;
    tile += gridDim.x;
    if (tile >= a.tiles) break;
    in_load(tile);
    lds_barrier();                      // every wave is done with the previous tile's buffers
    in_store();
  }
  if (threadIdx.x == 0) { slab[a.tail] = ls; slab[a.tail + 1] = cs; slab[a.tail + 2] = qs; slab[a.tail + 3] = 0.f; }
  OPE_STAMP(8)
  if (a.opt.on) tile_opt_tail(a.opt, a.slabs, a.slab_stride, a.tail, 0, lds);
  OPE_STAMP(9)
#undef OPE_STAMP
}

// ---- actor update -------------------------------------------------------------------------------------------------------
template <int NCA, int NCC, bool ONE>
__global__ void __launch_bounds__(256, 1) ddpg_actor_tile_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave: uniform (SGPR)
  const int j = lane & 15, g = lane >> 4;
  const int N = a.N, A = a.A, S = a.S, B = a.B;
  const int64_t R = (int64_t)N * B;
  WFwd<NCA> wa;
  WFwd<NCC> wc;
  WBwd<NCC> wcb;
  WBwd<NCA> wab;
  float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
  float ls = 0.f, cs = 0.f, qs = 0.f;
  bool first = true;
  const float invB = 1.0f / (float)B;
  const int xsa = a.n0.xs, xsc = a.n1.xs;
  float ia[NCA], ic[NCC];
  auto in_load = [&](int t) {
    rows_load<NCA>(ia, a.bt.obs, (int64_t)t * kRT, R, a.D);
    cin_load<NCC>(ic, a.bt, (int64_t)t * kRT, R, B, S, A, a.Din);
  };
  auto in_store = [&]() {
    rows_store<NCA>(ia, lds + a.o_xa, xsa);
    rows_store<NCC>(ic, lds + a.o_xl, xsc);
  };
  int tile = blockIdx.x;                // the launch has at most `tiles` workgroups
  in_load(tile);                        // all global loads of the staging phase before the first LDS store; weights behind them
  {
    VecRegs v0, v1;
    vecs_load(a.n0, v0); vecs_load(a.n1, v1);
    load_wfwd<NCA>(a.n0, wave, j, g, wa);
    in_store();
    vecs_store(a.n0, v0, lds); vecs_store(a.n1, v1, lds);
  }
  load_wfwd<NCC>(a.n1, wave, j, g, wc);
  load_wbwd<NCC>(a.n1, wave, j, g, wcb);
  load_wbwd<NCA>(a.n0, wave, j, g, wab);
  for (;;) {
    const int64_t r0 = (int64_t)tile * kRT, row = r0 + j;
    const bool live = row < R;
    const int64_t rr = live ? row : 0;
    const int ag = (int)(((float)rr + 0.5f) * invB);
    float av[4] = {1.f, 1.f, 1.f, 1.f}, uu[4];
    if (a.bt.avail_acts) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < A) av[r] = a.bt.avail_acts[rr * A + 4 * g + r];
    }
    a.noise.at4(rr, A, g, uu);
    const float vld = live ? a.bt.valid_transition[rr] : 0.f;
    lds_barrier();
    const TBuf ta{lds + a.o_xa, lds + a.o_xna, lds + a.o_r1s, lds + a.o_a1s, lds + a.o_r2s};
    const f32x4 logit = tile_forward<true, NCA>(a.n0, lds, ta, wave, lane, wa);
    f32x4 y;
    const f32x4 act = select_action_frag(logit, A, g, av, uu, 1, y);
    // critic input: the agent's own action block replaced by the actor's sample (maddpg.py:207-227)
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < A) lds[a.o_xl + j * xsc + S + ag * A + 4 * g + r] = live ? act[r] : 0.f;
    }
    lds_barrier();
    const TBuf tc{lds + a.o_xl, nullptr, lds + a.o_r1c, nullptr, lds + a.o_r2c};
    const f32x4 q = tile_forward<false, NCC>(a.n1, lds, tc, wave, lane, wc);
    const float q1 = __shfl(q[0], j, 64);
    ls += row16_sum(-q1 * vld); cs += row16_sum(vld); qs += row16_sum(q1 * vld);     // loss = -sum(Q_1 valid) / sum(valid) (maddpg.py:229-232)
    f32x4 dh = {0.f, 0.f, 0.f, 0.f};
    if (g == 0) dh[0] = -vld;
    f32x4 dxc[NCC];
    tile_backward<false, true, NCC>(a.n1, lds, tc, lds + a.o_dz, lds + a.o_d1, lds + a.o_da, dh, nullptr, true, wave, lane, wcb, dxc);
    // adjoint of the agent's own action block -> straight-through gumbel adjoint: dlogit_i = y_i (dx_i - sum_m dx_m y_m)
    lds_barrier();                      // every wave has read d xn out of `da`
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < NCC; ++c) *reinterpret_cast<f32x4*>(lds + a.o_da + j * xsc + 16 * c + 4 * g) = dxc[c];
    }
    lds_barrier();
    f32x4 dxi = {0.f, 0.f, 0.f, 0.f};
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < A) {
        dxi[r] = lds[a.o_da + j * xsc + S + ag * A + 4 * g + r];
        dot = fmaf(dxi[r], y[r], dot);
      }
    dot = xg_sum(dot);
    f32x4 dl;
#pragma unroll
    for (int r = 0; r < 4; ++r) dl[r] = (4 * g + r < A) ? y[r] * (dxi[r] - dot) : 0.f;
    lds_barrier();                      // `da` is free again
    f32x4 dxa[NCA];
    tile_backward<true, false, NCA>(a.n0, lds, ta, lds + a.o_dz, lds + a.o_d1, lds + a.o_da, dl, slab, first, wave, lane, wab, dxa);
    if (ONE) break;
    first = false;
    tile += gridDim.x;
    if (tile >= a.tiles) break;
    in_load(tile);
    lds_barrier();                      // every wave is done with the previous tile's buffers
    in_store();
  }
  if (threadIdx.x == 0) { slab[a.tail] = ls; slab[a.tail + 1] = cs; slab[a.tail + 2] = qs; slab[a.tail + 3] = 0.f; }
  if (a.opt.on) tile_opt_tail(a.opt, a.slabs, a.slab_stride, a.tail, 1, lds);
}

// ---- slabs -> flat gradient -----------------------------------------------------------------------------------------------
// Workgroup = 64 consecutive output elements (lane = element: every load is one 256-byte run of a slab) x four waves: wave w forms
// the partial sums 2 w and 2 w + 1 of EIGHT interleaved ones (partial q = slabs q, q + 8, ... in that order), all of a lane's loads in
// flight before its first add, and the four pairs meet through LDS as ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)) -- the
// arithmetic of the first version (8 adjacent lanes per element and three xor-shuffles: 32-byte pieces of 8 slabs per load, the slab
// walk as a load / wait / add loop) and of tile_opt_tail, bit for bit. The registered-but-unused fc_h block [skip_begin, skip_end) is
// never written by the tile kernels: zero. Each workgroup also leaves the sum of squares of the gradient elements it produced --
// gsq[block] for the trunk, gsq[nb + block] for the head block (frozen upstream for the critic, SURVEY A-4, so the optimiser may want
// the trunk alone): the clip norm of ope_adam_step without a pass of its own over the gradient.
__global__ void __launch_bounds__(256) ddpg_tile_reduce_kernel(const float* __restrict__ slabs, int ns, int64_t stride, int P, int tail,
                                                                int skip_begin, int skip_end, int head_begin, float* __restrict__ grad,
                                                                float* __restrict__ gsq) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const bool live = e < P + 4;
  const bool skip = e >= skip_begin && e < skip_end;
  const int ee = (live && !skip) ? e : 0;
  float p0 = 0.f, p1 = 0.f;
  for (int s0 = 2 * wave; s0 < ns; s0 += 32) {
    float t0[4], t1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sa = s0 + 8 * u;
      t0[u] = sa < ns ? slabs[(int64_t)sa * stride + ee] : 0.f;
      t1[u] = sa + 1 < ns ? slabs[(int64_t)(sa + 1) * stride + ee] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { p0 += t0[u]; p1 += t1[u]; }
  }
  part[wave][lane] = p0 + p1;
  __syncthreads();
  if (wave != 0) return;
  float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  float sq_trunk = 0.f, sq_head = 0.f;
  if (live) {
    if (skip) v = 0.f;
    if (e >= P) {
      grad[tail + (e - P)] = e - P < 3 ? v : 0.f;
    } else {
      grad[e] = v;
      if (e >= head_begin) sq_head = v * v; else sq_trunk = v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    sq_trunk += __shfl_xor(sq_trunk, o, 64);
    sq_head += __shfl_xor(sq_head, o, 64);
  }
  if (lane == 0) {
    gsq[blockIdx.x] = sq_trunk;
    gsq[gridDim.x + blockIdx.x] = sq_head;
  }
}

int bucket_a(int D) { const int nc = ope_cdiv(D, 16); return nc <= 2 ? 2 : (nc <= 4 ? 4 : 8); }
int bucket_c(int Din) { return ope_cdiv(Din, 16) <= 5 ? 5 : 8; }
int stride_for(int nc) { const int w = 16 * nc / 4; return 4 * (w | 1); }      // 4 x odd >= 16 nc

TNet make_net(const float* th, int K0, int Hout, int vec, int nc) {
  TNet n;
  n.th = th; n.K0 = K0; n.Hout = Hout; n.nc0 = nc; n.vec = vec; n.xs = stride_for(nc);
  return n;
}
constexpr int kMaxTilesPerLaunch = 1024;    // more tiles than this are walked by a grid-stride loop (slabs stay small)

int slab_len(int D, int A, int Din, int K) {
  const int pa = ope_agent_layout_mlp(D, A, 0).end, pc = ope_agent_layout_mlp(Din, K, 0).end;
  return (pa > pc ? pa : pc) + 4;
}

// LDS plans (float offsets into the dynamic segment); return the total
int plan_critic(TileArgs& a, const float* tat, const float* tct, const float* tc) {
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  const int nca = bucket_a(a.D), ncc = bucket_c(a.Din);
  a.n0 = make_net(tat, a.D, a.A, take(V_SIZE), nca);
  a.n1 = make_net(tct, a.Din, a.K, take(V_SIZE), ncc);
  a.n2 = make_net(tc, a.Din, a.K, take(V_SIZE), ncc);
  const int xa = kRT * a.n0.xs, xc = kRT * a.n1.xs;
  a.o_xa = take(a.N * xa); a.o_xt = take(xc); a.o_xl = take(xc); a.o_xnl = take(xc);
  a.o_r1 = take(kRT * kHS); a.o_r2 = take(kRT * kHS); a.o_r1s = take(kRT * kHS); a.o_a1s = take(kRT * kHS); a.o_r2s = take(kRT * kHS);
  a.o_dz = take(kRT * kHS); a.o_d1 = take(kRT * kHS); a.o_da = take(xc);
  return o;
}
int plan_actor(TileArgs& a, const float* ta, const float* tc) {
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  const int nca = bucket_a(a.D), ncc = bucket_c(a.Din);
  a.n0 = make_net(ta, a.D, a.A, take(V_SIZE), nca);
  a.n1 = make_net(tc, a.Din, a.K, take(V_SIZE), ncc);
  const int xa = kRT * a.n0.xs, xc = kRT * a.n1.xs;
  a.o_xa = take(xa); a.o_xna = take(xa); a.o_xl = take(xc);
  a.o_r1s = take(kRT * kHS); a.o_a1s = take(kRT * kHS); a.o_r2s = take(kRT * kHS); a.o_r1c = take(kRT * kHS); a.o_r2c = take(kRT * kHS);
  a.o_dz = take(kRT * kHS); a.o_d1 = take(kRT * kHS); a.o_da = take(xc > xa ? xc : xa);
  return o;
}
void fill_dims(TileArgs& a, int N, int A, int D, int S, int K, int B) {
  memset(&a, 0, sizeof(a));
  a.N = N; a.A = A; a.D = D; a.S = S; a.B = B; a.K = K; a.Din = S + N * A;
}

}  // namespace

bool ddpg_fused_ok(int N, int A, int D, int S, int K) {
  static const int on = getenv("OPE_DDPG_FUSED") ? atoi(getenv("OPE_DDPG_FUSED")) : 1;
  if (!(on && D <= 128 && S + N * A <= 128 && S >= 1 && A <= kHO && K <= kHO && A >= 1 && K >= 1)) return false;
  auto same = [](int K0, int Hout) {
    const AgentLayout R = ope_agent_layout_mlp(K0, Hout, 0);
    const TL L = tl_of(K0, Hout);
    return L.fn_w == R.fn_w && L.fn_b == R.fn_b && L.fc1_w == R.fc1_w && L.fc1_b == R.fc1_b && L.ln1_w == R.ln1_w && L.ln1_b == R.ln1_b &&
           L.fc2_w == R.fc2_w && L.fc2_b == R.fc2_b && L.ln2_w == R.ln2_w && L.ln2_b == R.ln2_b && L.q_w == R.q_w && L.q_b == R.q_b &&
           L.end == R.end;
  };
  if (!same(D, A) || !same(S + N * A, K)) return false;      // the device-side offset arithmetic must be the flat layout's
  TileArgs a;
  fill_dims(a, N, A, D, S, K, kRT);
  return (size_t)plan_critic(a, nullptr, nullptr, nullptr) * sizeof(float) <= 160 * 1024;     // many agents: their staged inputs
}

static int device_cus() {
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  return cus;
}
bool ddpg_tile_opt_ok(int N, int A, int D, int S, int K, int B) {
  return ddpg_fused_ok(N, A, D, S, K) && B >= 1 && ope_cdiv((int64_t)N * B, kRT) <= device_cus();      // one workgroup per CU: all co-resident
}

int ddpg_fused_gsq_blocks(int N, int A, int D, int S, int K, bool critic) {
  const int P = critic ? ope_agent_layout_mlp(S + N * A, K, 0).end : ope_agent_layout_mlp(D, A, 0).end;
  return ope_cdiv((int64_t)(P + 4), 64);      // workgroups of the slab-sum launch (64 elements each)
}

int64_t ddpg_fused_slab_floats(int N, int A, int D, int S, int K, int B) {
  int64_t tiles = ope_cdiv((int64_t)N * B, kRT);
  if (tiles > kMaxTilesPerLaunch) tiles = kMaxTilesPerLaunch;
  return tiles * slab_len(D, A, S + N * A, K);
}

static int launch_reduce(const float* slabs, int ns, int64_t stride, const AgentLayout& L, float* grad, float* gsq, hipStream_t st) {
  const int P = L.end;
  kprof_work(0.0, 4.0 * ((double)ns * stride + P));
  OPE_LAUNCH(ddpg_tile_reduce_kernel, dim3(ope_cdiv((int64_t)(P + 4), 64)), dim3(256), 0, st, slabs, ns, stride, P, P, L.fch_w,
                     L.fc2_w, L.q_w, grad, gsq);
  return hipGetLastError() == hipSuccess ? OPE_OK : OPE_ELAUNCH;
}

template <typename KERN>
static int launch_tile(KERN kern, const TileArgs& a, int blocks, size_t lds, hipStream_t st) {
  if (lds > 160 * 1024) return OPE_EINVAL;
  // > 64 KB of dynamic LDS needs the attribute, per kernel instantiation; it is set when the request grows (a runtime call per
  // launch is measurable on a step whose whole host side is ~120 us)
  static int granted[16];
  static const void* owner[16];
  int slot = 0;
  while (slot < 15 && owner[slot] && owner[slot] != (const void*)kern) ++slot;
  if (owner[slot] != (const void*)kern) { owner[slot] = (const void*)kern; granted[slot] = 0; }
  if ((int)lds > granted[slot]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return OPE_ELAUNCH;
    granted[slot] = (int)lds;
  }
  OPE_LAUNCH(kern, dim3(blocks), dim3(256), lds, st, a);
  return hipGetLastError() == hipSuccess ? OPE_OK : OPE_ELAUNCH;
}
#define OPE_TILE_DISPATCH1(KERNEL, ONE, nca, ncc, ...)                                  \
  ((nca) == 2 ? ((ncc) == 5 ? launch_tile(KERNEL<2, 5, ONE>, __VA_ARGS__) : launch_tile(KERNEL<2, 8, ONE>, __VA_ARGS__))    \
   : (nca) == 4 ? ((ncc) == 5 ? launch_tile(KERNEL<4, 5, ONE>, __VA_ARGS__) : launch_tile(KERNEL<4, 8, ONE>, __VA_ARGS__))  \
                : ((ncc) == 5 ? launch_tile(KERNEL<8, 5, ONE>, __VA_ARGS__) : launch_tile(KERNEL<8, 8, ONE>, __VA_ARGS__)))
#define OPE_TILE_DISPATCH(KERNEL, one, nca, ncc, ...) ((one) ? OPE_TILE_DISPATCH1(KERNEL, true, nca, ncc, __VA_ARGS__) : OPE_TILE_DISPATCH1(KERNEL, false, nca, ncc, __VA_ARGS__))

int launch_ddpg_critic_fused(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor_tgt, const float* theta_critic,
                             const float* theta_critic_tgt, const float* U, const float* per_w, float* slabs, float* grad, float* prio_out,
                             float* gsq, hipStream_t st, const TileOpt* opt) {
  const ope_dims& d = cfg->dims;
  TileArgs a;
  fill_dims(a, d.n_agents, d.act_dim, d.obs_dim, d.state_dim, cfg->num_q, cfg->batch);
  const int lds = plan_critic(a, theta_actor_tgt, theta_critic_tgt, theta_critic);
  if (opt) {
    if (!ddpg_tile_opt_ok(a.N, a.A, a.D, a.S, a.K, a.B)) return OPE_EINVAL;
    const AgentLayout L = ope_agent_layout_mlp(a.Din, a.K, 0);
    a.opt = *opt; a.opt.on = 1; a.opt.grad = grad; a.opt.gsq = gsq; a.opt.skip_begin = L.fch_w; a.opt.skip_end = L.fc2_w;
    if (a.opt.n_opt < 4 || a.opt.n_opt > L.end || (a.opt.n_opt & 3)) return OPE_EINVAL;
  }
  a.bt = *bt; a.noisy = cfg->target_gumbel; a.noise = NoiseSrc{U, cfg->noise_seed, cfg->noise_counter, 0};
  a.per_w = cfg->use_per ? per_w : nullptr; a.prio_out = prio_out; a.slabs = slabs;
  a.use_huber = cfg->use_huber; a.gamma = cfg->gamma; a.huber_delta = cfg->huber_delta; a.per_eps = cfg->per_eps;
  a.tiles = ope_cdiv(a.B, kRT);
  a.slab_stride = slab_len(a.D, a.A, a.Din, a.K);
  a.tail = ope_agent_layout_mlp(a.Din, a.K, 0).end;
  static const bool dbg_on = getenv("OPE_DDPG_DBG") != nullptr;
  a.dbg = dbg_on ? reinterpret_cast<long long*>(slabs + ddpg_fused_slab_floats(a.N, a.A, a.D, a.S, a.K, a.B)) : nullptr;
  const int blocks = a.tiles < kMaxTilesPerLaunch ? a.tiles : kMaxTilesPerLaunch;
  {   // GEMM-shaped work of the launch: N target-actor passes, target critic, live critic forward, its weight gradients and hidden adjoints
    const double am = (double)a.D * OPE_H + OPE_H * OPE_H + (double)OPE_H * a.A, cm = (double)a.Din * OPE_H + OPE_H * OPE_H + (double)OPE_H * a.K;
    kprof_work(2.0 * ((double)a.N * a.B * am + 2.0 * a.B * cm + a.B * (cm + OPE_H * OPE_H + (double)OPE_H * a.K)));
  }
  const int rc = OPE_TILE_DISPATCH(ddpg_critic_tile_kernel, blocks == a.tiles, a.n0.nc0, a.n1.nc0, a, blocks, (size_t)lds * sizeof(float), st);
  if (rc || opt) return rc;
  return launch_reduce(slabs, blocks, a.slab_stride, ope_agent_layout_mlp(a.Din, a.K, 0), grad, gsq, st);
}

int launch_ddpg_actor_fused(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor, const float* theta_critic,
                            const float* U, float* slabs, float* grad, float* gsq, hipStream_t st, const TileOpt* opt) {
  const ope_dims& d = cfg->dims;
  TileArgs a;
  fill_dims(a, d.n_agents, d.act_dim, d.obs_dim, d.state_dim, cfg->num_q, cfg->batch);
  const int lds = plan_actor(a, theta_actor, theta_critic);
  if (opt) {
    if (!ddpg_tile_opt_ok(a.N, a.A, a.D, a.S, a.K, a.B)) return OPE_EINVAL;
    const AgentLayout L = ope_agent_layout_mlp(a.D, a.A, 0);
    a.opt = *opt; a.opt.on = 1; a.opt.grad = grad; a.opt.gsq = gsq; a.opt.skip_begin = L.fch_w; a.opt.skip_end = L.fc2_w;
    if (a.opt.n_opt < 4 || a.opt.n_opt > L.end || (a.opt.n_opt & 3)) return OPE_EINVAL;
  }
  a.bt = *bt; a.noisy = 1; a.noise = NoiseSrc{U, cfg->noise_seed, cfg->noise_counter, 1}; a.slabs = slabs;
  a.tiles = ope_cdiv((int64_t)a.N * a.B, kRT);
  a.slab_stride = slab_len(a.D, a.A, a.Din, a.K);
  a.tail = ope_agent_layout_mlp(a.D, a.A, 0).end;
  const int blocks = a.tiles < kMaxTilesPerLaunch ? a.tiles : kMaxTilesPerLaunch;
  {   // actor forward, critic forward on the substituted joint action, critic adjoint down to the action block, actor weight gradients + adjoints
    const double am = (double)a.D * OPE_H + OPE_H * OPE_H + (double)OPE_H * a.A, cm = (double)a.Din * OPE_H + OPE_H * OPE_H + (double)OPE_H * a.K;
    kprof_work(2.0 * (double)a.N * a.B * (am + cm + ((double)OPE_H * a.K + OPE_H * OPE_H + (double)OPE_H * a.A) + am + ((double)a.A * OPE_H + OPE_H * OPE_H)));
  }
  const int rc = OPE_TILE_DISPATCH(ddpg_actor_tile_kernel, blocks == a.tiles, a.n0.nc0, a.n1.nc0, a, blocks, (size_t)lds * sizeof(float), st);
  if (rc || opt) return rc;
  return launch_reduce(slabs, blocks, a.slab_stride, ope_agent_layout_mlp(a.D, a.A, 0), grad, gsq, st);
}

}  // namespace ope
