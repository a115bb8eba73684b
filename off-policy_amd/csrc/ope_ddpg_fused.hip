// Fused MLP MADDPG / MATD3 update for small networks (MPE-sized: input widths <= 128, <= 8 actions): the whole critic
// update -- target actor, target action, target critic, live critic, TD error, critic backward, weight-gradient
// partials -- is ONE launch, and so is the whole actor update (actor, hard gumbel-softmax, critic on the substituted joint
// action, critic input adjoint, straight-through adjoint, actor backward, weight-gradient partials).
//
// Replaces (reference): MADDPG.shared_train_policy_on_batch, offpolicy/algorithms/maddpg/maddpg.py:90-249 (+ get_update_info
// 38-81, MADDPG_Actor / MADDPG_Critic actor_critic.py:7-87, gumbel_softmax / onehot_from_logits util.py:156-214). The
// unfused path (ope_ddpg.hip: ~14 launches of 4-8 us per network update through the shared trunk / wgrad / finalize
// kernels) stays for wide inputs; here everything of an update is row-local except the sum over rows of the weight
// gradients, so the structure is:
//   * one WAVE per data row (a transition for the critic update, an (agent, transition) pair for the actor update);
//     lane j owns hidden unit j of the 64-wide layers and input columns j, j + 64; a row's activations, LayerNorm
//     statistics and ReLU masks never leave the wave's registers between forward and backward;
//   * mat-vecs are "readlane broadcast x LDS weight" FMA loops (v_readlane -> SGPR operand, one ds_read_b32 per FMA);
//     the networks' weights sit in LDS in their original [out][in] orientation with ODD row strides, which makes both
//     access patterns conflict-free (lane = out row for the forward product, lane = in column for the adjoint);
//   * the row's outer-product contributions to every weight gradient accumulate in the wave's registers (<= 96 + 64 + ...
//     accumulators: one wave per SIMD, the 512-entry register file is there); at the end the 4 waves of a workgroup dump
//     them to LDS, the workgroup sums them in a fixed order and writes ONE compact slab;
//   * a small second launch sums the slabs in a fixed order into the flat gradient (+ the [loss_sum, count, q_sum] tail).
// No atomics: bitwise deterministic. 64 FLOP/clk/SIMD f32 VALU only -- at 256-768 rows of 64-wide layers (91 MFLOP per
// update) the update is bound by launch count and dependent-instruction latency, not by any throughput roof.
#include <stdlib.h>
#include <string.h>

#include "ope_ddpg.h"

namespace ope {
namespace {

constexpr int kWaves = 4;
constexpr int kHB = 8;        // head outputs (actions / q heads) supported by the fused path
constexpr float kEps = OPE_LN_EPS;

// One MLP net (feature-norm -> fc1 -> ReLU -> LN -> fc2 -> ReLU -> LN -> Linear head) as laid out in LDS (weights, padded odd
// strides) or in a compact gradient slab (dense strides). Offsets in floats relative to `base`.
struct NetImg {
  int Din, Hout, s1, s2;   // input width, head outputs, row strides of fc1_w / fc2_w
  int base;
  int fn_w, fn_b, fc1_w, fc1_b, ln1_w, ln1_b, fc2_w, fc2_b, ln2_w, ln2_b, q_w, q_b, size;
};
NetImg make_img(int Din, int Hout, int base, bool padded) {
  NetImg n;
  n.Din = Din; n.Hout = Hout; n.s1 = padded ? (Din | 1) : Din; n.s2 = padded ? (OPE_H + 1) : OPE_H; n.base = base;
  int o = 0;
  auto take = [&](int k) { int r = o; o += (k + 3) & ~3; return r; };   // 16-byte aligned tensors
  n.fn_w = take(Din); n.fn_b = take(Din); n.fc1_w = take(OPE_H * n.s1); n.fc1_b = take(OPE_H); n.ln1_w = take(OPE_H); n.ln1_b = take(OPE_H);
  n.fc2_w = take(OPE_H * n.s2); n.fc2_b = take(OPE_H); n.ln2_w = take(OPE_H); n.ln2_b = take(OPE_H);
  n.q_w = take(Hout * OPE_H); n.q_b = take(Hout);
  n.size = (o + 3) & ~3;
  return n;
}

__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wsum(float v) {          // fixed order, uniform result
  v = row16_sum(v);
  return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
__device__ __forceinline__ float wmax(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
}
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// z += sum_{k < n} wrow[k * stride] * x[k], x[k] broadcast from lane k of xv. The LDS reads of a chunk of 8 are issued
// together (a ds_read_b32 returns after ~64+ cycles: one read per dependent FMA would expose that latency 133 times per
// layer pass), two accumulators halve the dependent FMA chain.
__device__ __forceinline__ float dot_bcast(const float* wrow, int stride, float xv, int n, float z0) {
  float z1 = 0.f;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = wrow[(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      z0 = fmaf(wv[u], rl(xv, k + u), z0);
      z1 = fmaf(wv[u + 1], rl(xv, k + u + 1), z1);
    }
  }
  for (; k < n; ++k) z0 = fmaf(wrow[k * stride], rl(xv, k), z0);
  return z0 + z1;
}

// theta (flat, AgentLayout_mlp offsets, dense [out][in]) -> LDS image (all 256 threads of the workgroup). EVERY global load
// of the net (<= 8 + 4 sixteen-byte loads of the two matrices and a handful of scalars per thread) is issued before the
// first LDS store: a workgroup stages ~100 KB of weights, and a loop that loads and stores per iteration pays one L2 round
// trip (~2 k cycles) per iteration -- 60 us of the first version's 66.
struct StageRegs {
  f32x4 v1[8], v2[4];
  float g0, b0, s0, s1v, h0, h1, hb;
};
__device__ __forceinline__ void stage_load(const float* __restrict__ th, const AgentLayout& L, const NetImg& n, StageRegs& r) {
  const int tid = threadIdx.x;
  constexpr int NT = kWaves * 64;
  const int n1 = (OPE_H * n.Din) >> 2;                 // float4 pieces of fc1_w (Din <= 128: <= 2048 = 8 per thread)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int q = tid + u * NT;
    r.v1[u] = *reinterpret_cast<const f32x4*>(th + L.fc1_w + 4 * (q < n1 ? q : 0));
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) r.v2[u] = *reinterpret_cast<const f32x4*>(th + L.fc2_w + 4 * (tid + u * NT));
  const int k0 = tid < n.Din ? tid : 0;                  // Din <= 128 < NT
  r.g0 = th[L.fn_w + k0]; r.b0 = th[L.fn_b + k0];
  const int j = tid & 63, which = tid >> 6;              // 4 waves x 64: six 64-vectors
  const int off6[6] = {L.fc1_b, L.ln1_w, L.ln1_b, L.fc2_b, L.ln2_w, L.ln2_b};
  r.s0 = th[off6[which] + j];
  r.s1v = which < 2 ? th[off6[4 + which] + j] : 0.f;
  const int hq = n.Hout * OPE_H;                          // Hout <= 8: hq <= 512 = 2 per thread
  r.h0 = tid < hq ? th[L.q_w + tid] : 0.f; r.h1 = tid + NT < hq ? th[L.q_w + tid + NT] : 0.f;
  r.hb = tid < n.Hout ? th[L.q_b + tid] : 0.f;
}
__device__ __forceinline__ void stage_store(const NetImg& n, const StageRegs& r, float* lds) {
  float* d = lds + n.base;
  const int tid = threadIdx.x;
  constexpr int NT = kWaves * 64;
  const int n1 = (OPE_H * n.Din) >> 2;
  const float invD = 1.0f / (float)n.Din;
  if (n.s1 == n.Din) {
    // odd input width: the dense row stride is already conflict-free for both access patterns -> straight 16-byte copies
    // (one ds_write_b128 per piece instead of four scalar stores with index arithmetic: with ONE wave per SIMD every
    // instruction costs ~5 cycles of issue, and the scatter was 10 k of the kernel's 75 k cycles)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = tid + u * NT;
      if (q < n1) *reinterpret_cast<f32x4*>(d + n.fc1_w + 4 * q) = r.v1[u];
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = tid + u * NT;
      if (q < n1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = 4 * q + c;
          const int jj = fdiv(e, invD), k = e - jj * n.Din;
          d[n.fc1_w + jj * n.s1 + k] = r.v1[u][c];
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = 4 * (tid + u * NT);
#pragma unroll
    for (int c = 0; c < 4; ++c) d[n.fc2_w + ((e + c) >> 6) * n.s2 + ((e + c) & 63)] = r.v2[u][c];
  }
  const int j = tid & 63, which = tid >> 6;
  const int dst6[6] = {n.fc1_b, n.ln1_w, n.ln1_b, n.fc2_b, n.ln2_w, n.ln2_b};
  const int hq = n.Hout * OPE_H;
  if (tid < n.Din) { d[n.fn_w + tid] = r.g0; d[n.fn_b + tid] = r.b0; }
  d[dst6[which] + j] = r.s0;
  if (which < 2) d[dst6[4 + which] + j] = r.s1v;
  if (tid < hq) d[n.q_w + tid] = r.h0;
  if (tid + NT < hq) d[n.q_w + tid + NT] = r.h1;
  if (tid < n.Hout) d[n.q_b + tid] = r.hb;
}

// What the backward pass needs of one row's forward pass (all in the wave's registers).
struct RowSave {
  float xn_a, xn_b;     // LN0 output (normalised * gamma + beta) of input columns lane, lane + 64 (0 beyond Din)
  float xh_a, xh_b;     // LN0 normalised values
  float rstd0;
  float xh1, rstd1, a1; // LN1: normalised value, 1/std, output of hidden unit `lane`
  float xh2, rstd2, a2;
  bool p1, p2;          // ReLU of fc1 / fc2 on
};

// Forward of one row. xa / xb: raw input columns lane, lane + 64 (anything beyond Din is ignored). Returns in lane i < Hout the
// head output i.
__device__ __forceinline__ float net_forward(const float* lds, const NetImg& n, float xa, float xb, int lane, RowSave& s) {
  const float* w = lds + n.base;
  const int Din = n.Din;
  const bool va = lane < Din, vb = lane + 64 < Din;
  const float invD = 1.0f / (float)Din;
  const float mu = wsum((va ? xa : 0.f) + (vb ? xb : 0.f)) * invD;
  const float da = va ? xa - mu : 0.f, db = vb ? xb - mu : 0.f;
  const float rs0 = 1.0f / sqrtf(wsum(da * da + db * db) * invD + kEps);
  s.rstd0 = rs0;
  s.xh_a = da * rs0; s.xh_b = db * rs0;
  s.xn_a = va ? fmaf(s.xh_a, w[n.fn_w + lane], w[n.fn_b + lane]) : 0.f;
  s.xn_b = vb ? fmaf(s.xh_b, w[n.fn_w + lane + 64], w[n.fn_b + lane + 64]) : 0.f;
  // fc1
  float z = w[n.fc1_b + lane];
  const float* w1 = w + n.fc1_w + lane * n.s1;
  const int k64 = Din < 64 ? Din : 64;
  z = dot_bcast(w1, 1, s.xn_a, k64, z);
  if (Din > 64) z = dot_bcast(w1 + 64, 1, s.xn_b, Din - 64, z);
  s.p1 = z > 0.f;
  float r = fmaxf(z, 0.f);
  float m = wsum(r) * (1.0f / OPE_H);
  float d = r - m;
  s.rstd1 = 1.0f / sqrtf(wsum(d * d) * (1.0f / OPE_H) + kEps);
  s.xh1 = d * s.rstd1;
  s.a1 = fmaf(s.xh1, w[n.ln1_w + lane], w[n.ln1_b + lane]);
  // fc2
  z = w[n.fc2_b + lane];
  const float* w2 = w + n.fc2_w + lane * n.s2;
  z = dot_bcast(w2, 1, s.a1, OPE_H, z);
  s.p2 = z > 0.f;
  r = fmaxf(z, 0.f);
  m = wsum(r) * (1.0f / OPE_H);
  d = r - m;
  s.rstd2 = 1.0f / sqrtf(wsum(d * d) * (1.0f / OPE_H) + kEps);
  s.xh2 = d * s.rstd2;
  s.a2 = fmaf(s.xh2, w[n.ln2_w + lane], w[n.ln2_b + lane]);
  // head
  float hv = 0.f;
  for (int i = 0; i < n.Hout; ++i) {
    const float v = wsum(w[n.q_w + i * OPE_H + lane] * s.a2) + w[n.q_b + i];
    if (lane == i) hv = v;
  }
  return hv;
}

// Per-wave gradient accumulators of one net: lane j holds row j of dW1 / dW2, column j of the head weight gradients, entry j
// of the 64-long vectors, and entries j, j + 64 of the input LayerNorm's.
template <int KB>
struct Acc {
  float W1[KB], W2[OPE_H], Wh[kHB];
  float b1, g1, be1, b2, g2, be2, bh, g0a, g0b, be0a, be0b;
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int k = 0; k < KB; ++k) W1[k] = 0.f;
#pragma unroll
    for (int k = 0; k < OPE_H; ++k) W2[k] = 0.f;
#pragma unroll
    for (int k = 0; k < kHB; ++k) Wh[k] = 0.f;
    b1 = g1 = be1 = b2 = g2 = be2 = bh = g0a = g0b = be0a = be0b = 0.f;
  }
};

// Backward of one row. dhead: lane i < Hout holds d loss / d head_i (0 elsewhere). ACC: accumulate the parameter gradients;
// DX: also return the adjoint of the RAW input columns lane, lane + 64 in dxa / dxb.
template <int KB, bool ACC, bool DX>
__device__ __forceinline__ void net_backward(const float* lds, const NetImg& n, const RowSave& s, float dhead, int lane, Acc<KB>& g,
                                             float& dxa, float& dxb) {
  const float* w = lds + n.base;
  const int Din = n.Din;
  // head
  float da2 = 0.f;
  for (int i = 0; i < n.Hout; ++i) {
    const float di = rl(dhead, i);
    da2 = fmaf(di, w[n.q_w + i * OPE_H + lane], da2);
  }
  if (ACC) {
#pragma unroll
    for (int i = 0; i < kHB; ++i)
      if (i < n.Hout) g.Wh[i] = fmaf(rl(dhead, i), s.a2, g.Wh[i]);
    g.bh += dhead;
  }
  // LN2 -> ReLU -> fc2
  if (ACC) { g.g2 = fmaf(da2, s.xh2, g.g2); g.be2 += da2; }
  float dxh = da2 * w[n.ln2_w + lane];
  float m1 = wsum(dxh) * (1.0f / OPE_H), m2 = wsum(dxh * s.xh2) * (1.0f / OPE_H);
  const float dz2 = s.p2 ? s.rstd2 * (dxh - m1 - s.xh2 * m2) : 0.f;
  if (ACC) {
    g.b2 += dz2;
#pragma unroll
    for (int k = 0; k < OPE_H; ++k) g.W2[k] = fmaf(dz2, rl(s.a1, k), g.W2[k]);
  }
  float da1 = 0.f;
  {
    da1 = dot_bcast(w + n.fc2_w + lane, n.s2, dz2, OPE_H, 0.f);   // column `lane`
  }
  // LN1 -> ReLU -> fc1
  if (ACC) { g.g1 = fmaf(da1, s.xh1, g.g1); g.be1 += da1; }
  dxh = da1 * w[n.ln1_w + lane];
  m1 = wsum(dxh) * (1.0f / OPE_H);
  m2 = wsum(dxh * s.xh1) * (1.0f / OPE_H);
  const float dz1 = s.p1 ? s.rstd1 * (dxh - m1 - s.xh1 * m2) : 0.f;
  if (ACC) {
    g.b1 += dz1;
#pragma unroll
    for (int k = 0; k < KB; ++k) g.W1[k] = fmaf(dz1, k < 64 ? rl(s.xn_a, k) : rl(s.xn_b, k - 64), g.W1[k]);   // xn = 0 beyond Din
  }
  // adjoint of the LN0 output: columns lane, lane + 64
  const bool va = lane < Din, vb = lane + 64 < Din;
  float dna = 0.f, dnb = 0.f;
  {
    const float* w1a = w + n.fc1_w + (va ? lane : 0);
    const float* w1b = w + n.fc1_w + (vb ? lane + 64 : 0);
    dna = dot_bcast(w1a, n.s1, dz1, OPE_H, 0.f);
    if (Din > 64) dnb = dot_bcast(w1b, n.s1, dz1, OPE_H, 0.f);
    if (!va) dna = 0.f;
    if (!vb) dnb = 0.f;
  }
  if (ACC) {
    g.g0a = fmaf(dna, s.xh_a, g.g0a); g.be0a += dna;
    g.g0b = fmaf(dnb, s.xh_b, g.g0b); g.be0b += dnb;
  }
  if (DX) {
    const float ha = va ? dna * w[n.fn_w + lane] : 0.f, hb = vb ? dnb * w[n.fn_w + lane + 64] : 0.f;
    const float invD = 1.0f / (float)Din;
    m1 = wsum(ha + hb) * invD;
    m2 = wsum(ha * s.xh_a + hb * s.xh_b) * invD;
    dxa = va ? s.rstd0 * (ha - m1 - s.xh_a * m2) : 0.f;
    dxb = vb ? s.rstd0 * (hb - m1 - s.xh_b * m2) : 0.f;
  }
}

// The wave's accumulators -> its gradient image in LDS (same padded layout as the weight images).
template <int KB>
__device__ __forceinline__ void dump_acc(const Acc<KB>& g, const NetImg& c, float* buf, int lane) {
  const int Din = c.Din;
  // (odd row strides: 64 lanes storing element k of 64 different rows hit 64 different banks; with the dense stride 64 every
  // store of the fc2 block was a 32-way bank conflict -- the dump took 20 k cycles)
#pragma unroll
  for (int k = 0; k < KB; ++k)
    if (k < Din) buf[c.fc1_w + lane * c.s1 + k] = g.W1[k];
#pragma unroll
  for (int k = 0; k < OPE_H; ++k) buf[c.fc2_w + lane * c.s2 + k] = g.W2[k];
#pragma unroll
  for (int i = 0; i < kHB; ++i)
    if (i < c.Hout) buf[c.q_w + i * OPE_H + lane] = g.Wh[i];
  buf[c.fc1_b + lane] = g.b1; buf[c.ln1_w + lane] = g.g1; buf[c.ln1_b + lane] = g.be1;
  buf[c.fc2_b + lane] = g.b2; buf[c.ln2_w + lane] = g.g2; buf[c.ln2_b + lane] = g.be2;
  if (lane < c.Hout) buf[c.q_b + lane] = g.bh;
  if (lane < Din) { buf[c.fn_w + lane] = g.g0a; buf[c.fn_b + lane] = g.be0a; }
  if (lane + 64 < Din) { buf[c.fn_w + lane + 64] = g.g0b; buf[c.fn_b + lane + 64] = g.be0b; }
}

// Target action of one (agent, next-transition) row from logits held by lanes < A. mode 0: one-hot of (masked logit == max),
// ties give several ones (util.py:156-175); mode 1: hard gumbel-softmax, straight-through value (util.py:178-214).
// Returns the action value in lane i < A; `y_out` the soft sample (mode 1).
// avail / u: lane i holds the availability flag and the uniform noise of action i.
__device__ __forceinline__ float select_action(float logit, int A, int lane, float avail, float u, int mode, float& y_out) {
  const bool on = lane < A;
  float v = logit;
  if (mode == 1 && on) v += -logf(-logf(u + 1e-20f) + 1e-20f);
  if (on && avail == 0.f) v = -1e10f;
  const float mx = wmax(on ? v : -3.0e38f);
  y_out = 0.f;
  if (mode == 0) return (on && v == mx) ? 1.f : 0.f;
  const float e = on ? expf(v - mx) : 0.f;
  const float den = wsum(e);
  const float y = on ? e / den : 0.f;
  const float ymax = wmax(on ? y : 0.f);
  const float hard = (on && y == ymax) ? 1.f : 0.f;
  y_out = y;
  return on ? (hard - y) + y : 0.f;
}

struct FusedArgs {
  NetImg at, ct, cl, ac;          // LDS images: actor target, critic target, critic live, actor live (as used by the kernel)
  NetImg gimg;                    // compact gradient image (dense) of the differentiated net
  AgentLayout AL, CL;
  const float* theta_actor; const float* theta_actor_tgt; const float* theta_critic; const float* theta_critic_tgt;
  ope_mlp_batch bt;
  const float* U;                 // uniform noise [N*B][A] (target gumbel / actor gumbel) or null
  const float* per_w;             // [B] or null
  float* prio_out;                // [B] or null
  float* slabs;                   // [gridDim.x][gimg.size + 4]
  int N, A, D, S, B, K, Din, target_gumbel, use_huber;
  float gamma, huber_delta, per_eps;
  int buf_base;                   // LDS offset of the 4 per-wave dump buffers (may overlap the weight images)
  long long* dbg;                 // optional: s_memtime stamps of workgroup 0 / wave 0 (tools/ddpg_phases.py)
};

// ---- critic update -------------------------------------------------------------------------------------------------
template <int KB>
__global__ void __launch_bounds__(kWaves * 64, 1) ddpg_critic_fused_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool stamp = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
#define OPE_STAMP(i) if (stamp) a.dbg[i] = (long long)__builtin_amdgcn_s_memtime();
  OPE_STAMP(0)
  {   // the global loads of all three nets are in flight together (one L2 round trip), then the LDS scatters
    StageRegs r0, r1, r2;
    stage_load(a.theta_actor_tgt, a.AL, a.at, r0);
    stage_load(a.theta_critic_tgt, a.CL, a.ct, r1);
    stage_load(a.theta_critic, a.CL, a.cl, r2);
    stage_store(a.at, r0, lds);
    stage_store(a.ct, r1, lds);
    stage_store(a.cl, r2, lds);
  }
  __syncthreads();
  OPE_STAMP(1)
  Acc<KB> g;
  g.zero();
  float ls = 0.f, cs = 0.f, qs = 0.f;
  const int N = a.N, A = a.A, S = a.S, B = a.B, K = a.K;
  for (int b = blockIdx.x * kWaves + wave; b < B; b += gridDim.x * kWaves) {
    // joint next action from the target actor: column p = S + agent*A + i of the target critic's input
    float xa = lane < S ? a.bt.next_share_obs[(int64_t)b * S + lane] : 0.f;
    float xb = lane + 64 < S ? a.bt.next_share_obs[(int64_t)b * S + lane + 64] : 0.f;
    // every agent's inputs are requested one agent ahead (a row's loads are otherwise a serial 2-3 k cycle stall each)
    auto in_obs = [&](int ag, float& oa, float& ob, float& av, float& uu) {
      const int64_t r = (int64_t)ag * B + b;
      oa = lane < a.D ? a.bt.next_obs[r * a.D + lane] : 0.f;
      ob = lane + 64 < a.D ? a.bt.next_obs[r * a.D + lane + 64] : 0.f;
      av = (a.bt.next_avail_acts && lane < A) ? a.bt.next_avail_acts[r * A + lane] : 1.f;
      uu = (a.U && lane < A) ? a.U[r * A + lane] : 0.5f;
    };
    float noa, nob, nav, nuu;
    in_obs(0, noa, nob, nav, nuu);
    for (int ag = 0; ag < N; ++ag) {
      const float oa = noa, ob = nob, av = nav, uu = nuu;
      if (ag + 1 < N) in_obs(ag + 1, noa, nob, nav, nuu);
      RowSave sv;
      const float logit = net_forward(lds, a.at, oa, ob, lane, sv);
      float ydummy;
      const float act = select_action(logit, A, lane, av, uu, a.target_gumbel ? 1 : 0, ydummy);
      for (int i = 0; i < A; ++i) {
        const int p = S + ag * A + i;
        const float v = rl(act, i);
        if (p < 64) { if (lane == p) xa = v; } else { if (lane == p - 64) xb = v; }
      }
    }
    OPE_STAMP(2)
    RowSave st;
    const float qt = net_forward(lds, a.ct, xa, xb, lane, st);
    OPE_STAMP(3)
    float qn = rl(qt, 0);
    for (int k = 1; k < K; ++k) qn = fminf(qn, rl(qt, k));
    // live critic on [cent_obs | joint buffer action]
    {
      const int p0 = lane, p1 = lane + 64;
      auto col = [&](int p) -> float {
        if (p < S) return a.bt.share_obs[(int64_t)b * S + p];
        if (p >= a.Din) return 0.f;
        const int q = p - S, ag = q / A, i = q - ag * A;
        return a.bt.acts[((int64_t)ag * B + b) * A + i];
      };
      xa = col(p0); xb = col(p1);
    }
    RowSave sl;
    const float q = net_forward(lds, a.cl, xa, xb, lane, sl);
    OPE_STAMP(4)
    // TD error (maddpg.py:112-157): target = r + gamma (1 - done) min_k Q'_k ; e_k = target - Q_k
    const float target = a.bt.rewards[b] + a.gamma * (1.0f - a.bt.dones_env[b]) * qn;
    const float wgt = a.per_w ? a.per_w[b] : 1.0f;
    float dq = 0.f, fe = 0.f, ae = 0.f;
    if (lane < K) {
      const float e = target - q;
      float dfe;
      if (a.use_huber) {
        const float x = fabsf(e), dl = a.huber_delta;
        if (x <= dl) { fe = e * e * 0.5f; dfe = e; } else { fe = dl * (x - dl * 0.5f); dfe = dl * (e > 0.f ? 1.f : -1.f); }
      } else {
        fe = e * e;
        dfe = 2.0f * e;
      }
      dq = -dfe * wgt;
      ae = fabsf(e);
    }
    float l1 = 0.f, p1s = 0.f, q1 = 0.f;
    for (int k = 0; k < K; ++k) { l1 += wgt * rl(fe, k); p1s += rl(ae, k); q1 += rl(q, k); }   // same order as critic_td_kernel
    ls += l1; cs += 1.0f; qs += q1;
    if (a.prio_out && lane == 0) a.prio_out[b] = p1s / (float)K + a.per_eps;
    float dxa, dxb;
    net_backward<KB, true, false>(lds, a.cl, sl, dq, lane, g, dxa, dxb);
    OPE_STAMP(5)
  }
  // workgroup reduction of the four waves' accumulators, then one slab
  __syncthreads();                      // the weight images are dead: their LDS is reused
  OPE_STAMP(6)
  float* buf = lds + a.buf_base;
  const int G = a.gimg.size;
  dump_acc<KB>(g, a.gimg, buf + wave * (G + 4), lane);
  if (lane == 0) { float* t = buf + wave * (G + 4) + G; t[0] = ls; t[1] = cs; t[2] = qs; t[3] = 0.f; }
  __syncthreads();
  OPE_STAMP(7)
  float* out = a.slabs + (int64_t)blockIdx.x * (G + 4);
  for (int e = threadIdx.x; e < G + 4; e += blockDim.x)
    out[e] = (buf[e] + buf[(G + 4) + e]) + (buf[2 * (G + 4) + e] + buf[3 * (G + 4) + e]);
  OPE_STAMP(8)
}

// ---- actor update ----------------------------------------------------------------------------------------------------
template <int KB>
__global__ void __launch_bounds__(kWaves * 64, 1) ddpg_actor_fused_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    StageRegs r0, r1;
    stage_load(a.theta_actor, a.AL, a.ac, r0);
    stage_load(a.theta_critic, a.CL, a.cl, r1);
    stage_store(a.ac, r0, lds);
    stage_store(a.cl, r1, lds);
  }
  __syncthreads();
  Acc<KB> g;
  g.zero();
  Acc<1> none;
  float ls = 0.f, cs = 0.f, qs = 0.f;
  const int N = a.N, A = a.A, S = a.S, B = a.B;
  const int R = N * B;
  for (int r = blockIdx.x * kWaves + wave; r < R; r += gridDim.x * kWaves) {
    const int ag = r / B, b = r - ag * B;
    const float oa = lane < a.D ? a.bt.obs[(int64_t)r * a.D + lane] : 0.f;
    const float ob = lane + 64 < a.D ? a.bt.obs[(int64_t)r * a.D + lane + 64] : 0.f;
    const float av_in = (a.bt.avail_acts && lane < A) ? a.bt.avail_acts[(int64_t)r * A + lane] : 1.f;
    const float u_in = lane < A ? a.U[(int64_t)r * A + lane] : 0.5f;
    const float vld = a.bt.valid_transition[r];
    RowSave sa;
    const float logit = net_forward(lds, a.ac, oa, ob, lane, sa);
    float y;
    const float act = select_action(logit, A, lane, av_in, u_in, 1, y);
    // critic input: [cent_obs | joint action with agent `ag`'s block replaced by the actor's sample] (maddpg.py:207-227)
    float xa, xb;
    {
      auto col = [&](int p) -> float {
        if (p < S) return a.bt.share_obs[(int64_t)b * S + p];
        if (p >= a.Din) return 0.f;
        const int q = p - S, a2 = q / A, i = q - a2 * A;
        return a.bt.acts[((int64_t)a2 * B + b) * A + i];
      };
      xa = col(lane); xb = col(lane + 64);
      for (int i = 0; i < A; ++i) {
        const int p = S + ag * A + i;
        const float v = rl(act, i);
        if (p < 64) { if (lane == p) xa = v; } else { if (lane == p - 64) xb = v; }
      }
    }
    RowSave sc;
    const float q = net_forward(lds, a.cl, xa, xb, lane, sc);
    const float q1 = rl(q, 0);
    ls += -q1 * vld; cs += vld; qs += q1 * vld;                       // loss = -sum(Q_1 valid) / sum(valid) (maddpg.py:229-232)
    float dxa, dxb;
    net_backward<1, false, true>(lds, a.cl, sc, lane == 0 ? -vld : 0.f, lane, none, dxa, dxb);
    // adjoint of the agent's own action block -> straight-through gumbel adjoint: dlogit_j = y_j (dx_j - sum_m dx_m y_m)
    float dxi = 0.f;
    for (int i = 0; i < A; ++i) {
      const int p = S + ag * A + i;
      const float v = p < 64 ? rl(dxa, p) : rl(dxb, p - 64);
      if (lane == i) dxi = v;
    }
    const float dot = wsum(lane < A ? dxi * y : 0.f);
    const float dlogit = lane < A ? y * (dxi - dot) : 0.f;
    net_backward<KB, true, false>(lds, a.ac, sa, dlogit, lane, g, dxa, dxb);
  }
  __syncthreads();
  float* buf = lds + a.buf_base;
  const int G = a.gimg.size;
  dump_acc<KB>(g, a.gimg, buf + wave * (G + 4), lane);
  if (lane == 0) { float* t = buf + wave * (G + 4) + G; t[0] = ls; t[1] = cs; t[2] = qs; t[3] = 0.f; }
  __syncthreads();
  float* out = a.slabs + (int64_t)blockIdx.x * (G + 4);
  for (int e = threadIdx.x; e < G + 4; e += blockDim.x)
    out[e] = (buf[e] + buf[(G + 4) + e]) + (buf[2 * (G + 4) + e] + buf[3 * (G + 4) + e]);
}

// ---- slabs -> flat gradient ---------------------------------------------------------------------------------------------
struct SlabMap {
  int n;                 // segments: [rows][cols] blocks stored with row stride `stride` in the slab, dense in the gradient
  int src[16], size[16], dst[16], cols[16], stride[16];   // slab offset, slab length (rows * stride), flat-gradient offset
  int G, ns;             // compact image size (tail follows at G), number of slabs
  int tail_dst;          // flat offset of [loss_sum, count, q_sum, 0]
};
// 8 adjacent lanes share an output element: lane s sums slabs s, s + 8, ... (fixed order), the 8 partial sums meet through
// three xor-shuffles (fixed order): deterministic, and the slab walk is 8x shorter than one thread per element.
__global__ void __launch_bounds__(256) ddpg_slab_reduce_kernel(SlabMap m, const float* __restrict__ slabs, float* __restrict__ grad) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int e = gid >> 3, s8 = gid & 7;
  const bool live = e < m.G + 4;
  const int ee = live ? e : 0;
  float v = 0.f;
  const int64_t stride = m.G + 4;
  for (int s = s8; s < m.ns; s += 8) v += slabs[s * stride + ee];
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  if (!live || s8 != 0) return;
  if (e >= m.G) { grad[m.tail_dst + (e - m.G)] = e - m.G < 3 ? v : 0.f; return; }
  int q = 0;
#pragma unroll
  for (int i = 1; i < 16; ++i)
    if (i < m.n && e >= m.src[i]) q = i;
  const int local = e - m.src[q];
  if (local >= m.size[q]) return;
  const int j = fdiv(local, 1.0f / (float)m.stride[q]), k = local - j * m.stride[q];
  if (k < m.cols[q]) grad[m.dst[q] + j * m.cols[q] + k] = v;
}

SlabMap make_map(const NetImg& c, const AgentLayout& L, int ns) {
  SlabMap m;
  memset(&m, 0, sizeof(m));
  int k = 0;
  auto seg = [&](int src, int rows, int cols, int stride, int dst) {
    m.src[k] = src; m.size[k] = rows * stride; m.dst[k] = dst; m.cols[k] = cols; m.stride[k] = stride; ++k;
  };
  seg(c.fn_w, 1, c.Din, c.Din, L.fn_w); seg(c.fn_b, 1, c.Din, c.Din, L.fn_b); seg(c.fc1_w, OPE_H, c.Din, c.s1, L.fc1_w);
  seg(c.fc1_b, 1, OPE_H, OPE_H, L.fc1_b); seg(c.ln1_w, 1, OPE_H, OPE_H, L.ln1_w); seg(c.ln1_b, 1, OPE_H, OPE_H, L.ln1_b);
  seg(c.fc2_w, OPE_H, OPE_H, c.s2, L.fc2_w); seg(c.fc2_b, 1, OPE_H, OPE_H, L.fc2_b); seg(c.ln2_w, 1, OPE_H, OPE_H, L.ln2_w);
  seg(c.ln2_b, 1, OPE_H, OPE_H, L.ln2_b); seg(c.q_w, 1, c.Hout * OPE_H, c.Hout * OPE_H, L.q_w); seg(c.q_b, 1, c.Hout, c.Hout, L.q_b);
  m.n = k; m.G = c.size; m.ns = ns; m.tail_dst = L.end;
  return m;
}

}  // namespace

bool ddpg_fused_ok(int N, int A, int D, int S, int K) {
  static const int on = getenv("OPE_DDPG_FUSED") ? atoi(getenv("OPE_DDPG_FUSED")) : 1;
  return on && D <= 128 && S + N * A <= 128 && A <= kHB && K <= kHB && A >= 1 && K >= 1;
}

int64_t ddpg_fused_slab_floats(int N, int A, int D, int S, int K, int B) {
  const NetImg ga = make_img(D, A, 0, true), gc = make_img(S + N * A, K, 0, true);
  const int64_t blocks = ope_cdiv((int64_t)N * B, kWaves);
  const int64_t g = ga.size > gc.size ? ga.size : gc.size;
  return blocks * (g + 4);
}

static int bucket(int w) { return w <= 32 ? 32 : (w <= 64 ? 64 : (w <= 96 ? 96 : 128)); }

int launch_ddpg_critic_fused(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor_tgt, const float* theta_critic,
                             const float* theta_critic_tgt, const float* U, const float* per_w, float* slabs, float* grad, float* prio_out,
                             hipStream_t st) {
  const ope_dims& d = cfg->dims;
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.N = d.n_agents; a.A = d.act_dim; a.D = d.obs_dim; a.S = d.state_dim; a.B = cfg->batch; a.K = cfg->num_q; a.Din = a.S + a.N * a.A;
  a.AL = ope_agent_layout_mlp(a.D, a.A, 0);
  a.CL = ope_agent_layout_mlp(a.Din, a.K, 0);
  a.at = make_img(a.D, a.A, 0, true);
  a.ct = make_img(a.Din, a.K, a.at.size, true);
  a.cl = make_img(a.Din, a.K, a.at.size + a.ct.size, true);
  a.gimg = make_img(a.Din, a.K, 0, true);
  a.theta_actor_tgt = theta_actor_tgt; a.theta_critic = theta_critic; a.theta_critic_tgt = theta_critic_tgt;
  a.bt = *bt; a.U = cfg->target_gumbel ? U : nullptr; a.per_w = cfg->use_per ? per_w : nullptr; a.prio_out = prio_out; a.slabs = slabs;
  a.target_gumbel = cfg->target_gumbel; a.use_huber = cfg->use_huber; a.gamma = cfg->gamma; a.huber_delta = cfg->huber_delta;
  a.per_eps = cfg->per_eps;
  a.buf_base = 0;
  static const bool dbg_on = getenv("OPE_DDPG_DBG") != nullptr;
  a.dbg = dbg_on ? reinterpret_cast<long long*>(slabs + ddpg_fused_slab_floats(a.N, a.A, a.D, a.S, a.K, a.B)) : nullptr;
  const int wfloats = a.at.size + a.ct.size + a.cl.size, bfloats = kWaves * (a.gimg.size + 4);
  const size_t lds = (size_t)(wfloats > bfloats ? wfloats : bfloats) * sizeof(float);
  if (lds > 160 * 1024) return OPE_EINVAL;
  const int blocks = ope_cdiv(a.B, kWaves);
  const int kb = bucket(a.Din);
#define OPE_LAUNCH_C(KB)                                                                                                  \
  {                                                                                                                       \
    static int attr_lds = 0;   /* > 64 KB of dynamic LDS needs the attribute; set when it grows */                        \
    if ((int)lds > attr_lds) {                                                                                            \
      if (hipFuncSetAttribute((const void*)ddpg_critic_fused_kernel<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return OPE_ELAUNCH;                                                                                               \
      attr_lds = (int)lds;                                                                                                \
    }                                                                                                                     \
    hipLaunchKernelGGL((ddpg_critic_fused_kernel<KB>), dim3(blocks), dim3(kWaves * 64), lds, st, a);                      \
  }
  if (kb == 32) OPE_LAUNCH_C(32) else if (kb == 64) OPE_LAUNCH_C(64) else if (kb == 96) OPE_LAUNCH_C(96) else OPE_LAUNCH_C(128)
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  const SlabMap m = make_map(a.gimg, a.CL, blocks);
  hipLaunchKernelGGL(ddpg_slab_reduce_kernel, dim3(ope_cdiv((int64_t)(m.G + 4) * 8, 256)), dim3(256), 0, st, m, slabs, grad);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int launch_ddpg_actor_fused(const ope_ddpg_cfg* cfg, const ope_mlp_batch* bt, const float* theta_actor, const float* theta_critic,
                            const float* U, float* slabs, float* grad, hipStream_t st) {
  const ope_dims& d = cfg->dims;
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.N = d.n_agents; a.A = d.act_dim; a.D = d.obs_dim; a.S = d.state_dim; a.B = cfg->batch; a.K = cfg->num_q; a.Din = a.S + a.N * a.A;
  a.AL = ope_agent_layout_mlp(a.D, a.A, 0);
  a.CL = ope_agent_layout_mlp(a.Din, a.K, 0);
  a.ac = make_img(a.D, a.A, 0, true);
  a.cl = make_img(a.Din, a.K, a.ac.size, true);
  a.gimg = make_img(a.D, a.A, 0, true);
  a.theta_actor = theta_actor; a.theta_critic = theta_critic;
  a.bt = *bt; a.U = U; a.slabs = slabs;
  a.buf_base = 0;
  const int wfloats = a.ac.size + a.cl.size, bfloats = kWaves * (a.gimg.size + 4);
  const size_t lds = (size_t)(wfloats > bfloats ? wfloats : bfloats) * sizeof(float);
  if (lds > 160 * 1024) return OPE_EINVAL;
  const int blocks = ope_cdiv((int64_t)a.N * a.B, kWaves);
  const int kb = bucket(a.D);
#define OPE_LAUNCH_A(KB)                                                                                                 \
  {                                                                                                                      \
    static int attr_lds = 0;                                                                                             \
    if ((int)lds > attr_lds) {                                                                                           \
      if (hipFuncSetAttribute((const void*)ddpg_actor_fused_kernel<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return OPE_ELAUNCH;                                                                                              \
      attr_lds = (int)lds;                                                                                               \
    }                                                                                                                    \
    hipLaunchKernelGGL((ddpg_actor_fused_kernel<KB>), dim3(blocks), dim3(kWaves * 64), lds, st, a);                      \
  }
  if (kb == 32) OPE_LAUNCH_A(32) else if (kb == 64) OPE_LAUNCH_A(64) else if (kb == 96) OPE_LAUNCH_A(96) else OPE_LAUNCH_A(128)
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  const SlabMap m = make_map(a.gimg, a.AL, blocks);
  hipLaunchKernelGGL(ddpg_slab_reduce_kernel, dim3(ope_cdiv((int64_t)(m.G + 4) * 8, 256)), dim3(256), 0, st, m, slabs, grad);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

}  // namespace ope
