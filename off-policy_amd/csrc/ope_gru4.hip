// GRU scans for few rows ("gru4"): one workgroup per row, the 192 x 64 recurrent mat-vec split by OUTPUT over four waves.
//
// Replaces (reference): nn.GRU inside RNNLayer.forward, offpolicy/algorithms/utils/rnn.py:19-23, stepped over the
// T+1 entries of an episode by AgentQFunction.forward (qmix/algorithm/agent_q_function.py:34-67), and the autograd of it.
//
// Measured on gfx950 (tools/microbench_lat.hip, tools/microbench_lds.hip): a lone wave issues one instruction per ~5
// cycles whatever the unit (96 v_pk_fma_f32 = 488 cycles, every ds_read / s_waitcnt between them costs another slot),
// an LDS write -> read round trip is ~136 cycles, the sigmoid/sigmoid/tanh chain ~144, a workgroup barrier ~44. With
// 256-512 rows in flight (QMIX, B=32) the scans are bound by that per-step chain, so the work of a step is spread over
// as many issue ports as one exchange allows:
//   * compute wave q (0..3) owns hidden features 16q..16q+15. Lane (j, g) = (lane >> 2, lane & 3) holds, for feature
//     f = 16q + j, the K-quarter [16g, 16g+16) of the three W_hh rows of f (forward) / of column f over the gate-row
//     quarter (backward): 48 VGPRs. A step is 4 (12) broadcast ds_read_b128, 24 v_pk_fma_f32, a quad reduction with two
//     DPP adds per sum (no LDS), and the gates / gate adjoints of feature f evaluated once per quad lane;
//   * the exchange is the step's 64-vector itself (h_t, or the three gate adjoints): every wave publishes its 16
//     features to a parity-double-buffered LDS vector, ONE workgroup barrier per step, everybody reads its K-quarter;
//   * wave 4 only loads (8 steps ahead, compiler-invisible asm loads, s_waitcnt vmcnt(N) counted in loads only; the idiom
//     relies on hipcc never copying a destination register between the load and its wait -- true in this low-pressure
//     branch, check the ISA for v_mov of the `pre` registers after any change: DESIGN.md section 4) and
//     hands one step per step to the compute waves through LDS; wave 5 only stores (the previous step's results, read
//     back from LDS), so no wave's vmcnt mixes loads and stores and nothing waits on HBM inside the chain.
// Fixed summation order (four partial sums per lane, pairwise; quad reduction commutative-symmetric), so all four lanes
// of a quad hold bit-identical values and results are deterministic.
//   r = sigma(gi_r + gh_r), z = sigma(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h' = (1-z) n + z h      (nn.GRU)
#include <stdlib.h>

#include <type_traits>

#include "ope_agent.h"

namespace ope {
namespace {

constexpr int kAhead = 8;   // loader prefetch distance in steps

__device__ __forceinline__ float hsum4(f32x2 a, f32x2 b) { return (a[0] + a[1]) + (b[0] + b[1]); }
// sum over the G = 2 or 4 adjacent lanes that share a feature (DPP, every lane ends with the same bits)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if (G == 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  return v;
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int W, bool DBG>   // W = compute waves per row = lanes per feature (2 or 4)
__global__ void __launch_bounds__((W + 2) * 64) gru_fwd4_kernel(GruFwdArgs a) {
  constexpr int FPW = OPE_H / W;     // features per compute wave
  constexpr int KS = OPE_H / W;      // K-slice per lane
  constexpr int NP = KS / 2;         // weight pairs per gate per lane
  __shared__ __attribute__((aligned(16))) float sm[2 * OPE_H + 6 * OPE_H + 8 * OPE_H];
  float(*hs)[OPE_H] = reinterpret_cast<float(*)[OPE_H]>(sm);                        // [2][64]    h_t, parity of t
  float(*gis)[3][OPE_H] = reinterpret_cast<float(*)[3][OPE_H]>(sm + 2 * OPE_H);     // [2][3][64] gi of step t
  float(*sv)[4][OPE_H] = reinterpret_cast<float(*)[4][OPE_H]>(sm + 8 * OPE_H);      // [2][4][64] r, z, n, gh_n of step t
  constexpr int kHs = 0, kSv = 8 * OPE_H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rid = blockIdx.x;
  const int net = rid / a.NB;
  const int row = rid - net * a.NB;
  const int L = a.L;
  const int nchunks = (L + kAhead - 1) / kAhead;

  if (wave == W) {   // ---- loader: lane = feature
    const float* __restrict__ gi = net == 0 ? a.gi0 : a.gi1;
    const int64_t stride_t = (int64_t)a.NB * (3 * OPE_H);
    const float* gp = gi + (int64_t)row * (3 * OPE_H) + lane;
    float pre[kAhead][3];
    auto load_step = [&](float (&d)[3], int t) {
      const float* p = gp + (int64_t)min(t, L - 1) * stride_t;
      gload_async(d[0], p);
      gload_async(d[1], p + OPE_H);
      gload_async(d[2], p + 2 * OPE_H);
    };
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_step(pre[s], s);
    OPE_GWAIT24(pre);
    gis[0][0][lane] = pre[0][0]; gis[0][1][lane] = pre[0][1]; gis[0][2][lane] = pre[0][2];
    load_step(pre[0], kAhead);
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
      for (int s = 0; s < kAhead; ++s) {
        const int t = c * kAhead + s;
        if (t < L) {
          float(&d)[3] = pre[(s + 1) % kAhead];      // holds step t+1; 7 x 3 younger loads are in flight behind it
          asm volatile("s_waitcnt vmcnt(21)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])::"memory");
          gis[(s + 1) & 1][0][lane] = d[0];
          gis[(s + 1) & 1][1][lane] = d[1];
          gis[(s + 1) & 1][2][lane] = d[2];
          load_step(d, t + 1 + kAhead);
          lds_barrier();
        }
      }
    }
    return;
  }
  if (wave == W + 1) {   // ---- storer: lane = feature; step t-1's results leave during step t
    float* __restrict__ hout = net == 0 ? a.h0out : a.h1out;
    const bool save = (net == 0) && (a.rg != nullptr);
    lds_barrier();
    for (int t = 0; t <= L; ++t) {
      if (t > 0) {
        const int p = (t - 1) & 1;
        const int64_t o = ((int64_t)(t - 1) * a.NB + row) * OPE_H + lane;
        hout[o] = hs[p ^ 1][lane];   // h_t was published as the next step's input
        if (save) {
          a.rg[o] = sv[p][0][lane];
          a.zg[o] = sv[p][1][lane];
          a.ng[o] = sv[p][2][lane];
          a.ghn[o] = sv[p][3][lane];
        }
      }
      if (t < L) lds_barrier();
    }
    return;
  }

  // ---- compute wave q: feature f = FPW q + j, K-slice g
  const int j = lane / W, g = lane % W;
  const int f = FPW * wave + j;
  const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;
  f32x2 wr[NP], wz[NP], wn[NP];
  {
    const float* w = th + a.whh_off + KS * g;
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
      const f32x4 vr = *reinterpret_cast<const f32x4*>(w + (int64_t)f * OPE_H + 4 * k);
      const f32x4 vz = *reinterpret_cast<const f32x4*>(w + (int64_t)(OPE_H + f) * OPE_H + 4 * k);
      const f32x4 vn = *reinterpret_cast<const f32x4*>(w + (int64_t)(2 * OPE_H + f) * OPE_H + 4 * k);
      wr[2 * k] = f32x2{vr[0], vr[1]}; wr[2 * k + 1] = f32x2{vr[2], vr[3]};
      wz[2 * k] = f32x2{vz[0], vz[1]}; wz[2 * k + 1] = f32x2{vz[2], vz[3]};
      wn[2 * k] = f32x2{vn[0], vn[1]}; wn[2 * k + 1] = f32x2{vn[2], vn[3]};
    }
  }
  const float br = th[a.bhh_off + f], bz = th[a.bhh_off + OPE_H + f], bn = th[a.bhh_off + 2 * OPE_H + f];
  const float* hin = net == 0 ? a.hinit : a.hinit1;
  float h = hin ? hin[(int64_t)row * OPE_H + f] : 0.f;
  // The sigmoids are evaluated as rcp(1 + exp2(x')) with x' = -log2(e) x: the scale is folded into the r/z rows of W_hh,
  // their biases and (per step, off the chain) gi, so that after the quad reduction a gate is exp2, add, rcp.
  constexpr float kL = -1.4426950408889634f;
#pragma unroll
  for (int i = 0; i < NP; ++i) { wr[i] *= kL; wz[i] *= kL; }
  const float brs = br * kL, bzs = bz * kL;
  const float bn0 = (g == 0) ? bn : 0.f;   // biases and gi enter the sums once per quad: through lane g == 0's accumulator
  if (g == 0) hs[0][f] = h;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  // one step at static parity P of t (the loop below is unrolled by two so every LDS address is loop-invariant)
  long long ph[4] = {0, 0, 0, 0}, tprev = 0;   // DBG: cycles in (reads+FMA | reduce+gates | publish | barrier)
#define OPE_PHASE(i)                                              \
  if (DBG) {                                                      \
    __builtin_amdgcn_sched_barrier(0);                            \
    const long long now = (long long)__builtin_amdgcn_s_memtime(); \
    ph[i] += now - tprev;                                         \
    tprev = now;                                                  \
    __builtin_amdgcn_sched_barrier(0);                            \
  }
  if (DBG) tprev = (long long)__builtin_amdgcn_s_memtime();
  auto step = [&](auto P) {
    constexpr int p = decltype(P)::value;
    const float gir = gis[p][0][f], giz = gis[p][1][f], gin = gis[p][2][f];   // LDS returns in order: gi first
    __builtin_amdgcn_sched_barrier(0);
    const float* hp = &hs[p][KS * g];
    f32x4 hq[KS / 4];
#pragma unroll
    for (int v = 0; v < KS / 4; ++v) hq[v] = *reinterpret_cast<const f32x4*>(hp + 4 * v);
    __builtin_amdgcn_sched_barrier(0);   // all LDS reads issue before anything waits
    const float xr0 = (g == 0) ? fmaf(gir, kL, brs) : 0.f;
    const float xz0 = (g == 0) ? fmaf(giz, kL, bzs) : 0.f;
    const float gin2 = gin * (2.0f * kL);
    f32x2 ar0 = {xr0, 0.f}, ar1 = {0.f, 0.f}, az0 = {xz0, 0.f}, az1 = {0.f, 0.f}, an0 = {bn0, 0.f}, an1 = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < KS / 4; ++v) {
      const f32x2 lo = {hq[v][0], hq[v][1]}, hi = {hq[v][2], hq[v][3]};
      ar0 = __builtin_elementwise_fma(wr[2 * v], lo, ar0);
      az0 = __builtin_elementwise_fma(wz[2 * v], lo, az0);
      an0 = __builtin_elementwise_fma(wn[2 * v], lo, an0);
      ar1 = __builtin_elementwise_fma(wr[2 * v + 1], hi, ar1);
      az1 = __builtin_elementwise_fma(wz[2 * v + 1], hi, az1);
      an1 = __builtin_elementwise_fma(wn[2 * v + 1], hi, an1);
    }
    if (DBG) asm volatile("" : "+v"(ar0), "+v"(ar1), "+v"(az0), "+v"(az1), "+v"(an0), "+v"(an1));
    OPE_PHASE(0)
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(group_sum<W>(hsum4(ar0, ar1))));
    sm[kSv + (4 * p + 0) * OPE_H + f] = r;     // the saves leave as soon as they exist (4 lanes, same value, same address)
    const float an = group_sum<W>(hsum4(an0, an1));
    sm[kSv + (4 * p + 3) * OPE_H + f] = an;
    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(group_sum<W>(hsum4(az0, az1))));
    sm[kSv + (4 * p + 1) * OPE_H + f] = z;
    // tanh(x) = 2 / (1 + exp(-2x)) - 1 (absolute error ~1e-7)
    const float n = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r, an * (2.0f * kL), gin2))), -1.0f);
    sm[kSv + (4 * p + 2) * OPE_H + f] = n;
    h = fmaf(z, h - n, n);                     // (1 - z) n + z h
    if (DBG) asm volatile("" : "+v"(h));
    OPE_PHASE(1)
    sm[kHs + (p ^ 1) * OPE_H + f] = h;
    if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OPE_PHASE(2)
    lds_barrier();
    OPE_PHASE(3)
  };
  for (int t = 0; t < L; t += 2) {
    step(std::integral_constant<int, 0>{});
    if (t + 1 < L) step(std::integral_constant<int, 1>{});
  }
  if (DBG && a.dbg && lane == 0) {
    long long* o = a.dbg + ((int64_t)blockIdx.x * W + wave) * 8;
    o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3]; o[4] = L;
  }
}

// ---------------------------------------------------------------------------------------------------------
// BPTT, step i = 0 .. T-1-t_lo at time tt = T-1-i:
//   dh_{tt-1}[k] = dh_tt[k] z[k] + sum_i ( W_hr[i][k] dr_pre[i] + W_hz[i][k] dz_pre[i] + W_hn[i][k] dghn[i] )
// ---------------------------------------------------------------------------------------------------------
template <int W, bool DBG>
__global__ void __launch_bounds__((W + 2) * 64) gru_bwd4_kernel(GruBwdArgs a) {
  constexpr int FPW = OPE_H / W;     // features per compute wave
  constexpr int IS = OPE_H / W;      // gate rows per gate per lane
  constexpr int NP = IS / 2;
  __shared__ __attribute__((aligned(16))) float sm[8 * OPE_H + 12 * OPE_H];
  float(*ds)[4][OPE_H] = reinterpret_cast<float(*)[4][OPE_H]>(sm);                  // [2][4][64] dr_pre, dz_pre, dgn (broadcast) + dn_pre
  float(*sav)[6][OPE_H] = reinterpret_cast<float(*)[6][OPE_H]>(sm + 8 * OPE_H);     // [2][6][64] r, z, n, gh_n, dh_out, h_prev
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const int64_t NB = a.NB;
  const int nsteps = a.T - a.t_lo;
  const int nchunks = (nsteps + kAhead - 1) / kAhead;

  if (wave == W) {   // ---- loader
    float pre[kAhead][6];
    auto load_step = [&](float (&d)[6], int i) {
      const int t = max(a.T - 1 - i, a.t_lo);
      const int64_t o = ((int64_t)t * NB + row) * OPE_H + lane;
      gload_async(d[0], a.rg + o);
      gload_async(d[1], a.zg + o);
      gload_async(d[2], a.ng + o);
      gload_async(d[3], a.ghn + o);
      gload_async(d[4], a.dh_out + o);
      gload_async(d[5], a.h + (t > 0 ? o - NB * OPE_H : o));
    };
    auto publish = [&](float (&d)[6], int i) {
      const int p = i & 1;
      const int t = a.T - 1 - i;
#pragma unroll
      for (int k = 0; k < 5; ++k) sav[p][k][lane] = d[k];
      sav[p][5][lane] = t > 0 ? d[5] : 0.f;   // h_{-1} = 0
    };
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_step(pre[s], s);
    asm volatile("s_waitcnt vmcnt(42)" : "+v"(pre[0][0]), "+v"(pre[0][1]), "+v"(pre[0][2]), "+v"(pre[0][3]), "+v"(pre[0][4]), "+v"(pre[0][5])::"memory");
    publish(pre[0], 0);
    load_step(pre[0], kAhead);
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
      for (int s = 0; s < kAhead; ++s) {
        const int i = c * kAhead + s;
        if (i < nsteps) {
          float(&d)[6] = pre[(s + 1) % kAhead];   // holds step i+1; 7 x 6 younger loads behind it
          asm volatile("s_waitcnt vmcnt(42)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])::"memory");
          publish(d, i + 1);
          load_step(d, i + 1 + kAhead);
          lds_barrier();
        }
      }
    }
    return;
  }
  if (wave == W + 1) {   // ---- storer: step i-1's adjoints leave during step i
    lds_barrier();
    for (int i = 0; i <= nsteps; ++i) {
      if (i > 0) {
        const int p = (i - 1) & 1;
        const int t = a.T - i;
        float* gout = a.dgi + ((int64_t)t * NB + row) * (3 * OPE_H) + lane;
        gout[0] = ds[p][0][lane];
        gout[OPE_H] = ds[p][1][lane];
        gout[2 * OPE_H] = ds[p][3][lane];
        a.dghn[((int64_t)t * NB + row) * OPE_H + lane] = ds[p][2][lane];
      }
      if (i < nsteps) lds_barrier();
    }
    return;
  }

  // ---- compute wave q: feature k = FPW q + j; reduction slice g (gate rows IS g .. IS g + IS - 1 of each gate)
  const int j = lane / W, g = lane % W;
  const int k = FPW * wave + j;
  f32x2 wr[NP], wz[NP], wn[NP];
  {
    const float* w = a.theta + a.whh_off + (int64_t)(IS * g) * OPE_H + k;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      wr[i] = f32x2{w[(int64_t)(2 * i) * OPE_H], w[(int64_t)(2 * i + 1) * OPE_H]};
      wz[i] = f32x2{w[(int64_t)(OPE_H + 2 * i) * OPE_H], w[(int64_t)(OPE_H + 2 * i + 1) * OPE_H]};
      wn[i] = f32x2{w[(int64_t)(2 * OPE_H + 2 * i) * OPE_H], w[(int64_t)(2 * OPE_H + 2 * i + 1) * OPE_H]};
    }
  }
  float dh = a.dh_in ? a.dh_in[(int64_t)row * OPE_H + k] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  // Per-step factors that do not depend on dh, formed one step ahead (the loader publishes step i+1's saves before
  // barrier i): A = d n_pre / d h', Bz = d z_pre / d h', Cr = d r_pre / d n_pre.
  float r, z, dho, A, Bz, Cr;
  {
    const float r_ = sav[0][0][k], z_ = sav[0][1][k], n_ = sav[0][2][k];
    const float gn_ = sav[0][3][k], dho_ = sav[0][4][k], hp_ = sav[0][5][k];
    const float omz = 1.0f - z_;
    r = r_; z = z_; dho = dho_;
    A = omz * (1.0f - n_ * n_);
    Bz = (hp_ - n_) * z_ * omz;
    Cr = gn_ * r_ * (1.0f - r_);
  }
  long long ph[4] = {0, 0, 0, 0}, tprev = 0;   // DBG: cycles in (adjoints | publish | barrier | reads+FMA+reduce)
  if (DBG) tprev = (long long)__builtin_amdgcn_s_memtime();
  auto step = [&](auto P) {
    constexpr int p = decltype(P)::value;
    const float dht = dh + dho;
    const float dz_pre = dht * Bz;
    sm[(4 * p + 1) * OPE_H + k] = dz_pre;       // four lanes of a quad write the same value to the same address
    const float dn_pre = dht * A;
    sm[(4 * p + 3) * OPE_H + k] = dn_pre;
    const float dr_pre = dn_pre * Cr;
    sm[(4 * p + 0) * OPE_H + k] = dr_pre;
    float dgn = dn_pre * r;
    if (DBG) asm volatile("" : "+v"(dgn));
    OPE_PHASE(0)
    sm[(4 * p + 2) * OPE_H + k] = dgn;
    const float c00 = (g == 0) ? dht * z : 0.f;  // the carry term enters the sum through lane g == 0's accumulator
    if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OPE_PHASE(1)
    lds_barrier();
    OPE_PHASE(2)
    const float* base = &ds[p][0][IS * g];
    f32x4 rv[IS / 4], zv[IS / 4], nv[IS / 4];
#pragma unroll
    for (int v = 0; v < IS / 4; ++v) {
      rv[v] = *reinterpret_cast<const f32x4*>(base + 4 * v);
      zv[v] = *reinterpret_cast<const f32x4*>(base + OPE_H + 4 * v);
      nv[v] = *reinterpret_cast<const f32x4*>(base + 2 * OPE_H + 4 * v);
    }
    const float r1 = sav[p ^ 1][0][k], z1 = sav[p ^ 1][1][k], n1 = sav[p ^ 1][2][k];   // next step's saves
    const float gn1 = sav[p ^ 1][3][k], dho1 = sav[p ^ 1][4][k], hp1 = sav[p ^ 1][5][k];
    __builtin_amdgcn_sched_barrier(0);   // all reads in flight before the first FMA waits
    f32x2 c0 = {c00, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c3 = {0.f, 0.f}, c4 = {0.f, 0.f}, c5 = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < IS / 4; ++v) {
      c0 = __builtin_elementwise_fma(wr[2 * v], f32x2{rv[v][0], rv[v][1]}, c0);
      c1 = __builtin_elementwise_fma(wr[2 * v + 1], f32x2{rv[v][2], rv[v][3]}, c1);
      c2 = __builtin_elementwise_fma(wz[2 * v], f32x2{zv[v][0], zv[v][1]}, c2);
      c3 = __builtin_elementwise_fma(wz[2 * v + 1], f32x2{zv[v][2], zv[v][3]}, c3);
      c4 = __builtin_elementwise_fma(wn[2 * v], f32x2{nv[v][0], nv[v][1]}, c4);
      c5 = __builtin_elementwise_fma(wn[2 * v + 1], f32x2{nv[v][2], nv[v][3]}, c5);
    }
    dh = group_sum<W>((hsum4(c0, c1) + hsum4(c2, c3)) + hsum4(c4, c5));
    {
      const float omz = 1.0f - z1;
      r = r1; z = z1; dho = dho1;
      A = omz * (1.0f - n1 * n1);
      Bz = (hp1 - n1) * z1 * omz;
      Cr = gn1 * r1 * (1.0f - r1);
    }
    if (DBG) asm volatile("" : "+v"(dh));
    OPE_PHASE(3)
  };
  for (int i = 0; i < nsteps; i += 2) {
    step(std::integral_constant<int, 0>{});
    if (i + 1 < nsteps) step(std::integral_constant<int, 1>{});
  }
  if (a.dh_carry && g == 0) a.dh_carry[(int64_t)row * OPE_H + k] = dh;
  if (DBG && a.dbg && lane == 0) {
    long long* o = a.dbg + ((int64_t)blockIdx.x * W + wave) * 8;
    o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3]; o[4] = nsteps;
  }
}

}  // namespace

// Compute waves per row: as many as keep the launch at about one compute wave per SIMD (1024 SIMDs): the kernels are
// bound by the per-step issue/latency chain of a wave, and a second wave on the same SIMD stretches both.
static int waves_per_row(int64_t rows, int asked) {
  if (asked == 2 || asked == 4) return asked;
  if (g_scan_waves) return g_scan_waves;
  return rows <= 512 ? 4 : 2;   // measured: 3s5z (512 / 256 rows) prefers 4, MMM2 (640 / 320 rows) 2 forward
}

template <int W>
static void launch_fwd(const GruFwdArgs& a, hipStream_t st) {
  if (a.dbg)
    hipLaunchKernelGGL((gru_fwd4_kernel<W, true>), dim3(a.nets * a.NB), dim3((W + 2) * 64), 0, st, a);
  else
    hipLaunchKernelGGL((gru_fwd4_kernel<W, false>), dim3(a.nets * a.NB), dim3((W + 2) * 64), 0, st, a);
}
template <int W>
static void launch_bwd(const GruBwdArgs& a, hipStream_t st) {
  if (a.dbg)
    hipLaunchKernelGGL((gru_bwd4_kernel<W, true>), dim3(a.NB), dim3((W + 2) * 64), 0, st, a);
  else
    hipLaunchKernelGGL((gru_bwd4_kernel<W, false>), dim3(a.NB), dim3((W + 2) * 64), 0, st, a);
}

int launch_gru_fwd4(const GruFwdArgs& a, hipStream_t st) {
  if (waves_per_row((int64_t)a.nets * a.NB, a.waves) == 4) launch_fwd<4>(a, st); else launch_fwd<2>(a, st);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int launch_gru_bwd4(const GruBwdArgs& a, hipStream_t st) {
  if (waves_per_row(a.NB, a.waves) == 4) launch_bwd<4>(a, st); else launch_bwd<2>(a, st);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

}  // namespace ope
