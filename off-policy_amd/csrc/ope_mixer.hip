// QMIX mixing network + TD target + loss, forward and backward
//   QMixer.forward                     offpolicy/algorithms/qmix/algorithm/q_mixer.py:68-94
//   TD target / mask / Huber|MSE / PER offpolicy/algorithms/qmix/qmix.py:158-187, utils/util.py:103-110
//   VDN mixing (sum over agents)       offpolicy/algorithms/vdn/algorithm/vdn_mixer.py:28-40 (with the A-2 shape fix)
//
// A workgroup owns 16 (t,b) rows of one net. The four hyper-network first layers (S -> 64,64,64,32) are 14 f32-MFMA
// output tiles over the state row, split over the 4 waves; the second layers (64 -> N*32, 32), the per-row agent-Q x |w1|
// contraction, ELU, |w2| dot and b2 follow with the agents split over the waves (4 lanes share a row: two cross-lane adds).
#include <stdlib.h>

#include "ope_rowops.h"

namespace ope {

// ---------------------------------------------------------------------------------------------------------
// mixer_fwd, workgroup-cooperative form. (One wave per 16 rows was ~600 waves for 3s5z/B=32 -- 0.6 per SIMD -- each a
// chain of ~1050 dependent-issue MFMAs.) A workgroup owns the 16 (t,b) rows and its four waves split the hyper-networks:
//   stage A   wave 0: hyper_w1.0 (4 tiles)   wave 1: hyper_w2.0 (4)   wave 2: hyper_b2.0 (4)   wave 3: hyper_b1 (2)
//   stage B   hw1 goes through LDS; wave w takes agents w, w+4, ...: v1_a = W1b_a hw1 + b, hidden partial += q_a |v1_a|
//             (wave 3's partial starts from b1); wave 1 also forms v2 = W2b hw2 + b; wave 2 the b2 head dot
//   combine   wave 1 adds the four hidden partials in fixed order, ELU, dot with |v2|, + b2  ->  Q_tot
// Two workgroup barriers.
// ---------------------------------------------------------------------------------------------------------
constexpr int kHwPitch = OPE_HYP + 4;
constexpr int kHidPitch = OPE_MIX + 4;

template <int VEC, int RT>
__global__ void __launch_bounds__(256) mixer_fwd2_kernel(MixerFwdArgs a) {
  // RT row tiles (16*RT rows) per workgroup: every weight fragment a wave loads feeds RT MFMAs. The weights of the four
  // hyper-networks are ~260 KB per workgroup pass through the CU's L1 in 64-byte row segments, which (s_memtime stamps)
  // is what the one-tile form spends its time on: 14 state chunks at ~2 800 cycles each against 512 cycles of MFMA.
  constexpr int TR = 16 * RT;
  __shared__ __attribute__((aligned(16))) float hw1s[TR * kHwPitch];
  __shared__ __attribute__((aligned(16))) float hidp[4][TR * kHidPitch];
  __shared__ float pbs[TR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int groups = (a.TB + TR - 1) / TR;
  const int net = blockIdx.x / groups;
  const int m0 = (blockIdx.x - net * groups) * TR;
  const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;
  const MixerLayout& L = a.L;
  const int S = a.S, N = a.N;
  const bool save = (net == 0) && (a.hw1 != nullptr);
  int m[RT];
  bool valid[RT];
  const float* srow[RT];
  const float* qrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    m[t] = m0 + 16 * t + j;
    valid[t] = m[t] < a.TB;
    const int mm = valid[t] ? m[t] : a.TB - 1;
    const int tt = mm / a.B, b = mm - tt * a.B;
    srow[t] = a.share + ((int64_t)(tt + net) * a.B + b) * S;
    qrow[t] = (net == 0 ? a.agent_q : a.agent_nq) + (int64_t)mm * N;
  }
  long long* dbg = a.dbg ? a.dbg + ((int64_t)blockIdx.x * 4 + wave) * 8 : nullptr;
  if (dbg && lane == 0) dbg[0] = __builtin_amdgcn_s_memtime();

  // ---- stage A: this wave's tiles (wave 3 owns only two: it computes them twice, the copies are discarded) ----
  int tile[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tile[q] = wave < 3 ? 4 * wave + q : 12 + (q & 1);
  f32x4 acc[RT][4];
  const float* wrow[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bq = *reinterpret_cast<const f32x4*>(stageA_bias(th, L, tile[q]) + 4 * g);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t][q] = bq;
    wrow[q] = stageA_row(th, L, S, tile[q], j);
  }
  const int KC = (S + 15) >> 4;
  if (a.wide_slab) {
    // Wide-state path: the K reduction was done by mixer_wide_gemm_kernel (ope_mixer_wide.hip); add the tile's stream-K partial
    // slabs in ascending K order on top of the bias. Lane (j, g) takes the four features 16 tile[q] + 4 g .. + 3 of its row.
    const WidePlan P = wide_plan(a.TB, S);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int mm = valid[t] ? m[t] : a.TB - 1;
      const int rb = mm / kWideBM, rin = mm - rb * kWideBM;
      const int wt = 2 * rb + net;
      const int w_lo = wide_owner(P, wt * P.nst), w_hi = wide_owner(P, (wt + 1) * P.nst - 1);
      for (int w = w_lo; w <= w_hi; ++w) {
        const int seg = wt - wide_bound(P, w) / P.nst;
        const float* __restrict__ sl = a.wide_slab + (((int64_t)w * P.maxseg + seg) * kWideBM + rin) * kWideBN + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sl + 16 * tile[q]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][q][r] += v[r];
        }
      }
    }
  } else {
    struct Chunk { f32x4 w[4]; f32x4 x[RT]; };
    auto fetch = [&](Chunk& c, int ci) {
      const int k = 16 * ci + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) c.w[q] = load4c<VEC>(wrow[q], k, S);
#pragma unroll
      for (int t = 0; t < RT; ++t) c.x[t] = load4c<VEC>(srow[t], k, S);
    };
    auto compute = [&](const Chunk& c, int ci) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const f32x4 xs = mask4(c.x[t], 16 * ci + 4 * g, S);   // chunks past KC-1 multiply a zero-masked state vector
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[t][q] = mfma16(c.w[q][r], xs[r], acc[t][q]);
      }
    };
    // All workgroups walk the SAME weight rows; started together they would hit the same L2 lines in the same cycles.
    // Each workgroup therefore starts its K loop at a different chunk and wraps around (summation order per row depends on
    // the workgroup index only: still deterministic).
    const int NIT = ((KC + 2) / 3) * 3;
    const int rot = a.k_stagger ? (int)(blockIdx.x % (unsigned)NIT) : 0;
    auto cid = [&](int ci) { const int c = ci + rot; return c >= NIT ? c - NIT : c; };
    Chunk c0, c1, c2;
    fetch(c0, cid(0));
    fetch(c1, cid(1));
    for (int ci = 0; ci < NIT; ci += 3) {
      fetch(c2, cid(ci + 2));
      __builtin_amdgcn_sched_barrier(0);
      compute(c0, cid(ci));
      __builtin_amdgcn_sched_barrier(0);
      fetch(c0, cid(ci + 3 < NIT ? ci + 3 : 0));
      __builtin_amdgcn_sched_barrier(0);
      compute(c1, cid(ci + 1));
      __builtin_amdgcn_sched_barrier(0);
      fetch(c1, cid(ci + 4 < NIT ? ci + 4 : 0));
      __builtin_amdgcn_sched_barrier(0);
      compute(c2, cid(ci + 2));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (dbg && lane == 0) dbg[1] = __builtin_amdgcn_s_memtime();
  if (wave < 3) {   // ReLU of the three hidden layers; saved for backward
    float* dst = wave == 0 ? a.hw1 : (wave == 1 ? a.hw2 : a.hb2);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][q][r] = fmaxf(acc[t][q][r], 0.f);
      if (save && valid[t]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + (int64_t)m[t] * OPE_HYP + 16 * q + 4 * g) = acc[t][q];
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(hw1s + (16 * t + j) * kHwPitch + 16 * q + 4 * g) = acc[t][q];
  }
  if (wave == 2) {   // b2 head: b2b_w . relu(hb2) per row
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float pb = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(th + L.b2b_w + 16 * q + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) pb = fmaf(wv[r], acc[t][q][r], pb);
      }
      pb = rowsum4(pb);
      if (g == 0) pbs[16 * t + j] = pb;
    }
  }
  lds_barrier();
  if (dbg && lane == 0) dbg[2] = __builtin_amdgcn_s_memtime();

  // ---- stage B: agents wave, wave+4, ... ----
  f32x4 hid[RT][2];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    if (wave == 3) { hid[t][0] = acc[t][0]; hid[t][1] = acc[t][1]; }
    else { hid[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hid[t][1] = hid[t][0]; }
  }
  for (int ag = wave; ag < N; ag += 4) {
    f32x4 w[2][4], bq[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bq[kh] = *reinterpret_cast<const f32x4*>(th + L.w1b_b + ag * OPE_MIX + 16 * kh + 4 * g);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
        w[kh][ft] = *reinterpret_cast<const f32x4*>(th + L.w1b_w + (int64_t)(ag * OPE_MIX + 16 * kh + j) * OPE_HYP + 16 * ft + 4 * g);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      f32x4 v[2] = {bq[0], bq[1]};
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hw1s + (16 * t + j) * kHwPitch + 16 * ft + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) v[kh] = mfma16(w[kh][ft][r], hv[r], v[kh]);
      }
      if (save && valid[t]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
          *reinterpret_cast<f32x4*>(a.v1 + (int64_t)m[t] * (N * OPE_MIX) + ag * OPE_MIX + 16 * kh + 4 * g) = v[kh];
      }
      const float qa = qrow[t][ag];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[t][kh][r] = fmaf(qa, fabsf(v[kh][r]), hid[t][kh][r]);
    }
  }
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) *reinterpret_cast<f32x4*>(hidp[wave] + (16 * t + j) * kHidPitch + 16 * kh + 4 * g) = hid[t][kh];
  f32x4 v2[RT][2];
  if (wave == 1) {   // w2 = |W2b hw2 + b| (hw2 is this wave's stage-A result)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) v2[t][kh] = *reinterpret_cast<const f32x4*>(th + L.w2b_b + 16 * kh + 4 * g);
      gemm64<2>(th + L.w2b_w, OPE_HYP, j, g, acc[t], v2[t]);
      if (save && valid[t]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) *reinterpret_cast<f32x4*>(a.v2 + (int64_t)m[t] * OPE_MIX + 16 * kh + 4 * g) = v2[t][kh];
      }
    }
  }
  if (dbg && lane == 0) dbg[3] = __builtin_amdgcn_s_memtime();
  lds_barrier();
  if (dbg && lane == 0) dbg[4] = __builtin_amdgcn_s_memtime();
  if (wave != 1) return;

  // ---- combine (wave 1) ----
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    float part = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      f32x4 h = *reinterpret_cast<const f32x4*>(hidp[3] + (16 * t + j) * kHidPitch + 16 * kh + 4 * g);   // b1 + agents 3, 7, ..
#pragma unroll
      for (int w2 = 0; w2 < 3; ++w2) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(hidp[w2] + (16 * t + j) * kHidPitch + 16 * kh + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] += o[r];
      }
      if (save && valid[t]) *reinterpret_cast<f32x4*>(a.hpre + (int64_t)m[t] * OPE_MIX + 16 * kh + 4 * g) = h;
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(elu1(h[r]), fabsf(v2[t][kh][r]), part);
    }
    const float qtot = rowsum4(part) + (pbs[16 * t + j] + th[L.b2b_b]);
    if (valid[t] && g == 0) (net == 0 ? a.qtot : a.nqtot)[m[t]] = qtot;
  }
  if (dbg && lane == 0) dbg[5] = __builtin_amdgcn_s_memtime();
}

// ---------------------------------------------------------------------------------------------------------
// mixer_fwd3: weights stationary in registers, row tiles streamed (3s5z-sized states: S <= 16 KCM, S % 4 == 0, N <= 8).
// mixer_fwd2 re-streams the 260 KB of hyper-network weights out of L2 for every 16-row workgroup (156 MB per launch at 3s5z,
// 600 workgroups reading the same lines at once); its K loop waits on the L2 channels that hold them (stage A = 34 k of a
// wave's 45 k cycles; deeper prefetch changes nothing). Here the grid is persistent -- one 8-wave
// workgroup per CU, the even ones on the live net, the odd ones on the target -- every wave loads ITS slice of the weights
// once and then walks over row tiles; per tile only the 16 state rows (HBM -> registers -> LDS, one tile ahead) and the results
// move:
//   stage A   waves 0..6 own two of the 14 first-layer output tiles each (2 x KC fragments = 112 registers at S = 216): two
//             independent MFMA chains per wave over the state chunks, operands read from the LDS state tile
//   stage B   wave w = agent w: v1_w = W1b_w hw1 + b (its 8 fragments resident), q_w |v1_w| -> LDS; wave 7 also forms
//             v2 = W2b hw2 + b; waves 4, 5 the two halves of the b2 head dot
//   combine   wave 7 adds b1 and the agents' terms in agent order, ELU, dot with |v2|, + b2 -> Q_tot, while the other waves
//             already run stage A of the next tile. Two workgroup barriers per tile.
// Summation order differs from mixer_fwd2's (K ascending instead of rotated per workgroup; agents 0..N-1 instead of by wave):
// rounding-level differences, fixed for a given shape.
// ---------------------------------------------------------------------------------------------------------
template <int KCM, bool FULL>   // FULL: ceil(S / 16) == KCM exactly (no per-chunk guard: the K loop is straight-line code)
__global__ void __launch_bounds__(512, 1) mixer_fwd3_kernel(MixerFwdArgs a) {
  constexpr int TR = 16;
  constexpr int Sp = 16 * KCM + 4;
  __shared__ __attribute__((aligned(16))) float xs[TR * Sp];
  __shared__ __attribute__((aligned(16))) float hw1s[TR * kHwPitch];
  __shared__ __attribute__((aligned(16))) float hw2s[TR * kHwPitch];
  __shared__ __attribute__((aligned(16))) float b1s[TR * kHidPitch];
  __shared__ __attribute__((aligned(16))) float hidp[8][TR * kHidPitch];
  __shared__ float pbs[2][TR];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int net = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;
  const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;
  const MixerLayout& L = a.L;
  const int S = a.S, N = a.N, S4 = S >> 2;
  const int KC = (S + 15) >> 4;
  const int ntiles = (a.TB + TR - 1) / TR;
  const bool save = (net == 0) && (a.hw1 != nullptr);
  const float* __restrict__ qsrc = net == 0 ? a.agent_q : a.agent_nq;
  float* __restrict__ qdst = net == 0 ? a.qtot : a.nqtot;
  // optional s_memtime stamps (ope_set_debug; tools/mixer3_phases.py): [workgroup][wave][8] = start, weights + first state tile in
  // place, then for the FIRST tile: stage A done, after barrier 1, stage B done, after barrier 2; last: end of the wave
  long long* dbg = a.dbg ? a.dbg + ((int64_t)blockIdx.x * 8 + wave) * 8 : nullptr;
  auto stamp = [&](int k) { if (dbg && lane == 0) dbg[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);

  // ---- this wave's weights, fetched row-contiguously and spread to their owner lanes (see trunk_fwd3) ----
  const int lj = lane >> 2, lg = lane & 3, src = 4 * j + g;
  auto spread = [&](const f32x4& t) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = __shfl(t[r], src, 64);
    return o;
  };
  f32x4 wA[2][KCM], bA[2];
  const int itA = 2 * wave;                 // stage-A output tiles 2 wave, 2 wave + 1 (waves 0..6)
  if (wave < 7) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* __restrict__ wr = stageA_row(th, L, S, itA + q, lj);
#pragma unroll
      for (int c = 0; c < KCM; ++c) wA[q][c] = spread(load4c<4>(wr, 16 * c + 4 * lg, S));   // clamped; columns >= S meet zeros in xs
      bA[q] = *reinterpret_cast<const f32x4*>(stageA_bias(th, L, itA + q) + 4 * g);
    }
  }
  const int ag = wave < N ? wave : N - 1;    // stage-B agent of this wave (waves >= N idle there)
  f32x4 w1b[2][4], b1b[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    b1b[kh] = *reinterpret_cast<const f32x4*>(th + L.w1b_b + ag * OPE_MIX + 16 * kh + 4 * g);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
      w1b[kh][ft] = spread(*reinterpret_cast<const f32x4*>(th + L.w1b_w + (int64_t)(ag * OPE_MIX + 16 * kh + lj) * OPE_HYP + 16 * ft + 4 * lg));
  }
  f32x4 hb2w[2];
  if (wave == 7) {   // wave 7 has no stage-A tiles: its W2b fragments live in the registers of wA[0][0..7], the bias in bA
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bA[kh] = *reinterpret_cast<const f32x4*>(th + L.w2b_b + 16 * kh + 4 * g);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
        wA[0][4 * kh + ft] = spread(*reinterpret_cast<const f32x4*>(th + L.w2b_w + (int64_t)(16 * kh + lj) * OPE_HYP + 16 * ft + 4 * lg));
    }
  }
  if (wave == 4 || wave == 5) {              // b2 head weights of this wave's 32 hb2 features
#pragma unroll
    for (int q = 0; q < 2; ++q) hb2w[q] = *reinterpret_cast<const f32x4*>(th + L.b2b_w + 16 * ((itA + q) - 8) + 4 * g);
  }
  const float b2b_bias = th[L.b2b_b];
  // zero the padding columns of the state tile once (columns S .. 16 KCM + 3; the tile loads never touch them)
  for (int p = tid; p < TR * (Sp - S); p += 512) {
    const int r = p / (Sp - S), c = p - r * (Sp - S);
    xs[r * Sp + S + c] = 0.f;
  }
  // state rows of a tile: 16 rows x S/4 16-byte pieces, at most two per thread (S <= 16 KCM <= 224 -> 896 pieces)
  constexpr int kPieces = 2;
  f32x4 pre[kPieces];
  auto request = [&](int tile) {
#pragma unroll
    for (int u = 0; u < kPieces; ++u) {
      const int p = tid + 512 * u;
      const int pp = p < TR * S4 ? p : TR * S4 - 1;
      const int r = pp / S4, c = pp - r * S4;
      const int m = tile * TR + r;
      const int mm = m < a.TB ? m : a.TB - 1;
      const int tt = mm / a.B, b = mm - tt * a.B;
      pre[u] = *reinterpret_cast<const f32x4*>(a.share + ((int64_t)(tt + net) * a.B + b) * S + 4 * c);
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int u = 0; u < kPieces; ++u) {
      const int p = tid + 512 * u;
      if (p < TR * S4) {
        const int r = p / S4, c = p - r * S4;
        *reinterpret_cast<f32x4*>(xs + r * Sp + 4 * c) = pre[u];
      }
    }
  };
  int tile = wg;
  if (tile < ntiles) {
    request(tile);
    deposit();
    if (tile + nwg < ntiles) request(tile + nwg);
  }
  lds_barrier();
  stamp(1);
  const int first_tile = tile;
  for (; tile < ntiles; tile += nwg) {
    const bool st1 = tile == first_tile;
    const int m = tile * TR + j;
    const bool valid = m < a.TB;
    const int mm = valid ? m : a.TB - 1;
    // ---- stage A ----
    f32x4 acc[2];
    if (wave < 7) {
      acc[0] = bA[0]; acc[1] = bA[1];
#pragma unroll
      for (int c = 0; c < KCM; ++c) {
        if (FULL || c < KC) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + j * Sp + 16 * c + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[0] = mfma16(wA[0][c][r], xv[r], acc[0]);
            acc[1] = mfma16(wA[1][c][r], xv[r], acc[1]);
          }
        }
      }
      if (wave < 6) {   // ReLU of the three hidden layers (tiles 0..11); saved for backward
        float* dst = wave < 2 ? a.hw1 : (wave < 4 ? a.hw2 : a.hb2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[q][r] = fmaxf(acc[q][r], 0.f);
          const int f0 = 16 * ((itA + q) & 3) + 4 * g;
          if (save && valid) *reinterpret_cast<f32x4*>(dst + (int64_t)m * OPE_HYP + f0) = acc[q];
          if (wave < 2) *reinterpret_cast<f32x4*>(hw1s + j * kHwPitch + f0) = acc[q];
          else if (wave < 4) *reinterpret_cast<f32x4*>(hw2s + j * kHwPitch + f0) = acc[q];
        }
        if (wave >= 4) {   // b2 head: this wave's half of b2b_w . relu(hb2)
          float pb = 0.f;
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) pb = fmaf(hb2w[q][r], acc[q][r], pb);
          pb = rowsum4(pb);
          if (g == 0) pbs[wave - 4][j] = pb;
        }
      } else {          // wave 6: hyper_b1 (no activation)
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(b1s + j * kHidPitch + 16 * q + 4 * g) = acc[q];
      }
    }
    if (st1) stamp(2);
    lds_barrier();      // stage-A results visible; everybody is done reading xs
    if (st1) stamp(3);
    if (tile + nwg < ntiles) {
      deposit();        // next tile's rows (requested one tile ago)
      if (tile + 2 * nwg < ntiles) request(tile + 2 * nwg);
    }
    // ---- stage B ----
    f32x4 v2[2], hb1[2];
    float pbsum = 0.f;
    if (wave < N) {
      f32x4 v[2] = {b1b[0], b1b[1]};
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hw1s + j * kHwPitch + 16 * ft + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) v[kh] = mfma16(w1b[kh][ft][r], hv[r], v[kh]);
      }
      const float qa = qsrc[(int64_t)mm * N + ag];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (save && valid) *reinterpret_cast<f32x4*>(a.v1 + (int64_t)m * (N * OPE_MIX) + ag * OPE_MIX + 16 * kh + 4 * g) = v[kh];
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = qa * fabsf(v[kh][r]);
        *reinterpret_cast<f32x4*>(hidp[wave] + j * kHidPitch + 16 * kh + 4 * g) = t;
      }
    }
    if (wave == 7) {   // w2 = |W2b hw2 + b|
      v2[0] = bA[0]; v2[1] = bA[1];
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hw2s + j * kHwPitch + 16 * ft + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) v2[kh] = mfma16(wA[0][4 * kh + ft][r], hv[r], v2[kh]);
      }
      if (save && valid) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) *reinterpret_cast<f32x4*>(a.v2 + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = v2[kh];
      }
      // b1 and the b2 head partials are rewritten by the NEXT tile's stage A while this wave combines: take them now
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) hb1[kh] = *reinterpret_cast<const f32x4*>(b1s + j * kHidPitch + 16 * kh + 4 * g);
      pbsum = pbs[0][j] + pbs[1][j];
    }
    if (st1) stamp(4);
    lds_barrier();      // agents' terms visible; the next tile's state is in xs
    if (st1) stamp(5);
    if (wave == 7) {    // ---- combine (the other waves go on to the next tile's stage A) ----
      float part = 0.f;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        f32x4 h = hb1[kh];
        for (int w = 0; w < N; ++w) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(hidp[w] + j * kHidPitch + 16 * kh + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] += o[r];
        }
        if (save && valid) *reinterpret_cast<f32x4*>(a.hpre + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = h;
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(elu1(h[r]), fabsf(v2[kh][r]), part);
      }
      const float qtot = rowsum4(part) + (pbsum + b2b_bias);
      if (valid && g == 0) qdst[m] = qtot;
    }
  }
  stamp(6);
}

template <int VEC>
static void launch_mixer2(const MixerFwdArgs& a0, hipStream_t st) {
  MixerFwdArgs a = a0;
  static const int stag = getenv("OPE_STAGGER") ? atoi(getenv("OPE_STAGGER")) : 1;
  a.k_stagger = stag;
  static const int forced = getenv("OPE_MIXER_RT") ? atoi(getenv("OPE_MIXER_RT")) : 0;
  const int rt = forced ? forced : 1;
  // weights-in-registers persistent form for 3s5z-sized problems (OPE_MIXER_PERSIST=0: the re-streaming kernel)
  // (read per launch: tests switch it between calls; 2 = also for small problems, where one workgroup per row tile is as good)
  const char* pe = getenv("OPE_MIXER_PERSIST");
  const int persist = a.path == 1 ? 2 : (a.path == 2 ? 0 : (pe ? atoi(pe) : 1));     // cfg->mixer_path overrides the environment
  // both nets; with the wide-state slabs the first hyper-layers' products were done by mixer_wide_gemm
  kprof_work(2.0 * 2.0 * a.TB * (((double)a.S * (3.0 * OPE_HYP + OPE_MIX) + (double)OPE_HYP * a.N * OPE_MIX + OPE_HYP * OPE_MIX + OPE_HYP + (double)a.N * OPE_MIX + OPE_MIX) - (a.wide_slab ? (double)a.S * (3.0 * OPE_HYP + OPE_MIX) : 0.0)));
  if (VEC == 4 && persist && !forced && !a.wide_slab && a.N <= 8 && a.S <= 16 * 14 && (a.TB >= 16 * 64 || persist == 2)) {
    note_launch("mixer_fwd3", 14, ((a.S + 15) >> 4) == 14);
    const int tiles = ope_cdiv(a.TB, 16);
    static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
    const int per_net = tiles < cus / 2 ? tiles : cus / 2;
    if (((a.S + 15) >> 4) == 14) OPE_LAUNCH((mixer_fwd3_kernel<14, true>), dim3(2 * per_net), dim3(512), 0, st, a);
    else OPE_LAUNCH((mixer_fwd3_kernel<14, false>), dim3(2 * per_net), dim3(512), 0, st, a);
    return;
  }
  note_launch(a.wide_slab ? "mixer_fwd2_wide" : "mixer_fwd2", VEC, rt);
  if (rt == 4) OPE_LAUNCH((mixer_fwd2_kernel<VEC, 4>), dim3(2 * ope_cdiv(a.TB, 64)), dim3(256), 0, st, a);
  else if (rt == 2) OPE_LAUNCH((mixer_fwd2_kernel<VEC, 2>), dim3(2 * ope_cdiv(a.TB, 32)), dim3(256), 0, st, a);
  else OPE_LAUNCH((mixer_fwd2_kernel<VEC, 1>), dim3(2 * ope_cdiv(a.TB, 16)), dim3(256), 0, st, a);
}

int launch_mixer_fwd(const MixerFwdArgs& a0, hipStream_t st) {
  if (a0.TB < 1 || a0.N < 1 || a0.S < 1) return OPE_EINVAL;
  MixerFwdArgs a = a0;
  const bool wide = a.wide_slab && (a.path == 3 || (a.path == 0 && a.S > kWideAutoS));
  if (wide) {
    const int rc = launch_mixer_wide_gemm(a, st);
    if (rc) return rc;
    a.path = 2;                  // the second stage = mixer_fwd2 reading the slabs
    a.dbg = nullptr;             // (the stamp region belongs to the GEMM kernel in this mode)
  } else {
    if (a.path == 3) return OPE_EINVAL;          // wide-state path asked for without its slab region (the plan decides both)
    a.wide_slab = nullptr;
  }
  const int vec = ope_vec_of(a.S);
  // an explicit request for the resident-weight kernel that the shape does not allow: no silent fall-back (tests pin kernels by path)
  if (a.path == 1 && !(vec == 4 && a.N <= 8 && a.S <= 16 * 14)) return OPE_EINVAL;
  if (vec == 4) launch_mixer2<4>(a, st);
  else if (vec == 2) launch_mixer2<2>(a, st);
  else launch_mixer2<1>(a, st);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// Mixer adjoint: a 16-row tile spread over the 4 waves of a workgroup. (One wave per tile was 300 waves at 3s5z, each a
// serial chain over the N agents -- 2 + 8 loads and 8 MFMAs per agent, every load waited for in turn: 25 us of one wave's
// latency on a third of the SIMDs.) Wave w takes agents w, w+4, ..., the four partial W1b^T dv1 tiles meet in LDS (fixed
// order), and wave w finishes feature tile w of the three hyper-net adjoints.
constexpr int kPartPitch = OPE_HYP + 4;
__global__ void __launch_bounds__(256) mixer_bwd4_kernel(MixerBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float part[4][16][kPartPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int m0 = tile * 16;
  const int m = m0 + j;
  const bool valid = m < a.TB;
  const int mm = valid ? m : m0;
  const int t = mm / a.td.B, b = mm - t * a.td.B;
  const float* __restrict__ th = a.theta;
  const MixerLayout& L = a.L;
  const int N = a.N;
  const int NM = N * OPE_MIX;
  const float* w1bT = a.thetaT;                 // [64][N*32]
  const float* w2bT = a.thetaT + OPE_HYP * NM;  // [64][32]
  const int fo = 16 * wave + 4 * g;             // the 4 hyper-net features this lane finishes

  // loads with no dependency on the TD error go out first: this wave's share of the final stage
  const f32x4 h1 = *reinterpret_cast<const f32x4*>(a.hw1 + (int64_t)mm * OPE_HYP + fo);
  const f32x4 h2 = *reinterpret_cast<const f32x4*>(a.hw2 + (int64_t)mm * OPE_HYP + fo);
  const f32x4 h3 = *reinterpret_cast<const f32x4*>(a.hb2 + (int64_t)mm * OPE_HYP + fo);
  const f32x4 wb = *reinterpret_cast<const f32x4*>(th + L.b2b_w + fo);
  f32x4 hp[2], v2[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    hp[kh] = *reinterpret_cast<const f32x4*>(a.hpre + (int64_t)mm * OPE_MIX + 16 * kh + 4 * g);
    v2[kh] = *reinterpret_cast<const f32x4*>(a.v2 + (int64_t)mm * OPE_MIX + 16 * kh + 4 * g);
  }
  const float qtot = a.qtot[mm];
  TdOut td = td_row(a.td, t, b, qtot, a.nqtot[mm]);
  if (!valid) { td.err = 0.f; td.keep = 0.f; td.lossel = 0.f; td.dq = 0.f; }
  const float dQ = td.dq;
  if (wave == 0) {
    const float ls = tilesum16(g == 0 ? td.lossel : 0.f);
    const float cs = tilesum16(g == 0 ? td.keep : 0.f);
    const float qs = tilesum16(g == 0 ? qtot * td.keep : 0.f);
    if (lane == 0) {
      a.loss_part[tile * 4 + 0] = ls;
      a.loss_part[tile * 4 + 1] = cs;
      a.loss_part[tile * 4 + 2] = qs;
      a.loss_part[tile * 4 + 3] = 0.f;
    }
    if (valid && g == 0) {
      a.err_abs[m] = fabsf(td.err);
      *reinterpret_cast<f32x4*>(a.dqtot + 4 * (int64_t)m) = f32x4{dQ, 0.f, 0.f, 0.f};   // [TB][4]: lda = 4 for the wgrad kernel
    }
  }
  // lane-local 32-vectors (k = 16kh + 4g + r), evaluated by every wave; wave 0 stores them
  f32x4 dpre[2], dv2[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float hdn = elu1(hp[kh][r]);
      dv2[kh][r] = dQ * hdn * sgn(v2[kh][r]);
      const float dh = dQ * fabsf(v2[kh][r]);
      dpre[kh][r] = dh * (hp[kh][r] > 0.f ? 1.0f : expf(hp[kh][r]));
    }
    if (valid && wave == 0) {
      *reinterpret_cast<f32x4*>(a.d_b1 + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = dpre[kh];
      *reinterpret_cast<f32x4*>(a.d_v2 + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = dv2[kh];
    }
  }
  // this wave's agents: dq_a, dv1, and its partial of dhw1 = W1b^T dv1
  f32x4 dh1[4];
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) dh1[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ag = wave; ag < N; ag += 4) {
    const float qa = a.agent_q[(int64_t)mm * N + ag];
    float dqa = 0.f;
    f32x4 dv1[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.v1 + (int64_t)mm * NM + ag * OPE_MIX + 16 * kh + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dqa = fmaf(dpre[kh][r], fabsf(v[r]), dqa);
        dv1[kh][r] = dpre[kh][r] * qa * sgn(v[r]);
      }
      if (valid) *reinterpret_cast<f32x4*>(a.d_v1 + (int64_t)m * NM + ag * OPE_MIX + 16 * kh + 4 * g) = dv1[kh];
    }
    dqa = rowsum4(dqa);
    if (valid && g == 0) a.d_agent_q[(int64_t)m * N + ag] = dqa;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w1bT + (int64_t)(16 * ft + j) * NM + ag * OPE_MIX + 16 * kh + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) dh1[ft] = mfma16(wv[r], dv1[kh][r], dh1[ft]);
      }
  }
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) *reinterpret_cast<f32x4*>(&part[wave][j][16 * ft + 4 * g]) = dh1[ft];
  // feature tile `wave` of dhw2 = W2b^T dv2 while the partials settle
  f32x4 dh2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w2bT + (int64_t)(16 * wave + j) * OPE_MIX + 16 * kh + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) dh2 = mfma16(wv[r], dv2[kh][r], dh2);
  }
  lds_barrier();
  if (valid) {
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(&part[0][j][fo]), p1 = *reinterpret_cast<const f32x4*>(&part[1][j][fo]);
    const f32x4 p2 = *reinterpret_cast<const f32x4*>(&part[2][j][fo]), p3 = *reinterpret_cast<const f32x4*>(&part[3][j][fo]);
    f32x4 o1, o2, o3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s1 = (p0[r] + p1[r]) + (p2[r] + p3[r]);
      o1[r] = h1[r] > 0.f ? s1 : 0.f;
      o2[r] = h2[r] > 0.f ? dh2[r] : 0.f;
      o3[r] = h3[r] > 0.f ? dQ * wb[r] : 0.f;
    }
    *reinterpret_cast<f32x4*>(a.d_hw1 + (int64_t)m * OPE_HYP + fo) = o1;
    *reinterpret_cast<f32x4*>(a.d_hw2 + (int64_t)m * OPE_HYP + fo) = o2;
    *reinterpret_cast<f32x4*>(a.d_hb2 + (int64_t)m * OPE_HYP + fo) = o3;
  }
}

int launch_mixer_bwd(const MixerBwdArgs& a, hipStream_t st) {
  if (a.TB < 1) return OPE_EINVAL;
  kprof_work(2.0 * a.TB * ((double)a.N * OPE_MIX * OPE_HYP + OPE_MIX * OPE_HYP));      // W1b^T dv1 and W2b^T dv2
  OPE_LAUNCH(mixer_bwd4_kernel, dim3(ope_cdiv(a.TB, 16)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("mixer_bwd4");
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// VDN: Q_tot = sum_a q_a (both nets), TD, and d agent_q = dQ_tot broadcast. One thread per (t,b); loss partials
// per 16-row group in the same [tile][4] format the QMIX path uses.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vdn_kernel(VdnArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = m < a.TB;
  const int mm = valid ? m : 0;
  const int t = mm / a.td.B, b = mm - t * a.td.B;
  float q = 0.f, nq = 0.f;
  for (int ag = 0; ag < a.N; ++ag) {
    q += a.agent_q[(int64_t)mm * a.N + ag];
    nq += a.agent_nq[(int64_t)mm * a.N + ag];
  }
  TdOut td = td_row(a.td, t, b, q, nq);
  if (!valid) { td.err = 0.f; td.keep = 0.f; td.lossel = 0.f; td.dq = 0.f; }
  const float ls = tilesum16(td.lossel), cs = tilesum16(td.keep), qs = tilesum16(q * td.keep);
  if ((threadIdx.x & 15) == 0 && (m >> 4) < ((a.TB + 15) >> 4)) {
    const int tile = m >> 4;
    a.loss_part[tile * 4 + 0] = ls;
    a.loss_part[tile * 4 + 1] = cs;
    a.loss_part[tile * 4 + 2] = qs;
    a.loss_part[tile * 4 + 3] = 0.f;
  }
  if (valid) {
    a.err_abs[m] = fabsf(td.err);
    for (int ag = 0; ag < a.N; ++ag) a.d_agent_q[(int64_t)m * a.N + ag] = td.dq;
  }
}

int launch_vdn(const VdnArgs& a, hipStream_t st) {
  if (a.TB < 1) return OPE_EINVAL;
  OPE_LAUNCH(vdn_kernel, dim3(ope_cdiv(a.TB, 256)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// per-episode [mean_t |err|, max_t |err|] for the R2D2-style priorities (qmix.py:179-181): one wave per episode, lanes stride
// over time, fixed-order tree (the thread-per-episode form walked T strided loads serially: 43 us at T = 180)
__global__ void __launch_bounds__(256) td_stats_kernel(const float* __restrict__ err_abs, int T, int B, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float s = 0.f, mx = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float e = err_abs[(int64_t)t * B + b];
    s += e;
    mx = fmaxf(mx, e);
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if (lane == 0) {
    out[2 * b] = s / (float)T;
    out[2 * b + 1] = mx;
  }
}

int launch_td_stats(const float* err_abs, int T, int B, float* out, hipStream_t st) {
  OPE_LAUNCH(td_stats_kernel, dim3(ope_cdiv(B, 4)), dim3(256), 0, st, err_abs, T, B, out);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

}  // namespace ope
