// The (t, b)-row chain of a QMIX / VDN update in ONE launch: agent q heads of both nets, chosen-action / double-Q target selection, the
// mixing networks' second stage of both nets, TD target + mask + MSE / Huber (+ PER weights), the mixer's adjoint and the q head's adjoint.
//
// Replaces (reference), for one tile of 16 (t, b) rows per workgroup:
//   ACTLayer.forward on RNNBase's output LayerNorm      offpolicy/algorithms/utils/act.py:21-37, utils/rnn.py:33-47
//   QMixPolicy.q_values_from_actions                     qmix/algorithm/QMixPolicy.py:69-93
//   double-Q / plain target selection                    qmix/qmix.py:138-148, QMixPolicy.actions_from_q QMixPolicy.py:102-174, util.py:297-302
//   QMixer.forward (live on s_t, target on s_{t+1})      qmix/algorithm/q_mixer.py:68-94      | VDNMixer.forward vdn_mixer.py:28-40 (A-2 fix)
//   TD target, mask, loss, PER error                     qmix/qmix.py:158-187, utils/util.py:103-110
//   loss.backward() through all of the above             qmix/qmix.py:191
// This is the "fused mixer + TD-target + Huber-loss kernel" of the north star, extended by the two row-local stages either side of it.
//
// Why one launch (rounds 1-3 ran head_fwd -> mixer_fwd3 -> mixer_bwd4 -> head_bwd: 15.4 + 28.1 + 13.5 + 8.6 us at 3s5z, B = 32): every one of
// those is a latency-bound chain over 4 800 (t, b) rows = 300 tiles -- about one tile per CU -- so each pays a launch boundary, a ramp
// and a tail for ~10 us of dependent work, and the forward mixer additionally re-stages 268 KB of hyper-network weights per CU for 1.2 tiles.
// Split of the work:
//   * mixer_hyp_kernel: the four FIRST hyper-layers of both nets ([2 x T*B x S] . [S x 224]: 72 % of the mixer's multiply-adds). They depend
//     on the centralized state only -- not on any agent network -- so they are a plain GEMM that can run anywhere between the gather and the
//     chain (ope_api.hip launches it beside the GRU scan, whose launch leaves the matrix pipes idle); post-ReLU outputs go to HBM (8.6 MB).
//   * qchain_kernel: everything that depends on the agents' q values. A workgroup = 16 (t, b) rows, wave w = agent w (w + 8, ...): its three
//     16-row head evaluations (live at t: chosen q; live at t+1: greedy action; target at t+1), the agent's slice of W1b for both nets
//     (32 MFMAs each), the agent's adjoint slice; the few cross-agent sums meet in LDS in fixed order. What is left of the mixers' weights
//     (W1b, W2b, their transposes: 72 KB per net) comes out of L2 as MFMA fragments.
// Deterministic: no atomics, fixed summation orders (agents by wave then by pass; waves 0..7).
#include <stdlib.h>

#include "ope_rowops.h"

namespace ope {
namespace {

// ---------------------------------------------------------------------------------------------------------
// First hyper-layers of both nets. Wave = 16 (t, b) rows x (up to) 7 of the output tiles (a table, hyp_tiles_for: two-layer hyper-networks
// 14 = hyper_w1.0 x 4, hyper_w2.0 x 4, hyper_b2.0 x 4, hyper_b1 x 2; one-layer ones 2 N + 8 = hyper_w1 x 2 N, hyper_w2 x 2, hyper_b1 x 2,
// hyper_b2.0 x 4); a workgroup = 4+ row tiles of one (net, column group), so its waves stream the same weight rows through the CU's L1. K in chunks of 16, two chunks in flight. Extra workgroups carry the weight transposes the backward kernels read.
// (A persistent, weight-stationary form -- one wave per SIMD owning one (net, output tile) for its whole life and walking over every 37th row
// tile: 8 400 evenly spread units -- was built and measured SLOWER: 28.9 us against 23.7 us at 3s5z, B = 32; each state row is then re-read
// by the 14 waves of its net and the launch is bound by how fast the CUs' texture-address units accept the scattered 64-byte row pieces of
// the MFMA operand layout, not by the matrix pipes. Removed.)
// ---------------------------------------------------------------------------------------------------------
// PK: the packed (t, b) rows of the live plan. The grid is what the padded batch needs (the host does not know the live count); the live
// tiles are dealt out over the launch's row groups again on the device -- rpw_live = ceil(live tiles / groups) waves of a workgroup work, the
// rest leave at once -- so a batch with a quarter of its rows dead runs four waves per CU instead of five (at 3s5z the launch is bound by
// the weight rows its waves stream through the CU's L1: 12.9 us with four waves at B = 24 against 20.7 us with five at B = 32).
template <int VEC, bool PK>
__global__ void __launch_bounds__(512) mixer_hyp_kernel(HypFirstArgs a) {
  if ((int)blockIdx.x >= a.main_blocks) {
    transpose4_element(a.side, ((int)blockIdx.x - a.main_blocks) * (int)blockDim.x + (int)threadIdx.x);
    return;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int TBr = PK ? a.lp.hdr[2] : a.TB;        // rows of this launch
  const int rpw_full = (int)(blockDim.x >> 6);    // row tiles per workgroup (set by the launcher so that the grid is ONE round of the chip)
  const int groups = ((a.TB + 15) / 16 + rpw_full - 1) / rpw_full;
  const int tiles = (TBr + 15) >> 4;
  const int rpw = PK ? (tiles + groups - 1) / groups : rpw_full;
  const int ncg = (a.ntiles + 6) / 7;            // column groups of (up to) 7 output tiles
  // The column groups of one (net, row group) read the same state rows: with two of them, blocks b and b + 8 form the pair, so that both
  // run on the same XCD (a workgroup lands on XCD b % 8) and the second reader finds the rows in its L2.
  int bid = blockIdx.x;
  int cg;
  if (ncg == 2 && (a.main_blocks & 15) == 0) {
    cg = (bid >> 3) & 1;
    bid = (bid & 7) | ((bid >> 4) << 3);
  } else {
    cg = bid % ncg;
    bid /= ncg;
  }
  const int net = bid / groups, grp = bid - net * groups;
  const int tile = grp * rpw + wave;
  if (wave >= rpw || tile >= tiles) return;     // (no barrier in this kernel)
  const int m = tile * 16 + j;
  const bool valid = m < TBr;
  const int mm = valid ? m : TBr - 1;
  int tt, b;
  if (PK) {
    const int2 tb = *reinterpret_cast<const int2*>(a.lp.tbrec + 8 * (int64_t)mm);
    tt = tb.x; b = tb.y;
  } else {
    tt = mm / a.B; b = mm - tt * a.B;
  }
  const int S = a.S;
  const float* __restrict__ srow = a.share + ((int64_t)(tt + net) * a.B + b) * S;      // live: s_t, target: s_{t+1} (qmix.py:155-156)
  const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;
  const int it0 = 7 * cg;
  const int cnt = a.ntiles - it0 < 7 ? a.ntiles - it0 : 7;      // (uniform: the last group may hold fewer tiles; its spare slots repeat the last one)
  f32x4 acc[7];
  const float* wrow[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const HypTile& T = a.tile[it0 + (q < cnt ? q : cnt - 1)];
    acc[q] = *reinterpret_cast<const f32x4*>(th + T.b_off + 4 * g);
    wrow[q] = th + T.w_off + (int64_t)j * S;
  }
  struct Chunk { f32x4 w[7]; f32x4 x; };
  auto fetch = [&](Chunk& c, int ci) {           // clamped addresses: chunks past the end re-read valid data and meet a zero-masked state
    const int k = 16 * ci + 4 * g;
#pragma unroll
    for (int q = 0; q < 7; ++q) c.w[q] = load4c<VEC>(wrow[q], k, S);
    c.x = load4c<VEC>(srow, k, S);
  };
  auto compute = [&](const Chunk& c, int ci) {
    const f32x4 xs = mask4(c.x, 16 * ci + 4 * g, S);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 7; ++q) acc[q] = mfma16(c.w[q][r], xs[r], acc[q]);
  };
  // Two chunks in flight. (A ring of four was measured slower: 23.9 us against 19.6 us at 3s5z -- every workgroup streams the same 96 KB of
  // weight rows, and more requests in flight only queue up at the L2 channels that hold them.)
  const int KC = (S + 15) >> 4;
  const int NIT = (KC + 1) & ~1;
  Chunk c0, c1;
  fetch(c0, 0);
  for (int ci = 0; ci < NIT; ci += 2) {
    fetch(c1, ci + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(c0, ci);
    __builtin_amdgcn_sched_barrier(0);
    fetch(c0, ci + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(c1, ci + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!valid) return;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    if (q >= cnt) break;
    const HypTile& T = a.tile[it0 + q];
    f32x4 v = acc[q];
    if (T.relu) {                                // the hidden layers (q_mixer.py:41,46,62); hyper_b1 and the one-layer hyper_w1 / hyper_w2 are plain Linears
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    float* base = T.dst == HYP_HW1 ? a.out[net][HYP_HW1] : (T.dst == HYP_HW2 ? a.out[net][HYP_HW2] : (T.dst == HYP_HB2 ? a.out[net][HYP_HB2] :
                  (T.dst == HYP_HB1 ? a.out[net][HYP_HB1] : (T.dst == HYP_V1 ? a.out[net][HYP_V1] : a.out[net][HYP_V2]))));
    const int ldo = T.dst == HYP_HB1 || T.dst == HYP_V2 ? OPE_MIX : (T.dst == HYP_V1 ? a.ld[HYP_V1] : OPE_HYP);
    *reinterpret_cast<f32x4*>(base + (int64_t)m * ldo + T.col + 4 * g) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The chain kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int kCW = 8;                 // waves per workgroup = agents handled side by side
constexpr int kHidP = OPE_MIX + 4;     // LDS pitch of a 32-vector row
constexpr int kPartP = OPE_HYP + 4;    // LDS pitch of a 64-vector row
constexpr int kQaP = 17;               // [row][agent] pitch (N <= 16)
constexpr int kMx = 100;               // small mixer vectors staged per net: w2b_b [32], b2b_w [64], b2b_b [1] (+ pad)

// (value, index) maximum over the 4 lanes of a row -- greater value wins, equal values -> lower index (first max) -- with gfx950's VALU
// lane swaps instead of ds_bpermute round trips: each swap leaves {own, partner} in the two results in a lane-dependent order, the same for
// the value and the index, so the best of the two pairs is picked without knowing which is which.
__device__ __forceinline__ void row_argmax4p(float& v, int& k) {
  auto pick = [&](unsigned v0, unsigned v1, unsigned k0, unsigned k1) {
    const float f0 = __uint_as_float(v0), f1 = __uint_as_float(v1);
    const int i0 = (int)k0, i1 = (int)k1;
    const bool first = f0 > f1 || (f0 == f1 && i0 <= i1);
    v = first ? f0 : f1;
    k = first ? i0 : i1;
  };
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  auto b = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false);
  pick(a[0], a[1], b[0], b[1]);
  auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  auto d = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
  pick(c[0], c[1], d[0], d[1]);
}

// LayerNorm of this lane's 16 features of a row: y = affine output, xh = normalised row, returns 1 / std
__device__ __forceinline__ float ln_row16x(const float* __restrict__ hrow, const float* __restrict__ th, int lno_w, int lno_b, int g, f32x4 (&y)[4],
                                           f32x4 (&xh)[4]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    y[c] = *reinterpret_cast<const f32x4*>(hrow + 16 * c + 4 * g);
    s += (y[c][0] + y[c][1]) + (y[c][2] + y[c][3]);
  }
  const float mu = rowsum4(s) * (1.0f / OPE_H);
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = y[c][r] - mu;
      v = fmaf(d, d, v);
    }
  const float rstd = 1.0f / sqrtf(rowsum4(v) * (1.0f / OPE_H) + OPE_LN_EPS);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(th + lno_w + 16 * c + 4 * g);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(th + lno_b + 16 * c + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[c][r] = (y[c][r] - mu) * rstd;
      y[c][r] = fmaf(xh[c][r], gm[r], bt[r]);
    }
  }
  return rstd;
}

// Latency is what this kernel is about: a tile is ~600 MFMAs and a few thousand vector instructions per wave, but the first version walked
// ~20 DEPENDENT global round trips (row -> LayerNorm parameters -> head weights -> actions -> ... ) at 1.5-2 us each: 45.6 us per launch at
// 3s5z. Now everything a wave will need is requested in the first few hundred cycles -- one round trip --:
//   * what the 8 waves share goes through LDS, staged once per workgroup: both nets' head weights, biases and LayerNorm parameters, W2b of
//     both mixers, the tile's hw1 rows of both nets;
//   * what is private to a wave stays in flight in registers: its agent's three GRU-state rows, action / availability rows, its W1b slices of
//     both nets, the TD scalars of its rows, the rows of its side job (hw2 / hb2 / hyper_b1);
//   * the agent's W1b^T slice for the adjoint is requested at the end of the forward mixer stage, a phase ahead of its use (its registers
//     are the forward slices'); the chosen action's W_q row for the head's adjoint is read from the staged copy in LDS.
// One workgroup per CU (up to 256 registers a wave).
// H1: one-layer hyper-networks (hypernet_layers = 1, q_mixer.py:39-44): w1 and w2 come straight from the first-layer kernel (pre-abs values in
// v1x / v2x), so the second-stage products, their transposes in the adjoint and the hw1 / hw2 ReLU masks all drop out.
// PK: the packed rows of the live plan (LivePlan, ope_common.h). A workgroup takes 16 consecutive PACKED (t, b) rows -- the grid is sized for
// the padded batch, workgroups past the live rows leave at once (300 -> 225 workgroups at 3s5z, B = 32 with a quarter of the rows dead: ONE
// round on 256 CUs instead of two) --; a row's record gives its (t, b) in the batch (actions, availability, rewards, flags, PER weight) and
// the packed agent rows of steps t and t + 1 (GRU states in, saves and adjoints out). Where step t + 1 of the episode is not computed
// (t + 1 >= len_b, i.e. dones_env[t, b] = 1) the reference multiplies the target by 1 - dones_env = 0: the rows of step t stand in, finite.
template <int NT, int APW, bool VDN, bool H1, bool PK>     // NT: 16-action tiles of the head (A <= 16 NT); APW: agents per wave (N <= 8 APW)
__global__ void __launch_bounds__(64 * kCW, 2) qchain_kernel(ChainArgs a) {
  constexpr int QR = 16 * NT;                                                    // staged rows of W_q (zero beyond A)
  constexpr bool PFH = APW == 1;               // GRU-state rows requested at kernel start (one agent per wave: 48 registers)
  constexpr bool PFW = APW == 1 && NT == 1 && !H1;    // ... and the agent's W1b slices of both nets (64 registers); else loaded where they are used
  __shared__ __attribute__((aligned(16))) float wq_s[2][QR * kPartP];            // [net][action][64] head weights
  __shared__ __attribute__((aligned(16))) float qb_s[2][QR];                     // [net][action] head bias
  __shared__ __attribute__((aligned(16))) float ln_s[2][2][OPE_H];               // [net][gamma | beta][64] rnn.norm
  __shared__ __attribute__((aligned(16))) float hw1_s[2][16 * kPartP];           // [net][row][64] relu(hyper_w1.0) of the tile's rows
  __shared__ __attribute__((aligned(16))) float w2b_s[2][OPE_MIX * kPartP];      // [net][32][64] hyper_w2.2 weights
  __shared__ __attribute__((aligned(16))) float mx_s[2][kMx];                    // [net][hyper_w2.2 bias 32 | hyper_b2.2 weight 64 | its bias 1]
  __shared__ float qa_s[2][16 * kQaP];                                           // [net][row][agent]: chosen q (live), target q at t+1
  __shared__ __attribute__((aligned(16))) float wk_s[2 * kCW * 16 * kHidP];      // forward: [net][wave][16][36] partial hidden layers;
                                                                                 // backward: [wave][16][68] partial W1b^T dv1 (aliased)
  __shared__ __attribute__((aligned(16))) float hp_s[16 * kHidP];                // pre-ELU hidden layer of the live mixer
  __shared__ __attribute__((aligned(16))) float hd_s[16 * kHidP];                // ELU of it (the hidden layer itself)
  __shared__ __attribute__((aligned(16))) float v2_s[16 * kHidP];                // pre-abs w2 of the live mixer
  __shared__ float qt_s[2][2][16];                                               // [net][kh] halves of hidden . |w2| of the live / target net
  __shared__ float pb_s[2][16];                                                  // hyper_b2 head dot of both nets
  static_assert(2 * kCW * 16 * kHidP >= kCW * 16 * kPartP, "the backward partials alias the forward ones");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int A = a.A, N = a.N, B = a.B, NB = a.NB;
  const int tile = blockIdx.x;
  const int TBr = PK ? __builtin_amdgcn_readfirstlane(a.lp.hdr[2]) : a.TB;
  if (PK && tile * 16 >= TBr) {            // past the live rows: nothing to add to the loss sums (err_abs is cleared with / beside the plan)
    if (tid < 4) a.loss_part[tile * 4 + tid] = 0.f;
    return;
  }
  const int m = tile * 16 + j;
  const bool valid = m < TBr;
  const int mm = valid ? m : TBr - 1;
  int t, b, pr0 = 0, pn0 = 0, pr1 = 0, pn1 = 0;
  if (PK) {
    const int4 q0 = reinterpret_cast<const int4*>(a.lp.tbrec)[2 * (int64_t)mm], q1 = reinterpret_cast<const int4*>(a.lp.tbrec)[2 * (int64_t)mm + 1];
    t = q0.x; b = q0.y; pr0 = q0.z; pn0 = q0.w;
    pr1 = q1.z ? q1.x : q0.z; pn1 = q1.z ? q1.y : q0.w;      // no step t + 1: step t's rows stand in (their weight is exactly zero)
  } else {
    t = mm / B; b = mm - t * B;
  }
  // agent `ag`'s rows of steps t / t + 1: in the batch (r0s: actions; + NB: availability at t + 1) and where the step's arrays hold them
  auto row_s = [&](int ag) -> int64_t { return ((int64_t)t * N + ag) * B + b; };
  auto row_0 = [&](int ag) -> int64_t { return PK ? (int64_t)pr0 + (int64_t)ag * pn0 : ((int64_t)t * N + ag) * B + b; };
  auto row_1 = [&](int ag) -> int64_t { return PK ? (int64_t)pr1 + (int64_t)ag * pn1 : ((int64_t)t * N + ag) * B + b + NB; };
  const bool first = valid && g == 0;
  const AgentLayout& AL = a.AL;
  const MixerLayout& ML = a.ML;
  const float* __restrict__ th0 = a.theta0;
  const float* __restrict__ th1 = a.theta1;
  const int NM = N * OPE_MIX;
  const int A4 = ope_round4_dev(A);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // optional s_memtime stamps (ope_qmix_cfg.debug; tools/chain_phases.py): start | requests issued | shared data staged (after the barrier) |
  // heads done | after barrier | mixer stage 2 done | after the combine's two barriers | TD + adjoints done | after barrier | end
  long long* dbg = a.dbg ? a.dbg + ((int64_t)tile * kCW + wave) * 10 : nullptr;
  auto stamp = [&](int k) { if (dbg && lane == 0) dbg[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);

  // ---- requests: shared data (one 16-byte piece per thread and array), then this wave's private data ---------------------------------
  f32x4 st_wq[NT], st_hw1, st_w2b[2];
  float st_ln = 0.f, st_qb = 0.f;
#pragma unroll
  for (int u = 0; u < NT; ++u) {          // W_q of both nets: 2 x QR rows x 16 pieces
    const int p = tid + 512 * u;
    const int net = p / (QR * 16), row = (p >> 4) % QR, piece = p & 15;
    const float* __restrict__ th = net == 0 ? th0 : th1;
    st_wq[u] = *reinterpret_cast<const f32x4*>(th + AL.q_w + (int64_t)(row < A ? row : A - 1) * OPE_H + 4 * piece);
  }
  {                                       // hw1 rows of both nets: 2 x 16 rows x 16 pieces
    const int net = tid >> 8, row = (tid >> 4) & 15, piece = tid & 15;
    int mr = tile * 16 + row;
    mr = mr < TBr ? mr : TBr - 1;
    // (pointer chosen by a select of the two kernel arguments: indexing the argument array with a per-lane value would be a LOAD of the pointer)
    if (!VDN && !H1) st_hw1 = *reinterpret_cast<const f32x4*>((net ? a.hw1[1] : a.hw1[0]) + (int64_t)mr * OPE_HYP + 4 * piece);
  }
  if (!VDN && !H1) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {         // W2b of both nets: 2 x 32 rows x 16 pieces
      const int p = tid + 512 * u;
      const int net = p >> 9, row = (p >> 4) & 31, piece = p & 15;
      const float* __restrict__ th = net == 0 ? th0 : th1;
      st_w2b[u] = *reinterpret_cast<const f32x4*>(th + ML.w2b_w + (int64_t)row * OPE_HYP + 4 * piece);
    }
  }
  {   // rnn.norm gamma / beta of both nets (threads 0..255 keep theirs) and the head biases (threads 256 .. 256 + 2 QR): branch-free requests
      // with clamped addresses -- a load inside a divergent branch makes hipcc drain every outstanding load at the join
    const int net = (tid >> 7) & 1, which = (tid >> 6) & 1, f = tid & 63;
    st_ln = (net == 0 ? th0 : th1)[(which ? AL.lno_b : AL.lno_w) + f];
    const int q = (tid - 256) & (2 * QR - 1), netq = q / QR, k = q - netq * QR;
    st_qb = (netq == 0 ? th0 : th1)[AL.q_b + (k < A ? k : A - 1)];
    st_qb = k < A ? st_qb : 0.f;
  }
  float st_mx = 0.f;
  if (!VDN) {   // the mixers' small vectors (threads 0 .. 2 kMx keep theirs)
    const int e2 = tid < 2 * kMx ? tid : 2 * kMx - 1, net = e2 / kMx, e = e2 - net * kMx;
    const int off = e < OPE_MIX ? (H1 ? ML.b2b_b : ML.w2b_b + e) : (e < OPE_MIX + OPE_HYP ? ML.b2b_w + (e - OPE_MIX) : ML.b2b_b);
    st_mx = (net == 0 ? th0 : th1)[off];
  }
  // private: GRU-state rows (live t, live t + 1, target t + 1), action / availability rows of this wave's agents
  f32x4 hrow[3][4];
  float acv[APW][NT][4], avl[APW][NT][4];
  const float* __restrict__ avp = a.avail ? a.avail : a.h0;      // (no availability mask: read SOMETHING valid of at least that size, select below)
#pragma unroll
  for (int ia = 0; ia < APW; ++ia) {
    const int ag = wave + kCW * ia;
    const int agc = ag < N ? ag : N - 1;
    const int64_t r0 = row_0(agc), r1 = row_1(agc);
    const int64_t r0s = row_s(agc), r1s = r0s + NB;
    if (PFH) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        hrow[0][c] = *reinterpret_cast<const f32x4*>(a.h0 + r0 * OPE_H + 16 * c + 4 * g);
        hrow[1][c] = *reinterpret_cast<const f32x4*>(a.h0 + r1 * OPE_H + 16 * c + 4 * g);
        hrow[2][c] = *reinterpret_cast<const f32x4*>(a.h1 + r1 * OPE_H + 16 * c + 4 * g);
      }
    }
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int k = 16 * it + 4 * g + rr;
        const int kc = k < A ? k : A - 1;
        acv[ia][it][rr] = a.acts[r0s * A + kc];
        avl[ia][it][rr] = avp[r1s * A + kc];      // (no mask: inside h0's allocation, which is sized for the padded batch; never used)
      }
  }
  // TD scalars of this lane's (t, b) row (qmix.py:159-164)
  const float td_rew = a.td.rewards[((int64_t)t * N + 0) * B + b];
  const float td_den = a.td.dones_env[(int64_t)t * B + b];
  float td_bad = a.td.dones_env[(int64_t)(t > 0 ? t - 1 : 0) * B + b];
  td_bad = t == 0 ? 0.f : td_bad;
  float td_w = (a.td.per_weights ? a.td.per_weights : a.td.dones_env)[b];
  td_w = a.td.per_weights ? td_w : 1.0f;
  __builtin_amdgcn_sched_barrier(0);      // (keep the requests above where they are: the scheduler must not sink them towards their uses)
  stamp(1);

  // ---- deposit the shared pieces ---------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int p = tid + 512 * u;
    const int net = p / (QR * 16), row = (p >> 4) % QR, piece = p & 15;
    *reinterpret_cast<f32x4*>(&wq_s[net][row * kPartP + 4 * piece]) = row < A ? st_wq[u] : zero4;
  }
  if (!VDN && !H1) {
    const int net = tid >> 8, row = (tid >> 4) & 15, piece = tid & 15;
    *reinterpret_cast<f32x4*>(&hw1_s[net][row * kPartP + 4 * piece]) = st_hw1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = tid + 512 * u;
      const int net2 = p >> 9, row2 = (p >> 4) & 31, piece2 = p & 15;
      *reinterpret_cast<f32x4*>(&w2b_s[net2][row2 * kPartP + 4 * piece2]) = st_w2b[u];
    }
  }
  if (tid < 256) ln_s[tid >> 7][(tid >> 6) & 1][tid & 63] = st_ln;
  else if (tid < 256 + 2 * QR) qb_s[(tid - 256) / QR][(tid - 256) % QR] = st_qb;      // (stores in a branch are harmless)
  if (!VDN && tid < 2 * kMx) mx_s[tid / kMx][tid % kMx] = st_mx;
  lds_barrier();
  stamp(2);
  // ---- second wave of requests, in flight while the heads compute: this wave's agents' W1b slices of both nets (A operand fragments),
  // their biases, and the rows of the wave's side job:
  //   waves 0..3 = (net, kh) = (wave >> 1, wave & 1): w2[16 kh ..] = W2b relu(hw2) + b of that net, then that half of the combine;
  //   waves 4, 5: the hyper_b2 head dot of the live / target net
  f32x4 w1b[2][2][4], b1b[VDN ? 1 : APW][2][2];
  f32x4 side[4], hb1v, mk2, mk3;
  if (!VDN) {
#pragma unroll
    for (int ia = 0; ia < APW; ++ia) {
      const int ag = wave + kCW * ia;
      const int agc = ag < N ? ag : N - 1;
#pragma unroll
      for (int net = 0; net < 2; ++net) {
        const float* __restrict__ th = net == 0 ? th0 : th1;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          if (H1) {      // the pre-abs w1 slice itself
            b1b[ia][net][kh] = *reinterpret_cast<const f32x4*>((net ? a.v1x[1] : a.v1x[0]) + (int64_t)mm * NM + agc * OPE_MIX + 16 * kh + 4 * g);
            continue;
          }
          b1b[ia][net][kh] = *reinterpret_cast<const f32x4*>(th + ML.w1b_b + agc * OPE_MIX + 16 * kh + 4 * g);
          if (PFW) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
              w1b[net][kh][ft] = *reinterpret_cast<const f32x4*>(th + ML.w1b_w + (int64_t)(agc * OPE_MIX + 16 * kh + j) * OPE_HYP + 16 * ft + 4 * g);
          }
        }
      }
    }
    {
      const int snet = wave < 4 ? (wave >> 1) : (wave & 1);
      if (H1 && wave < 4) {      // this wave's half of the pre-abs w2
        side[0] = *reinterpret_cast<const f32x4*>((snet ? a.v2x[1] : a.v2x[0]) + (int64_t)mm * OPE_MIX + 16 * (wave & 1) + 4 * g);
        side[1] = side[2] = side[3] = side[0];
      } else {
        const float* __restrict__ src = ((wave < 4 && !H1) ? (snet ? a.hw2[1] : a.hw2[0]) : (snet ? a.hb2[1] : a.hb2[0])) + (int64_t)mm * OPE_HYP;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) side[ft] = *reinterpret_cast<const f32x4*>(src + 16 * ft + 4 * g);
      }
      hb1v = *reinterpret_cast<const f32x4*>((snet ? a.hb1[1] : a.hb1[0]) + (int64_t)mm * OPE_MIX + 16 * (wave & 1) + 4 * g);
      // the live net's relu(hw2) / relu(hb2) at the feature tile waves 0..3 finish in the adjoint (their ReLU masks)
      const int fo = 16 * (wave & 3) + 4 * g;
      mk2 = H1 ? zero4 : *reinterpret_cast<const f32x4*>(a.hw2[0] + (int64_t)mm * OPE_HYP + fo);
      mk3 = *reinterpret_cast<const f32x4*>(a.hb2[0] + (int64_t)mm * OPE_HYP + fo);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // LayerNorm of a prefetched row with the staged parameters: y = affine output, xh = normalised row, returns 1 / std
  auto ln_regs = [&](const f32x4 (&hr)[4], int net, f32x4 (&y)[4], f32x4 (&xn)[4]) -> float {
    float s0 = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s0 += (hr[c][0] + hr[c][1]) + (hr[c][2] + hr[c][3]);
    const float mu = rowsum4(s0) * (1.0f / OPE_H);
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = hr[c][r] - mu;
        v = fmaf(d, d, v);
      }
    const float rstd = __builtin_amdgcn_rsqf(rowsum4(v) * (1.0f / OPE_H) + OPE_LN_EPS);      // v_rsq_f32: 1 ulp
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(&ln_s[net][0][16 * c + 4 * g]);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(&ln_s[net][1][16 * c + 4 * g]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xn[c][r] = (hr[c][r] - mu) * rstd;
        y[c][r] = fmaf(xn[c][r], gm[r], bt[r]);
      }
    }
    return rstd;
  };
  // q[16 it + 4 g + r] of row j: bias + W_q y on the matrix pipe, weights from LDS (rows beyond A are zero)
  auto q_lds = [&](int net, const f32x4 (&y)[4], f32x4 (&q)[NT]) {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      q[it] = *reinterpret_cast<const f32x4*>(&qb_s[net][16 * it + 4 * g]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&wq_s[net][(16 * it + j) * kPartP + 16 * c + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) q[it] = mfma16(w[r], y[c][r], q[it]);
      }
    }
  };

  // ---- phase 1: the q heads of this wave's agents --------------------------------------------------------
  f32x4 xh[APW][4];
  float rstd_k[APW];
  int chosen_k[APW];
#pragma unroll
  for (int ia = 0; ia < APW; ++ia) {
    const int ag = wave + kCW * ia;
    if (ag < N) {
      const int64_t r0 = row_0(ag);                             // row (t, agent, b) of the [T+1][N*B] stacks (or its packed place)
      const int64_t r1 = row_1(ag);                             // (t + 1, agent, b)
      // chosen action = first maximum of the one-hot row (QMixPolicy.q_values_from_actions, QMixPolicy.py:69-93)
      float cv = kNegInf;
      int chosen = 1 << 30;
#pragma unroll
      for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int k = 16 * it + 4 * g + rr;
          if (k < A && acv[ia][it][rr] > cv) { cv = acv[ia][it][rr]; chosen = k; }      // ascending k within the lane: strict > keeps the first
        }
      row_argmax4p(cv, chosen);
      chosen_k[ia] = chosen;
      f32x4 y[4], q[NT];
      // live net at t: q of the action taken
      f32x4 hr[4];
      auto row_of = [&](int which) {      // the prefetched row, or (two agents per wave) the row loaded now
#pragma unroll
        for (int c = 0; c < 4; ++c)
          hr[c] = PFH ? hrow[which][c] : *reinterpret_cast<const f32x4*>((which == 2 ? a.h1 : a.h0) + (which == 0 ? r0 : r1) * OPE_H + 16 * c + 4 * g);
      };
      row_of(0);
      rstd_k[ia] = ln_regs(hr, 0, y, xh[ia]);
      if (valid) {
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(a.xhat_o + r0 * OPE_H + 16 * c + 4 * g) = xh[ia][c];
        if (g == 0) a.rstd_o[r0] = rstd_k[ia];
      }
      q_lds(0, y, q);
      if (a.q_all && valid) {
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (16 * it + 4 * g + rr < A) a.q_all[r0 * A + 16 * it + 4 * g + rr] = q[it][rr];
      }
      float qc = 0.f;
#pragma unroll
      for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qc += (16 * it + 4 * g + rr == chosen) ? q[it][rr] : 0.f;
      qc = rowsum4(qc);
      if (g == 0) qa_s[0][j * kQaP + ag] = qc;
      if (first) {
        a.act_idx[r0] = chosen;
        if (a.agent_q) a.agent_q[(int64_t)m * N + ag] = qc;
      }
      // live net at t + 1: greedy action over the available ones (double Q, qmix.py:138-146)
      float gv = kNegInf;
      int greedy = 1 << 30;
      f32x4 xd[4];
      if (a.double_q) {
        row_of(1);
        ln_regs(hr, 0, y, xd);
        q_lds(0, y, q);
        if (a.q_all && valid && t + 1 == a.T) {      // (debug output: the rows of the last time step are only ever seen as a "t + 1")
#pragma unroll
          for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              if (16 * it + 4 * g + rr < A) a.q_all[r1 * A + 16 * it + 4 * g + rr] = q[it][rr];
        }
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int k = 16 * it + 4 * g + rr;
            if (k < A) {
              const float qm = (a.avail && avl[ia][it][rr] == 0.f) ? -1e10f : q[it][rr];      // util.py:297-302
              if (qm > gv) { gv = qm; greedy = k; }
            }
          }
        row_argmax4p(gv, greedy);
      }
      // target net at t + 1: q at the live net's greedy action, or the plain maximum (qmix.py:148)
      row_of(2);
      ln_regs(hr, 1, y, xd);
      q_lds(1, y, q);
      float tq;
      if (a.double_q) {
        tq = 0.f;
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) tq += (16 * it + 4 * g + rr == greedy) ? q[it][rr] : 0.f;
        tq = rowsum4(tq);
      } else {
        tq = kNegInf;
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (16 * it + 4 * g + rr < A) tq = fmaxf(tq, (a.mask_target_max && a.avail && avl[ia][it][rr] == 0.f) ? -1e10f : q[it][rr]);
        tq = fmaxf(tq, __shfl_xor(tq, 16, 64));
        tq = fmaxf(tq, __shfl_xor(tq, 32, 64));
      }
      if (g == 0) qa_s[1][j * kQaP + ag] = tq;
      if (first && a.agent_nq) a.agent_nq[(int64_t)m * N + ag] = tq;
    }
  }
  stamp(3);
  lds_barrier();
  stamp(4);

  float qtot, nqtot;
  f32x4 v1k[APW][2];          // pre-abs w1 slices of this wave's agents (live mixer): needed again by the adjoint
  f32x4 w1t[VDN ? 1 : APW][4][2];   // this wave's agents' slices of W1b^T (the adjoint's A operand), requested at the end of phase 2
  f32x4 w2t[2];                     // ... and feature tile (wave & 3) of W2b^T
  if (VDN) {
    // Q_tot = sum over agents (every wave, redundantly: the TD below is lane-local)
    qtot = 0.f;
    nqtot = 0.f;
    for (int ag = 0; ag < N; ++ag) {
      qtot += qa_s[0][j * kQaP + ag];
      nqtot += qa_s[1][j * kQaP + ag];
    }
  } else {
    // ---- phase 2: second stage of both mixers, this wave's agents ----------------------------------------
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      f32x4 hv[4];
      if (!H1) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) hv[ft] = *reinterpret_cast<const f32x4*>(&hw1_s[net][j * kPartP + 16 * ft + 4 * g]);
      }
      f32x4 hid[2] = {zero4, zero4};
#pragma unroll
      for (int ia = 0; ia < APW; ++ia) {
        const int ag = wave + kCW * ia;
        if (ag < N) {
          f32x4 v[2] = {b1b[ia][net][0], b1b[ia][net][1]};
          const float* __restrict__ thn = net == 0 ? th0 : th1;
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) {
            if (H1) break;         // (v is the first-layer kernel's output already)
            f32x4 w[2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
              w[kh] = PFW ? w1b[net][kh][ft]
                          : *reinterpret_cast<const f32x4*>(thn + ML.w1b_w + (int64_t)(ag * OPE_MIX + 16 * kh + j) * OPE_HYP + 16 * ft + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int kh = 0; kh < 2; ++kh) v[kh] = mfma16(w[kh][r], hv[ft][r], v[kh]);
          }
          if (net == 0) {
            v1k[ia][0] = v[0];
            v1k[ia][1] = v[1];
            if (a.v1 && valid) {
#pragma unroll
              for (int kh = 0; kh < 2; ++kh) *reinterpret_cast<f32x4*>(a.v1 + (int64_t)m * NM + ag * OPE_MIX + 16 * kh + 4 * g) = v[kh];
            }
          }
          const float qa = qa_s[net][j * kQaP + ag];
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 4; ++r) hid[kh][r] = fmaf(qa, fabsf(v[kh][r]), hid[kh][r]);     // q_a |w1_a|  (q_mixer.py:82-86)
        }
      }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) *reinterpret_cast<f32x4*>(wk_s + ((net * kCW + wave) * 16 + j) * kHidP + 16 * kh + 4 * g) = hid[kh];
    }
    // the adjoint's W1b^T slices: requested now, used after the TD
    if (!H1) {
      const float* __restrict__ w1bT = a.mixT;                    // [64][N*32]
#pragma unroll
      for (int ia = 0; ia < APW; ++ia) {
        const int ag = wave + kCW * ia;
        const int agc = ag < N ? ag : N - 1;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
            w1t[ia][ft][kh] = *reinterpret_cast<const f32x4*>(w1bT + (int64_t)(16 * ft + j) * NM + agc * OPE_MIX + 16 * kh + 4 * g);
      }
      const float* __restrict__ w2bT = a.mixT + OPE_HYP * NM;     // [64][32]
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) w2t[kh] = *reinterpret_cast<const f32x4*>(w2bT + (int64_t)(16 * (wave & 3) + j) * OPE_MIX + 16 * kh + 4 * g);
    }
    // side jobs (see the requests above): waves 0..3 their half of w2 = W2b relu(hw2) + b, waves 4 / 5 the hyper_b2 head dots
    f32x4 v2h = zero4;
    if (wave < 4 && H1) {
      v2h = side[0];
    } else if (wave < 4) {
      const int net = wave >> 1, kh = wave & 1;
      v2h = *reinterpret_cast<const f32x4*>(&mx_s[net][16 * kh + 4 * g]);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(&w2b_s[net][(16 * kh + j) * kPartP + 16 * ft + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v2h = mfma16(w[r], side[ft][r], v2h);
      }
    } else if (wave < 6) {
      const int net = wave & 1;
      float pb = 0.f;
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&mx_s[net][OPE_MIX + 16 * ft + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) pb = fmaf(wv[r], side[ft][r], pb);
      }
      pb = rowsum4(pb);
      if (g == 0) pb_s[net][j] = pb;
    }
    stamp(5);
    lds_barrier();

    // ---- phase 3: combine, one (net, kh) half of the 32 hidden units per wave 0..3 ---------------------------
    if (wave < 4) {
      const int net = wave >> 1, kh = wave & 1;
      f32x4 h = hb1v;
#pragma unroll
      for (int w = 0; w < kCW; ++w) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(wk_s + ((net * kCW + w) * 16 + j) * kHidP + 16 * kh + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] += o[r];
      }
      f32x4 hd;
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hd[r] = elu1(h[r]);
        part = fmaf(hd[r], fabsf(v2h[r]), part);            // ELU(.) |w2|  (q_mixer.py:86-93)
      }
      if (net == 0) {
        *reinterpret_cast<f32x4*>(hp_s + j * kHidP + 16 * kh + 4 * g) = h;
        *reinterpret_cast<f32x4*>(hd_s + j * kHidP + 16 * kh + 4 * g) = hd;
        *reinterpret_cast<f32x4*>(v2_s + j * kHidP + 16 * kh + 4 * g) = v2h;
        if (a.hpre && valid) *reinterpret_cast<f32x4*>(a.hpre + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = h;
        if (a.v2 && valid) *reinterpret_cast<f32x4*>(a.v2 + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = v2h;
      }
      part = rowsum4(part);
      if (g == 0) qt_s[net][kh][j] = part;
    }
    lds_barrier();
    qtot = (qt_s[0][0][j] + qt_s[0][1][j]) + (pb_s[0][j] + mx_s[0][OPE_MIX + OPE_HYP]);
    nqtot = (qt_s[1][0][j] + qt_s[1][1][j]) + (pb_s[1][j] + mx_s[1][OPE_MIX + OPE_HYP]);
  }
  stamp(6);

  // ---- phase 4: TD target, mask, loss (every wave, lane-local; qmix.py:158-176) and the adjoints ------------
  TdOut td;
  {
    td.keep = 1.0f - td_bad;
    const float target = td_rew + (1.0f - td_den) * a.td.gamma * nqtot;
    const float e = (qtot - target) * td.keep;
    td.err = e;
    float fe, dfe;
    if (a.td.use_huber) {
      const float ae = fabsf(e), dl = a.td.huber_delta;
      if (ae <= dl) { fe = e * e * 0.5f; dfe = e; }
      else { fe = dl * (ae - dl * 0.5f); dfe = dl * sgn(e); }
    } else {
      fe = e * e;
      dfe = 2.0f * e;
    }
    td.lossel = td_w * fe;
    td.dq = td_w * dfe * td.keep;
  }
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
    if (first) {
      a.err_abs[PK ? t * B + b : m] = fabsf(td.err);
      if (!VDN) *reinterpret_cast<f32x4*>(a.dqtot + 4 * (int64_t)m) = f32x4{dQ, 0.f, 0.f, 0.f};       // [TB][4]: lda = 4 for the wgrad kernel
      if (a.qtot) { a.qtot[m] = qtot; a.nqtot[m] = nqtot; }
    }
  }
  float dqa_k[APW];
  if (VDN) {
#pragma unroll
    for (int ia = 0; ia < APW; ++ia) dqa_k[ia] = dQ;                                 // d Q_tot / d q_a = 1
  } else {
    f32x4 dpre[2], dv2[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const f32x4 hp = *reinterpret_cast<const f32x4*>(hp_s + j * kHidP + 16 * kh + 4 * g);
      const f32x4 hd = *reinterpret_cast<const f32x4*>(hd_s + j * kHidP + 16 * kh + 4 * g);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(v2_s + j * kHidP + 16 * kh + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dv2[kh][r] = dQ * hd[r] * sgn(vv[r]);
        const float dh = dQ * fabsf(vv[r]);
        dpre[kh][r] = dh * (hp[r] > 0.f ? 1.0f : hd[r] + 1.0f);      // ELU'(x) = exp(x) = ELU(x) + 1 for x <= 0 (the combine's value: no second exponential)
      }
      if (valid && wave == 0) {
        *reinterpret_cast<f32x4*>(a.d_b1 + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = dpre[kh];
        *reinterpret_cast<f32x4*>(a.d_v2 + (int64_t)m * OPE_MIX + 16 * kh + 4 * g) = dv2[kh];
      }
    }
    f32x4 dh1[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) dh1[ft] = zero4;
#pragma unroll
    for (int ia = 0; ia < APW; ++ia) {
      const int ag = wave + kCW * ia;
      dqa_k[ia] = 0.f;
      if (ag < N) {
        const float qa = qa_s[0][j * kQaP + ag];
        float dqa = 0.f;
        f32x4 dv1[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dqa = fmaf(dpre[kh][r], fabsf(v1k[ia][kh][r]), dqa);
            dv1[kh][r] = dpre[kh][r] * qa * sgn(v1k[ia][kh][r]);
          }
          if (valid) *reinterpret_cast<f32x4*>(a.d_v1 + (int64_t)m * NM + ag * OPE_MIX + 16 * kh + 4 * g) = dv1[kh];
        }
        dqa = rowsum4(dqa);
        dqa_k[ia] = dqa;
        if (first && a.d_agent_q) a.d_agent_q[(int64_t)m * N + ag] = dqa;
        if (!H1) {
#pragma unroll
          for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
              for (int r = 0; r < 4; ++r) dh1[ft] = mfma16(w1t[ia][ft][kh][r], dv1[kh][r], dh1[ft]);
        }
      }
    }
    // (all reads of the forward partials in wk_s happened before the barrier that closed phase 3)
    if (!H1) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) *reinterpret_cast<f32x4*>(wk_s + (wave * 16 + j) * kPartP + 16 * ft + 4 * g) = dh1[ft];
    }
    f32x4 dh2 = zero4;
    if (wave < 4 && !H1) {          // feature tile `wave` of dhw2 = W2b^T dv2
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh2 = mfma16(w2t[kh][r], dv2[kh][r], dh2);
    }
    stamp(7);
    if (!H1) lds_barrier();
    stamp(8);
    if (wave < 4 && valid && H1) {   // only hyper_b2 has a hidden layer: its pre-activation adjoint
      const int fo = 16 * wave + 4 * g;
      const f32x4 wb = *reinterpret_cast<const f32x4*>(&mx_s[0][OPE_MIX + fo]);
      f32x4 o3;
#pragma unroll
      for (int r = 0; r < 4; ++r) o3[r] = mk3[r] > 0.f ? dQ * wb[r] : 0.f;
      *reinterpret_cast<f32x4*>(a.d_hb2 + (int64_t)m * OPE_HYP + fo) = o3;
    }
    if (wave < 4 && valid && !H1) {   // feature tile `wave` of the three hyper-net adjoints (pre-activation: zero where the ReLU was off)
      const int fo = 16 * wave + 4 * g;
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(&hw1_s[0][j * kPartP + fo]);
      const f32x4 h2 = mk2, h3 = mk3;
      const f32x4 wb = *reinterpret_cast<const f32x4*>(&mx_s[0][OPE_MIX + fo]);
      f32x4 s1 = *reinterpret_cast<const f32x4*>(wk_s + (0 * 16 + j) * kPartP + fo);
#pragma unroll
      for (int w = 1; w < kCW; ++w) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(wk_s + (w * 16 + j) * kPartP + fo);
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[r] += p[r];
      }
      f32x4 o1, o2, o3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o1[r] = h1[r] > 0.f ? s1[r] : 0.f;
        o2[r] = h2[r] > 0.f ? dh2[r] : 0.f;
        o3[r] = h3[r] > 0.f ? dQ * wb[r] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.d_hw1 + (int64_t)m * OPE_HYP + fo) = o1;
      *reinterpret_cast<f32x4*>(a.d_hw2 + (int64_t)m * OPE_HYP + fo) = o2;
      *reinterpret_cast<f32x4*>(a.d_hb2 + (int64_t)m * OPE_HYP + fo) = o3;
    }
  }

  // ---- phase 5: adjoint of the q head of this wave's agents: dq at the chosen action -> LayerNorm adjoint -> dh_out ----
#pragma unroll
  for (int ia = 0; ia < APW; ++ia) {
    const int ag = wave + kCW * ia;
    if (ag < N && valid) {
      const int64_t r0 = row_0(ag);
      const float dq = dqa_k[ia];
      const int act = chosen_k[ia];
      for (int k0 = 4 * g; k0 < A4; k0 += 16)
        *reinterpret_cast<f32x4*>(a.dqoh + r0 * A4 + k0) =
            f32x4{k0 == act ? dq : 0.f, k0 + 1 == act ? dq : 0.f, k0 + 2 == act ? dq : 0.f, k0 + 3 == act ? dq : 0.f};
      f32x4 d[4];
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(&ln_s[0][0][16 * c + 4 * g]);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(&wq_s[0][act * kPartP + 16 * c + 4 * g]);      // W_q row of the chosen action (staged)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          d[c][rr] = dq * wa[rr];
          d[c][rr] *= gm[rr];
          m1 += d[c][rr];
          m2 = fmaf(d[c][rr], xh[ia][c][rr], m2);
        }
      }
      // (rows drop out of this block as whole quads of lanes -- `valid` depends on j only -- so the 4-lane sums stay complete)
      m1 = rowsum4(m1) * (1.0f / OPE_H);
      m2 = rowsum4(m2) * (1.0f / OPE_H);
      const float rs = rstd_k[ia];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) d[c][rr] = rs * (d[c][rr] - m1 - xh[ia][c][rr] * m2);
        *reinterpret_cast<f32x4*>(a.dh_out + r0 * OPE_H + 16 * c + 4 * g) = d[c];
      }
    }
  }
  stamp(9);
}

}  // namespace

bool qchain_shape_ok(int N, int A) { return N >= 1 && N <= 2 * kCW && A >= 1 && A <= 32; }

void hyp_tiles_for(const MixerLayout& L, int N, int S, HypFirstArgs* a) {
  int n = 0;
  auto add = [&](int w, int b, int rows, int dst, int relu) {
    for (int r = 0; r < rows; r += 16) {
      HypTile& T = a->tile[n++];
      T.w_off = w + r * S; T.b_off = b + r; T.dst = dst; T.col = r; T.relu = relu;
    }
  };
  if (L.one_layer) {
    add(L.w1a_w, L.w1a_b, N * OPE_MIX, HYP_V1, 0);
    add(L.w2a_w, L.w2a_b, OPE_MIX, HYP_V2, 0);
    add(L.b1_w, L.b1_b, OPE_MIX, HYP_HB1, 0);
    add(L.b2a_w, L.b2a_b, OPE_HYP, HYP_HB2, 1);
  } else {
    add(L.w1a_w, L.w1a_b, OPE_HYP, HYP_HW1, 1);
    add(L.w2a_w, L.w2a_b, OPE_HYP, HYP_HW2, 1);
    add(L.b2a_w, L.b2a_b, OPE_HYP, HYP_HB2, 1);
    add(L.b1_w, L.b1_b, OPE_MIX, HYP_HB1, 0);
  }
  a->ntiles = n;
  a->ld[HYP_HW1] = a->ld[HYP_HW2] = a->ld[HYP_HB2] = OPE_HYP;
  a->ld[HYP_HB1] = a->ld[HYP_V2] = OPE_MIX;
  a->ld[HYP_V1] = N * OPE_MIX;
}

int launch_mixer_hyp(const HypFirstArgs& a0, hipStream_t st) {
  if (a0.TB < 1 || a0.S < 1 || a0.B < 1 || a0.ntiles < 1 || a0.ntiles > kHypMaxTiles) return OPE_EINVAL;
  HypFirstArgs a = a0;
  const int tiles = ope_cdiv(a.TB, 16);
  const int ncg = ope_cdiv(a.ntiles, 7);
  // Row tiles (= waves) per workgroup: 4, or as many more as it takes for the 2 * ncg * ceil(tiles / rpw) workgroups to be at most one per CU.
  // Measured at 3s5z: 300 workgroups on 256 CUs (B = 32) run as two rounds, 23.7 us, where 228 (B = 24) take 12.9 us.
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  int rpw = 4;
  while (rpw < 8 && 2 * ncg * ope_cdiv(tiles, rpw) > cus) ++rpw;
  a.main_blocks = 2 * ncg * ope_cdiv(tiles, rpw);
  const int threads = 64 * rpw;
  const int blocks = a.main_blocks + (a.side.total > 0 ? ope_cdiv(a.side.total, threads) : 0);
  const int vec = ope_vec_of(a.S);
  kprof_work(2.0 * 2.0 * a.TB * (double)a.S * 16.0 * a.ntiles);
  if (a.lp.hdr) kprof_rows(3);
  const bool pk = a.lp.hdr != nullptr;
  if (pk) {
    if (vec == 4) OPE_LAUNCH((mixer_hyp_kernel<4, true>), dim3(blocks), dim3(threads), 0, st, a);
    else if (vec == 2) OPE_LAUNCH((mixer_hyp_kernel<2, true>), dim3(blocks), dim3(threads), 0, st, a);
    else OPE_LAUNCH((mixer_hyp_kernel<1, true>), dim3(blocks), dim3(threads), 0, st, a);
  } else {
    if (vec == 4) OPE_LAUNCH((mixer_hyp_kernel<4, false>), dim3(blocks), dim3(threads), 0, st, a);
    else if (vec == 2) OPE_LAUNCH((mixer_hyp_kernel<2, false>), dim3(blocks), dim3(threads), 0, st, a);
    else OPE_LAUNCH((mixer_hyp_kernel<1, false>), dim3(blocks), dim3(threads), 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(pk ? "mixer_hyp_live" : "mixer_hyp", vec);
  return OPE_OK;
}

int launch_qchain(const ChainArgs& a, hipStream_t st) {
  if (a.TB < 1 || !qchain_shape_ok(a.N, a.A)) return OPE_EINVAL;
  const int blocks = ope_cdiv(a.TB, 16);
  const int nt = a.A <= 16 ? 1 : 2, apw = a.N <= kCW ? 1 : 2;
  // heads (three 16-row evaluations per agent and tile), W1b of both nets, W2b, the two transposed products
  kprof_work(2.0 * a.TB * ((double)a.N * 3.0 * OPE_H * a.A + ((a.vdn || a.ML.one_layer) ? 0.0 : (2.0 * ((double)a.N * OPE_MIX * OPE_HYP + OPE_MIX * OPE_HYP) +
                                                                             (double)a.N * OPE_MIX * OPE_HYP + OPE_MIX * OPE_HYP))));
  const bool pk = a.lp.hdr != nullptr;
  if (pk && (a.q_all || a.agent_q || a.agent_nq || a.qtot || a.v1 || a.v2 || a.hpre || a.d_agent_q)) return OPE_EINVAL;      // (the debug outputs are batch-indexed)
#define OPE_QCHAIN2(NT_, APW_, PK_)                                                                                 \
  do {                                                                                                              \
    if (a.vdn) OPE_LAUNCH((qchain_kernel<NT_, APW_, true, false, PK_>), dim3(blocks), dim3(64 * kCW), 0, st, a);    \
    else if (a.ML.one_layer) OPE_LAUNCH((qchain_kernel<NT_, APW_, false, true, PK_>), dim3(blocks), dim3(64 * kCW), 0, st, a); \
    else OPE_LAUNCH((qchain_kernel<NT_, APW_, false, false, PK_>), dim3(blocks), dim3(64 * kCW), 0, st, a);         \
  } while (0)
  if (pk) kprof_rows(3);
#define OPE_QCHAIN(NT_, APW_)                   \
  do {                                          \
    if (pk) OPE_QCHAIN2(NT_, APW_, true);       \
    else OPE_QCHAIN2(NT_, APW_, false);         \
  } while (0)
  if (nt == 1 && apw == 1) OPE_QCHAIN(1, 1);
  else if (nt == 2 && apw == 1) OPE_QCHAIN(2, 1);
  else if (nt == 1) OPE_QCHAIN(1, 2);
  else OPE_QCHAIN(2, 2);
#undef OPE_QCHAIN
#undef OPE_QCHAIN2
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(pk ? (a.vdn ? "qchain_vdn_live" : (a.ML.one_layer ? "qchain_h1_live" : "qchain_live")) : (a.vdn ? "qchain_vdn" : (a.ML.one_layer ? "qchain_h1" : "qchain")), nt, apw);
  return OPE_OK;
}

}  // namespace ope
