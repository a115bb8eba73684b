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
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "ope_agent.h"

namespace ope {
namespace {

constexpr int kAhead = 8;   // loader prefetch distance in steps

__device__ __forceinline__ float hsum4(f32x2 a, f32x2 b) { return (a[0] + a[1]) + (b[0] + b[1]); }
// sum over the G = 2 or 4 adjacent lanes that share a feature (DPP, every lane ends with the same bits)
template <int G>
__device__ __forceinline__ float group_sum_from(float v) {   // the stages after the first (lanes l, l^1 already hold their pair's sum)
  if (G >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  if (G == 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad of the eight
  return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  return group_sum_from<G>(v);
}

// d = w * {hp[HALF], hp[HALF]} + c: a packed FMA whose second operand is one half of a register pair, broadcast by op_sel
// (hipcc folds most such splats itself but copies some through a v_mov; written out so that none is left on the chain)
template <int HALF>
__device__ __forceinline__ f32x2 pk_fma_bcast(f32x2 w, f32x2 hp, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,%4,0] op_sel_hi:[1,%4,1]" : "=v"(d) : "v"(w), "v"(hp), "v"(c), "n"(HALF));
  return d;
}
template <int HALF>
__device__ __forceinline__ f32x2 pk_mul_bcast(f32x2 w, f32x2 hp) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,%3] op_sel_hi:[1,%3]" : "=v"(d) : "v"(w), "v"(hp), "n"(HALF));
  return d;
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// PK: packed rows of the live plan (LivePlan, ope_common.h): the workgroup of (agent, ranked episode j) walks the len[j] steps the episode
// needs and stops -- every later (t, b) of that episode is multiplied by a zero mask in the loss (qmix.py:161-166) --; step t's rows are
// at N * cum[t] + agent * n[t] + j, a table the workgroup builds in LDS before the chain starts (the loader / storer read it, off the chain).
// EXP (instantiated only in builds with -DOPE_EXPERIMENTS; TIMING variants, results WRONG; OPE_GRU_EXP, profiles/r06_gru4_decomposition.txt):
// leave one part of the step out and see what the launch loses -- 1 the gates (exp2 / rcp chains become one FMA each), 2 the per-step barrier
// (all roles run free: races), 4 the saves' LDS publishes (r, z, n, gh_n: h only), 8 the h reads from LDS (registers instead: no exchange
// latency), 16 the 24 packed FMAs of the mat-vec.
template <int W, bool DBG, bool PK, int EXP = 0>   // W = compute waves per row = lanes per feature (2 or 4)
__global__ void __launch_bounds__((W + 2) * 64) gru_fwd4_kernel(GruFwdArgs a) {
  constexpr int FPW = OPE_H / W;     // features per compute wave
  constexpr int KS = OPE_H / W;      // K-slice per lane
  constexpr int NP = KS / 2;         // weight pairs per gate per lane
  // LDS (floats): h_t [2][64] | r, z, n, gh_n' of step t [2][4][64] | gate inputs of step t: gx [2][2][65] = x_r, x_z planes
  // (x = -log2(e) (gi + b_hh); entry 64 stays zero: the lanes that must not inject read it) | gn [2][64] = -2 log2(e) gi_n |
  // a dump area the lanes that hold no valid n / h write to (same offsets as the real ones, shifted by kDump)
  // Bank placement (round 4; PMC: 0.93 M SQ_LDS_BANK_CONFLICT cycles per launch in this kernel, none in the backward one). The LDS
  // serves a ds_write half a wave at a time over 32 banks (measured: moving the planes by a multiple of 32 floats changes nothing, and
  // a dump shift of 32 DOUBLED the count); half a wave holds 8 (W = 4) or 16 (W = 2) consecutive features. The r / z publish put z
  // exactly 64 floats behind r: same bank for the same feature, 2-way on every step. Planes are now kPl = 80 floats apart (z lands 16
  // banks beside r); the dump shift, 8 banks before, becomes 16 so that it clears the 16-feature half-wave of W = 2 as well.
  constexpr int kPl = OPE_H + 16;
  constexpr int kHs = 0, kSv = 2 * OPE_H, kGx = kSv + 8 * kPl, kGxP = 2 * (OPE_H + 1) + 2, kGxZ = OPE_H + 1, kGn = kGx + 2 * kGxP,
                kDump = ((kGn + 2 * OPE_H + 63) / 64) * 64 + 16;
  __shared__ __attribute__((aligned(16))) float sm[kDump + kSv + 8 * kPl];
  float(*hs)[OPE_H] = reinterpret_cast<float(*)[OPE_H]>(sm);                        // [2][64]    h_t, parity of t
  float(*sv)[4][kPl] = reinterpret_cast<float(*)[4][kPl]>(sm + kSv);                // [2][4][64 (+ 16 unused)]
  constexpr float kL = -1.4426950408889634f;
  __builtin_amdgcn_s_setprio(3);      // every instruction of a scan wave is on the step chain: win the issue arbitration against whatever shares the CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Which (net, agent, ranked episode) a workgroup walks. Padded rows: block = net * NB + row. Packed rows (PK): the dispatcher hands block b to
  // CU b % 256 (tools/microbench_placement.hip: blocks b and b + 256 of this launch shape share a CU on every run, XCD = b % 8), two
  // workgroups per CU at 512 rows, and a step costs a scan 0.41 us beside a neighbour, 0.29 us alone -- so the launch ends earliest when every
  // CU holds a LONG episode beside a SHORT one (the same episode of both nets side by side, the plain mapping, keeps the 16 longest chains
  // paired for all of their steps: 60 us against 54). The 2 NB chains, ordered by length (episode rank major), are dealt out in a snake over
  // the CUs: slot s = b / C takes entries s C .. s C + C - 1, odd slots backwards. Placement only: any other assignment is as correct.
  int net, pk_ag = 0, pk_j = 0, row;
  if (PK && a.pair_cus > 0) {
    const int C = a.pair_cus, G = a.nets * a.NB;
    const int s = blockIdx.x / C, c = blockIdx.x - s * C;
    const int lo = s * C, cnt = min(C, G - lo);
    const int idx = lo + ((s & 1) ? cnt - 1 - c : c);
    pk_j = idx / (a.nets * a.N);
    const int r = idx - pk_j * (a.nets * a.N);
    net = r / a.N;
    pk_ag = r - net * a.N;
    row = pk_ag * a.B + pk_j;
  } else {
    net = blockIdx.x / a.NB;
    row = blockIdx.x - net * a.NB;
    if (PK) { pk_ag = row / a.B; pk_j = row - pk_ag * a.B; }
  }
  __shared__ int rows_s[PK ? kLiveMaxT + 2 : 1];
  int L = a.L;
  if (PK) L = __builtin_amdgcn_readfirstlane(a.lp.len[pk_j]);
  // The step -> row table: every wave fills its share and meets the others at ONE extra barrier -- the loader and the storer at once, the
  // compute waves behind the requests for their W_hh rows (the table's round trip to L2 then overlaps the weights').
  auto build_rows = [&]() {
    if (!PK) return;
    for (int t = threadIdx.x; t < L; t += (W + 2) * 64) rows_s[t] = (a.N * a.lp.cum[t] + pk_ag * a.lp.nn[t] + pk_j) * (4 * OPE_H);      // byte offset of the row in a [rows][64] array
    __syncthreads();
  };
  const int nchunks = (L + kAhead - 1) / kAhead;
  const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;

  if (wave == W) {   // ---- loader: lane = feature
    build_rows();
    const float* __restrict__ gi = net == 0 ? a.gi0 : a.gi1;
    const int64_t stride_t = (int64_t)a.NB * (3 * OPE_H);
    const float* gp = gi + (int64_t)row * (3 * OPE_H);          // uniform: the lane enters as the 32-bit offset of the load
    const unsigned lo4 = 4u * lane;
    const float brs = th[a.bhh_off + lane] * kL, bzs = th[a.bhh_off + OPE_H + lane] * kL;
    float pre[kAhead][3];
    auto row_at = [&](int t) -> int { return PK ? rows_s[min(t, L - 1)] : 0; };
    auto load_step = [&](float (&d)[3], int t, int prow) {      // prow (PK): the step's row, read from the table ahead of time
      const float* p = PK ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(gi) + 3ull * (unsigned)__builtin_amdgcn_readfirstlane(prow)) : gp + (int64_t)min(t, L - 1) * stride_t;
      gload_async_s<0>(d[0], p, lo4);
      gload_async_s<4 * OPE_H>(d[1], p, lo4);
      gload_async_s<8 * OPE_H>(d[2], p, lo4);
    };
    // the scaling that turns a sigmoid into rcp(1 + exp2(.)) and the b_hh of the r / z gates are applied here, off the chain
    auto publish = [&](const float (&d)[3], int q) {
      // scalar, and to separate planes, on purpose: anything that wants the two values in a register PAIR (a packed FMA, a
      // ds_write_b64) makes hipcc copy the landed registers into the pair BEFORE the s_waitcnt that guards them
      float xr = fmaf(d[0], kL, brs), xz = fmaf(d[1], kL, bzs);
      asm("" : "+v"(xr), "+v"(xz));
      sm[kGx + q * kGxP + lane] = xr;
      sm[kGx + q * kGxP + kGxZ + lane] = xz;
      sm[kGn + q * OPE_H + lane] = d[2] * (2.0f * kL);
    };
    if (lane < 4) sm[kGx + (lane >> 1) * kGxP + (lane & 1) * kGxZ + OPE_H] = 0.f;
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_step(pre[s], s, row_at(s));
    OPE_GWAIT24(pre);
    publish(pre[0], 0);
    load_step(pre[0], kAhead, row_at(kAhead));
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
      for (int s = 0; s < kAhead; ++s) {
        const int t = c * kAhead + s;
        if (t < L) {
          float(&d)[3] = pre[(s + 1) % kAhead];      // holds step t+1; 7 x 3 younger loads are in flight behind it
          // the table entry of the step requested below: asked for before the wait, so the LDS round trip hides behind it
          const int prow = row_at(t + 1 + kAhead);
          if (PK) __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt vmcnt(21)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])::"memory");
          publish(d, (s + 1) & 1);
          load_step(d, t + 1 + kAhead, prow);
          if (!(EXP & 2)) lds_barrier();
        }
      }
    }
    return;
  }
  if (wave == W + 1) {   // ---- storer: lane = feature; step t-1's results leave during step t
    build_rows();
    float* __restrict__ hout = net == 0 ? a.h0out : a.h1out;
    const bool save = (net == 0) && (a.rg != nullptr);
    lds_barrier();
    for (int t = 0; t <= L; ++t) {
      if (t > 0) {
        const int p = (t - 1) & 1;
        const int64_t o = (PK ? (int64_t)((unsigned)__builtin_amdgcn_readfirstlane(rows_s[t - 1]) >> 2) : ((int64_t)(t - 1) * a.NB + row) * OPE_H) + lane;
        hout[o] = hs[p ^ 1][lane];   // h_t was published as the next step's input
        if (save) {
          a.rg[o] = sv[p][0][lane];
          a.zg[o] = sv[p][1][lane];
          a.ng[o] = sv[p][2][lane];
          a.ghn[o] = sv[p][3][lane] * (1.0f / (2.0f * kL));   // the n rows of W_hh carry the tanh's -2 log2(e)
        }
      }
      if (t < L && !(EXP & 2)) lds_barrier();
    }
    return;
  }

  // ---- compute wave q: feature f = FPW q + j, K-slice g
  const int j = lane / W, g = lane % W;
  const int f = FPW * wave + j;
  // Weights of the lane: the r and z rows of feature f PAIRED per k ({W_hr[f][k], W_hz[f][k]}: one packed FMA with h[k] broadcast
  // to both halves advances both gates, and no horizontal add is needed for them), the n row in k-pairs.
  f32x2 wrz[KS], wn[NP];
  {
    const float* w = th + a.whh_off + KS * g;
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
      const f32x4 vr = *reinterpret_cast<const f32x4*>(w + (int64_t)f * OPE_H + 4 * k);
      const f32x4 vz = *reinterpret_cast<const f32x4*>(w + (int64_t)(OPE_H + f) * OPE_H + 4 * k);
      const f32x4 vn = *reinterpret_cast<const f32x4*>(w + (int64_t)(2 * OPE_H + f) * OPE_H + 4 * k);
#pragma unroll
      for (int e = 0; e < 4; ++e) wrz[4 * k + e] = f32x2{vr[e], vz[e]};
      wn[2 * k] = f32x2{vn[0], vn[1]}; wn[2 * k + 1] = f32x2{vn[2], vn[3]};
    }
  }
  const float bn = th[a.bhh_off + 2 * OPE_H + f];
  build_rows();
  const float* hin = net == 0 ? a.hinit : a.hinit1;
  float h = hin ? hin[(int64_t)row * OPE_H + f] : 0.f;
  // The sigmoids are evaluated as rcp(1 + exp2(x')) with x' = -log2(e) x and tanh(x) as 2 rcp(1 + exp2(-2 log2(e) x)) - 1
  // (absolute error ~1e-7): the scales are folded into the rows of W_hh, into b_hn and (by the loader) into gi and the r / z
  // biases, so that after the quad reduction a gate is exp2, add, rcp. The storer takes the scale out of the saved gh_n.
#pragma unroll
  for (int i = 0; i < KS; ++i) wrz[i] *= kL;
#pragma unroll
  for (int i = 0; i < NP; ++i) wn[i] *= 2.0f * kL;
  // gi and the biases enter the sums once per quad: as the addend of lane g == 0's first FMA. The other lanes read zeros
  // through their own (loop-invariant) address, so no select sits on the chain.
  const f32x2 bn0 = {(g == 0) ? bn * (2.0f * kL) : 0.f, 0.f};
  const int gx_off = kGx + (g == 0 ? f : OPE_H);
  // Lane roles after the reduction: the even lanes of a feature evaluate r, then n and h'; the odd lanes evaluate z (one exp2 / rcp
  // sequence serves both sigmoids). In the first exchange of the reduction (lane ^ 1) every lane keeps the gate of its role and
  // hands the other one over; the later stage stays inside a role. Only the even lanes hold a valid n / h': the odd ones write
  // theirs to the dump area (their h stays a bounded mix of gate values and is never read).
  constexpr int kSwap = 0xB1;       // quad_perm [1,0,3,2]
  const bool even = (g & 1) == 0;
  const int rz_off = kSv + (even ? 0 : kPl) + f;        // r from one side, z from the other: one ds_write
  const int val_off = f + (even ? 0 : kDump);
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
    // LDS returns in order: what the first FMAs need (h[0..3] of the slice, the injected pair) is asked for first, gi_n last
    const float* hp = &hs[p][KS * g];
    f32x4 hq[KS / 4];
    if (EXP & 8) {      // no exchange: every value from the lane's own state (timing only)
#pragma unroll
      for (int v = 0; v < KS / 4; ++v) hq[v] = f32x4{h, h * 0.5f, -h, h * 0.25f};
    } else
    hq[0] = *reinterpret_cast<const f32x4*>(hp);
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 x = {sm[gx_off + p * kGxP], sm[gx_off + p * kGxP + kGxZ]};      // {x_r, x_z} or zeros (one ds_read2)
    __builtin_amdgcn_sched_barrier(0);
    if (!(EXP & 8)) {
#pragma unroll
    for (int v = 1; v < KS / 4; ++v) hq[v] = *reinterpret_cast<const f32x4*>(hp + 4 * v);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float gin2 = sm[kGn + p * OPE_H + f];
    __builtin_amdgcn_sched_barrier(0);   // all LDS reads issue before anything waits
    f32x2 a0, a1, an0, an1;              // {r, z} partials over even / odd k; n partials over the two k-pairs of a quad of k
    if (EXP & 16) {      // no mat-vec: the partials from one value each (timing only)
      a0 = f32x2{hq[0][0], hq[0][1]} + x; a1 = f32x2{hq[0][2], hq[0][3]}; an0 = f32x2{hq[KS / 4 - 1][0], hq[KS / 4 - 1][1]} + bn0; an1 = f32x2{hq[KS / 4 - 1][2], hq[KS / 4 - 1][3]};
    } else
#pragma unroll
    for (int v = 0; v < KS / 4; ++v) {
      const f32x2 lo = {hq[v][0], hq[v][1]}, hi = {hq[v][2], hq[v][3]};
      if (v == 0) {
        a0 = pk_fma_bcast<0>(wrz[0], lo, x);
        a1 = pk_mul_bcast<1>(wrz[1], lo);
        an0 = __builtin_elementwise_fma(wn[0], lo, bn0);
        an1 = wn[1] * hi;
      } else {
        a0 = pk_fma_bcast<0>(wrz[4 * v], lo, a0);
        a1 = pk_fma_bcast<1>(wrz[4 * v + 1], lo, a1);
        an0 = __builtin_elementwise_fma(wn[2 * v], lo, an0);
        an1 = __builtin_elementwise_fma(wn[2 * v + 1], hi, an1);
      }
      a0 = pk_fma_bcast<0>(wrz[4 * v + 2], hi, a0);
      a1 = pk_fma_bcast<1>(wrz[4 * v + 3], hi, a1);
    }
    if (DBG) asm volatile("" : "+v"(a0), "+v"(a1), "+v"(an0), "+v"(an1));
    OPE_PHASE(0)
    // quad reduction of {r, z}: every lane keeps the gate of its role and hands the other one to its neighbour
    const f32x2 t = a0 + a1;
    const float mine = even ? t[0] : t[1], theirs = even ? t[1] : t[0];
    float u = mine + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, theirs), kSwap, 0xF, 0xF, true));
    u = group_sum_from<W>(u);
    const float y = (EXP & 1) ? fmaf(u, 0.001f, 0.5f) : __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));   // r (even lanes) | z (odd lanes)
    if (!(EXP & 4)) sm[rz_off + 4 * p * kPl] = y;              // the saves leave as soon as they exist (r from one side, z from the other)
    const f32x2 tn = an0 + an1;
    float sn = tn[0] + tn[1];
    const float an = group_sum<W>(sn);         // -2 log2(e) (W_hn h + b_hn), all lanes
    if (!(EXP & 4)) sm[kSv + (4 * p + 3) * kPl + f] = an;
    const float n = (EXP & 1) ? fmaf(fmaf(y, an, gin2), 0.001f, 0.1f) : fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(y, an, gin2))), -1.0f);
    if (!(EXP & 4)) sm[kSv + (4 * p + 2) * kPl + val_off] = n;
    const float z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), kSwap, 0xF, 0xF, true));
    h = fmaf(z, h - n, n);                     // (1 - z) n + z h
    if (DBG) asm volatile("" : "+v"(h));
    OPE_PHASE(1)
    sm[kHs + (p ^ 1) * OPE_H + val_off] = h;
    if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OPE_PHASE(2)
    if (!(EXP & 2)) lds_barrier();
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
// EXP: timing-only variants as in gru_fwd4_kernel -- 2 no per-step barrier, 4 no publishes of the gate adjoints (the storer stores stale values), 8 the 12
// broadcast reads of the adjoints replaced by register values, 16 the 24 packed FMAs left out.
template <int W, bool DBG, bool PK, int EXP = 0>   // PK: as gru_fwd4_kernel -- the chain starts at the episode's last live step, min(len[j], T) - 1
__global__ void __launch_bounds__((W + 2) * 64) gru_bwd4_kernel(GruBwdArgs a) {
  constexpr int FPW = OPE_H / W;     // features per compute wave
  constexpr int IS = OPE_H / W;      // gate rows per gate per lane
  constexpr int NP = IS / 2;
  __shared__ __attribute__((aligned(16))) float sm[8 * OPE_H + 12 * OPE_H];
  float(*ds)[4][OPE_H] = reinterpret_cast<float(*)[4][OPE_H]>(sm);                  // [2][4][64] dr_pre, dz_pre, dgn (broadcast) + dn_pre
  float(*sav)[6][OPE_H] = reinterpret_cast<float(*)[6][OPE_H]>(sm + 8 * OPE_H);     // [2][6][64] the step's factors (loader): a_n, a_r, a_g, a_z, z, dh_out
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const int64_t NB = a.NB;
  __shared__ int rows_s[PK ? kLiveMaxT + 2 : 1];
  int Tl = a.T;                       // the chain walks t = Tl - 1 .. t_lo
  const int pk_ag = PK ? row / a.B : 0, pk_j = PK ? row - pk_ag * a.B : 0;
  if (PK) Tl = min(__builtin_amdgcn_readfirstlane(a.lp.len[pk_j]), a.T);
  auto build_rows = [&]() {           // (as in gru_fwd4_kernel)
    if (!PK) return;
    for (int t = threadIdx.x; t < Tl; t += (W + 2) * 64) rows_s[t] = (a.N * a.lp.cum[t] + pk_ag * a.lp.nn[t] + pk_j) * (4 * OPE_H);      // byte offsets, as in gru_fwd4_kernel
    __syncthreads();
  };
  const int nsteps = Tl - a.t_lo;
  const int nchunks = (nsteps + kAhead - 1) / kAhead;

  if (wave == W) {   // ---- loader
    build_rows();
    float pre[kAhead][6];
    auto row_at = [&](int i, int back) -> int { return PK ? rows_s[max(max(Tl - 1 - i, a.t_lo) - back, 0)] : 0; };      // row of step i's t (- back)
    auto load_step = [&](float (&d)[6], int i, int prow, int prow1) {      // prow / prow1 (PK): rows of steps t / t - 1, read from the table ahead of time
      const int t = max(Tl - 1 - i, a.t_lo);
      if (PK) {      // (uniform array base) + (32-bit byte offset of the row + lane): two vector adds for the six loads of a step
        const unsigned vo = (unsigned)prow + 4u * lane, vo1 = (unsigned)prow1 + 4u * lane;
        gload_async_s<0>(d[0], a.rg, vo);
        gload_async_s<0>(d[1], a.zg, vo);
        gload_async_s<0>(d[2], a.ng, vo);
        gload_async_s<0>(d[3], a.ghn, vo);
        gload_async_s<0>(d[4], a.dh_out, vo);
        gload_async_s<0>(d[5], a.h, vo1);      // (t = 0: row 0's own h, unused)
        return;
      }
      const int64_t o = ((int64_t)t * NB + row) * OPE_H + lane;
      gload_async(d[0], a.rg + o);
      gload_async(d[1], a.zg + o);
      gload_async(d[2], a.ng + o);
      gload_async(d[3], a.ghn + o);
      gload_async(d[4], a.dh_out + o);
      gload_async(d[5], a.h + (t > 0 ? o - NB * OPE_H : o));
    };
    // The factors of step i that do not depend on dh are formed here, off the compute waves' chain (every instruction of a
    // compute wave is on it): with dht = dh_t + dh_out_t,
    //   dn_pre = a_n dht, a_n = (1 - z)(1 - n^2);  dr_pre = a_r dht, a_r = a_n gh_n r (1 - r);  dgh_n = a_g dht, a_g = a_n r;
    //   dz_pre = a_z dht, a_z = (h_prev - n) z (1 - z).     (scalar arithmetic on purpose: see the forward loader)
    auto publish = [&](float (&d)[6], int i) {
      const int p = i & 1;
      const int t = Tl - 1 - i;
      const float r = d[0], z = d[1], n = d[2], gn = d[3];
      const float hp = t > 0 ? d[5] : 0.f;         // h_{-1} = 0
      const float omz = 1.0f - z;
      float an = omz * fmaf(-n, n, 1.0f);
      float ar = an * (gn * (r * (1.0f - r)));
      float ag = an * r;
      float az = (hp - n) * (z * omz);
      asm("" : "+v"(an), "+v"(ar), "+v"(ag), "+v"(az));
      sav[p][0][lane] = an;
      sav[p][1][lane] = ar;
      sav[p][2][lane] = ag;
      sav[p][3][lane] = az;
      sav[p][4][lane] = z;
      sav[p][5][lane] = i < nsteps ? d[4] : 0.f;   // nothing enters behind the last step: the carried value leaves as dh_carry
    };
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_step(pre[s], s, row_at(s, 0), row_at(s, 1));
    asm volatile("s_waitcnt vmcnt(42)" : "+v"(pre[0][0]), "+v"(pre[0][1]), "+v"(pre[0][2]), "+v"(pre[0][3]), "+v"(pre[0][4]), "+v"(pre[0][5])::"memory");
    publish(pre[0], 0);
    load_step(pre[0], kAhead, row_at(kAhead, 0), row_at(kAhead, 1));
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
      for (int s = 0; s < kAhead; ++s) {
        const int i = c * kAhead + s;
        if (i < nsteps) {
          float(&d)[6] = pre[(s + 1) % kAhead];   // holds step i+1; 7 x 6 younger loads behind it
          // the table entries of the step requested below, asked for before the wait
          const int prow = row_at(i + 1 + kAhead, 0), prow1 = row_at(i + 1 + kAhead, 1);
          if (PK) __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt vmcnt(42)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])::"memory");
          publish(d, i + 1);
          load_step(d, i + 1 + kAhead, prow, prow1);
          if (!(EXP & 2)) lds_barrier();
        }
      }
    }
    return;
  }
  if (wave == W + 1) {   // ---- storer: step i-1's adjoints leave during step i
    build_rows();
    lds_barrier();
    for (int i = 0; i <= nsteps; ++i) {
      if (i > 0) {
        const int p = (i - 1) & 1;
        const int t = Tl - i;
        const int64_t ro64 = PK ? (int64_t)((unsigned)__builtin_amdgcn_readfirstlane(rows_s[t]) >> 2) : ((int64_t)t * NB + row) * OPE_H;      // float offset of the row in a [rows][64] array
        float* gout = a.dgi + 3 * ro64 + lane;
        gout[0] = ds[p][0][lane];
        gout[OPE_H] = ds[p][1][lane];
        gout[2 * OPE_H] = ds[p][3][lane];
        a.dghn[ro64 + lane] = ds[p][2][lane];
      }
      if (i < nsteps && !(EXP & 2)) lds_barrier();
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
  build_rows();
  float dh = a.dh_in ? a.dh_in[(int64_t)row * OPE_H + k] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  // The factors of a step arrive one step ahead (the loader publishes step i+1's before barrier i).
  f32x2 fnr = {sav[0][0][k], sav[0][1][k]}, fgz = {sav[0][2][k], sav[0][3][k]};   // {a_n, a_r}, {a_g, a_z}
  float z = sav[0][4][k];
  long long ph[4] = {0, 0, 0, 0}, tprev = 0;   // DBG: cycles in (adjoints | publish | barrier | reads+FMA+reduce)
  if (DBG) tprev = (long long)__builtin_amdgcn_s_memtime();
  // dht = dh_t + dh_out_t is the loop-carried value: dht' = S + (dht z + dho') where S is the mat-vec sum of this step; the
  // bracket is formed while the LDS reads are in flight, so one add follows the quad reduction on the chain.
  float dht = dh + sav[0][5][k];
  auto step = [&](auto P) {
    constexpr int p = decltype(P)::value;
    const f32x2 d2 = {dht, dht};
    const f32x2 nr = fnr * d2, gz = fgz * d2;   // {dn_pre, dr_pre}, {dgh_n, dz_pre}: two packed multiplies
    if (!(EXP & 4)) {
    sm[(4 * p + 3) * OPE_H + k] = nr[0];        // four lanes of a quad write the same value to the same address
    sm[(4 * p + 0) * OPE_H + k] = nr[1];
    sm[(4 * p + 1) * OPE_H + k] = gz[1];
    }
    float dgn = gz[0];
    if (DBG) asm volatile("" : "+v"(dgn));
    OPE_PHASE(0)
    if (!(EXP & 4)) sm[(4 * p + 2) * OPE_H + k] = dgn;
    if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    OPE_PHASE(1)
    if (!(EXP & 2)) lds_barrier();
    OPE_PHASE(2)
    const float* base = &ds[p][0][IS * g];
    f32x4 rv[IS / 4], zv[IS / 4], nv[IS / 4];
#pragma unroll
    for (int v = 0; v < IS / 4; ++v) {
      if (EXP & 8) {      // no exchange (timing only)
        rv[v] = f32x4{nr[1], nr[1] * 0.5f, -nr[1], nr[1] * 0.25f}; zv[v] = f32x4{gz[1], gz[1] * 0.5f, -gz[1], gz[1] * 0.25f}; nv[v] = f32x4{dgn, dgn * 0.5f, -dgn, dgn * 0.25f};
        continue;
      }
      rv[v] = *reinterpret_cast<const f32x4*>(base + 4 * v);
      zv[v] = *reinterpret_cast<const f32x4*>(base + OPE_H + 4 * v);
      nv[v] = *reinterpret_cast<const f32x4*>(base + 2 * OPE_H + 4 * v);
    }
    const f32x2 fnr1 = {sav[p ^ 1][0][k], sav[p ^ 1][1][k]}, fgz1 = {sav[p ^ 1][2][k], sav[p ^ 1][3][k]};   // next step's factors
    const float z1 = sav[p ^ 1][4][k], dho1 = sav[p ^ 1][5][k];
    __builtin_amdgcn_sched_barrier(0);   // all reads in flight before the first FMA waits
    // three chains (one per gate), eight packed FMAs deep: interleaved they issue back to back, and two packed adds and one
    // add fold them (six four-deep chains cost nine more instructions per step to fold)
    f32x2 cr, cz, cn;
    if (EXP & 16) {      // no mat-vec (timing only)
      cr = f32x2{rv[0][0], rv[IS / 4 - 1][1]}; cz = f32x2{zv[0][2], zv[IS / 4 - 1][3]}; cn = f32x2{nv[0][0], nv[IS / 4 - 1][3]};
    } else
#pragma unroll
    for (int v = 0; v < IS / 4; ++v) {
      if (v == 0) {
        cr = wr[0] * f32x2{rv[0][0], rv[0][1]};
        cz = wz[0] * f32x2{zv[0][0], zv[0][1]};
        cn = wn[0] * f32x2{nv[0][0], nv[0][1]};
      } else {
        cr = __builtin_elementwise_fma(wr[2 * v], f32x2{rv[v][0], rv[v][1]}, cr);
        cz = __builtin_elementwise_fma(wz[2 * v], f32x2{zv[v][0], zv[v][1]}, cz);
        cn = __builtin_elementwise_fma(wn[2 * v], f32x2{nv[v][0], nv[v][1]}, cn);
      }
      cr = __builtin_elementwise_fma(wr[2 * v + 1], f32x2{rv[v][2], rv[v][3]}, cr);
      cz = __builtin_elementwise_fma(wz[2 * v + 1], f32x2{zv[v][2], zv[v][3]}, cz);
      cn = __builtin_elementwise_fma(wn[2 * v + 1], f32x2{nv[v][2], nv[v][3]}, cn);
    }
    const f32x2 ct = (cr + cz) + cn;
    float s1 = ct[0] + ct[1];
    asm("" : "+v"(s1));                  // keeps the SLP vectoriser from pairing this add with anything else
    dht = group_sum<W>(s1) + fmaf(dht, z, dho1);   // the loader publishes dho = 0 behind the last step: dht ends as dh_{t_lo - 1}
    fnr = fnr1; fgz = fgz1; z = z1;
    if (DBG) asm volatile("" : "+v"(dht));
    OPE_PHASE(3)
  };
  for (int i = 0; i < nsteps; i += 2) {
    step(std::integral_constant<int, 0>{});
    if (i + 1 < nsteps) step(std::integral_constant<int, 1>{});
  }
  if (a.dh_carry && g == 0) a.dh_carry[(int64_t)row * OPE_H + k] = dht;
  if (DBG && a.dbg && lane == 0) {
    long long* o = a.dbg + ((int64_t)blockIdx.x * W + wave) * 8;
    o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3]; o[4] = nsteps;
  }
}

}  // namespace

// Compute waves per row: four while at most two rows share a CU (512 rows), two beyond (fewer lanes repeat the gate arithmetic).
// The kernels are bound by the per-step issue / latency chain of a wave until the SIMDs fill up. Measured at 3s5z (512 / 256 rows
// forward / backward at batch 32, and the 16 / 4 episode shares of a multi-GPU job) and MMM2 (640 / 320 rows); eight waves per row
// (half the LDS reads and FMAs per lane, a ten-wave barrier) were slower at every size (forward 46 vs 41 us at batch 4, 70 vs 58 at 32).
static int waves_per_row(int64_t rows, int asked) {
  if (asked == 2 || asked == 4) return asked;
  if (g_scan_waves) return g_scan_waves;
  return rows <= 512 ? 4 : 2;
}

static int scan_device_cus() {
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  return cus;
}

template <int W>
static void launch_fwd(const GruFwdArgs& a, hipStream_t st) {
#ifdef OPE_EXPERIMENTS
  static const int gexp = getenv("OPE_GRU_EXP") ? atoi(getenv("OPE_GRU_EXP")) : 0;
  if (gexp && !a.lp.hdr && !a.dbg && W == 4) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "libope: OPE_GRU_EXP=%d -- a timing-only variant of gru_fwd4_kernel runs: its outputs are WRONG\n", gexp); warned = true; }
    const int nets = (gexp & 32) ? 1 : a.nets;      // 32: the live net's rows only (one workgroup per CU at 256 rows)
#define OPE_GRU_CASE(E) case E: OPE_LAUNCH((gru_fwd4_kernel<4, false, false, E>), dim3(nets * a.NB), dim3((4 + 2) * 64), 0, st, a); return;
    switch (gexp & 31) {
      OPE_GRU_CASE(0) OPE_GRU_CASE(1) OPE_GRU_CASE(2) OPE_GRU_CASE(4) OPE_GRU_CASE(8) OPE_GRU_CASE(16) OPE_GRU_CASE(3) OPE_GRU_CASE(10) OPE_GRU_CASE(17) OPE_GRU_CASE(31)
      default: break;
    }
#undef OPE_GRU_CASE
  }
#endif
  if (a.lp.hdr) {
    static const int pair_env = getenv("OPE_GRU_PAIR") ? atoi(getenv("OPE_GRU_PAIR")) : 1;      // 0: the plain block -> (net, row) mapping (A/B runs)
    GruFwdArgs b = a;
    b.pair_cus = pair_env ? scan_device_cus() : 0;
    OPE_LAUNCH((gru_fwd4_kernel<W, false, true>), dim3(a.nets * a.NB), dim3((W + 2) * 64), 0, st, b);
  } else if (a.dbg)
    OPE_LAUNCH((gru_fwd4_kernel<W, true, false>), dim3(a.nets * a.NB), dim3((W + 2) * 64), 0, st, a);
  else
    OPE_LAUNCH((gru_fwd4_kernel<W, false, false>), dim3(a.nets * a.NB), dim3((W + 2) * 64), 0, st, a);
}
template <int W>
static void launch_bwd(const GruBwdArgs& a, hipStream_t st);
// Eight compute waves per row, BACKWARD scan only, "by shape" only (no pinned wave count) and only while every row has a CU to itself: half
// the broadcast reads (6 instead of 12 ds_read_b128) and packed FMAs (12 instead of 24) per wave, a ten-wave barrier. Round 6, 3s5z, B = 32 (256
// rows): 45.5 -> 43.8 us (two same-box pairs). The FORWARD scan with eight waves is slower (57.6 vs 48.8 us at two rows per CU: the gate
// arithmetic is repeated by twice the lanes and twenty waves share a CU), as round 3 had found. OPE_GRUB_W = 4 keeps four.
static bool launch_bwd8(const GruBwdArgs& a, hipStream_t st) {
  static const int w8 = getenv("OPE_GRUB_W") ? atoi(getenv("OPE_GRUB_W")) : 8;
  if (w8 != 8 || a.dbg || a.waves || g_scan_waves || a.NB > scan_device_cus()) return false;
  if (a.lp.hdr) OPE_LAUNCH((gru_bwd4_kernel<8, false, true>), dim3(a.NB), dim3((8 + 2) * 64), 0, st, a);
  else OPE_LAUNCH((gru_bwd4_kernel<8, false, false>), dim3(a.NB), dim3((8 + 2) * 64), 0, st, a);
  return true;
}
template <int W>
static void launch_bwd(const GruBwdArgs& a, hipStream_t st) {
#ifdef OPE_EXPERIMENTS
  static const int gexp = getenv("OPE_GRUB_EXP") ? atoi(getenv("OPE_GRUB_EXP")) : 0;
  if (gexp && !a.lp.hdr && !a.dbg && W == 4) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "libope: OPE_GRUB_EXP=%d -- a timing-only variant of gru_bwd4_kernel runs: its outputs are WRONG\n", gexp); warned = true; }
#define OPE_GRU_CASE(E) case E: OPE_LAUNCH((gru_bwd4_kernel<4, false, false, E>), dim3(a.NB), dim3((4 + 2) * 64), 0, st, a); return;
    switch (gexp) {
      OPE_GRU_CASE(2) OPE_GRU_CASE(4) OPE_GRU_CASE(8) OPE_GRU_CASE(16) OPE_GRU_CASE(24) OPE_GRU_CASE(30)
      default: break;
    }
#undef OPE_GRU_CASE
  }
#endif
  if (a.lp.hdr)
    OPE_LAUNCH((gru_bwd4_kernel<W, false, true>), dim3(a.NB), dim3((W + 2) * 64), 0, st, a);
  else if (a.dbg)
    OPE_LAUNCH((gru_bwd4_kernel<W, true, false>), dim3(a.NB), dim3((W + 2) * 64), 0, st, a);
  else
    OPE_LAUNCH((gru_bwd4_kernel<W, false, false>), dim3(a.NB), dim3((W + 2) * 64), 0, st, a);
}

int launch_gru_fwd4(const GruFwdArgs& a, hipStream_t st) {
  const int w = waves_per_row((int64_t)a.nets * a.NB, a.waves);
  kprof_work(2.0 * a.nets * a.NB * (double)a.L * 3.0 * OPE_H * OPE_H);      // W_hh h per row and step
  if (a.lp.hdr && (a.dbg || a.hinit || a.hinit1 || a.B < 1 || a.N < 1 || a.NB != a.N * a.B || a.L > kLiveMaxT + 1)) return OPE_EINVAL;
  if (a.lp.hdr) kprof_rows(1);
  if (w == 4) launch_fwd<4>(a, st); else launch_fwd<2>(a, st);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(a.lp.hdr ? "gru_fwd4_live" : "gru_fwd4", w == 4 ? 4 : 2);
  return OPE_OK;
}

int launch_gru_bwd4(const GruBwdArgs& a, hipStream_t st) {
  const int w = waves_per_row(a.NB, a.waves);
  kprof_work(2.0 * a.NB * (double)(a.T - a.t_lo) * 3.0 * OPE_H * OPE_H);     // W_hh^T (gate adjoints) per row and step
  if (a.lp.hdr && (a.dbg || a.dh_in || a.dh_carry || a.t_lo != 0 || a.B < 1 || a.N < 1 || a.NB != a.N * a.B || a.T > kLiveMaxT)) return OPE_EINVAL;
  if (a.lp.hdr) kprof_rows(2);
  const bool w8 = w == 4 && launch_bwd8(a, st);
  if (w8) { /* launched */ }
  else if (w == 4) launch_bwd<4>(a, st); else launch_bwd<2>(a, st);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(a.lp.hdr ? "gru_bwd4_live" : "gru_bwd4", w8 ? 8 : (w == 4 ? 4 : 2));
  return OPE_OK;
}

}  // namespace ope
