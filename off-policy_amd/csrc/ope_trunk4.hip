// trunk_fwd4: the agent network's row-parallel trunk (feature LayerNorm -> fc1 -> ReLU -> LN -> fc2 -> ReLU -> LN -> W_ih) of the LIVE and the
// TARGET net in ONE launch, weights resident in LDS, one wave per 16-row tile.
//   MLPBase.forward / MLPLayer.forward   offpolicy/algorithms/utils/mlp.py:25-29, 52-89
//   RNNLayer's input projection (nn.GRU weight_ih_l0 / bias_ih_l0)   offpolicy/algorithms/utils/rnn.py:19-23
//   called once per net by QMix.train_policy_on_batch   offpolicy/algorithms/qmix/qmix.py:127-148
//
// Why another trunk kernel. trunk_fwd3 (ope_trunk2.hip) keeps a net's weights in REGISTERS (128 VGPRs of fragments at D = 252), four waves
// share a 16-row tile (16 output features each) and meet through LDS behind 4-5 workgroup barriers per tile; with 254 registers only two such
// workgroups fit a CU, and a tile spends 4.1 k of its 14.4 k cycles issuing MFMAs: 41 + 38 us for the two nets at
// 3s5z, 39-42 % of the f32 matrix pipe, and the observation rows are read from HBM twice.
// Here a CU holds ONE net's weights in LDS (fc1 64 x D, fc2, W_ih: 128 KB at D <= 256), staged once per launch, even CUs the live net, odd CUs
// the target; a WAVE owns whole 16-row tiles and walks them through all three layers alone:
//   * no barrier after the prologue. A layer's output fragment (lane (j, g): features 16 it + 4 g .. + 3 of row j) IS the next layer's MFMA
//     B operand (the "transposed chain" convention, ope_common.h), LayerNorm statistics are 16 local values + two lane swaps: nothing of a
//     tile ever leaves the wave's registers except the saves;
//   * weights are the MFMA A operand, read from LDS as needed: one ds_read_b128 per four MFMAs (160 reads, 640 LDS-array cycles, per 512 MFMAs
//     = 16.4 k matrix-pipe cycles of a tile), conflict-free by an XOR swizzle of the 16-byte column slot with 2 (row & 7) (the ds_read_b128
//     lane groups hold rows {0-3, 12-15} at slot P and {4-11} at P ^ 1: the swizzle spreads each set over the 8 even residues);
//   * latency is hidden by the OTHER wave of the SIMD (two per SIMD, each in its own tile, in whatever phase it happens to be) and by
//     requesting the next tile's observation rows right after fc1 has consumed this tile's: they land behind fc2 / W_ih, in the registers
//     fc1 just freed;
//   * the two waves of a SIMD draw their tiles from one LDS counter, so a SIMD's share of the 2 416 tiles per net (4.72) is what balances,
//     not a wave's.
// The observation rows are still read once per net (the nets sit on different CUs), but within one launch and largely out of L2.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ope_agent.h"

namespace ope {

namespace {

__device__ __forceinline__ int wslot(int row, int p) { return p ^ ((row & 7) << 1); }   // swizzled 16-byte slot of logical slot p in `row`

template <int EXP>
__device__ __forceinline__ f32x4 t4_mfma(float w, float x, f32x4 acc) {
  if (EXP & 1) {
    asm volatile("" : "+v"(acc) : "v"(w), "v"(x));      // the operands stay live (the LDS reads are not dead code), the matrix pipe stays idle
    return acc;
  }
  return mfma16(w, x, acc);
}

template <int KCM>
struct T4 {
  static constexpr int RS1 = 4 * KCM;                 // 16-byte slots per fc1 row
  static constexpr int W1F = OPE_H * 16 * KCM;        // floats
  static constexpr int W2F = OPE_H * OPE_H;
  // W_ih rows resident in LDS: all 192, except at the widest input (KCM = 24: fc1 alone is 96 KB) where the last two 16-gate tiles
  // (8 KB) stay in L2 and are read as MFMA operands straight from there -- 8 loads per lane and tile, requested a phase ahead
  static constexpr int W3R = KCM > 16 ? 160 : 3 * OPE_H;
  static constexpr int W3F = W3R * OPE_H;
  static constexpr int FNP = 2 * 16 * KCM;            // input LayerNorm gamma / beta, zero beyond D
  static constexpr int LNP = 6 * OPE_H;               // b1, ln1 w, ln1 b, b2, ln2 w, ln2 b
  static constexpr int BIH = 3 * OPE_H;
  static constexpr int TOTAL = W1F + W2F + W3F + FNP + LNP + BIH + 16;
  static constexpr int TOTAL_REF = TOTAL + kObsRefMaxB;        // + the sampled episode slots (rows read from the store)
};

}  // namespace

// lean argument block (two full TrunkFwdArgs cost ~170 spilled SGPRs): what differs between the nets is the parameter vector and the gi
// destination; the saves belong to net 0
struct TrunkPairArgs {
  const float* x; int R, D; int nets;
  const float* theta[2];
  float* gi[2];
  int fn_w, fn_b, fc1_w, fc1_b, ln1_w, ln1_b, fc2_w, fc2_b, ln2_w, ln2_b, wih, bih;
  float* mu0; float* rstd0; float* xhat1; float* rstd1; float* mu1; uint64_t* mask1; float* xhat2; float* rstd2; uint64_t* mask2;
  long long* dbg;
  int no_fn;                      // no input LayerNorm (OPE_DIMS_NO_FEATURE_NORM)
  int save0;                      // net 0 writes the saves (a single-net launch of a target / rollout net does not)
  ObsRef ref; int ref_tn0;        // LAZY instantiation: x = the store's obs ring; first (t, agent) index of the launch's row range
  const int* live_hdr; const int* live_src;   // PK instantiation: the live plan's header (hdr[0] = rows of this launch) and packed row -> batch row
};

// KCM = ceil(D / 16) exactly and D % 4 == 0: every 16-column chunk but the last is complete, and a 16-byte piece of the last one is inside
// the row or entirely past it -- one per-lane predicate instead of 4 KCM hoisted column masks (which cost ~130 spilled SGPRs + 46 VGPRs)
// NW = waves per workgroup (8 or 12: two or three per SIMD), PF = request the next tile's rows behind fc1 (keeps the 4 KCM row
// registers live through the whole tile: 248 VGPRs, two waves per SIMD) or at the top of the tile (<= 168 VGPRs, three per SIMD).
// LAZY: the observation rows are read in place from the episode-major store (ObsRef, ope_common.h) instead of a gathered batch: the same
// 16 KCM-byte rows, at (per-lane 64-bit row pointer) + (immediate) instead of (uniform base) + (32-bit offset).
// VEC = 4: D % 4 == 0, rows and weight rows read as 16-byte pieces; VEC = 2 (D % 2 == 0: MMM2's 370): as two 8-byte halves.
// EXP (instantiated only in builds with -DOPE_EXPERIMENTS; TIMING variants, results WRONG; OPE_T4_EXP, profiles/r05_trunk4_decomposition.txt):
// 1 no MFMAs (their LDS weight reads stay), 2 no saves for the backward pass, 4 no weight staging, 8 no gi stores, 16 no ReLU / LayerNorm
// between the layers, 32 no observation-row loads
// PK: the packed rows of the live plan (LivePlan, ope_common.h). The row count comes from the plan's header on the device; packed row p reads
// batch row live_src[p] (the batch keeps the reference's padded layout) and writes every output at p. A tile's 16 source indices are
// requested one tile ahead -- at the top of the tile before, where the next tile is now drawn -- so they have landed when the rows
// themselves are requested behind fc1.
template <int KCM, int NW, bool PF, bool LAZY, int VEC = 4, int EXP = 0, bool PK = false>
__global__ void __launch_bounds__(64 * NW, NW / 4) trunk_fwd4_kernel(TrunkPairArgs pa) {
  using C = T4<KCM>;
  static_assert(!(LAZY && VEC != 4), "rows are read in place from the store as 16-byte pieces only");
  static_assert(!(PK && (LAZY || !PF)), "packed rows: gathered batch, rows prefetched behind fc1");
  constexpr int NT = 64 * NW;
  __shared__ __attribute__((aligned(16))) float sm[LAZY ? C::TOTAL_REF : C::TOTAL];
  float* const W1s = sm;
  float* const W2s = W1s + C::W1F;
  float* const W3s = W2s + C::W2F;
  float* const fnp = W3s + C::W3F;
  float* const lnp = fnp + C::FNP;
  float* const bih = lnp + C::LNP;
  int* const ctr = reinterpret_cast<int*>(bih + C::BIH);
  int* const eps = ctr + 16;      // LAZY: episode slot of batch column b

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  // Two nets: workgroups b and b + 8 form a pair -- the same tiles (same wgn), one net each, and the SAME XCD (a workgroup lands on XCD
  // b % 8), so the observation rows one of them has fetched are in that XCD's L2 when the other asks for them (they walk their tiles
  // at about the same pace). (With net = b & 1 the pair sat on neighbouring XCDs and every row crossed the fabric twice.)
  const bool paired = pa.nets == 2 && (gridDim.x & 15) == 0;
  const int net = pa.nets == 2 ? (paired ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 1)) : 0;
  const int wgn = pa.nets == 2 ? (paired ? (int)((blockIdx.x & 7) | ((blockIdx.x >> 4) << 3)) : (int)(blockIdx.x >> 1)) : (int)blockIdx.x;
  const int nwg = pa.nets == 2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
  const TrunkPairArgs& a = pa;
  const float* __restrict__ th = net ? pa.theta[1] : pa.theta[0];
  float* __restrict__ gi_out = net ? pa.gi[1] : pa.gi[0];
  const int D = a.D, R = PK ? __builtin_amdgcn_readfirstlane(pa.live_hdr[0]) : a.R;
  const bool save = net == 0 && pa.save0 && !(EXP & 2);

  // optional s_memtime stamps (ope_qmix_cfg.debug; tools/trunk4_phases.py): [workgroup][wave][16] = start, weights staged, then for the
  // wave's FIRST tile: rows arrived + statistics, fc1, LN1 + saves, fc2 + LN2 + saves, W_ih + gi stores; last: all tiles done, tiles done
  long long* dbg = a.dbg ? a.dbg + ((int64_t)blockIdx.x * 12 + wave) * 16 : nullptr;
  auto stamp = [&](int k) { if (dbg && lane == 0) dbg[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  // ---- tiles: the two waves of a SIMD (waves w and w + 4) draw from counter w & 3 ----
  // LAZY: a tile = 16 consecutive (t, agent) rows of ONE sampled episode b (contiguous in the store: one 16 x 4 D-byte run, one page)
  // instead of 16 consecutive batch rows (16 episodes); tile = b * tpe + q covers tn = 16 q .. 16 q + 15 of the launch's TNc = R / B
  const int TNc = LAZY ? R / a.ref.B : 0, tpe = LAZY ? (TNc + 15) >> 4 : 1;
  const int ntiles = LAZY ? a.ref.B * tpe : (R + 15) >> 4;
  auto rows_of = [&](int tile, int& row, bool& valid, int& b, int& tnj) {
    if (LAZY) {
      b = tile / tpe;
      tnj = 16 * (tile - b * tpe) + j;
      valid = tnj < TNc;
      tnj = valid ? tnj : TNc - 1;
      row = tnj * a.ref.B + b;
    } else {
      row = tile * 16 + j;
      valid = row < R;
      b = 0; tnj = 0;
    }
  };
  const int nslots = 4 * nwg, slot0 = 4 * wgn + (wave & 3);
  auto grab = [&]() -> int {
    int k = 0;
    if (lane == 0) k = __hip_atomic_fetch_add(&ctr[wave & 3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    k = __builtin_amdgcn_readfirstlane(k);
    const int64_t t = (int64_t)slot0 + (int64_t)nslots * k;
    return t < ntiles ? (int)t : ntiles;
  };
  const float inv_d = 1.0f / (float)D;
  f32x4 xv[KCM];
  const bool tail_ok = 16 * (KCM - 1) + 4 * g < D;       // this lane's piece of the last chunk lies inside the row
  const bool tail_ok1 = VEC == 4 ? tail_ok : 16 * (KCM - 1) + 4 * g + 2 < D;      // ... its second 8-byte half (VEC = 2)
  // Every global access of the tile loop is (uniform base pointer) + (32-bit byte offset): all arrays are < 4 GB (R * 768 B the
  // largest), and 64-bit per-lane addresses cost twice the registers -- the first version spilled five of them, and a scratch
  // reload is a VMEM load: its vmcnt wait also waits for the 16 row loads just requested for the NEXT tile (13 k cycles per tile).
  const char* __restrict__ xb = reinterpret_cast<const char*>(a.x);
  // part (KCM > 16 only): 0 = the whole row, 1 = chunks [0, 16) -- what is requested behind fc1 for the NEXT tile and held through fc2 /
  // W_ih (a whole 24-chunk row would be 96 registers: 89 spilled) --, 2 = the rest, requested at the top of the tile itself
  constexpr int PFC = KCM > 16 ? 16 : KCM;
  if (EXP & 32) {
#pragma unroll
    for (int c = 0; c < KCM; ++c) xv[c] = f32x4{0.01f * (float)(lane + c), -0.02f * (float)(j + 3 * c), 0.5f - 0.03f * (float)g, 0.25f * (float)(c & 3)};
  }
  auto request = [&](int tile, bool staged = true, int part = 0, int src = 0) {
    if (EXP & 32) return;
    const int row = tile * 16 + j;
    if (LAZY) {
      int rw, b, tnj;
      bool ok;
      rows_of(tile, rw, ok, b, tnj);
      const int ep = staged ? eps[b] : obs_ref_slot(a.ref, a.ref.inds[b]);       // (the first tile is requested before the slots are staged)
      const char* rp = xb + (((int64_t)ep * a.ref.TTN + (a.ref_tn0 + tnj)) * (int64_t)(4 * D) + 16 * g);
      int gg = g;
      asm volatile("" : "+v"(gg));
      xv[KCM - 1] = *reinterpret_cast<const f32x4*>(rp + (16 * (KCM - 1) + 4 * gg < D ? 64 * (KCM - 1) : 0));
#pragma unroll
      for (int c = 0; c < KCM - 1; ++c) xv[c] = *reinterpret_cast<const f32x4*>(rp + 64 * c);
      return;
    }
    const uint32_t xo = (uint32_t)(PK ? src : (row < R ? row : R - 1)) * (uint32_t)(4 * D) + 16u * g;
    if (VEC == 2) {
      // two 8-byte halves per piece; the last chunk's halves are clamped to offset 0 of the row when they lie past its end (zeroed below)
      int gg = g;
      asm volatile("" : "+v"(gg));
      const int k0 = 16 * (KCM - 1) + 4 * gg;
      if (part != 1) {
        const f32x2 t0 = *reinterpret_cast<const f32x2*>(xb + (xo + (k0 < D ? 64u * (KCM - 1) : 0u)));
        const f32x2 t1 = *reinterpret_cast<const f32x2*>(xb + (xo + (k0 + 2 < D ? 64u * (KCM - 1) + 8u : 0u)));
        xv[KCM - 1] = f32x4{t0[0], t0[1], t1[0], t1[1]};
      }
#pragma unroll
      for (int c = 0; c < KCM - 1; ++c) {
        if ((c < PFC && part == 2) || (c >= PFC && part == 1)) continue;
        const f32x2 h0 = *reinterpret_cast<const f32x2*>(xb + (xo + 64u * c)), h1 = *reinterpret_cast<const f32x2*>(xb + (xo + 64u * c + 8u));
        xv[c] = f32x4{h0[0], h0[1], h1[0], h1[1]};
      }
      return;
    }
    // the last chunk's piece first, its offset recomputed here (a hoisted copy got spilled, and the reload's vmcnt wait sat in the
    // middle of this burst of loads)
    int gg = g;
    asm volatile("" : "+v"(gg));
    xv[KCM - 1] = *reinterpret_cast<const f32x4*>(xb + (xo + (16 * (KCM - 1) + 4 * gg < D ? 64u * (KCM - 1) : 0u)));   // (zeroed below when outside)
#pragma unroll
    for (int c = 0; c < KCM - 1; ++c) xv[c] = *reinterpret_cast<const f32x4*>(xb + (xo + 64u * c));
  };
  // every wave's FIRST tile is assigned statically (slot, round = wave / 4), so its rows can be requested before the weights are
  // staged: the HBM latency of the first tile hides behind the 10 k-cycle prologue
  int tile;
  {
    const int64_t t = (int64_t)slot0 + (int64_t)nslots * (wave >> 2);
    tile = t < ntiles ? (int)t : ntiles;
  }
  int src_cur = 0, src_next = 0;      // PK: batch rows of this lane's row of the current / the next tile
  if (PK) src_cur = pa.live_src[min(tile * 16 + j, R - 1)];
  if (PF && tile < ntiles) request(tile, false, KCM > PFC ? 1 : 0, src_cur);
  // ---- prologue: this net's weights -> LDS (swizzled), parameters, tile counters. Every thread requests ALL its pieces before the
  // first LDS store (8 KCM / 16 + 2 + 6 independent 16-byte loads in flight per thread instead of one round trip per piece) ----
  if (!(EXP & 4)) {
    constexpr int P1 = OPE_H * C::RS1, P2 = OPE_H * 16, P3 = C::W3R * 16;                         // 16-byte pieces of the three matrices (their LDS-resident rows)
    constexpr int N1 = (P1 + NT - 1) / NT, N2 = (P2 + NT - 1) / NT, N3 = (P3 + NT - 1) / NT;       // per thread (a partly used last round)
    f32x4 p1[N1], p2[N2], p3[N3];
#pragma unroll
    for (int u = 0; u < N1; ++u) {
      const int i = min(tid + NT * u, P1 - 1), row = i / C::RS1, p = i - row * C::RS1;
      p1[u] = load4c<VEC>(th + a.fc1_w + (int64_t)row * D, 4 * p < D ? 4 * p : 0, D);
    }
#pragma unroll
    for (int u = 0; u < N2; ++u) p2[u] = *reinterpret_cast<const f32x4*>(th + a.fc2_w + 4 * min(tid + NT * u, P2 - 1));
#pragma unroll
    for (int u = 0; u < N3; ++u) p3[u] = *reinterpret_cast<const f32x4*>(th + a.wih + 4 * min(tid + NT * u, P3 - 1));
#pragma unroll
    for (int u = 0; u < N1; ++u) {
      const int i = tid + NT * u, row = i / C::RS1, p = i - row * C::RS1;
      f32x4 v = mask4(p1[u], 4 * p, D);
      if (4 * p >= D) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < P1) *reinterpret_cast<f32x4*>(W1s + row * (4 * C::RS1) + 4 * wslot(row, p)) = v;
    }
#pragma unroll
    for (int u = 0; u < N2; ++u) {
      const int i = tid + NT * u, row = i >> 4, p = i & 15;
      if (i < P2) *reinterpret_cast<f32x4*>(W2s + row * OPE_H + 4 * wslot(row, p)) = p2[u];
    }
#pragma unroll
    for (int u = 0; u < N3; ++u) {
      const int i = tid + NT * u, row = i >> 4, p = i & 15;
      if (i < P3) *reinterpret_cast<f32x4*>(W3s + row * OPE_H + 4 * wslot(row, p)) = p3[u];
    }
  }
  for (int f = tid; f < 16 * KCM; f += NT) {
    fnp[f] = f < D ? th[a.fn_w + f] : 0.f;
    fnp[16 * KCM + f] = f < D ? th[a.fn_b + f] : 0.f;
  }
  if (tid < OPE_H) {
    lnp[tid] = th[a.fc1_b + tid]; lnp[OPE_H + tid] = th[a.ln1_w + tid]; lnp[2 * OPE_H + tid] = th[a.ln1_b + tid];
    lnp[3 * OPE_H + tid] = th[a.fc2_b + tid]; lnp[4 * OPE_H + tid] = th[a.ln2_w + tid]; lnp[5 * OPE_H + tid] = th[a.ln2_b + tid];
  }
  if (tid < 3 * OPE_H) bih[tid] = th[a.bih + tid];
  if (tid < 4) ctr[tid] = NW / 4;        // every wave's first tile is assigned statically (below): the counters start behind them
  if (LAZY)
    for (int i = tid; i < a.ref.B; i += NT) eps[i] = obs_ref_slot(a.ref, a.ref.inds[i]);

  __syncthreads();                       // weights, parameters and counters are in place

  // Weight-fragment addresses. wslot(j, 4 c + g) = 4 (c ^ m) + (g ^ 2 (j & 1)), m = (j >> 1) & 3: only c's two low bits meet the lane, so
  // FOUR per-lane byte offsets (c & 3 = 0 .. 3) + compile-time immediates address every fragment of the three matrices (rows 16 it + j
  // and chunks 4 q + cl are immediates of up to 49 920 B). (Left to hipcc: one address register per chunk, 26 in all -- with three
  // waves per SIMD that alone was 15 % of the budget.)
  uint32_t wo1[4], wo2[4];
  {
    const int m = (j >> 1) & 3, gl = g ^ (2 * (j & 1));
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      wo1[cl] = (uint32_t)j * (64u * KCM) + 64u * (cl ^ m) + 16u * gl;       // fc1 rows: 16 KCM floats
      wo2[cl] = (uint32_t)j * (4u * OPE_H) + 64u * (cl ^ m) + 16u * gl;      // fc2 / W_ih rows: 64 floats
    }
  }
  const char* const W1b = reinterpret_cast<const char*>(W1s);
  const char* const W2b = reinterpret_cast<const char*>(W2s);
  const char* const W3b = reinterpret_cast<const char*>(W3s);
  auto lnp4 = [&](int i, int it) { return *reinterpret_cast<const f32x4*>(lnp + i * OPE_H + 16 * it + 4 * g); };
  // ReLU + LayerNorm over the 64 features of row j held as z[it][r] = feature 16 it + 4 g + r: one-pass statistics (sum, sum of
  // squares of the post-ReLU values; E[x^2] - mean^2 at O(1) magnitudes, as trunk_fwd3), the row's four lanes meet by two lane swaps
  auto relu_ln = [&](f32x4 (&z)[4], int gi_, int bi_, char* xhat_base, uint32_t xhat_off, float& rs, float& mu, uint64_t& bits) {
    // ReLU mask, bit f = z[f] > 0 with f = 16 it + 4 g + r: four lane-independent bits per output tile, one lane-dependent shift by 4 g
    // per 32-bit half (NOT sixteen hoisted per-lane 64-bit constants 1 << f: 32 registers, two of them spilled)
    uint32_t nib[4];
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      nib[it] = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        nib[it] |= (z[it][r] > 0.f ? 1u : 0u) << r;
        z[it][r] = fmaxf(z[it][r], 0.f);
        s += z[it][r];
        s2 = fmaf(z[it][r], z[it][r], s2);
      }
    }
    const uint64_t mb = ((uint64_t)((nib[2] | (nib[3] << 16)) << (4 * g)) << 32) | (uint64_t)((nib[0] | (nib[1] << 16)) << (4 * g));
    s = rowsum4(s);
    s2 = rowsum4(s2);
    mu = s * (1.0f / OPE_H);
    const float var = fmaxf(s2 * (1.0f / OPE_H) - mu * mu, 0.f);
    rs = __builtin_amdgcn_rsqf(var + OPE_LN_EPS);      // v_rsq_f32 (1 ulp); the IEEE 1 / sqrtf sequence is ~25 dependent instructions
    bits = mb;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 gm = lnp4(gi_, it), bt = lnp4(bi_, it);
      f32x4 xh;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[r] = (z[it][r] - mu) * rs;
        z[it][r] = fmaf(xh[r], gm[r], bt[r]);          // z becomes the layer's output = the next layer's B operand
      }
      if (xhat_base) *reinterpret_cast<f32x4*>(xhat_base + (xhat_off + 64u * it)) = xh;   // saved for the backward pass (live net, valid rows)
    }
  };
  auto or4 = [&](uint64_t b) -> uint64_t {             // OR over the row's four lanes (j, 0..3)
    uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    lo |= __shfl_xor((int)lo, 16, 64); hi |= __shfl_xor((int)hi, 16, 64);
    lo |= __shfl_xor((int)lo, 32, 64); hi |= __shfl_xor((int)hi, 32, 64);
    return ((uint64_t)hi << 32) | lo;
  };

  stamp(1);
  int done = 0;
  while (tile < ntiles) {
    // LDS is read-only after the prologue, so LLVM hoists every parameter fragment read (biases, LayerNorm gamma / beta, b_ih: ~270
    // registers' worth) out of the tile loop as loop invariants and then spills them; the clobber makes them per-tile reads again
    asm volatile("" ::: "memory");
    int row, b_, tnj_;
    bool valid;
    rows_of(tile, row, valid, b_, tnj_);
    if (!PF) request(tile);
    if (PF && KCM > PFC) request(tile, true, 2, src_cur);       // (unconditional: a "not for the first tile" test keeps the 32 registers live around the loop)
    int next = 0;
    if (PK) {      // the next tile is drawn a phase early: its source rows' indices travel while fc1 runs
      next = grab();
      src_next = pa.live_src[min(next * 16 + j, R - 1)];
    }
    // ---- input LayerNorm statistics of row j: 4 lanes x KCM pieces ----
    float s = 0.f;
    if (!tail_ok) xv[KCM - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (VEC == 2 && !tail_ok1) { xv[KCM - 1][2] = 0.f; xv[KCM - 1][3] = 0.f; }
#pragma unroll
    for (int c = 0; c < KCM; ++c) s += (xv[c][0] + xv[c][1]) + (xv[c][2] + xv[c][3]);
    const float mean = a.no_fn ? 0.f : rowsum4(s) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < KCM; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (c < KCM - 1 || (r < 2 ? tail_ok : tail_ok1)) ? xv[c][r] - mean : 0.f;
        xv[c][r] = d;
        sq = fmaf(d, d, sq);
      }
    }
    const float rstd = a.no_fn ? 1.0f : __builtin_amdgcn_rsqf(fmaf(rowsum4(sq), inv_d, OPE_LN_EPS));
    if (save && valid && g == 0) {
      a.mu0[row] = mean;
      a.rstd0[row] = rstd;
    }
    // the normalised, affine-transformed row, in place (exactly 0 beyond D): the fc1 loop below is LDS reads and MFMAs only
#pragma unroll
    for (int c = 0; c < KCM; ++c) {
      if ((c & 3) == 0) __builtin_amdgcn_sched_barrier(0);     // (at most 8 parameter reads in flight: all 2 KCM of them are 128 registers)
      const f32x4 gm = *reinterpret_cast<const f32x4*>(fnp + 16 * c + 4 * g), bt = *reinterpret_cast<const f32x4*>(fnp + 16 * KCM + 16 * c + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[c][r] = fmaf(xv[c][r] * rstd, gm[r], bt[r]);
      // KCM = 24: LLVM sinks these products towards their uses in the fc1 loop (across the scheduling fences, which only bind the machine
      // scheduler), so all 48 parameter reads -- 192 registers -- end up in flight next to the 96-register row: 89 spills. An empty asm
      // that "modifies" each product pins it here.
      if (KCM > 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(xv[c][r]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (done == 0) stamp(2);
    // ---- fc1: 4 output tiles x KCM chunks ----
    f32x4 z[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) z[it] = lnp4(0, it);
#pragma unroll
    for (int c = 0; c < KCM; ++c) {
      // keep the weight reads at most two chunks ahead of their MFMAs (hipcc otherwise hoists all 4 KCM of them: 256 registers, spills)
      if (PF ? (c & 1) == 0 : true) {
        // the three-wave variant must not read more than one chunk of weights ahead (168 registers): a compiler-level memory
        // barrier as well, LLVM otherwise clusters all 64 LDS reads in front of the 256 MFMAs and spills them
        if (!PF) asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 wv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) wv[it] = *reinterpret_cast<const f32x4*>(W1b + (wo1[c & 3] + (uint32_t)(16 * it * 64 * KCM + 256 * (c >> 2))));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int it = 0; it < 4; ++it) z[it] = t4_mfma<EXP>(wv[it][r], xv[c][r], z[it]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (done == 0) stamp(3);
    // the observation registers are free: request the next tile's rows now, they arrive behind fc2 / W_ih
    if (!PK) next = grab();
    if (PF && next < ntiles) request(next, true, KCM > PFC ? 1 : 0, src_next);

    float rs, mu;
    uint64_t bits;
    const uint32_t xh_off = (uint32_t)row * (4u * OPE_H) + 16u * g;
    if (EXP & 16) { rs = 1.0f; mu = 0.f; bits = 0; }
    else relu_ln(z, 1, 2, (save && valid) ? reinterpret_cast<char*>(a.xhat1) : nullptr, xh_off, rs, mu, bits);
    if (save) {
      const uint64_t m = or4(bits);
      if (valid) {
        if (g == 0) {
          a.mask1[row] = m;
          a.rstd1[row] = rs;
          if (a.mu1) a.mu1[row] = mu;
        }
      }
    }
    if (done == 0) stamp(4);
    // ---- fc2 ----
    f32x4 z2[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) z2[it] = lnp4(3, it);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      f32x4 wv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) wv[it] = *reinterpret_cast<const f32x4*>(W2b + (wo2[ft] + (uint32_t)(16 * it * 4 * OPE_H)));
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int it = 0; it < 4; ++it) z2[it] = t4_mfma<EXP>(wv[it][r], z[ft][r], z2[it]);
    }
    if (EXP & 16) { rs = 1.0f; mu = 0.f; bits = 0; }
    else relu_ln(z2, 4, 5, (save && valid) ? reinterpret_cast<char*>(a.xhat2) : nullptr, xh_off, rs, mu, bits);
    if (save) {
      const uint64_t m = or4(bits);
      if (valid) {
        if (g == 0) {
          a.mask2[row] = m;
          a.rstd2[row] = rs;
        }
      }
    }
    if (done == 0) stamp(5);
    // ---- gi = W_ih a2 + b_ih: 12 output tiles, four at a time ----
    constexpr int NG = (3 * OPE_H - C::W3R) / 16;           // 16-gate tiles read from L2 (0, or 2 at KCM = 24)
    f32x4 w3g[NG > 0 ? NG : 1][4];
#pragma unroll
    for (int u0 = 0; u0 < 12; u0 += 4) {
      if (NG > 0 && u0 == 8) {      // the group that holds the L2-resident tiles: their fragments, requested here (32 registers for a third of the phase)
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
          for (int ft = 0; ft < 4; ++ft)
            w3g[u][ft] = *reinterpret_cast<const f32x4*>(th + a.wih + (C::W3R + 16 * u + j) * OPE_H + 16 * ft + 4 * g);
      }
      f32x4 o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = *reinterpret_cast<const f32x4*>(bih + 16 * (u0 + u) + 4 * g);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        if ((ft & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        f32x4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (16 * (u0 + u) < C::W3R) wv[u] = *reinterpret_cast<const f32x4*>(W3b + (wo2[ft] + (uint32_t)(16 * (u0 + u) * 4 * OPE_H)));
          else wv[u] = w3g[u0 + u - C::W3R / 16][ft];      // (the tiles that stay in L2: requested at the top of the W_ih phase)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int u = 0; u < 4; ++u) o[u] = t4_mfma<EXP>(wv[u][r], z2[ft][r], o[u]);
      }
      if (EXP & 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(o[u]));
      } else if (valid) {
        const uint32_t go = (uint32_t)row * (12u * OPE_H) + 16u * g + 64u * u0;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(gi_out) + (go + 64u * u)) = o[u];
      }
    }
    if (done == 0) stamp(6);
    ++done;
    tile = next;
    src_cur = src_next;
  }
  stamp(7);
  if (dbg && lane == 0) dbg[8] = done;
}

// Rows from which "by shape" picks the pair launch. Round 4: 2 048 (it was 16 384, "a few tiles per SIMD"): one launch with the weights
// staged in LDS also beats the two register-resident launches when most waves get no tile at all -- 3m batch 32 (5 856 rows): 12.8 us
// against 11.6 + 10.0 us, step 0.1387 -> 0.1305 ms; 3s5z batch 8 (9 664 rows): 27.4 against 17.5 + 15.7 us. (The adjoint kernel keeps its
// 16 384: trunk_bwd4 9.3 us against trunk_bwd3 7.9 us at 3m.) OPE_TRUNK4_MINROWS overrides.
int trunk4_pair_min_rows() {
  static const int v = getenv("OPE_TRUNK4_MINROWS") ? atoi(getenv("OPE_TRUNK4_MINROWS")) : 2048;
  return v;
}

// Both nets' trunks in one launch when the shape allows it (recurrent nets, D <= 256, enough rows to give every SIMD of the chip a few
// tiles); otherwise the two trunk_fwd3 / trunk_fwd2 launches.
int launch_trunk_fwd_pair(const TrunkFwdArgs& live, const TrunkFwdArgs& tgt, int path, hipStream_t st) {
  // path (ope_qmix_cfg.trunk_path): 0 by shape, 3 the two trunk_fwd3 launches, 4 this kernel whenever it can run the shape.
  // Process default of "by shape": OPE_TRUNK4 = 1 | 0 (read once).
  static const int on = getenv("OPE_TRUNK4") ? atoi(getenv("OPE_TRUNK4")) : 1;
  const int KC = (live.D + 15) >> 4;
  const bool can = live.gi && tgt.gi && !live.a2_out && !tgt.a2_out && live.D == tgt.D && live.R == tgt.R && live.x == tgt.x && live.no_fn == tgt.no_fn &&
                   ((live.D % 4 == 0 && (KC == 4 || KC == 8 || KC == 12 || KC == 16)) || (live.D % 2 == 0 && KC == 24)) && live.xhat1 && live.mu0 && live.rstd0 && live.rstd1 && live.mask1 && live.xhat2 && live.rstd2 && live.mask2;
  if (path == 4 && (!can || live.tanh_act)) return OPE_EINVAL;      // an explicit request the shape does not allow: no silent fall-back (tests pin kernels by path)
  const bool ok = can && !live.tanh_act && (path == 4 || (path == 0 && on && live.R >= trunk4_pair_min_rows()));
  const bool lazy = live.ref.inds != nullptr;
  if (lazy && (!ok || KC == 24 || live.ref.B < 1 || live.ref.B > kObsRefMaxB || live.R % live.ref.B != 0 || live.ref_row0 % live.ref.B != 0 || live.ref.cap < 1))
    return OPE_EINVAL;                           // only this kernel reads rows from the store (ope_qmix_obs_ref_ok tells the caller beforehand)
  if (!ok) {
    int rc = launch_trunk_fwd(live, true, st, false);
    if (rc) return rc;
    return launch_trunk_fwd(tgt, false, st, false);
  }
  TrunkPairArgs pa;
  pa.x = live.x; pa.R = live.R; pa.D = live.D; pa.nets = 2;
  pa.theta[0] = live.theta; pa.theta[1] = tgt.theta; pa.gi[0] = live.gi; pa.gi[1] = tgt.gi;
  const AgentLayout& L = live.L;
  pa.fn_w = L.fn_w; pa.fn_b = L.fn_b; pa.fc1_w = L.fc1_w; pa.fc1_b = L.fc1_b; pa.ln1_w = L.ln1_w; pa.ln1_b = L.ln1_b;
  pa.fc2_w = L.fc2_w; pa.fc2_b = L.fc2_b; pa.ln2_w = L.ln2_w; pa.ln2_b = L.ln2_b; pa.wih = L.wih; pa.bih = L.bih;
  pa.mu0 = live.mu0; pa.rstd0 = live.rstd0; pa.xhat1 = live.xhat1; pa.rstd1 = live.rstd1; pa.mu1 = live.mu1; pa.mask1 = live.mask1;
  pa.xhat2 = live.xhat2; pa.rstd2 = live.rstd2; pa.mask2 = live.mask2;      // the target net saves nothing
  pa.dbg = live.dbg;
  pa.no_fn = live.no_fn;
  pa.save0 = 1;
  pa.ref = live.ref; pa.ref_tn0 = lazy ? live.ref_row0 / live.ref.B : 0;
  const bool pk = live.lp.hdr != nullptr;
  if (pk && (lazy || !ok)) return OPE_EINVAL;      // (the caller checks trunk4_pair_can first)
  pa.live_hdr = live.lp.hdr; pa.live_src = live.lp.srcrow;
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  const int grid = cus & ~1;     // one workgroup per CU, even ones the live net, odd ones the target
  // Eight waves (two per SIMD, rows prefetched behind fc1). Measured against twelve waves without the prefetch (three per SIMD at the 168
  // register cap, 78 of them spilled): 59.5 us vs 107 us at 3s5z batch 32, so only this variant is built. (Round 4: with the input-LayerNorm
  // products pinned -- see the KCM = 24 note in the kernel -- the twelve-wave forms fit 144 / 168 registers without spills and run 60.8 /
  // 61.2 us against 61.3 us: a third wave per SIMD buys nothing, the kernel is not waiting on latency.)
  kprof_work(2.0 * 2.0 * live.R * ((double)live.D * OPE_H + OPE_H * OPE_H + 3.0 * OPE_H * OPE_H));     // both nets
  // (KC = 24: rows prefetched behind fc1 up to chunk 16, the rest at the top of the tile: 117.8 us at MMM2 batch 32 against 119.7 us without
  // any prefetch and 168.6 us for the two trunk_fwd3<2, 24> launches)
#ifdef OPE_EXPERIMENTS
  static const int t4exp = getenv("OPE_T4_EXP") ? atoi(getenv("OPE_T4_EXP")) : 0;
  if (t4exp && KC == 16 && !lazy) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "libope: OPE_T4_EXP=%d -- a timing-only variant of trunk_fwd4_kernel runs: its outputs are WRONG\n", t4exp); warned = true; }
#define OPE_T4_CASE(E) case E: OPE_LAUNCH((trunk_fwd4_kernel<16, 8, true, false, 4, E>), dim3(grid), dim3(512), 0, st, pa); break;
    switch (t4exp) {
      OPE_T4_CASE(1) OPE_T4_CASE(2) OPE_T4_CASE(4) OPE_T4_CASE(8) OPE_T4_CASE(16) OPE_T4_CASE(32) OPE_T4_CASE(17) OPE_T4_CASE(58) OPE_T4_CASE(62) OPE_T4_CASE(59) OPE_T4_CASE(63)
      default: return OPE_EINVAL;
    }
#undef OPE_T4_CASE
    if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
    note_launch("trunk_fwd4", KC);
    return OPE_OK;
  }
#endif
  if (pk) kprof_rows(1);
  if (pk) {
    if (KC == 24) OPE_LAUNCH((trunk_fwd4_kernel<24, 8, true, false, 2, 0, true>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 4) OPE_LAUNCH((trunk_fwd4_kernel<4, 8, true, false, 4, 0, true>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 8) OPE_LAUNCH((trunk_fwd4_kernel<8, 8, true, false, 4, 0, true>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 12) OPE_LAUNCH((trunk_fwd4_kernel<12, 8, true, false, 4, 0, true>), dim3(grid), dim3(512), 0, st, pa);
    else OPE_LAUNCH((trunk_fwd4_kernel<16, 8, true, false, 4, 0, true>), dim3(grid), dim3(512), 0, st, pa);
  } else if (KC == 24) OPE_LAUNCH((trunk_fwd4_kernel<24, 8, true, false, 2>), dim3(grid), dim3(512), 0, st, pa);
  else if (lazy) {
    if (KC == 4) OPE_LAUNCH((trunk_fwd4_kernel<4, 8, true, true>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 8) OPE_LAUNCH((trunk_fwd4_kernel<8, 8, true, true>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 12) OPE_LAUNCH((trunk_fwd4_kernel<12, 8, true, true>), dim3(grid), dim3(512), 0, st, pa);
    else OPE_LAUNCH((trunk_fwd4_kernel<16, 8, true, true>), dim3(grid), dim3(512), 0, st, pa);
  } else {
    if (KC == 4) OPE_LAUNCH((trunk_fwd4_kernel<4, 8, true, false>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 8) OPE_LAUNCH((trunk_fwd4_kernel<8, 8, true, false>), dim3(grid), dim3(512), 0, st, pa);
    else if (KC == 12) OPE_LAUNCH((trunk_fwd4_kernel<12, 8, true, false>), dim3(grid), dim3(512), 0, st, pa);
    else OPE_LAUNCH((trunk_fwd4_kernel<16, 8, true, false>), dim3(grid), dim3(512), 0, st, pa);
  }
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(lazy ? "trunk_fwd4_store" : (pk ? "trunk_fwd4_live" : "trunk_fwd4"), KC);
  return OPE_OK;
}

// Would launch_trunk_fwd_pair run trunk_fwd4 on a gathered batch of this shape? (what the live-row path needs to know before it plans)
bool trunk4_pair_can(int D, int64_t R, int path, bool tanh_act) {
  static const int on = getenv("OPE_TRUNK4") ? atoi(getenv("OPE_TRUNK4")) : 1;
  const int KC = (D + 15) >> 4;
  const bool shape = (D % 4 == 0 && (KC == 4 || KC == 8 || KC == 12 || KC == 16)) || (D % 2 == 0 && KC == 24);
  return shape && !tanh_act && (path == 4 || (path == 0 && on && R >= trunk4_pair_min_rows()));
}

// ONE net's trunk on the LDS-resident kernel (every CU that net's weights): the recurrent MADDPG / MATD3 actors, whose live, target and
// rollout trunks are separate launches over 10^5 rows (config 5: 231 680 rows of MMM2's 370-wide observations). Returns 1 when the
// launch is not this kernel's (shape, outputs, row count): the caller goes on to the register-resident forms.
int launch_trunk_fwd4_single(const TrunkFwdArgs& a, bool save, hipStream_t st) {
  static const int on = getenv("OPE_TRUNK4") ? atoi(getenv("OPE_TRUNK4")) : 1;
  const int KC = (a.D + 15) >> 4;
  const bool shape = (a.D % 4 == 0 && (KC == 4 || KC == 8 || KC == 12 || KC == 16)) || (a.D % 2 == 0 && KC == 24);
  if (!on || !shape || !a.gi || a.a2_out || a.head_out || a.ref.inds || a.tanh_act || a.R < 16 * 1024) return 1;
  if (save && !(a.xhat1 && a.mu0 && a.rstd0 && a.rstd1 && a.mask1 && a.xhat2 && a.rstd2 && a.mask2)) return 1;
  TrunkPairArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.x = a.x; pa.R = a.R; pa.D = a.D; pa.nets = 1;
  pa.theta[0] = pa.theta[1] = a.theta; pa.gi[0] = pa.gi[1] = a.gi;
  const AgentLayout& L = a.L;
  pa.fn_w = L.fn_w; pa.fn_b = L.fn_b; pa.fc1_w = L.fc1_w; pa.fc1_b = L.fc1_b; pa.ln1_w = L.ln1_w; pa.ln1_b = L.ln1_b;
  pa.fc2_w = L.fc2_w; pa.fc2_b = L.fc2_b; pa.ln2_w = L.ln2_w; pa.ln2_b = L.ln2_b; pa.wih = L.wih; pa.bih = L.bih;
  pa.save0 = save ? 1 : 0;
  pa.no_fn = a.no_fn;
  if (a.lp.hdr) return OPE_EINVAL;
  if (save) {
    pa.mu0 = a.mu0; pa.rstd0 = a.rstd0; pa.xhat1 = a.xhat1; pa.rstd1 = a.rstd1; pa.mu1 = a.mu1; pa.mask1 = a.mask1;
    pa.xhat2 = a.xhat2; pa.rstd2 = a.rstd2; pa.mask2 = a.mask2;
  }
  pa.dbg = a.dbg;
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  kprof_work(2.0 * a.R * ((double)a.D * OPE_H + OPE_H * OPE_H + 3.0 * OPE_H * OPE_H));
  if (KC == 24) OPE_LAUNCH((trunk_fwd4_kernel<24, 8, true, false, 2>), dim3(cus), dim3(512), 0, st, pa);
  else if (KC == 4) OPE_LAUNCH((trunk_fwd4_kernel<4, 8, true, false>), dim3(cus), dim3(512), 0, st, pa);
  else if (KC == 8) OPE_LAUNCH((trunk_fwd4_kernel<8, 8, true, false>), dim3(cus), dim3(512), 0, st, pa);
  else if (KC == 12) OPE_LAUNCH((trunk_fwd4_kernel<12, 8, true, false>), dim3(cus), dim3(512), 0, st, pa);
  else OPE_LAUNCH((trunk_fwd4_kernel<16, 8, true, false>), dim3(cus), dim3(512), 0, st, pa);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("trunk_fwd4_single", KC, save ? 1 : 0);
  return OPE_OK;
}

}  // namespace ope
