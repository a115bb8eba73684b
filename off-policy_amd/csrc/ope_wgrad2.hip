// Weight-gradient reductions, register-blocked form (round 5):  C[M][N] = sum_k A[k][m] * B[k][n]   (K = 10^3..10^5 data rows)
//
// Replaces autograd's Linear backward (grad_W = grad_out^T @ input; qmix.py:190-191, one aten::mm per layer) like wgrad_kernel
// (ope_wgrad.hip) does, with a different work split. Measured on wgrad_kernel (profiles/r04o_wgrad_decomposition.txt): a wave that
// owns ONE 64 x 64 tile loads 128 operand floats per reduction row for 16 MFMAs, and the launch is bound by those bytes through the
// CU's vector-memory path (15 B / cycle / CU: the K loop without any matrix math takes 40 of the 58 us), not by the matrix pipe
// (30-37 us). Here a wave owns a BLOCK of up to four tiles that share an operand -- fc1: one 64-column panel of dz1 against the four
// panels of the observation row (316 floats per row for 64 MFMAs), W_ih: three panels of dgi against xhat2 (256 for 48) -- so the
// operand floats per MFMA drop from 8 to 4.9-5.3 (2.4 from 1.5 k floats per reduction row over the agent problems) and the loop
// becomes matrix-pipe-bound. 256 accumulator registers per lane mean ONE wave per SIMD; 64 independent MFMAs (2 048 cycles) per
// 4-row stage cover a fetch on their own, a three-buffer ring is enough.
//
// Work split: the table's problems are cut into units (a unit = pa x pb tiles, pa * pb <= 4, of one problem), every unit gets a
// number of workgroups proportional to its matrix work (tiles x K; greedy, deterministic), a workgroup = 4 waves = 4 consecutive
// K chunks of its unit, summed in the workgroup (fixed order) and written as ONE slab; `w2_reduce_kernel` sums a unit's slabs in
// workgroup order into the same `rsum` vector split_reduce_kernel produced. No float atomics: bitwise deterministic.
//
// Per problem as before: column sums of A (bias gradients), LayerNorm-on-load of B (the never-materialised xhat0), row shift of
// B (h_{t-1} for dW_hh). Row validity is a 0/1 factor on the A fragment (B rows are clamped to real data), column validity needs
// nothing: out-of-range columns are never stored.
#include "ope_wgrad.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace ope {

namespace {

// acc += a (x) b as ONE in-place v_mfma_f32_16x16x4_f32 on an AGPR accumulator. The builtin leaves destination and addend to the register
// allocator, which with 192-256 live accumulator registers rotates them (vdst != srcC) and pays for it with v_accvgpr copies in the loop
// (116 per 192 MFMAs in the 4-tile bodies); a tied "+a" operand cannot be renamed. The statement is not volatile: the scheduler still moves
// loads and VALU work between the MFMAs. Hazards the compiler no longer sees: none inside the loop (an accumulator is touched again >= 16
// MFMAs later; A / B operands are read at issue), after it w2_mfma_drain() before the first accumulator read.
// operand buffers of the software pipeline: 4 with 16-byte loads, 6 with 8-byte ones (twice the load instructions per stage want the longer run-up;
// measured, same box: 3s5z (VEC 4) 54.1-54.9 us with 4 against 56.2-56.5 with 6, MMM2 (VEC 2) 99.8-100.5 with 4 against 89.8-90.0 with 6)
constexpr int w2_ring(int vec) { return vec == 4 ? 4 : 6; }
__device__ __forceinline__ void w2_mfma(float a, float b, f32x4& c) {
  asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// gfx950's f32-input MFMAs are not XDL operations: a VGPR written by a VALU instruction must not be read by one as SrcA / SrcB within two
// wait states (LLVM's hazard recognizer inserts them for the builtin; it does not look inside inline asm -- without this the first MFMAs
// behind the operand preparation read stale fragments: measured, one accumulator of a 4-tile unit off by up to 49 %).
__device__ __forceinline__ void w2_settle(f32x4& x) { asm volatile("s_nop 1" : "+v"(x)); }
__device__ __forceinline__ void w2_mfma_drain(f32x4& c) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(c)); }

template <int VEC>
__device__ __forceinline__ f32x4 w2_load(const float* __restrict__ row, int off, int off1) {
  if (VEC == 4) return *reinterpret_cast<const f32x4*>(row + off);
  const f32x2 a = *reinterpret_cast<const f32x2*>(row + off), b = *reinterpret_cast<const f32x2*>(row + off1);
  return f32x4{a[0], a[1], b[0], b[1]};
}

// One wave, one unit: acc[t = a * PB + b][mi][ni] over the reduction rows [k0, k1).
//
// Software pipeline, written out by hand (one wave per SIMD: nothing else hides anything). A STAGE is SUB groups of 4 reduction rows
// (48-64 MFMAs); four operand buffers rotate: while the MFMAs of stage s issue from buffer s % 4, the fragments of stage s + 1 are turned
// into MFMA operands IN PLACE (row masks, LayerNorm-on-load, column sums) and the loads of stage s + 3 go out into the buffer stage s - 1
// has just finished with -- both cut into pieces that sit between groups of four MFMAs (sched_barriers pin them; the matrix pipe takes
// 128 cycles per group, a piece issues in 20-60). A stage is 1-2 k cycles long, a fetch has three of them to land.
// EXP (only in builds with -DOPE_EXPERIMENTS; timing variants, results WRONG): 1 = no MFMAs, 2 = no loads inside the loop, 4 = no operand preparation
// MAP (packed rows of a live plan): B's row of reduction row k comes from P.b_map -- the batch row of a packed observation / state row, the
// packed row of the step before (h_{t-1} for dW_hh; negative at t = 0: no contribution) --, fetched ONE STAGE AHEAD of the operand fetch that
// uses it (a stage is 1-2 k cycles: the index has landed), into the index slot of the buffer the running stage computes from.
template <int VEC, int PA, int PB, int EXP, bool MAP>
__device__ __forceinline__ void w2_body(const WgProb& P, int mp0, int np0, int k0, int k1, f32x4 (&acc)[PA * PB][4][4], f32x4 (&cs)[PA]) {
  constexpr int NT = PA * PB, F = PA + PB;
  constexpr int SUB = NT == 1 ? 2 : 1;          // groups of 4 rows per stage
  constexpr int SR = 4 * SUB;                   // rows per stage
  constexpr int NB = w2_ring(VEC);              // operand buffers in rotation: the fetch of stage s + NB - 1 goes out during stage s
  constexpr int NG = NT * 8 * SUB;              // pairs of MFMAs per stage
  constexpr int NPF = SUB * (F + 2);            // fetch pieces per stage: per row group the row indices, one load per fragment, the LayerNorm pair
  constexpr int NP = NPF + SUB * F;             // + one preparation per fragment
  // (strictly fewer pieces than MFMA pairs: the last preparation of a stage then sits at least one MFMA group in front of the next stage's
  // first MFMA, which covers the two wait states a VALU write -> f32 MFMA read needs and the compiler does not see behind inline asm; the
  // prologue's w2_settle covers the first stage)
  static_assert(NP < NG, "a preparation piece directly in front of the next stage's first MFMA (VALU -> MFMA hazard)");
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  const int shift = MAP ? 0 : P.b_shift;
  const int* __restrict__ mapp = P.b_map;
  const bool map_on = MAP && P.map_on != 0;
  // per A panel: base pointer, leading dimension and this lane's column offsets (panels >= a2_from come from the second matrix A2)
  const float* abase[PA];
  int alda[PA], moff[PA], moff1[PA], noff[PB], noff1[PB];
#pragma unroll
  for (int a = 0; a < PA; ++a) {
    const bool second = P.A2 != nullptr && mp0 + a >= P.a2_from;
    abase[a] = second ? P.A2 : P.A;
    alda[a] = second ? P.lda2 : P.lda;
    const int c = 64 * (second ? mp0 + a - P.a2_from : mp0 + a) + 4 * i;
    moff[a] = VEC == 4 ? min(c, alda[a] - 4) : min(c, alda[a] - 2);
    moff1[a] = min(c + 2, alda[a] - 2);
  }
  const int ldb = P.ldb;
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    const int c = 64 * (np0 + b) + 4 * i;
    noff[b] = VEC == 4 ? min(c, ldb - 4) : min(c, ldb - 2);
    noff1[b] = min(c + 2, ldb - 2);
  }
  const float* __restrict__ Bp = P.B;
  const float* __restrict__ mup = P.ln_mu;     // (plain problems point at a zeros / a ones vector: no control flow in the loop -- a branch there
  const float* __restrict__ rsp = P.ln_rstd;   // splits it into blocks and the accumulators then travel through copies at every block boundary)
  struct Buf { f32x4 a[SUB][PA]; f32x4 b[SUB][PB]; float mu[SUB], rs[SUB]; };
  int kcS[SUB], krS[SUB];      // row indices of the fetch in progress (set by its first piece)
  int ixr[NB][SUB];            // MAP: b_map entries of the stage each buffer holds / is about to be fetched for
  auto idx_load = [&](int kb, int q) -> int { return mapp[min(kb + 4 * q + g, k1 - 1)]; };
  auto fetch_piece = [&](int kb, int q, int r, Buf& s, const int (&ixf)[SUB]) {      // rows kb + 4 q + g of every panel, one piece at a time
    if (r == 0) {
      kcS[q] = min(kb + 4 * q + g, k1 - 1);      // (the ring runs three stages past the chunk: those fetches hit the last row again, in L1)
      krS[q] = MAP ? max(map_on ? ixf[q] : kcS[q], 0) : max(kcS[q] - shift, 0);
    } else if (r <= PA) {
      s.a[q][r - 1] = w2_load<VEC>(abase[r - 1] + (int64_t)kcS[q] * alda[r - 1], moff[r - 1], moff1[r - 1]);
    } else if (r <= F) {
      s.b[q][r - 1 - PA] = w2_load<VEC>(Bp + (int64_t)krS[q] * ldb, noff[r - 1 - PA], noff1[r - 1 - PA]);
    } else {
      s.mu[q] = mup[MAP ? kcS[q] : krS[q]];      // (LayerNorm statistics were saved per packed row)
      s.rs[q] = rsp[MAP ? kcS[q] : krS[q]];
    }
  };
  auto fetch = [&](int kb, int q, Buf& s, const int (&ixf)[SUB]) {
#pragma unroll
    for (int r = 0; r < F + 2; ++r) fetch_piece(kb, q, r, s, ixf);
  };
  auto prep = [&](int kb, int q, int f, Buf& s, const int (&ixp)[SUB]) {      // fragment f of row group q becomes an MFMA operand, in place
    const int k = kb + 4 * q + g;
    if (f < PA) {
      const float ka = (k < k1) ? 1.f : 0.f;           // rows beyond the chunk contribute nothing (B rows are clamped to real data)
      s.a[q][f] = s.a[q][f] * ka;
      cs[f] += s.a[q][f];
    } else {
      const float kbm = MAP ? ((map_on ? ixp[q] : 0) >= 0 ? 1.f : 0.f) : ((k - shift >= 0) ? 1.f : 0.f);
      s.b[q][f - PA] = (s.b[q][f - PA] - s.mu[q]) * (s.rs[q] * kbm);
    }
  };
  auto piece = [&](int pz, int kb, Buf& nxt, Buf& fet, const int (&ixf)[SUB], const int (&ixp)[SUB], int (&ixn)[SUB]) {
    if (pz < NPF) {
      if (!(EXP & 2)) fetch_piece(kb + (NB - 1) * SR, pz / (F + 2), pz % (F + 2), fet, ixf);
      if (MAP && pz % (F + 2) == 0) ixn[pz / (F + 2)] = idx_load(kb + NB * SR, pz / (F + 2));      // the index of the fetch one stage on
    } else if (!(EXP & 4)) {
      prep(kb + SR, (pz - NPF) / F, (pz - NPF) % F, nxt, ixp);
    }
  };
  auto stage = [&](int kb, Buf& cur, Buf& nxt, Buf& fet, const int (&ixf)[SUB], const int (&ixp)[SUB], int (&ixn)[SUB]) {
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = 2 * grp + h, q = m / (NT * 16), t = (m / 16) % NT, mi = (m / 4) % 4, ni = m % 4;
        if (EXP & 1) asm volatile("" ::"v"(cur.a[q][t / PB][mi]), "v"(cur.b[q][t % PB][ni]));
        else w2_mfma(cur.a[q][t / PB][mi], cur.b[q][t % PB][ni], acc[t][mi][ni]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pz = 0; pz < NP; ++pz)
        if (pz * NG / NP == grp) piece(pz, kb, nxt, fet, ixf, ixp, ixn);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (k0 >= k1) return;
  Buf bufs[NB];
  if (EXP & 32) {      // (no prologue fetch)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int q = 0; q < SUB; ++q) {
#pragma unroll
        for (int a = 0; a < PA; ++a) bufs[n].a[q][a] = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int b = 0; b < PB; ++b) bufs[n].b[q][b] = f32x4{1.f, 1.f, 1.f, 1.f};
        bufs[n].mu[q] = 0.f;
        bufs[n].rs[q] = 1.f;
      }
  } else {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int q = 0; q < SUB; ++q) ixr[n][q] = MAP ? idx_load(k0 + n * SR, q) : 0;
#pragma unroll
    for (int n = 0; n < NB - 1; ++n)
#pragma unroll
      for (int q = 0; q < SUB; ++q) fetch(k0 + n * SR, q, bufs[n], ixr[n]);
  }
#pragma unroll
  for (int q = 0; q < SUB; ++q)
#pragma unroll
    for (int f = 0; f < F; ++f) prep(k0, q, f, bufs[0], ixr[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < SUB; ++q) {      // (VALU -> f32 MFMA needs two wait states the compiler does not see behind inline asm)
#pragma unroll
    for (int a = 0; a < PA; ++a) w2_settle(bufs[0].a[q][a]);
#pragma unroll
    for (int b = 0; b < PB; ++b) w2_settle(bufs[0].b[q][b]);
  }
#pragma unroll 1
  for (int kb = k0; kb < k1; kb += NB * SR) {      // (kchunk is a multiple of NB stages)
#pragma unroll
    for (int n = 0; n < NB; ++n) stage(kb + n * SR, bufs[n], bufs[(n + 1) % NB], bufs[(n + NB - 1) % NB], ixr[(n + NB - 1) % NB], ixr[(n + 1) % NB], ixr[n]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        if (t == NT - 1 && mi == 3 && ni == 3) w2_mfma_drain(acc[t][mi][ni]);      // (the last MFMA issued; every earlier one has retired by then)
}

// LDS of the workgroup's final sum: four exchange regions of 32 accumulator vectors (1 KB each: [lane][4]) + four column-sum regions
constexpr int kXchgVecs = 32;
struct W2Lds {
  float x[4][kXchgVecs][64][4];      // 128 KB
  float cs[4][4][64][4];             // 16 KB
  float stot[4][64];                 // fin: the workgroup's column sums of A, by panel
  float cp[2][4][4][64];             // fin: per wave, the column partials of dgamma / dbeta by B panel (8 KB)
};
// what the epilogue needs to fold the LayerNorm-fed-Linear identities into the slab (launch_wgrad2_fin; ope_wgrad.h: WgProb.fin_kind)
struct W2Fin { const float* theta; const WgProb* P; int mp0, np0; bool on; };

// The four waves of a workgroup hold four consecutive K chunks of the unit. Their sum, in rounds of 32 accumulator vectors: every wave
// publishes its copy of the round's vectors, then wave w adds the four copies of ITS share (two of the round's eight (tile, mi) groups) as
// (w0 + w2) + (w1 + w3) in temporaries and stores them -- all four waves add and store (the first version's serial tree, where one wave
// ended up adding and storing everything, took 4.9 us), and no accumulator register is indexed by the wave number (one code path).
// fin.on (a LayerNorm-fed Linear, finalize folded in): what is stored is this workgroup's share of the FINAL gradient, dW = C gamma + s (x) beta
// with C, s its partial sums (the identities are linear in them), and the column partials sum_m W[m][n] C[m][n] / sum_m s[m] W[m][n] of dgamma /
// dbeta are collected per wave in LDS on the way (fixed order: groups in program order, waves summed by the caller).
template <int NT, int PB>
__device__ __forceinline__ void w2_sum_store(f32x4 (&acc)[NT][4][4], W2Lds& L, int wave, int lane, float* __restrict__ slab, const W2Fin& fin) {
  constexpr int NV = NT * 16;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int v0 = 0; v0 < NV; v0 += kXchgVecs) {
    constexpr int dummy = 0;
    (void)dummy;
    const int rv = NV - v0 < kXchgVecs ? NV - v0 : kXchgVecs;      // vectors of this round: 32 or 16
    if (v0 > 0) __syncthreads();                                   // (the previous round's copies have been read)
#pragma unroll
    for (int j = 0; j < kXchgVecs; ++j)
      if (j < rv) {
        const int idx = v0 + j;
        *reinterpret_cast<f32x4*>(L.x[wave][j][lane]) = acc[idx / 16][(idx / 4) % 4][idx % 4];
        if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);
      }
    __syncthreads();
    const int per = rv / 16;                                       // (tile, mi) groups per wave: 2 or 1
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
      if (gq < per) {
        const int grp = wave * per + gq;                           // group of the round (runtime: only an LDS address and a store address)
        // s[ni][r] = C[64 a + 16 g + 4 r + mi][64 b + 4 i + ni]: per r one float4 over ni, row 16 g + 4 r + mi of tile t
        const int idx = v0 + grp * 4, t = idx >> 4, mi = (idx >> 2) & 3;
        f32x4 s[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int j = grp * 4 + ni;
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(L.x[0][j][lane]), c1 = *reinterpret_cast<const f32x4*>(L.x[1][j][lane]),
                      c2 = *reinterpret_cast<const f32x4*>(L.x[2][j][lane]), c3 = *reinterpret_cast<const f32x4*>(L.x[3][j][lane]);
          s[ni] = (c0 + c2) + (c1 + c3);
        }
        if (fin.on) {
          const WgProb& P = *fin.P;
          const int a = t / PB, b = t - a * PB;
          // (the parameter values are requested HERE, behind the LDS reads: hoisted in front of them -- as scalars or as 16-byte pieces -- they
          // stay live across the sums and the kernel's accumulator array goes to scratch: wgrad2 47 -> 110 us, measured)
          const int nb = 64 * (fin.np0 + b) + 4 * i, mb = 64 * (fin.mp0 + a) + 16 * g + mi;
          float gam[4], bet[4], st[4], cg[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int c = min(nb + ni, P.N - 1);
            gam[ni] = fin.theta[P.gam_off + c];
            bet[ni] = fin.theta[P.bet_off + c];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) st[r] = L.stot[a][16 * g + 4 * r + mi];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mb + 4 * r;
            const bool ok = m < P.M;      // (rows past M hold sums of clamped columns: never stored, and kept out of the column partials)
            const float* wr = fin.theta + P.g_off + (int64_t)min(m, P.M - 1) * P.ldc;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              const float wv = wr[min(nb + ni, P.N - 1)];
              cg[ni] = ok ? fmaf(wv, s[ni][r], cg[ni]) : cg[ni];
              cb[ni] = ok ? fmaf(wv, st[r], cb[ni]) : cb[ni];
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) s[ni][r] = fmaf(s[ni][r], gam[ni], st[r] * bet[ni]);
          }
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) { cg[ni] = rowsum4(cg[ni]); cb[ni] = rowsum4(cb[ni]); }
          if (g == 0) {
            f32x4* p0 = reinterpret_cast<f32x4*>(&L.cp[0][wave][b][4 * i]);
            f32x4* p1 = reinterpret_cast<f32x4*>(&L.cp[1][wave][b][4 * i]);
            *p0 = *p0 + f32x4{cg[0], cg[1], cg[2], cg[3]};
            *p1 = *p1 + f32x4{cb[0], cb[1], cb[2], cb[3]};
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<f32x4*>(slab + t * 4096 + (16 * g + 4 * r + mi) * 64 + 4 * i) = f32x4{s[0][r], s[1][r], s[2][r], s[3][r]};
      }
  }
}

// One workgroup, one unit of shape PA x PB.
template <int VEC, int PA, int PB, int EXP, bool MAP>
__device__ __forceinline__ void w2_unit(const WgProb& P, const W2Unit& U, int wg, int wave, W2Lds& L, float* __restrict__ raw, const float* __restrict__ theta) {
  constexpr int NT = PA * PB;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  const int chunk = (wg - U.wg_begin) * 4 + wave;
  int K = P.K, kchunk = U.kchunk;
  if (MAP && P.K_dev) {      // the rows of this step, known on the device only: the same split rule as w2_build's rows_of()
    K = min(__builtin_amdgcn_readfirstlane(*P.K_dev), P.K);
    constexpr int gr = (NT == 1 ? 8 : 4) * w2_ring(VEC);
    kchunk = gr * ((((K + 4 * U.nwg - 1) / (4 * U.nwg)) + gr - 1) / gr);
  }
  const int k0 = min(chunk * kchunk, K);
  const int k1 = min(K, k0 + kchunk);
  f32x4 acc[NT][4][4];
  f32x4 cs[PA];
#pragma unroll
  for (int a = 0; a < PA; ++a) cs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  w2_body<VEC, PA, PB, EXP, MAP>(P, U.mp0, U.np0, k0, k1, acc, cs);
  float* __restrict__ slab = raw + (int64_t)wg * kW2Slab;
  const bool colsum = P.s_off >= 0 && U.np0 == 0;
  if (EXP & 8) {
    if (acc[0][0][0][0] == 123.456f) raw[0] = 1.f;
    return;
  }
  W2Fin fin;
  fin.theta = theta; fin.P = &P; fin.mp0 = U.mp0; fin.np0 = U.np0;
  fin.on = theta != nullptr && P.fin_kind == 1;      // (uniform over the workgroup)
  if (colsum || fin.on) {
#pragma unroll
    for (int a = 0; a < PA; ++a) *reinterpret_cast<f32x4*>(L.cs[wave][a][lane]) = cs[a];
  }
  if (fin.on) {
    // the workgroup's column sums of A, needed by every wave in front of its stores; the per-wave partial vectors start at zero
    for (int q = threadIdx.x; q < 2 * 4 * 4 * 64; q += 256) (&L.cp[0][0][0][0])[q] = 0.f;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int a = 0; a < PA; ++a) {
        const f32x4 c0 = cs[a], c1 = *reinterpret_cast<const f32x4*>(L.cs[1][a][lane]), c2 = *reinterpret_cast<const f32x4*>(L.cs[2][a][lane]),
                    c3 = *reinterpret_cast<const f32x4*>(L.cs[3][a][lane]);
        const f32x4 c = (c0 + c2) + (c1 + c3);
        f32x4 sv;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[q] = rowsum4(c[q]);
        if (g == 0) *reinterpret_cast<f32x4*>(&L.stot[a][4 * i]) = sv;
      }
    }
    // (the first round's barrier inside w2_sum_store, behind the publishes, orders stot / cp in front of their first use)
  }
  w2_sum_store<NT, PB>(acc, L, wave, lane, slab, fin);
  if (fin.on) {      // dgamma / dbeta partials of the workgroup: (w0 + w2) + (w1 + w3) per column
    __syncthreads();
    for (int q = threadIdx.x; q < 2 * PB * 64; q += 256) {
      const int which = q / (PB * 64), rem = q - which * (PB * 64), b = rem >> 6, c = rem & 63;
      slab[kW2CpOff + which * 256 + b * 64 + c] = (L.cp[which][0][b][c] + L.cp[which][2][b][c]) + (L.cp[which][1][b][c] + L.cp[which][3][b][c]);
    }
  }
  if (colsum && wave == 0) {      // column sums of A: (w0 + w2) + (w1 + w3), then the four k-row groups g
#pragma unroll
    for (int a = 0; a < PA; ++a) {
      const f32x4 c0 = cs[a], c1 = *reinterpret_cast<const f32x4*>(L.cs[1][a][lane]), c2 = *reinterpret_cast<const f32x4*>(L.cs[2][a][lane]),
                  c3 = *reinterpret_cast<const f32x4*>(L.cs[3][a][lane]);
      const f32x4 c = (c0 + c2) + (c1 + c3);
      f32x4 sv;
#pragma unroll
      for (int q = 0; q < 4; ++q) sv[q] = rowsum4(c[q]);
      if (g == 0) *reinterpret_cast<f32x4*>(slab + 4 * 4096 + a * 64 + 4 * i) = sv;
    }
  }
}

template <int VEC, int EXP = 0, bool MAP = false>
__global__ void __launch_bounds__(256, 1) wgrad2_kernel(W2Table tb, float* __restrict__ raw, const float* __restrict__ theta) {
  __shared__ __attribute__((aligned(16))) W2Lds L;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wg = blockIdx.x;
  int u = 0;
#pragma unroll
  for (int q = 1; q < kMaxW2Units; ++q)      // (one burst of scalar loads over the contiguous start table, INT_MAX beyond the last unit)
    if (wg >= tb.ubegin[q]) u = q;
  const W2Unit& U = tb.u[u];
  const WgProb& P = tb.p[U.prob];
  switch (U.pa * 8 + U.pb) {
    case 1 * 8 + 1: w2_unit<VEC, 1, 1, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    case 1 * 8 + 2: w2_unit<VEC, 1, 2, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    case 1 * 8 + 3: w2_unit<VEC, 1, 3, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    case 1 * 8 + 4: w2_unit<VEC, 1, 4, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    case 2 * 8 + 1: w2_unit<VEC, 2, 1, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    case 3 * 8 + 1: w2_unit<VEC, 3, 1, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    case 4 * 8 + 1: w2_unit<VEC, 4, 1, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
    default: w2_unit<VEC, 2, 2, EXP, MAP>(P, U, wg, wave, L, raw, theta); break;
  }
}

// rsum[...] = sum over a unit's workgroup slabs. Workgroup = (unit, FOUR 64-float rows of its tiles / column sums): wave w of eight
// sums the slabs w, w + 8, ... in that order for each of the four rows (a lane = one column: 256-byte coalesced reads, up to 36 loads in
// flight per lane -- the chain of a lane that walks all ~70 slabs alone is what the first version waited for, and one row per workgroup
// (3 120 workgroups of 512 threads, three residency rounds of a ~2 us latency chain each) what the second one did), the partial sums meet
// through LDS as ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)), wave r finishing row r. Fixed order.
constexpr int kRedRows = 4;
__global__ void __launch_bounds__(512) w2_reduce_kernel(W2Table tb, const float* __restrict__ raw, float* __restrict__ rsum) {
  __shared__ float part[kRedRows][8][64];
  const W2Unit& U = tb.u[blockIdx.y];      // grid = (row quads of the largest unit, units): no search
  const WgProb& P = tb.p[U.prob];
  const int ntile = U.pa * U.pb;
  if ((int)blockIdx.x * kRedRows >= U.red_rows) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* __restrict__ src = raw + (int64_t)U.wg_begin * kW2Slab;
  float* __restrict__ dst = rsum + P.rs_base;
  const int nwg = U.nwg;
  const float* s[kRedRows];
  int64_t out[kRedRows];
  bool live[kRedRows];
#pragma unroll
  for (int r = 0; r < kRedRows; ++r) {
    const int local = (int)blockIdx.x * kRedRows + r;
    if (local >= U.red_rows) {              // (uniform) past the unit's rows: reads row 0 again, writes nothing
      s[r] = src + lane; out[r] = 0; live[r] = false;
    } else if (local >= ntile * 64) {       // a panel's column sums
      const int a = local - ntile * 64;
      const int m = 64 * (U.mp0 + a) + lane;
      s[r] = src + 4 * 4096 + a * 64 + lane;
      out[r] = P.s_off + m;
      live[r] = m < P.M;
    } else {
      const int t = local >> 6, ml = local & 63;
      const int a = t / U.pb, b = t - a * U.pb;
      const int m = 64 * (U.mp0 + a) + ml, n = 64 * (U.np0 + b) + lane;
      s[r] = src + t * 4096 + ml * 64 + lane;
      out[r] = P.out_off + (int64_t)m * P.ldc + n;
      live[r] = m < P.M && n < P.N;
    }
  }
  float v[kRedRows] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 9
  for (int q = wave; q < nwg; q += 8) {
#pragma unroll
    for (int r = 0; r < kRedRows; ++r) v[r] += s[r][(int64_t)q * kW2Slab];
  }
#pragma unroll
  for (int r = 0; r < kRedRows; ++r) part[r][wave][lane] = v[r];
  __syncthreads();
  int64_t o = 0;
  bool lv = false;
#pragma unroll
  for (int r = 0; r < kRedRows; ++r)      // (selects, not an index: the arrays stay in registers)
    if (wave == r) { o = out[r]; lv = live[r]; }
  if (lv) {
    const float(&p)[8][64] = part[wave];
    dst[o] = ((p[0][lane] + p[1][lane]) + (p[2][lane] + p[3][lane])) + ((p[4][lane] + p[5][lane]) + (p[6][lane] + p[7][lane]));
  }
}

// The slab sum with the finalize step folded in (launch_wgrad2_fin): what w2_reduce_kernel does, but the sums land in the FLAT GRADIENT -- a
// tile row at its weight's place (a LayerNorm-fed Linear's slabs already hold dW = C gamma + s (x) beta per workgroup), a column-sum vector at
// its bias', the dgamma / dbeta column partials at the LayerNorm's parameters -- and every workgroup leaves the sum of squares of what it
// wrote (the clip norm's partials: ope_adam_step needs no pass of its own). Row y = nu of the grid: workgroup 0 the loss tail (as
// finalize_kernel's), the next ones the zero ranges (grad-less tensors, padding). One launch instead of w2_reduce + finalize.
__global__ void __launch_bounds__(512) w2_fin_kernel(W2Table tb, const float* __restrict__ raw, float* __restrict__ grad, float* __restrict__ gsq, FinMisc mz) {
  __shared__ float part[kRedRows][8][64];
  __shared__ float sq[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.y == tb.nu) {      // ---- tail, zero ranges
    const int nzb = (mz.zcum[mz.nz] + 2047) / 2048;
    const int x = blockIdx.x;
    if (x > nzb) return;
    const int gidx = tb.fin_live_blocks - (nzb + 1) + x;
    if (x == 0) {
      __shared__ float red[512][3];
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int q = threadIdx.x; q < mz.n_loss_tiles; q += 512) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(mz.loss_part + (int64_t)q * 4);
        a0 += v[0]; a1 += v[1]; a2 += v[2];
      }
      red[threadIdx.x][0] = a0; red[threadIdx.x][1] = a1; red[threadIdx.x][2] = a2;
      __syncthreads();
      for (int o = 256; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
          for (int c = 0; c < 3; ++c) red[threadIdx.x][c] += red[threadIdx.x + o][c];
        __syncthreads();
      }
      if (mz.n_loss_tiles >= 0 && threadIdx.x < 4) grad[mz.tail_off + threadIdx.x] = threadIdx.x < 3 ? red[0][threadIdx.x] : 0.f;
      // whatever another path of the step left behind this launch's partials
      for (int q = tb.fin_live_blocks + (int)threadIdx.x; q < mz.n_gsq_total; q += 512) gsq[q] = 0.f;
    } else {
      for (int e = (x - 1) * 2048 + (int)threadIdx.x; e < min(x * 2048, mz.zcum[mz.nz]); e += 512) {
        int r = 0;
#pragma unroll
        for (int q = 1; q < kMaxFinZero; ++q)
          if (q < mz.nz && e >= mz.zcum[q]) r = q;
        grad[mz.zbegin[r] + (e - mz.zcum[r])] = 0.f;
      }
    }
    if (threadIdx.x == 0) gsq[gidx] = 0.f;
    return;
  }
  const W2Unit& U = tb.u[blockIdx.y];
  const WgProb& P = tb.p[U.prob];
  const int ntile = U.pa * U.pb;
  if ((int)blockIdx.x * kRedRows >= U.fin_rows) return;
  const float* __restrict__ src = raw + (int64_t)U.wg_begin * kW2Slab;
  const int nwg = U.nwg;
  const int ncs = U.red_rows - ntile * 64;      // column-sum vectors of this unit (0 or pa)
  const float* s[kRedRows];
  int64_t out[kRedRows];
  bool live[kRedRows];
#pragma unroll
  for (int r = 0; r < kRedRows; ++r) {
    const int local = (int)blockIdx.x * kRedRows + r;
    if (local >= U.fin_rows) {
      s[r] = src + lane; out[r] = 0; live[r] = false;
    } else if (local >= ntile * 64 + ncs) {      // dgamma / dbeta partial vectors: [which][b]
      const int q = local - ntile * 64 - ncs, which = q / U.pb, b = q - which * U.pb;
      const int n = 64 * (U.np0 + b) + lane;
      s[r] = src + kW2CpOff + which * 256 + b * 64 + lane;
      out[r] = (which == 0 ? P.gam_off : P.bet_off) + n;
      live[r] = n < P.N;
    } else if (local >= ntile * 64) {
      const int a = local - ntile * 64;
      const int m = 64 * (U.mp0 + a) + lane;
      s[r] = src + 4 * 4096 + a * 64 + lane;
      out[r] = P.gs_off + m;
      live[r] = m < P.M && P.gs_off >= 0;
    } else {
      const int t = local >> 6, ml = local & 63;
      const int a = t / U.pb, b = t - a * U.pb;
      const int m = 64 * (U.mp0 + a) + ml, n = 64 * (U.np0 + b) + lane;
      s[r] = src + t * 4096 + ml * 64 + lane;
      out[r] = P.g_off + (int64_t)m * P.ldc + n;
      live[r] = m < P.M && n < P.N;
    }
  }
  float v[kRedRows] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 9
  for (int q = wave; q < nwg; q += 8) {
#pragma unroll
    for (int r = 0; r < kRedRows; ++r) v[r] += s[r][(int64_t)q * kW2Slab];
  }
#pragma unroll
  for (int r = 0; r < kRedRows; ++r) part[r][wave][lane] = v[r];
  __syncthreads();
  int64_t o = 0;
  bool lv = false;
#pragma unroll
  for (int r = 0; r < kRedRows; ++r)
    if (wave == r) { o = out[r]; lv = live[r]; }
  float w = 0.f;
  if (lv) {
    const float(&p)[8][64] = part[wave];
    w = ((p[0][lane] + p[1][lane]) + (p[2][lane] + p[3][lane])) + ((p[4][lane] + p[5][lane]) + (p[6][lane] + p[7][lane]));
    grad[o] = w;
  }
  // sum of squares of the (up to four) rows this workgroup wrote: waves 0 .. 3, fixed order
  float q2 = w * w;
  q2 = row16_sum(q2);
  q2 = (__shfl(q2, 0, 64) + __shfl(q2, 16, 64)) + (__shfl(q2, 32, 64) + __shfl(q2, 48, 64));
  if (lane == 0) sq[wave] = q2;
  __syncthreads();
  if (threadIdx.x == 0) gsq[U.fin_blk + blockIdx.x] = (sq[0] + sq[1]) + (sq[2] + sq[3]);
}

}  // namespace

static int launch_wgrad2_impl(const W2Table& w, float* raw, float* rsum, const float* fin_theta, hipStream_t st);

bool w2_attach_fin(W2Table* wp, const FinTable& ft, FinMisc* misc) {
  W2Table& w = *wp;
  w.fin = 0;
  memset(misc, 0, sizeof(*misc));
  misc->tail_off = -1;
  if (ft.n < 1 || ft.n > kMaxFinSegs) return false;
  bool claimed[kMaxFinSegs] = {};
  auto find = [&](int kind, int src) -> int {
    for (int q = 0; q < ft.n; ++q)
      if (ft.seg[q].kind == kind && ft.seg[q].src == src && ft.seg[q].size > 0) return q;
    return -1;
  };
  auto find_begin = [&](int kind, int begin) -> int {
    for (int q = 0; q < ft.n; ++q)
      if (ft.seg[q].kind == kind && ft.seg[q].begin == begin) return q;
    return -1;
  };
  for (int q = 0; q < w.np; ++q) {
    WgProb& P = w.p[q];
    P.fin_kind = 0; P.g_off = -1; P.gs_off = -1; P.gam_off = P.bet_off = 0;
    const int cstart = P.rs_base + P.out_off;
    int sw = find(FIN_LNLIN_W, cstart);
    if (sw >= 0) {
      const FinSeg& F = ft.seg[sw];
      if (F.M != P.M || F.K != P.ldc || P.N > P.ldc || F.size != P.M * P.ldc || F.w != F.begin || P.s_off < 0 || F.src_s != P.rs_base + P.s_off) return false;
      P.fin_kind = 1; P.g_off = F.begin; P.gam_off = F.gamma; P.bet_off = F.beta;
      claimed[sw] = true;
      // the LayerNorm's own gradients: present unless its slots are constants (no feature norm: FIN_ZERO there)
      const int sg = find_begin(FIN_LNLIN_G, F.gamma), sb = find_begin(FIN_LNLIN_B, F.beta);
      if ((sg >= 0) != (sb >= 0)) return false;
      if (sg >= 0) {
        if (ft.seg[sg].src != cstart || ft.seg[sb].src_s != F.src_s || ft.seg[sg].size != P.N || ft.seg[sb].size != P.N) return false;
        claimed[sg] = claimed[sb] = true;
      } else {
        P.fin_kind = 2;      // transform only: nothing is stored for the LayerNorm
      }
    } else {
      sw = find(FIN_COPY, cstart);
      if (sw < 0 || ft.seg[sw].size != P.M * P.ldc || P.N > P.ldc) return false;
      P.g_off = ft.seg[sw].begin;
      claimed[sw] = true;
    }
    if (P.s_off >= 0) {
      const int ss = find(FIN_COPY, P.rs_base + P.s_off);
      if (ss >= 0) {
        if (ft.seg[ss].size != P.M) return false;
        P.gs_off = ft.seg[ss].begin;
        claimed[ss] = true;
      }
    }
  }
  // every segment with a value has a producer; order by position for the gaps
  int order[kMaxFinSegs];
  for (int q = 0; q < ft.n; ++q) order[q] = q;
  std::sort(order, order + ft.n, [&](int x, int y) { return ft.seg[x].begin < ft.seg[y].begin; });
  int nz = 0, zc = 0;
  auto zero = [&](int begin, int len) -> bool {
    if (len <= 0) return true;
    if (nz >= kMaxFinZero) return false;
    misc->zbegin[nz] = begin; misc->zlen[nz] = len; misc->zcum[nz] = zc;
    zc += len; ++nz;
    return true;
  };
  for (int oi = 0; oi < ft.n; ++oi) {
    const FinSeg& F = ft.seg[order[oi]];
    const int next = oi + 1 < ft.n ? ft.seg[order[oi + 1]].begin : (int)ft.total;
    switch (F.kind) {
      case FIN_ZERO: if (!zero(F.begin, next - F.begin)) return false; break;
      case FIN_TAIL: misc->tail_off = F.begin; break;
      case FIN_SKIP: return false;
      default:
        if (F.size > 0 && !claimed[order[oi]]) return false;
        if (!zero(F.begin + F.size, next - (F.begin + F.size))) return false;
    }
  }
  if (ft.seg[order[0]].begin != 0 && !zero(0, ft.seg[order[0]].begin)) return false;
  misc->nz = nz;
  misc->zcum[nz] = zc;
  for (int q = nz + 1; q <= kMaxFinZero; ++q) misc->zcum[q] = zc;
  if (misc->tail_off < 0) return false;
  // a LayerNorm-fed Linear: all of its rows in ONE row block (the column partials of a unit are then complete for its columns)
  int blk = 0, bx = 0;
  for (int q = 0; q < w.nu; ++q) {
    W2Unit& U = w.u[q];
    const WgProb& P = w.p[U.prob];
    if (P.fin_kind && (U.mp0 != 0 || U.pa != P.mt)) return false;
    U.fin_rows = U.red_rows + (P.fin_kind == 1 ? 2 * U.pb : 0);
    U.fin_blk = blk;
    const int nb = (U.fin_rows + kRedRows - 1) / kRedRows;
    blk += nb;
    bx = std::max(bx, nb);
  }
  const int nzb = (zc + 2047) / 2048;
  w.fin_blocks_x = std::max(bx, nzb + 1);
  w.fin_live_blocks = blk + nzb + 1;
  w.fin = 1;
  return true;
}

int w2_fin_blocks(const W2Table& w, const FinMisc&) { return w.fin ? w.fin_live_blocks : 0; }

int launch_wgrad2_fin(const W2Table& w0, float* raw, const float* theta, float* grad, float* gsq_part, const FinMisc& misc, hipStream_t st) {
  if (!w0.fin || !theta || !grad || !gsq_part) return OPE_EINVAL;
  W2Table w = w0;
  for (int q = 0; q < w.np; ++q)
    if (w.p[q].fin_kind == 2) w.p[q].fin_kind = 1;      // (the kernels' transform is the same; the unit's fin_rows say whether the partial vectors are stored)
  int rc = launch_wgrad2_impl(w, raw, nullptr, theta, st);
  if (rc) return rc;
  kprof_work(0.0, 4.0 * (double)w.total_wg * kW2Slab);
  OPE_LAUNCH(w2_fin_kernel, dim3(w.fin_blocks_x, w.nu + 1), dim3(512), 0, st, w, raw, grad, gsq_part, misc);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("w2_fin", 0);
  return OPE_OK;
}

int w2_max_workgroups() {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 1 ? n : 256;
  }();
  return cus;
}

// Can the register-blocked kernel take this table? (whole launch on one path: vector width 4 or 2, no rows read in place)
bool w2_ok(const WgTable& tb) {
  if (tb.n < 1 || tb.n > kMaxWgProbs) return false;
  for (int q = 0; q < tb.n; ++q) {
    const WgProb& P = tb.p[q];
    if (P.ref_row1 > 0) return false;
    const bool v4 = P.lda % 4 == 0 && P.ldb % 4 == 0 && !((uintptr_t)P.A & 15) && !((uintptr_t)P.B & 15);
    const bool v2 = P.lda % 2 == 0 && P.ldb % 2 == 0 && P.lda >= 4 && P.ldb >= 4 && !((uintptr_t)P.A & 7) && !((uintptr_t)P.B & 7);
    if (!v4 && !v2) return false;
  }
  return true;
}

// 16-byte loads when every operand row of the table allows them, 8-byte ones otherwise (w2_ok has checked that much)
static int w2_vec(const WgTable& tb) {
  for (int q = 0; q < tb.n; ++q) {
    const WgProb& P = tb.p[q];
    if (!(P.lda % 4 == 0 && P.ldb % 4 == 0 && !((uintptr_t)P.A & 15) && !((uintptr_t)P.B & 15) && (!P.A2 || (P.lda2 % 4 == 0 && !((uintptr_t)P.A2 & 15))))) return 2;
  }
  return 4;
}

// Units and their workgroups. `tb` must have gone through wg_finish (mt / nt). Returns OPE_EINVAL when the table does not fit.
int w2_build(const WgTable& tb, W2Table* out) {
  W2Table& w = *out;
  memset(&w, 0, sizeof(w));
  w.np = tb.n;
  for (int q = 0; q < tb.n; ++q) w.p[q] = tb.p[q];
  auto cut = [](int n, int cap, int* sizes) {      // n panels into ceil(n / cap) blocks of near-equal size
    const int nb = ope_cdiv(n, cap);
    for (int q = 0; q < nb; ++q) sizes[q] = n / nb + (q < n % nb ? 1 : 0);
    return nb;
  };
  for (int q = 0; q < tb.n; ++q) {
    const WgProb& P = tb.p[q];
    int ms[64], ns[64];
    if (P.mt > 64 || P.nt > 64) return OPE_EINVAL;
    const int capm = P.nt == 1 ? 4 : (P.mt == 1 ? 1 : 2), capn = P.mt == 1 ? 4 : (P.nt == 1 ? 1 : 2);
    const int nbm = cut(P.mt, capm, ms), nbn = cut(P.nt, capn, ns);
    int m0 = 0;
    for (int bm = 0; bm < nbm; ++bm) {
      int n0 = 0;
      for (int bn = 0; bn < nbn; ++bn) {
        if (w.nu >= kMaxW2Units) return OPE_EINVAL;
        W2Unit& U = w.u[w.nu++];
        U.prob = q; U.mp0 = (short)m0; U.np0 = (short)n0; U.pa = (short)ms[bm]; U.pb = (short)ns[bn];
        U.nwg = 1;
        n0 += ns[bn];
      }
      m0 += ms[bm];
    }
  }
  // workgroups: greedy on the per-wave cost tiles x (rows per wave + a fixed prologue), one more workgroup at a time to the unit
  // whose waves are longest; a wave keeps at least 32 rows
  const int cap = w2_max_workgroups();
  const int ring = w2_ring(w2_vec(tb));
  auto rows_of = [&](const W2Unit& U) { const int gr = (U.pa * U.pb == 1 ? 8 : 4) * ring; return gr * ope_cdiv(ope_cdiv(w.p[U.prob].K, 4 * U.nwg), gr); };      // (whole rotations of the operand ring)
  auto cost_of = [&](const W2Unit& U) { return (int64_t)U.pa * U.pb * (rows_of(U) + 16); };
  if (w.nu > cap) return OPE_EINVAL;
  bool frozen[kMaxW2Units] = {};
  auto greedy = [&]() {      // hands out the workgroups that are left; returns the longest wave's cost
    int total = 0;
    for (int q = 0; q < w.nu; ++q) total += w.u[q].nwg;
    while (total < cap) {
      int best = -1;
      int64_t bc = -1;
      for (int q = 0; q < w.nu; ++q) {
        const W2Unit& U = w.u[q];
        if (frozen[q] || rows_of(U) <= 8 * ring) continue;
        const int64_t c = cost_of(U);
        if (c > bc) { bc = c; best = q; }
      }
      if (best < 0) break;
      ++w.u[best].nwg;
      ++total;
    }
    int64_t mx = 0;
    for (int q = 0; q < w.nu; ++q) mx = std::max(mx, cost_of(w.u[q]));
    return mx;
  };
  greedy();
  // Two units that read the same A rows (W_ih and W_hh both walk dgi): give them the same number of workgroups, a multiple of 8, and put
  // them first, one behind the other -- workgroup c of either then reduces the same rows at the same time on the same XCD (block b runs on
  // XCD b % 8: a placement for speed only), and the second reader finds dgi's rows in that XCD's L2 instead of fetching 30 MB again.
  int pa_ = -1, pb_ = -1;
  for (int q = 0; q < w.nu && pa_ < 0; ++q)
    for (int r = q + 1; r < w.nu; ++r) {
      const WgProb &A = w.p[w.u[q].prob], &B = w.p[w.u[r].prob];
      if (A.A == B.A && A.K == B.K && A.lda == B.lda && w.u[q].mp0 == w.u[r].mp0 && w.u[q].pa == w.u[r].pa && w.u[q].pb == w.u[r].pb &&
          w.u[q].nwg >= 8 && w.u[r].nwg >= 8) { pa_ = q; pb_ = r; break; }
    }
  if (pa_ >= 0) {
    int saved[kMaxW2Units];
    for (int q = 0; q < w.nu; ++q) saved[q] = w.u[q].nwg;
    const int lo = std::min(saved[pa_], saved[pb_]) / 8 * 8;
    int64_t best_cost = -1;
    int best_n = lo;
    for (int n = lo; n <= lo + 8; n += 8) {      // the multiple of 8 below and the one above what the greedy pass gave them
      for (int q = 0; q < w.nu; ++q) w.u[q].nwg = 1;
      w.u[pa_].nwg = w.u[pb_].nwg = n;
      frozen[pa_] = frozen[pb_] = true;
      int rest = 0;
      for (int q = 0; q < w.nu; ++q) rest += w.u[q].nwg;
      if (rest > cap) continue;
      const int64_t c = greedy();
      if (best_cost < 0 || c < best_cost) { best_cost = c; best_n = n; }
    }
    for (int q = 0; q < w.nu; ++q) w.u[q].nwg = 1;
    w.u[pa_].nwg = w.u[pb_].nwg = best_n;
    frozen[pa_] = frozen[pb_] = true;
    greedy();
    // order: the pair first
    W2Unit tmp[kMaxW2Units];
    int k = 0;
    tmp[k++] = w.u[pa_];
    tmp[k++] = w.u[pb_];
    for (int q = 0; q < w.nu; ++q)
      if (q != pa_ && q != pb_) tmp[k++] = w.u[q];
    for (int q = 0; q < w.nu; ++q) w.u[q] = tmp[q];
  }
  int wg = 0, rb = 0;
  for (int q = 0; q < w.nu; ++q) {
    W2Unit& U = w.u[q];
    U.kchunk = rows_of(U);
    U.wg_begin = wg;
    wg += U.nwg;
    U.red_rows = U.pa * U.pb * 64 + ((w.p[U.prob].s_off >= 0 && U.np0 == 0) ? U.pa : 0);
    rb = std::max(rb, U.red_rows);
  }
  for (int q = 0; q < kMaxW2Units; ++q) w.ubegin[q] = q < w.nu ? w.u[q].wg_begin : 0x7fffffff;
  w.total_wg = wg;
  w.red_blocks = rb;
  w.vec = w2_vec(tb);
  return OPE_OK;
}

static int launch_wgrad2_impl(const W2Table& w, float* raw, float* rsum, const float* fin_theta, hipStream_t st) {
  if (w.nu < 1 || w.total_wg < 1) return OPE_EINVAL;
  const bool v4 = w.vec == 4;
  bool mapped = false;
  for (int q = 0; q < w.np; ++q) mapped = mapped || w.p[q].K_dev != nullptr;
  for (int q = 0; q < w.np && mapped; ++q)      // a live-plan table: every problem carries the device row count and something readable as its map
    if (!w.p[q].K_dev || !w.p[q].b_map || w.p[q].b_shift != 0) return OPE_EINVAL;
  if (g_kprof_on) {
    double fl = 0;
    for (int q = 0; q < w.np; ++q) fl += 2.0 * w.p[q].M * (double)w.p[q].N * w.p[q].K;
    kprof_work(fl);
  }
#ifdef OPE_EXPERIMENTS
  static const int exp_env = getenv("OPE_W2_EXP") ? atoi(getenv("OPE_W2_EXP")) : 0;
  if (exp_env && v4 && !mapped) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "libope: OPE_W2_EXP=%d -- a timing-only variant of wgrad2_kernel runs: the gradients are WRONG\n", exp_env); warned = true; }
    switch (exp_env) {
      case 1: OPE_LAUNCH((wgrad2_kernel<4, 1>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 2: OPE_LAUNCH((wgrad2_kernel<4, 2>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 3: OPE_LAUNCH((wgrad2_kernel<4, 3>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 4: OPE_LAUNCH((wgrad2_kernel<4, 4>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 6: OPE_LAUNCH((wgrad2_kernel<4, 6>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 7: OPE_LAUNCH((wgrad2_kernel<4, 7>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 15: OPE_LAUNCH((wgrad2_kernel<4, 15>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 31: OPE_LAUNCH((wgrad2_kernel<4, 31>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 63: OPE_LAUNCH((wgrad2_kernel<4, 63>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      case 39: OPE_LAUNCH((wgrad2_kernel<4, 39>), dim3(w.total_wg), dim3(256), 0, st, w, raw, (const float*)nullptr); break;
      default: return OPE_EINVAL;
    }
  } else
#endif
  if (mapped) kprof_rows(2);      // (the agent problems' count; the mixer problems' live share is the same to within a row per episode)
  const float* theta = fin_theta;      // (null: the slabs hold the raw products, w2_reduce + finalize follow)
  if (mapped) {
    if (v4) OPE_LAUNCH((wgrad2_kernel<4, 0, true>), dim3(w.total_wg), dim3(256), 0, st, w, raw, theta);
    else OPE_LAUNCH((wgrad2_kernel<2, 0, true>), dim3(w.total_wg), dim3(256), 0, st, w, raw, theta);
  } else if (v4)
    OPE_LAUNCH((wgrad2_kernel<4>), dim3(w.total_wg), dim3(256), 0, st, w, raw, theta);
  else
    OPE_LAUNCH((wgrad2_kernel<2>), dim3(w.total_wg), dim3(256), 0, st, w, raw, theta);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(mapped ? "wgrad2_live" : "wgrad2", v4 ? 4 : 2);
  if (fin_theta) return OPE_OK;      // (launch_wgrad2_fin sums the slabs itself)
  kprof_work(0.0, 4.0 * (double)w.total_wg * kW2Slab);
  OPE_LAUNCH(w2_reduce_kernel, dim3((w.red_blocks + kRedRows - 1) / kRedRows, w.nu), dim3(512), 0, st, w, raw, rsum);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int launch_wgrad2(const W2Table& w, float* raw, float* rsum, hipStream_t st) { return launch_wgrad2_impl(w, raw, rsum, nullptr, st); }

}  // namespace ope
