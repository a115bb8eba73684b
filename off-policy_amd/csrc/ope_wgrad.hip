// Weight-gradient reductions:  C[M][N] = sum_k A[k][m] * B[k][n]   (K = number of data rows, 10^3..10^5)
//
// The reference obtains these from autograd's Linear backward (grad_W = grad_out^T @ input), one aten::mm per
// layer. Here ALL of a trainer's weight gradients are one batched launch: a table of problems, each cut into
// 64x64 output tiles x NS K-splits; one wave computes one (tile, split) with 16 f32-MFMA accumulators, reading
// its operands straight from global memory (each operand element feeds 4 MFMAs from registers, so no LDS stage).
// Split partials go to raw[split][...] and are summed in fixed order afterwards (deterministic, no atomics).
//
// Options per problem: column-sum of A (bias gradients) written next to C; LayerNorm-on-load of B
// (xhat = (B - mu[k]) * rstd[k], for the input feature-norm whose xhat is never materialised); row shift of B
// (B[k - shift], zero for k < shift: the h_{t-1} operand of dW_hh).
#include "ope_wgrad.h"

namespace ope {

// Column mapping inside a 64x64 tile: MFMA (mi, ni) computes the 16x16 block whose rows are m = m0 + 4*rho + mi and
// columns n = n0 + 4*kappa + ni (rho, kappa = MFMA row / column index). With that permutation lane (i, g) needs, for
// reduction row k = kb+g, the FOUR CONSECUTIVE floats A[k][m0+4i .. +3] and B[k][n0+4i .. +3]: one 16-byte load per
// operand per step (16 lanes x 16 B = two full 128-B lines per k-row), and it ends up holding, for each (mi, r),
// the four consecutive outputs C[m0+16g+4r+mi][n0+4i .. +3]: float4 stores.
// VEC = 4: every problem's lda / ldb is a multiple of 4 (rows 16-byte aligned, one 16-byte load per operand per step);
// VEC = 2: multiples of 2 (two 8-byte loads); VEC = 1: four scalar loads.
// LAZY (VEC = 4 only): problems with ref_row1 > 0 read their B rows (observations) in place from the episode-major store: reduction row
// k = batch row tn * B + b is store row ep[b] * TTN + tn, same rows in the same order as the gathered launch (bit-identical sums). The
// sampled episode slots sit in LDS; the decode is branch-free (a select between the plain and the decoded row index), so the other
// problems of the launch only pay ~10 VALU instructions per fetch. (Walking K episode-major instead -- four contiguous rows of one
// episode per fetch -- was measured: no faster, 65.8 vs 64.3 us at 3s5z; what the row-reading launch loses against the gathered one,
// 58.6 us, is where the rows come from: the gather's freshly written batch sits in the 256 MB Infinity Cache, the store's rows do not.)
// EXP (timing experiments, OPE_WGRAD_EXP, results wrong): 1 / 2 = the A / B rows come from a 64-row window (the caches), 4 = the MFMAs are left out (operands kept alive), 16 = the VALU work beside them is
// (masks, LayerNorm-on-load, column sums), 4 + 8 = also without the two per-row scalar loads, 4 + 32 = no K loop at all (the launch's ramp), 4 + 8 + 64 = only the A operand is loaded:
// what each part of the kernel costs on its own. EXP = 0 is the kernel.
template <int VEC, bool LAZY, int EXP = 0>
__global__ void __launch_bounds__(256, 3) wgrad_kernel(WgTable tb, float* __restrict__ raw) {
  __shared__ __attribute__((aligned(16))) float red[2][17][64][4];   // two partial-tile slots: [quad][lane][4]
  __shared__ int eps[LAZY ? kObsRefMaxB : 1];
  if (LAZY) {
    for (int q = threadIdx.x; q < tb.ref.B; q += 256) eps[q] = obs_ref_slot(tb.ref, tb.ref.inds[q]);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the problem / tile / split indices derived from it live in SGPRs
  const int i = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * 4 + wave;
  if (wid >= tb.total_waves) return;
  int p = 0;
#pragma unroll
  for (int q = 1; q < kMaxWgProbs; ++q)      // (a loop over p[q].wave_begin is one dependent scalar load per problem: ~1 us for 13)
    if (wid >= tb.wbegin[q]) p = q;
  const WgProb& P = tb.p[p];
  const int local = wid - P.wave_begin;
  const int split = local % P.nsplit;
  const int tile = local / P.nsplit;
  const int tn = tile % P.nt, tm = tile / P.nt;
  const int m0 = 64 * tm, n0 = 64 * tn;
  const int k0 = split * P.kchunk;
  const int k1 = min(P.K, k0 + P.kchunk);
  const int lda = P.lda, ldb = P.ldb, shift = P.b_shift, Kmax = P.K - 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};   // column sums of A for m = m0 + 4i + mi (this lane's k-rows only)
  // 0/1 masks of the four columns this lane feeds, and in-row clamped base offsets (loads are unconditional)
  f32x4 mokf, nokf;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    mokf[q] = (m0 + 4 * i + q < P.M) ? 1.f : 0.f;
    nokf[q] = (n0 + 4 * i + q < P.N) ? 1.f : 0.f;
  }
  // vector loads are clamped so that they stay inside the row (lda/ldb >= 4 whenever VEC > 1 is selected)
  // (VEC = 4: a group of 4 columns is either entirely inside the row or entirely masked; VEC = 2: each 2-column half is)
  const int moff = VEC == 4 ? min(m0 + 4 * i, lda - 4) : (VEC == 2 ? min(m0 + 4 * i, lda - 2) : m0 + 4 * i);
  const int noff = VEC == 4 ? min(n0 + 4 * i, ldb - 4) : (VEC == 2 ? min(n0 + 4 * i, ldb - 2) : n0 + 4 * i);
  const int moff1 = min(m0 + 4 * i + 2, lda - 2), noff1 = min(n0 + 4 * i + 2, ldb - 2);   // second halves (VEC = 2)
  const float* __restrict__ Ap = P.A;
  const float* __restrict__ Bp = P.B;
  const float* __restrict__ mup = P.ln_mu;     // never null: plain problems point at a zeros / ones vector
  const float* __restrict__ rsp = P.ln_rstd;
  const bool ref_on = LAZY && P.ref_row1 > 0;
  const float inv_b = 1.0f / (float)(LAZY ? tb.ref.B : 1);
  const int ref_tn0 = ref_on ? (P.ref_row1 - 1) / tb.ref.B : 0;      // (the problem's row range starts on a whole time step)

  struct Rawv { f32x4 a, b; float mu, rs; };
  auto fetch = [&](int kb, Rawv& r) {
    const int kc = min(kb + g, Kmax);
    const int kr = max(kc - shift, 0);
    int64_t brow = kr;
    if (LAZY) {
      const int tn = div_small(kr, inv_b), b = kr - tn * tb.ref.B;
      const int64_t sr = (int64_t)eps[b] * tb.ref.TTN + (ref_tn0 + tn);
      brow = ref_on ? sr : brow;
    }
    const float* ar = Ap + (int64_t)((EXP & 1) ? (kc & 63) : kc) * lda;
    const float* br = Bp + ((EXP & 2) ? (brow & 63) : brow) * ldb;
    if (VEC == 4) {
      r.a = *reinterpret_cast<const f32x4*>(ar + moff);
      if (EXP & 64) r.b = f32x4{1.f, 1.f, 1.f, 1.f};      // (experiment: half the operand bytes through the vector-memory path)
      else r.b = *reinterpret_cast<const f32x4*>(br + noff);
    } else if (VEC == 2) {
      const f32x2 a0 = *reinterpret_cast<const f32x2*>(ar + moff), a1 = *reinterpret_cast<const f32x2*>(ar + moff1);
      const f32x2 b0 = *reinterpret_cast<const f32x2*>(br + noff), b1 = *reinterpret_cast<const f32x2*>(br + noff1);
      r.a = f32x4{a0[0], a0[1], a1[0], a1[1]};
      r.b = f32x4{b0[0], b0[1], b1[0], b1[1]};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r.a[q] = ar[min(moff + q, P.M - 1)];
        r.b[q] = br[min(noff + q, P.N - 1)];
      }
    }
    if (EXP & 8) {      // (experiment: without the two per-row scalar loads)
      r.mu = 0.f;
      r.rs = 1.f;
    } else {
      r.mu = mup[kr];
      r.rs = rsp[kr];
    }
  };
  auto compute = [&](const Rawv& c, int kb) {
    const int k = kb + g;
    // zero invalid rows / columns by multiplying with 0/1 masks (selects get turned back into guarded loads)
    const float ka = (k < k1) ? 1.f : 0.f;
    const float kbm = (k < k1 && k - shift >= 0) ? 1.f : 0.f;
    f32x4 av, bv;
    if (EXP & 16) {
      av = c.a;
      bv = c.b;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        av[q] = c.a[q] * (ka * mokf[q]);
        bv[q] = ((c.b[q] - c.mu) * c.rs) * (kbm * nokf[q]);
      }
      cs += av;
    }
    if (EXP & 4) {
      asm volatile("" ::"v"(av), "v"(bv));
      return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(av[mi], bv[ni], acc[mi][ni]);
  };
  // Four-deep ring of operand buffers, loop unrolled by four: three fetches (12 loads per lane) stay in flight behind
  // every MFMA group. With ~2 waves per SIMD at these problem sizes the loop is bound by memory latency, not by the
  // matrix pipe, so prefetch depth is what sets the speed. The sched_barriers pin each fetch BEFORE the MFMAs of the
  // buffer being consumed (the compiler otherwise sinks the loads next to their uses).
  Rawv b0, b1, b2, b3;
  fetch(k0, b0);
  fetch(k0 + 4, b1);
  fetch(k0 + 8, b2);
  for (int kb = k0; kb < ((EXP & 32) ? k0 : k1); kb += 16) {      // (experiment 32: no loop at all -- the launch's ramp)
    fetch(kb + 12, b3);
    __builtin_amdgcn_sched_barrier(0);
    compute(b0, kb);
    __builtin_amdgcn_sched_barrier(0);
    fetch(kb + 16, b0);
    __builtin_amdgcn_sched_barrier(0);
    compute(b1, kb + 4);   // rows >= k1 are masked to zero
    __builtin_amdgcn_sched_barrier(0);
    fetch(kb + 20, b1);
    __builtin_amdgcn_sched_barrier(0);
    compute(b2, kb + 8);
    __builtin_amdgcn_sched_barrier(0);
    fetch(kb + 24, b2);
    __builtin_amdgcn_sched_barrier(0);
    compute(b3, kb + 12);
    __builtin_amdgcn_sched_barrier(0);
  }

  // acc[mi][ni][r] = C[m0 + 16g + 4r + mi][n0 + 4i + ni]  ->  per (mi, r) one float4 over ni
  // The 4 waves of a workgroup hold 4 consecutive K-splits of the same tile (nsplit % 4 == 0 for every problem when
  // tb.wg_reduce): sum them here in a fixed order and write one slab instead of four.
  int slab = split;
  if (tb.wg_reduce) {
    // two-level tree through a 2-slot buffer (35 KB of LDS instead of 52 KB: one more workgroup per CU):
    //   waves 2,3 publish | waves 0,1 add (w0 += w2, w1 += w3) | wave 1 publishes | wave 0 adds, writes the slab
    auto publish = [&](int slot) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<f32x4*>(red[slot][mi * 4 + r][lane]) = f32x4{acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
      *reinterpret_cast<f32x4*>(red[slot][16][lane]) = cs;
    };
    auto absorb = [&](int slot) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(red[slot][mi * 4 + r][lane]);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni][r] += v[ni];
        }
      cs += *reinterpret_cast<const f32x4*>(red[slot][16][lane]);
    };
    if (wave >= 2) publish(wave - 2);
    __syncthreads();
    if (wave < 2) absorb(wave);
    __syncthreads();
    if (wave == 1) publish(0);
    __syncthreads();
    if (wave > 0) return;
    absorb(0);
    slab = split >> 2;
  }
  float* out = raw + P.raw_base + (int64_t)slab * P.raw_stride;
  const int nb = n0 + 4 * i;
  const bool n4 = VEC == 4 && (P.ldc % 4 == 0) && (P.out_off % 4 == 0) && (nb + 3 < P.N);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * g + 4 * r + mi;
      if (m < P.M) {
        float* o = out + P.out_off + (int64_t)m * P.ldc + nb;
        if (n4) {
          *reinterpret_cast<f32x4*>(o) = f32x4{acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
        } else {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            if (nb + ni < P.N) o[ni] = acc[mi][ni][r];
        }
      }
    }
  if (P.s_off >= 0 && tn == 0) {
    // column sum for m = m0 + 4i + q: add the four k-row groups g
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float sv = rowsum4(cs[q]);
      const int m = m0 + 4 * i + q;
      if (g == 0 && m < P.M) out[P.s_off + m] = sv;
    }
  }
}

int wg_finish(WgTable* tb) {
  int waves = 0;
  for (int q = 0; q < tb->n; ++q) {
    WgProb& P = tb->p[q];
    if (P.M < 1 || P.N < 1 || P.K < 1 || P.nsplit < 1) return OPE_EINVAL;
    P.mt = ope_cdiv(P.M, 64);
    P.nt = ope_cdiv(P.N, 64);
    P.kchunk = 4 * ope_cdiv(ope_cdiv(P.K, P.nsplit), 4);
    P.wave_begin = waves;
    waves += P.mt * P.nt * P.nsplit;
    if (P.ref_row1 > 0 && (!tb->ref.inds || tb->ref.B < 1 || tb->ref.B > kObsRefMaxB || P.K > kObsRefMaxRows || P.K % tb->ref.B != 0 ||
                           (P.ref_row1 - 1) % tb->ref.B != 0 || P.b_shift != 0))
      return OPE_EINVAL;
  }
  tb->total_waves = waves;
  for (int q = 0; q < kMaxWgProbs; ++q) tb->wbegin[q] = q < tb->n ? tb->p[q].wave_begin : 0x7fffffff;
  tb->wg_reduce = 1;
  for (int q = 0; q < tb->n; ++q)
    if (tb->p[q].nsplit % 4 != 0) tb->wg_reduce = 0;
  return OPE_OK;
}
// slabs actually written per problem after the optional in-workgroup reduction
int wg_slabs(const WgTable& tb, int nsplit) { return tb.wg_reduce ? nsplit / 4 : nsplit; }

int launch_wgrad(const WgTable& tb, float* raw, hipStream_t st) {
  if (tb.n < 1 || tb.total_waves < 1) return OPE_EINVAL;
  int vec = 4;
  for (int q = 0; q < tb.n; ++q) {
    const WgProb& P = tb.p[q];
    int v = 1;
    if (P.lda % 4 == 0 && P.ldb % 4 == 0 && !((uintptr_t)P.A & 15) && !((uintptr_t)P.B & 15)) v = 4;
    else if (P.lda % 2 == 0 && P.ldb % 2 == 0 && P.lda >= 4 && P.ldb >= 4 && !((uintptr_t)P.A & 7) && !((uintptr_t)P.B & 7)) v = 2;
    if (v < vec) vec = v;
  }
  if (g_kprof_on) {
    double fl = 0;
    for (int q = 0; q < tb.n; ++q) fl += 2.0 * tb.p[q].M * (double)tb.p[q].N * tb.p[q].K;
    kprof_work(fl);
  }
  bool lazy = false;
  for (int q = 0; q < tb.n; ++q) lazy = lazy || tb.p[q].ref_row1 > 0;
#ifdef OPE_EXPERIMENTS
  // (only in libope_exp.so, built with -DOPE_EXPERIMENTS: the release library carries no wrong-result instantiation)
  // where does the time go -- the matrix pipe or the operand traffic? OPE_WGRAD_EXP serves the A (bit 0) and / or B (bit 1) rows from a 64-row
  // window, i.e. from the caches. The results are WRONG: honoured only while the in-process kernel timer is on (bench.py's per-kernel table),
  // never for a plain training call, and announced on stderr.
  static const int exp_env = getenv("OPE_WGRAD_EXP") ? atoi(getenv("OPE_WGRAD_EXP")) : 0;
  if (exp_env && g_kprof_on && !lazy && vec == 4) {
    static bool warned = false;
    if (!warned) { fprintf(stderr, "libope: OPE_WGRAD_EXP=%d -- a timing-only variant of wgrad_kernel runs: the gradients are WRONG\n", exp_env); warned = true; }
    const dim3 grid(ope_cdiv(tb.total_waves, 4));
#define OPE_WG_EXP(E) case E: OPE_LAUNCH((wgrad_kernel<4, false, E>), grid, dim3(256), 0, st, tb, raw); break
    switch (exp_env) {
      OPE_WG_EXP(1); OPE_WG_EXP(2); OPE_WG_EXP(3); OPE_WG_EXP(4); OPE_WG_EXP(12); OPE_WG_EXP(16); OPE_WG_EXP(20); OPE_WG_EXP(36); OPE_WG_EXP(76);
      default: return OPE_EINVAL;
    }
#undef OPE_WG_EXP
    if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
    note_launch("wgrad", vec);
    return OPE_OK;
  }
#endif
  if (lazy)
    OPE_LAUNCH((wgrad_kernel<4, true>), dim3(ope_cdiv(tb.total_waves, 4)), dim3(256), 0, st, tb, raw);
  else if (vec == 4)
    OPE_LAUNCH((wgrad_kernel<4, false>), dim3(ope_cdiv(tb.total_waves, 4)), dim3(256), 0, st, tb, raw);
  else if (vec == 2)
    OPE_LAUNCH((wgrad_kernel<2, false>), dim3(ope_cdiv(tb.total_waves, 4)), dim3(256), 0, st, tb, raw);
  else
    OPE_LAUNCH((wgrad_kernel<1, false>), dim3(ope_cdiv(tb.total_waves, 4)), dim3(256), 0, st, tb, raw);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch(lazy ? "wgrad_store" : "wgrad", vec);
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// rsum[i] = sum_s raw[s][i]  (fixed order) for two slab regions (agent, mixer) in one launch.
// ---------------------------------------------------------------------------------------------------------
__global__ void split_reduce_kernel(SplitRed a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n0 + a.n1) return;
  const bool second = i >= a.n0;
  const float* raw = second ? a.raw1 : a.raw0;
  const int64_t j = second ? i - a.n0 : i, stride = second ? a.n1 : a.n0;
  const int ns = second ? a.ns1 : a.ns0;
  float s = 0.f;
#pragma unroll 8
  for (int q = 0; q < ns; ++q) s += raw[(int64_t)q * stride + j];
  a.rsum[i] = s;
}

int launch_split_reduce(const SplitRed& a, hipStream_t st) {
  kprof_work(0.0, 4.0 * ((double)a.n0 * (a.ns0 + 1) + (double)a.n1 * (a.ns1 + 1)));
  OPE_LAUNCH(split_reduce_kernel, dim3(ope_cdiv(a.n0 + a.n1, 256)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// finalize: map reduced raw quantities to the flat gradient vector.
//   COPY     grad = rsum[src + local]
//   LNLIN_W  Linear fed by a LayerNorm: dW[i][k] = P[i][k] gamma[k] + s[i] beta[k],  P = dOut^T xhat, s = colsum(dOut)
//   LNLIN_G  that LayerNorm's weight:   dgamma[k] = sum_i W[i][k] P[i][k]
//   LNLIN_B  that LayerNorm's bias:     dbeta[k]  = sum_i s[i] W[i][k]
//   ZERO     registered-but-unused tensors (fc_h) and padding
//   TAIL     [loss_sum, mask_count, qtot_sum, 0] summed over the per-tile partials
//   SKIP     left untouched (another call of a phased multi-policy step owns that block)
// ---------------------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave (fixed order: DPP row sums, then the four rows through readlane), result uniform.
__device__ __forceinline__ float wave64_sum(float v) {
  v = row16_sum(v);
  return (__shfl(v, 0, 64) + __shfl(v, 16, 64)) + (__shfl(v, 32, 64) + __shfl(v, 48, 64));
}

// gsq_part (optional): per-workgroup partial sums of grad[i]^2 over the elements THIS workgroup wrote (tail excluded), one
// float per workgroup of the launch in blockIdx order. With them ope_adam_step needs no separate norm pass over the gradient
// (single-GPU path only: a multi-GPU run all-reduces the gradient between this kernel and the optimizer).
__global__ void __launch_bounds__(256) finalize_kernel(FinTable ft, const float* __restrict__ rsum, const float* __restrict__ theta,
                                                       const float* __restrict__ loss_part, int n_loss_tiles, float* __restrict__ grad,
                                                       int n_main, float* __restrict__ gsq_part, int n_gsq_total) {
  __shared__ float sq[4];
  if ((int)blockIdx.x > n_main) {
    // Column reductions of the LayerNorm-fed Linears (LNLIN_G / LNLIN_B: M = 14 .. 192 rows, two loads per row). Workgroup = 16
    // consecutive elements (columns k) of ONE segment: thread (c, rg) = (tid & 15, tid >> 4) takes column k0 + c and rows rg, rg + 16, ...
    // -- a wave's load touches 4 rows x 64 contiguous bytes (round 5; one wave per element with lane = row read 64 lines per load:
    // these blocks were 3.3 of the launch's 7.8 us) -- all of a thread's loads (<= 12 per array per pass) issue before the first FMA, the
    // 16 partial sums of a column meet by two lane swaps and one LDS exchange. Fixed order.
    __shared__ float red[4][16];
    int blk = (int)blockIdx.x - n_main - 1, s = -1, k0 = 0;
    if (ft.ncb > 0) {
      if (blk < ft.ncb) { const int e = ft.cb[blk]; s = e & 63; k0 = 16 * (e >> 6); }
    } else {
#pragma unroll 1
      for (int q = 0; q < ft.n; ++q) {
        const int kind = ft.seg[q].kind;
        if (kind != FIN_LNLIN_G && kind != FIN_LNLIN_B) continue;
        const int nb = (ft.seg[q].size + 15) >> 4;
        if (s < 0) {
          if (blk < nb) { s = q; k0 = 16 * blk; }
          else blk -= nb;
        }
      }
    }
    const bool live = s >= 0;
    const FinSeg& F = ft.seg[live ? s : 0];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int k = k0 + c;
    const int M = live ? F.M : 0, K = F.K;
    const bool colsum = F.kind == FIN_LNLIN_B;
    const bool ok = live && k < F.size;
    float acc = 0.f;
    if (ok) {
      for (int i0 = 0; i0 < M; i0 += 192) {
        float wv[12], pv[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
          const int i = min(i0 + rg + 16 * u, M - 1);
          wv[u] = theta[F.w + (int64_t)i * K + k];
          pv[u] = colsum ? rsum[F.src_s + i] : rsum[F.src + (int64_t)i * K + k];
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) acc = (i0 + rg + 16 * u < M) ? fmaf(wv[u], pv[u], acc) : acc;
      }
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    const int wv_ = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 16) red[wv_][c] = acc;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x < 16) {
      tot = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
      if (ok) grad[F.begin + k] = tot;
      else tot = 0.f;
    }
    if (gsq_part) {
      float v = threadIdx.x < 16 ? tot * tot : 0.f;
      if (threadIdx.x < 64) v = wave64_sum(v);
      if (threadIdx.x == 0) gsq_part[blockIdx.x] = v;
    }
    return;
  }
  if ((int)blockIdx.x == n_main) {
    // extra block: [loss_sum, mask_count, qtot_sum, 0] = fixed-order strided sums over the per-tile partials, then a tree
    __shared__ float red[256][3];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int q = threadIdx.x; q < n_loss_tiles; q += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(loss_part + (int64_t)q * 4);
      a0 += v[0]; a1 += v[1]; a2 += v[2];
    }
    red[threadIdx.x][0] = a0; red[threadIdx.x][1] = a1; red[threadIdx.x][2] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o)
        for (int c = 0; c < 3; ++c) red[threadIdx.x][c] += red[threadIdx.x + o][c];
      __syncthreads();
    }
    if (n_loss_tiles >= 0 && threadIdx.x < 4) grad[ft.total - OPE_GRAD_TAIL + threadIdx.x] = threadIdx.x < 3 ? red[0][threadIdx.x] : 0.f;
    if (gsq_part && threadIdx.x == 0) gsq_part[blockIdx.x] = 0.f;   // the tail is not part of the gradient norm
    // (a step that took launch_wgrad2_fin on this workspace before left more partials than this launch has workgroups)
    if (gsq_part)
      for (int q = (int)gridDim.x + (int)threadIdx.x; q < n_gsq_total; q += 256) gsq_part[q] = 0.f;
    return;
  }
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (the segment table lives in kernel-argument memory: its starts are compared from one burst of scalar loads, and the
  // record is copied whole, so that the thread pays two memory round trips -- record, then data -- instead of one per field)
  int s = 0;
#pragma unroll
  for (int q = 1; q < kMaxFinSegs; ++q)
    if (idx >= ft.begin[q]) s = q;
  const FinSeg F = ft.seg[s];
  const int local = (int)(idx - F.begin);
  float out = 0.f;
  bool write = idx < ft.total;
  if (write && local < F.size) {
    switch (F.kind) {
      case FIN_COPY:
        out = rsum[F.src + local];
        break;
      case FIN_LNLIN_W: {
        const int i = local / F.K, k = local - i * F.K;
        out = rsum[F.src + local] * theta[F.gamma + k] + rsum[F.src_s + i] * theta[F.beta + k];
        break;
      }
      case FIN_LNLIN_G:
      case FIN_LNLIN_B:
        write = false;             // written by the reduction blocks above
        break;
      case FIN_TAIL:
        write = false;             // written by the extra block
        break;
      case FIN_SKIP:
        write = false;             // owned by another call (phased multi-policy steps)
        break;
      default:
        out = 0.f;
    }
  }
  if (write) grad[idx] = out;
  if (gsq_part) {
    float v = write ? out * out : 0.f;
    v = wave64_sum(v);
    if ((threadIdx.x & 63) == 0) sq[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) gsq_part[blockIdx.x] = (sq[0] + sq[1]) + (sq[2] + sq[3]);
  }
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int launch_fill(float* p, int64_t n, float v, hipStream_t st) {
  OPE_LAUNCH(fill_kernel, dim3(ope_cdiv(n, 256)), dim3(256), 0, st, p, n, v);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// dst[c][r] = src[r][c] for up to 4 matrices in one launch
__global__ void transpose4_kernel(Transp4 a) { transpose4_element(a, blockIdx.x * blockDim.x + threadIdx.x); }
int launch_transpose4(const Transp4& a, hipStream_t st) {
  OPE_LAUNCH(transpose4_kernel, dim3(ope_cdiv(a.total, 256)), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int finalize_blocks(const FinTable& ft) {
  int64_t n_red = 0;
  for (int q = 0; q < ft.n; ++q)
    if (ft.seg[q].kind == FIN_LNLIN_G || ft.seg[q].kind == FIN_LNLIN_B) n_red += ope_cdiv(ft.seg[q].size, 16);      // one workgroup per 16 columns
  return (int)ope_cdiv(ft.total, 256) + 1 + (int)n_red;
}

int launch_finalize(const FinTable& ft0, const float* rsum, const float* theta, const float* loss_part, int n_loss_tiles,
                    float* grad, hipStream_t st, float* gsq_part, int n_gsq_total) {
  FinTable t = ft0;
  for (int q = 0; q < kMaxFinSegs; ++q) t.begin[q] = q < t.n ? t.seg[q].begin : 0x7fffffff;
  t.ncb = 0;
  {
    int n = 0;
    bool fits = true;
    for (int q = 0; q < t.n && fits; ++q) {
      if (t.seg[q].kind != FIN_LNLIN_G && t.seg[q].kind != FIN_LNLIN_B) continue;
      const int nb = (int)ope_cdiv(t.seg[q].size, 16);
      if (q >= 64 || nb > 1024 || n + nb > kMaxColBlocks) { fits = false; break; }
      for (int b = 0; b < nb; ++b) t.cb[n++] = (unsigned short)(q | (b << 6));
    }
    t.ncb = fits ? n : 0;
  }
  const FinTable& ft = t;
  const int n_main = (int)ope_cdiv(ft.total, 256);
  // blocks [0, n_main): one thread per element; block n_main: the loss tail; the rest: one workgroup per 16 column reductions
  kprof_work(0.0, 4.0 * 3.0 * (double)ft.total);       // rsum + theta in, grad out (plus the small column reductions)
  int blocks = finalize_blocks(ft);
#ifdef OPE_EXPERIMENTS
  // OPE_FIN_EXP (timing only, the gradient is INCOMPLETE): 1 = the element-wise blocks alone, 2 = + the loss-tail block (no column reductions)
  static const int fexp = getenv("OPE_FIN_EXP") ? atoi(getenv("OPE_FIN_EXP")) : 0;
  if (fexp == 1) blocks = n_main;
  if (fexp == 2) blocks = n_main + 1;
#endif
  OPE_LAUNCH(finalize_kernel, dim3(blocks), dim3(256), 0, st, ft, rsum, theta, loss_part, n_loss_tiles, grad, n_main,
                     gsq_part, n_gsq_total);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

}  // namespace ope
