// trunk_fwd, workgroup-cooperative form (default). Same arithmetic as trunk_fwd_kernel (ope_agent_fwd.hip):
//   LN_D(x) -> fc1 + ReLU + LN -> fc2 + ReLU + LN -> gi = W_ih a2 + b_ih      (RNNBase.forward rnn.py:33-47, mlp.py:25-29)
//
// Why a second form: with one wave carrying 16*RT rows through all 64 output features, a 3s5z batch is only ~1200 waves
// -- about one per SIMD -- and each of them is a ~60 us serial chain (PMC: f32-MFMA pipe 22 % busy, the wave stalled or
// waiting the rest of the time). Here the 64 output features of every layer are split over the 4 waves of a workgroup
// (one 16-feature MFMA tile each; 3 tiles each for the 192-wide W_ih), so the same batch is ~4800 much shorter waves and
// the SIMDs have 3-4 of them to interleave:
//   phase 0  each wave normalises TR/4 input rows (two-pass LayerNorm, coalesced row reads) into LDS  xn[TR][Dp]
//   fc1      wave w: z1[:, 16w..16w+15] = W1[16w.., :] xn   -- A operand = weight rows from L2 (4-deep register
//            prefetch ring), B operand = ds_read_b128 from LDS, shared by the 4 waves
//   LN       per-row sums of the 4 waves meet through LDS (two barriers: mean, then centred second moment)
//   fc2, LN, W_ih the same way from an LDS copy of the activations.
// Transposed-chain lane convention as everywhere else: lane (j = l & 15, g = l >> 4) holds features 16*tile + 4g + r of
// data row j.
#include <stdlib.h>

#include "ope_agent.h"

namespace ope {

constexpr int kActPitch = OPE_H + 4;   // LDS row pitch of the 64-wide activations (bank shift of 4 per row)

// MODE 0: the plain trunk. MODE 2: producer of the replicated-rows form -- no normalisation, xn = gamma o x, stops after the
// first layer's product and writes u, s1, s2 per base row. MODE 1: consumer -- phase 0 and the first layer's product are replaced
// by the per-copy correction (RepIn, ope_agent.h), everything after is the plain kernel.
template <int VEC, int RT, bool SAVE, int MODE = 0>
__global__ void __launch_bounds__(256) trunk_fwd2_kernel(TrunkFwdArgs a, int Dp) {
  constexpr int TR = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xn = sm;                          // [TR][Dp]   normalised input (zero beyond D)
  float* actb = sm + TR * Dp;              // [TR][kActPitch]
  float* stat = actb + TR * kActPitch;     // [4][TR]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TR;
  const int D = a.D;
  const int KC = (D + 15) >> 4;
  const float* __restrict__ th = a.theta;

  // ---- phase 0: input LayerNorm of this wave's TR/4 rows -> LDS. All row loads are issued before the first reduction,
  // so the HBM latency is paid once per wave, not once per row. ----
  if constexpr (MODE != 1) {
    constexpr int NI = 8;                  // D <= 512
    constexpr int NR = TR / 4;             // rows per wave
    float v[NR][NI];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int row = row0 + wave * NR + q;
      const float* xr = a.x + (int64_t)(row < a.R ? row : a.R - 1) * D;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int k = lane + 64 * i;
        v[q][i] = xr[k < D ? k : D - 1];
      }
    }
    float gam[NI], bet[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = lane + 64 * i;
      const int kc = k < D ? k : D - 1;
      gam[i] = k < D ? th[a.L.fn_w + kc] : 0.f;
      bet[i] = k < D ? th[a.L.fn_b + kc] : 0.f;
    }
    float mean[NR], rstd[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (lane + 64 * i >= D) v[q][i] = 0.f;
        s += v[q][i];
      }
      mean[q] = s;
    }
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < NR; ++q) mean[q] += __shfl_xor(mean[q], o, 64);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      mean[q] = a.no_fn ? 0.f : mean[q] / (float)D;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float d = (lane + 64 * i < D) ? v[q][i] - mean[q] : 0.f;
        sq = fmaf(d, d, sq);
      }
      rstd[q] = sq;
    }
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < NR; ++q) rstd[q] += __shfl_xor(rstd[q], o, 64);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int rr = wave * NR + q;
      if constexpr (MODE == 2) {       // producer: xn = gamma o (x - mean) (centred, not scaled), plus the row's mean and M2
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int k = lane + 64 * i;
          if (k < 16 * KC) xn[rr * Dp + k] = (k < D ? v[q][i] - mean[q] : 0.f) * gam[i];
        }
        if (lane == 0 && row0 + rr < a.R) {
          a.rep.s12_out[2 * (int64_t)(row0 + rr)] = mean[q];
          a.rep.s12_out[2 * (int64_t)(row0 + rr) + 1] = rstd[q];       // here still sum (x - mean)^2
        }
        continue;
      }
      rstd[q] = a.no_fn ? 1.0f : 1.0f / sqrtf(rstd[q] / (float)D + OPE_LN_EPS);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int k = lane + 64 * i;
        if (k < 16 * KC) xn[rr * Dp + k] = fmaf((v[q][i] - mean[q]) * rstd[q], gam[i], bet[i]);   // gam = bet = 0 beyond D -> exactly 0
      }
      if (SAVE && lane == 0 && row0 + rr < a.R) {
        a.mu0[row0 + rr] = mean[q];
        a.rstd0[row0 + rr] = rstd[q];
      }
    }
  }
  lds_barrier();

  int row[RT];
  bool valid[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    row[t] = row0 + 16 * t + j;
    valid[t] = row[t] < a.R;
  }

  // ---- fc1: this wave's 16 output features, K = D ----
  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
    acc[t] = MODE == 2 ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(th + a.L.fc1_b + 16 * wave + 4 * g);
  if constexpr (MODE != 1) {
    const float* __restrict__ Wr = th + a.L.fc1_w + (int64_t)(16 * wave + j) * D;
    auto wload = [&](int c) { return load4c<VEC>(Wr, 16 * c + 4 * g, D); };   // clamped: chunks past KC-1 re-read the last one
    auto step = [&](const f32x4& wv, int c) {
      if (c >= KC) return;
      const int k = 16 * c + 4 * g;
      f32x4 xv[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) xv[t] = *reinterpret_cast<const f32x4*>(xn + (16 * t + j) * Dp + k);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = mfma16(wv[r], xv[t][r], acc[t]);
    };
    f32x4 w0 = wload(0), w1 = wload(1), w2 = wload(2), w3;
    for (int c = 0; c < KC; c += 4) {
      w3 = wload(c + 3);
      __builtin_amdgcn_sched_barrier(0);
      step(w0, c);
      __builtin_amdgcn_sched_barrier(0);
      w0 = wload(c + 4);
      __builtin_amdgcn_sched_barrier(0);
      step(w1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      w1 = wload(c + 5);
      __builtin_amdgcn_sched_barrier(0);
      step(w2, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      w2 = wload(c + 6);
      __builtin_amdgcn_sched_barrier(0);
      step(w3, c + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if constexpr (MODE == 2) {      // producer: u = W1 (gamma o x) of the base rows, nothing else
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t]) *reinterpret_cast<f32x4*>(a.rep.u_out + (int64_t)row[t] * OPE_H + 16 * wave + 4 * g) = acc[t];
    return;
  }
  if constexpr (MODE == 1) {
    // consumer: row r = (t*N + rep)*B + b is base row t*B + b with action block `rep` replaced by repl[r]:
    //   z1 = rstd (u + Wblk[:, rep] (new - old) - delta wsum) + cst,  u = W1 gamma (x_base - mean_base), delta = mean' - mean_base
    const RepIn& R = a.rep;
    const int NA = R.NT * R.A, A = R.A;
    float* wb = sm;                          // [64][NA]  (W1 gamma)[:, S:]
    float* dl = sm + OPE_H * NA;             // [TR][A]   new - old action block
    float* rinfo = dl + TR * A;              // [TR][4]   mean shift, rstd, rep, base row
    for (int e = threadIdx.x; e < OPE_H * NA; e += 256) wb[e] = R.wblk[e];
    if ((int)threadIdx.x < TR) {
      const int lr = threadIdx.x;
      const int64_t r = row0 + lr < a.R ? row0 + lr : a.R - 1;
      const int t = (int)(r / (R.N * R.B));
      const int rem = (int)(r - (int64_t)t * (R.N * R.B));
      const int rep = rem / R.B, b = rem - rep * R.B;
      const int64_t rb = (int64_t)t * R.B + b;
      const float* nw = R.repl + r * A;
      const float* od = R.acts + (((int64_t)t * R.NT + R.a0 + rep) * R.B + b) * A;
      // exact update of the base row's (mean, M2) for the replaced block: mean' = mean + delta, delta = sum(new - old) / D,
      // M2' = M2 + sum_blk [(new - mean)^2 - (old - mean)^2] - D delta^2   (all correction terms are small: no cancellation)
      const float mb = R.s12[2 * rb], M2 = R.s12[2 * rb + 1];
      float d1 = 0.f, d2 = 0.f;
      for (int q = 0; q < A; ++q) {
        const float vn = nw[q], vo = od[q];
        dl[lr * A + q] = vn - vo;
        d1 += vn - vo;
        d2 += (vn - mb) * (vn - mb) - (vo - mb) * (vo - mb);
      }
      const float invD = 1.0f / (float)D;
      const float delta = d1 * invD;
      const float var = fmaxf((M2 + d2) * invD - delta * delta, 0.f);
      const float rsd = 1.0f / sqrtf(var + OPE_LN_EPS);
      const float m = mb + delta;
      rinfo[4 * lr] = delta; rinfo[4 * lr + 1] = rsd; rinfo[4 * lr + 2] = __int_as_float(R.a0 + rep); rinfo[4 * lr + 3] = __int_as_float((int)rb);
      if (SAVE && row0 + lr < a.R) { a.mu0[row0 + lr] = m; a.rstd0[row0 + lr] = rsd; }
    }
    lds_barrier();
    const int fo = 16 * wave + 4 * g;
    const f32x4 ws = *reinterpret_cast<const f32x4*>(R.wsum + fo), cs = *reinterpret_cast<const f32x4*>(R.cst + fo);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int lr = 16 * t + j;
      const float m = rinfo[4 * lr], rsd = rinfo[4 * lr + 1];
      const int rep = __float_as_int(rinfo[4 * lr + 2]), rb = __float_as_int(rinfo[4 * lr + 3]);
      f32x4 uv = *reinterpret_cast<const f32x4*>(R.u + (int64_t)rb * OPE_H + fo);
      const float* wrow = wb + fo * NA + rep * A;
      const float* dr = dl + lr * A;
      for (int q = 0; q < A; ++q) {
        const float dq = dr[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) uv[r] = fmaf(wrow[r * NA + q], dq, uv[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(rsd, uv[r] - m * ws[r], cs[r]);
    }
  }

  // ReLU + LayerNorm over the 64 features of a row whose 16-feature slices live in the 4 waves. Leaves act (affine
  // output) in `o`, xhat in `xh`; returns rstd / mean / this wave's 16 ReLU bits per row through the references.
  auto relu_ln = [&](f32x4 (&z)[RT], const float* gamv, const float* betv, f32x4 (&xh)[RT], f32x4 (&o)[RT], float (&rs)[RT], float (&mu)[RT],
                     uint32_t (&bits)[RT]) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      uint32_t mb = 0;
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (z[t][r] > 0.f) mb |= 1u << (4 * g + r);
        z[t][r] = fmaxf(z[t][r], 0.f);
        s += z[t][r];
      }
      bits[t] = mb;
      s = rowsum4(s);
      if (g == 0) stat[wave * TR + 16 * t + j] = s;
    }
    lds_barrier();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int rr = 16 * t + j;
      mu[t] = ((stat[rr] + stat[TR + rr]) + (stat[2 * TR + rr] + stat[3 * TR + rr])) * (1.0f / OPE_H);
    }
    lds_barrier();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float q = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = z[t][r] - mu[t];
        q = fmaf(d, d, q);
      }
      q = rowsum4(q);
      if (g == 0) stat[wave * TR + 16 * t + j] = q;
    }
    lds_barrier();
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamv + 16 * wave + 4 * g);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(betv + 16 * wave + 4 * g);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int rr = 16 * t + j;
      const float var = ((stat[rr] + stat[TR + rr]) + (stat[2 * TR + rr] + stat[3 * TR + rr])) * (1.0f / OPE_H);
      rs[t] = 1.0f / sqrtf(var + OPE_LN_EPS);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[t][r] = (z[t][r] - mu[t]) * rs[t];
        o[t][r] = fmaf(xh[t][r], gm[r], bt[r]);
      }
    }
    // (the next write to `stat` happens after at least one more barrier)
  };
  // 16 ReLU bits of this wave -> its quarter of the row's 64-bit mask (bit f = feature f), plus rstd (and mean) once per row
  auto save_row = [&](uint64_t* mask, float* rstd_out, float* mu_out, const uint32_t (&bits)[RT], const float (&rs)[RT], const float (&mu)[RT]) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      uint32_t b = bits[t];
      b |= __shfl_xor((int)b, 16, 64);
      b |= __shfl_xor((int)b, 32, 64);
      if (g == 0 && valid[t]) {
        reinterpret_cast<uint16_t*>(mask + row[t])[wave] = (uint16_t)b;
        if (wave == 0) {
          rstd_out[row[t]] = rs[t];
          if (mu_out) mu_out[row[t]] = mu[t];
        }
      }
    }
  };

  f32x4 xh[RT], act[RT];
  float rs[RT], mu[RT];
  uint32_t bits[RT];
  relu_ln(acc, th + a.L.ln1_w, th + a.L.ln1_b, xh, act, rs, mu, bits);
  if (SAVE) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t]) *reinterpret_cast<f32x4*>(a.xhat1 + (int64_t)row[t] * OPE_H + 16 * wave + 4 * g) = xh[t];
    save_row(a.mask1, a.rstd1, a.mu1, bits, rs, mu);
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(actb + (16 * t + j) * kActPitch + 16 * wave + 4 * g) = act[t];
  lds_barrier();

  // ---- fc2 ----
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(th + a.L.fc2_b + 16 * wave + 4 * g);
  {
    const float* __restrict__ Wr = th + a.L.fc2_w + (int64_t)(16 * wave + j) * OPE_H + 4 * g;
    f32x4 wv[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) wv[ft] = *reinterpret_cast<const f32x4*>(Wr + 16 * ft);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      f32x4 xv[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) xv[t] = *reinterpret_cast<const f32x4*>(actb + (16 * t + j) * kActPitch + 16 * ft + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = mfma16(wv[ft][r], xv[t][r], acc[t]);
    }
  }
  relu_ln(acc, th + a.L.ln2_w, th + a.L.ln2_b, xh, act, rs, mu, bits);   // its first barrier also fences the fc2 reads of actb
  if (SAVE) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t]) *reinterpret_cast<f32x4*>(a.xhat2 + (int64_t)row[t] * OPE_H + 16 * wave + 4 * g) = xh[t];
    save_row(a.mask2, a.rstd2, nullptr, bits, rs, mu);
  }
  if (a.a2_out) {   // MLP nets stop here: the trunk output feeds the head directly
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t]) *reinterpret_cast<f32x4*>(a.a2_out + (int64_t)row[t] * OPE_H + 16 * wave + 4 * g) = act[t];
    if (!a.head_out) return;
    // fused small Linear head (<= 16 outputs = one MFMA tile): the activations meet in LDS, wave 0 does the 16 MFMAs
#pragma unroll
    for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(actb + (16 * t + j) * kActPitch + 16 * wave + 4 * g) = act[t];
    lds_barrier();
    if (wave != 0) return;
    const int hd = a.head_dim;
    f32x4 ho[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) ho[t][r] = (4 * g + r < hd) ? th[a.L.q_b + 4 * g + r] : 0.f;
    const float* __restrict__ Wh = th + a.L.q_w + (int64_t)(j < hd ? j : hd - 1) * OPE_H + 4 * g;   // rows >= hd: clamped, discarded below
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(Wh + 16 * ft);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(actb + (16 * t + j) * kActPitch + 16 * ft + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) ho[t] = mfma16(wv[r], xv[r], ho[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (valid[t])
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * g + r < hd) a.head_out[(int64_t)row[t] * hd + 4 * g + r] = ho[t][r];
    return;
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(actb + (16 * t + j) * kActPitch + 16 * wave + 4 * g) = act[t];
  lds_barrier();

  // ---- gi = W_ih a2 + b_ih : 12 output tiles, 3 per wave ----
  f32x4 o[RT][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int t = 0; t < RT; ++t) o[t][u] = *reinterpret_cast<const f32x4*>(th + a.L.bih + 16 * (3 * wave + u) + 4 * g);
#pragma unroll
  for (int ft = 0; ft < 4; ++ft) {
    f32x4 wv[3], xv[RT];
#pragma unroll
    for (int u = 0; u < 3; ++u)
      wv[u] = *reinterpret_cast<const f32x4*>(th + a.L.wih + (int64_t)(16 * (3 * wave + u) + j) * OPE_H + 16 * ft + 4 * g);
#pragma unroll
    for (int t = 0; t < RT; ++t) xv[t] = *reinterpret_cast<const f32x4*>(actb + (16 * t + j) * kActPitch + 16 * ft + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u) o[t][u] = mfma16(wv[u][r], xv[t][r], o[t][u]);
  }
#pragma unroll
  for (int t = 0; t < RT; ++t)
    if (valid[t]) {
#pragma unroll
      for (int u = 0; u < 3; ++u) *reinterpret_cast<f32x4*>(a.gi + (int64_t)row[t] * (3 * OPE_H) + 16 * (3 * wave + u) + 4 * g) = o[t][u];
    }
}

template <int VEC, bool SAVE>
static int launch2(const TrunkFwdArgs& a, hipStream_t st) {
  static const int forced = getenv("OPE_TRUNK_RT") ? atoi(getenv("OPE_TRUNK_RT")) : 0;
  const int KC = (a.D + 15) >> 4;
  const int Dp = 16 * KC + 4;
  // measured: 16-row workgroups win at QMIX sizes (3s5z, 38 k rows: 0.544 vs 0.562 ms/step), 32-row ones at the recurrent
  // MADDPG sizes (MMM2 B=128, 230 k rows: 3.51 vs 3.62 ms/step) where the weight re-reads per workgroup start to matter
  const bool two = forced ? forced == 2 : a.R >= 65536;
  const int TR = two ? 32 : 16;
  const size_t lds = (size_t)(TR * Dp + TR * kActPitch + 4 * TR) * sizeof(float);
  kprof_work(2.0 * a.R * ((double)a.D * OPE_H + OPE_H * OPE_H + (a.gi ? 3.0 * OPE_H * OPE_H : 0.0) + (a.head_out ? (double)OPE_H * a.head_dim : 0.0)));
  if (two)
    OPE_LAUNCH((trunk_fwd2_kernel<VEC, 2, SAVE>), dim3(ope_cdiv(a.R, 32)), dim3(256), lds, st, a, Dp);
  else
    OPE_LAUNCH((trunk_fwd2_kernel<VEC, 1, SAVE>), dim3(ope_cdiv(a.R, 16)), dim3(256), lds, st, a, Dp);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

// ---- replicated-rows form (RepIn, ope_agent.h) -----------------------------------------------------------------------------
// wblk[f][c] = W1[f][S + c] gamma[S + c],  wsum[f] = sum_k W1[f][k] gamma[k],  cst[f] = sum_k W1[f][k] beta[k] + b1[f]
__global__ void __launch_bounds__(64) trunk_rep_prep_kernel(const float* __restrict__ th, AgentLayout L, int D, int S, float* __restrict__ wblk,
                                                             float* __restrict__ wsum, float* __restrict__ cst) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const float* w = th + L.fc1_w + (int64_t)f * D;
  float a = 0.f, b = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float wg = w[k] * th[L.fn_w + k];
    a += wg;
    b = fmaf(w[k], th[L.fn_b + k], b);
    if (k >= S) wblk[(int64_t)f * (D - S) + (k - S)] = wg;
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if (lane == 0) { wsum[f] = a; cst[f] = b + th[L.fc1_b + f]; }
}

bool trunk_rep_ok(int D, int NT, int A, int copies) {
  static const int on = getenv("OPE_TRUNK_REP") ? atoi(getenv("OPE_TRUNK_REP")) : 1;
  // LDS of the consumer: 64 NT A weights + 32 rows of (A + 4) + the 64-wide activations
  const size_t lds = ((size_t)OPE_H * NT * A + 32 * (A + 4) + 32 * kActPitch + 4 * 32) * sizeof(float);
  return on && D <= 512 && copies >= 2 && lds <= 150 * 1024;
}

template <int VEC>
static int launch_rep(const TrunkFwdArgs& base, const TrunkFwdArgs& a0, float* scratch, hipStream_t st) {
  const RepIn& R = a0.rep;
  const int D = base.D, NA = R.NT * R.A;
  float* wblk = scratch;
  float* wsum = scratch + (int64_t)OPE_H * NA;
  float* cst = wsum + OPE_H;
  OPE_LAUNCH(trunk_rep_prep_kernel, dim3(OPE_H), dim3(64), 0, st, a0.theta, a0.L, D, R.S, wblk, wsum, cst);
  {   // producer over the T*B base rows
    const int KC = (D + 15) >> 4, Dp = 16 * KC + 4;
    const size_t lds = (size_t)(16 * Dp + 16 * kActPitch + 4 * 16) * sizeof(float);
    kprof_work(2.0 * base.R * (double)D * OPE_H);     // first layer once per base row
    OPE_LAUNCH((trunk_fwd2_kernel<VEC, 1, false, 2>), dim3(ope_cdiv(base.R, 16)), dim3(256), lds, st, base, Dp);
  }
  TrunkFwdArgs a = a0;
  a.rep.wblk = wblk; a.rep.wsum = wsum; a.rep.cst = cst;
  a.D = D;
  {   // consumer over the R = T*N*B copies; "Dp" only sizes the scratch in front of the activation buffers
    constexpr int TR = 32;
    const int need = OPE_H * NA + TR * R.A + 4 * TR;
    const int Dp = ((ope_cdiv(need, TR) + 3) / 4) * 4;
    const size_t lds = (size_t)(TR * Dp + TR * kActPitch + 4 * TR) * sizeof(float);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)trunk_fwd2_kernel<VEC, 2, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return OPE_ELAUNCH;
    // per copy: the A-column correction of the first layer + the remaining layers (the materialised form would be 2 R (D 64 + ...))
    kprof_work(2.0 * a.R * ((double)R.A * OPE_H + OPE_H * OPE_H + (a.gi ? 3.0 * OPE_H * OPE_H : 0.0) + (a.head_out ? (double)OPE_H * a.head_dim : 0.0)));
    OPE_LAUNCH((trunk_fwd2_kernel<VEC, 2, true, 1>), dim3(ope_cdiv(a.R, TR)), dim3(256), lds, st, a, Dp);
  }
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int launch_trunk_fwd_rep(const TrunkFwdArgs& base, const TrunkFwdArgs& a, float* scratch, hipStream_t st) {
  if (base.R < 1 || a.R < 1 || base.D < 1 || base.D > 512 || !scratch || base.no_fn || a.no_fn) return OPE_EINVAL;
  const int vec = ope_vec_of(base.D);
  if (vec == 4) return launch_rep<4>(base, a, scratch, st);
  if (vec == 2) return launch_rep<2>(base, a, scratch, st);
  return launch_rep<1>(base, a, scratch, st);
}

int launch_trunk_fwd2(const TrunkFwdArgs& a, bool save, hipStream_t st) {
  if (a.R < 1 || a.D < 1 || a.D > 512) return OPE_EINVAL;
  const int vec = ope_vec_of(a.D);
  note_launch("trunk_fwd2", vec);
  if (save) {
    if (vec == 4) return launch2<4, true>(a, st);
    if (vec == 2) return launch2<2, true>(a, st);
    return launch2<1, true>(a, st);
  }
  if (vec == 4) return launch2<4, false>(a, st);
  if (vec == 2) return launch2<2, false>(a, st);
  return launch2<1, false>(a, st);
}


// ---------------------------------------------------------------------------------------------------------
// trunk_fwd3: weights stationary in registers, row tiles streamed (default).
// s_memtime stamps in the cooperative kernels showed where their time goes: every workgroup re-streams the layer weights
// (128 KB per 16-row tile at 3s5z, 310 MB per launch) through the CU's L1/TA path in 64-byte row segments, and a 16-k chunk
// of the first layer costs ~2 800 cycles against 512 cycles of MFMA. Weights per wave are small, though: with the feature
// split of trunk_fwd2 a wave needs 16 rows of W1 (D/4 <= 128 VGPRs), 16 rows of W2 (16 VGPRs) and 48 rows of W_ih
// (48 VGPRs) as MFMA A-operand fragments. So here the grid is persistent (2 workgroups per CU), every wave loads its
// fragments ONCE and then walks over 16-row tiles: per tile only the observations (HBM -> LDS, normalised) and the
// outputs move. KCM = compile-time bound on D/16 (register array size); chunks beyond D multiply zero-padded activations.
// ---------------------------------------------------------------------------------------------------------
template <int VEC, int KCM, bool SAVE>
__global__ void __launch_bounds__(256, 2) trunk_fwd3_kernel(TrunkFwdArgs a) {
  constexpr int TR = 16;
  constexpr int Dp = 16 * KCM + 4;
  __shared__ __attribute__((aligned(16))) float xn[TR * Dp];
  __shared__ __attribute__((aligned(16))) float actb[TR * kActPitch];
  __shared__ __attribute__((aligned(8))) float stat[2 * 4 * TR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int D = a.D;
  const float* __restrict__ th = a.theta;
  const int ntiles = (a.R + TR - 1) / TR;

  long long* dbg = a.dbg ? a.dbg + ((int64_t)blockIdx.x * 4 + wave) * 16 : nullptr;
  int dbi = 0;
#define OPE_STAMP() do { if (dbg && lane == 0 && dbi < 16) dbg[dbi] = __builtin_amdgcn_s_memtime(); ++dbi; } while (0)
  OPE_STAMP();
  // ---- this wave's weight fragments, loaded once ----
  // wide inputs (KCM > 16): only the first-layer fragments stay resident; W2 / W_ih fragments (16 loads per tile) are
  // fetched where they are used, otherwise the register file spills
  constexpr bool WREG = KCM <= 16;
  f32x4 w1[KCM], w2[4], w3[3][4];
  if (VEC == 4 && WREG) {
    // A fragment is [row j][4 floats at column 4 g] of a 16 x 16 block: read in fragment order, the 16 lanes of a quarter-wave hit 16
    // DIFFERENT rows (16 lines, 16 bytes of each) and the address unit walks 64 line lookups per instruction -- 128 such loads made
    // the prologue 14.5 k cycles, 17 % of the launch (s_memtime stamps). Instead lane l FETCHES piece (l & 3) of
    // row (l >> 2) -- a quarter-wave covers four whole 64-byte row segments -- and the piece travels to the lane that owns it,
    // 4 j + g -> 16 g + j, through four ds_bpermute (no LDS memory). Same values in the same registers as the direct loads.
    const int lj = lane >> 2, lg = lane & 3, src = 4 * j + g;
    auto spread = [&](const f32x4& t) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __shfl(t[r], src, 64);
      return o;
    };
    const float* __restrict__ Wl = th + a.L.fc1_w + (int64_t)(16 * wave + lj) * D;
#pragma unroll
    for (int c = 0; c < KCM; ++c) w1[c] = spread(load4c<4>(Wl, 16 * c + 4 * lg, D));   // clamped; columns >= D meet zero activations
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
      w2[ft] = spread(*reinterpret_cast<const f32x4*>(th + a.L.fc2_w + (int64_t)(16 * wave + lj) * OPE_H + 16 * ft + 4 * lg));
    if (!a.a2_out) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
          w3[u][ft] = spread(*reinterpret_cast<const f32x4*>(th + a.L.wih + (int64_t)(16 * (3 * wave + u) + lj) * OPE_H + 16 * ft + 4 * lg));
    }
  } else {
    const float* __restrict__ Wr = th + a.L.fc1_w + (int64_t)(16 * wave + j) * D;
#pragma unroll
    for (int c = 0; c < KCM; ++c) w1[c] = load4c<VEC>(Wr, 16 * c + 4 * g, D);   // clamped; columns >= D meet zero activations
    if (WREG) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) w2[ft] = *reinterpret_cast<const f32x4*>(th + a.L.fc2_w + (int64_t)(16 * wave + j) * OPE_H + 16 * ft + 4 * g);
    }
    if (WREG && !a.a2_out) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
          w3[u][ft] = *reinterpret_cast<const f32x4*>(th + a.L.wih + (int64_t)(16 * (3 * wave + u) + j) * OPE_H + 16 * ft + 4 * g);
    }
  }
  // biases and LayerNorm affine parameters of the two hidden layers: 6 x 64 floats kept in LDS and re-read (one
  // ds_read_b128 each) where they are used, instead of 24 VGPRs held across the whole tile loop
  // (likewise b_ih and the input LayerNorm's affine parameters, zero-padded to 16*KCM: no compiler-tracked global load is
  // left inside the tile loop, whose vmcnt waits would drain the next tile's prefetched rows)
  __shared__ __attribute__((aligned(16))) float lnp[6][OPE_H];
  __shared__ __attribute__((aligned(16))) float bihs[3 * OPE_H];
  __shared__ __attribute__((aligned(16))) float fnp[2][16 * KCM];
  if (threadIdx.x < OPE_H) {
    const int f = threadIdx.x;
    lnp[0][f] = th[a.L.fc1_b + f]; lnp[1][f] = th[a.L.ln1_w + f]; lnp[2][f] = th[a.L.ln1_b + f];
    lnp[3][f] = th[a.L.fc2_b + f]; lnp[4][f] = th[a.L.ln2_w + f]; lnp[5][f] = th[a.L.ln2_b + f];
  }
  if (!a.a2_out)
    for (int f = threadIdx.x; f < 3 * OPE_H; f += blockDim.x) bihs[f] = th[a.L.bih + f];
  for (int f = threadIdx.x; f < 16 * KCM; f += blockDim.x) {
    fnp[0][f] = f < D ? th[a.L.fn_w + f] : 0.f;
    fnp[1][f] = f < D ? th[a.L.fn_b + f] : 0.f;
  }
  auto lnp4 = [&](int i) { return *reinterpret_cast<const f32x4*>(&lnp[i][16 * wave + 4 * g]); };
  // ReLU + LayerNorm over the 64 features of a row whose 16-feature slices live in the 4 waves (see trunk_fwd2)
  // One pass: per-row sum and sum of squares of the post-ReLU values (O(1) magnitudes: E[x^2] - mean^2 loses nothing that
  // matters at fp32), the 4 waves' partials meet through LDS behind ONE barrier.
  auto relu_ln = [&](f32x4& z, const f32x4& gm, const f32x4& bt, f32x4& xh, f32x4& o, float& rs, float& mu, uint32_t& bits) {
    uint32_t mb = 0;
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (z[r] > 0.f) mb |= 1u << (4 * g + r);
      // ReLU, or (OPE_DIMS_TANH) tanh(x) = 2 / (1 + 2^(-2 log2(e) x)) - 1: v_exp_f32 / v_rcp_f32, ~1e-7 absolute, as the GRU gates
      z[r] = a.tanh_act ? fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * z[r])), -1.0f) : fmaxf(z[r], 0.f);
      s += z[r];
      s2 = fmaf(z[r], z[r], s2);
    }
    bits = mb;
    s = rowsum4(s);
    s2 = rowsum4(s2);
    if (g == 0) *reinterpret_cast<f32x2*>(stat + 2 * (wave * TR + j)) = f32x2{s, s2};
    lds_barrier();
    const f32x2 p0 = *reinterpret_cast<const f32x2*>(stat + 2 * j), p1 = *reinterpret_cast<const f32x2*>(stat + 2 * (TR + j));
    const f32x2 p2 = *reinterpret_cast<const f32x2*>(stat + 2 * (2 * TR + j)), p3 = *reinterpret_cast<const f32x2*>(stat + 2 * (3 * TR + j));
    mu = ((p0[0] + p1[0]) + (p2[0] + p3[0])) * (1.0f / OPE_H);
    const float var = fmaxf(((p0[1] + p1[1]) + (p2[1] + p3[1])) * (1.0f / OPE_H) - mu * mu, 0.f);
    rs = 1.0f / sqrtf(var + OPE_LN_EPS);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[r] = (z[r] - mu) * rs;
      o[r] = fmaf(xh[r], gm[r], bt[r]);
    }
    // (stat is rewritten only after the next workgroup barrier: the actb hand-off)
  };

  // Touch every weight fragment once: the compiler then knows the prologue loads have landed and leaves no vmcnt wait of
  // its own inside the tile loop (it cannot see the asm prefetch loads, so such a wait would stall on them every tile).
#pragma unroll
  for (int c = 0; c < KCM; ++c) asm volatile("" : "+v"(w1[c]));
  if (WREG) {
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) asm volatile("" : "+v"(w2[ft]));
    if (!a.a2_out) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) asm volatile("" : "+v"(w3[u][ft]));
    }
  }
  constexpr bool PF = (VEC == 4) && (KCM <= 16);    // wider inputs go through trunk_fwd2 by default
  constexpr int XP = 16 * KCM;                      // floats per staged row
  __shared__ __attribute__((aligned(16))) float xraw[PF ? TR * XP : 4];
  // one LDS-DMA instruction moves 64 lanes x 16 B = 1 KB to [wave-uniform LDS base + 16 * lane]: 256 / XP whole rows
  auto request_rows = [&](int t0) {
    constexpr int RPI = PF ? 256 / XP : 1;          // rows per instruction (1 at KCM = 16, 2 at KCM = 8)
    const int sub = lane / (64 / RPI), pc = lane % (64 / RPI);
    const int k = 4 * pc;
#pragma unroll
    for (int u = 0; u < 4 / RPI; ++u) {
      const int rl = wave * 4 + u * RPI;            // first staged row of this instruction
      const int rw = t0 * TR + rl + sub;
      const float* src = a.x + (int64_t)(rw < a.R ? rw : a.R - 1) * D + (k + 4 <= D ? k : D - 4);   // clamped; masked below
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(xraw + rl * XP), 16, 0, 0);
    }
  };
  if (PF && (int)blockIdx.x < ntiles) request_rows(blockIdx.x);
  f32x4 go[3];        // gi of the previous tile (this lane's row j, output tiles 3 wave .. 3 wave + 2), not yet stored
  int go_row = -1;
  auto flush_gi = [&]() {
    if (go_row >= 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) *reinterpret_cast<f32x4*>(a.gi + (int64_t)go_row * (3 * OPE_H) + 16 * (3 * wave + u) + 4 * g) = go[u];
    }
    go_row = -1;
  };
  lds_barrier();   // lnp / bihs / fnp are complete (phase 0 of the first tile reads fnp before any other barrier)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TR;
    // ---- phase 0: input LayerNorm of this wave's 4 rows -> LDS. 16 lanes per row (one DPP row): each lane keeps
    // 4*KCM/4 = KCM values of its row, row sums are four DPP adds; every global access is a 16-byte (VEC-wide) piece of a
    // 256-byte contiguous run. With 16-byte pieces (PF) the raw rows arrive by LDS-DMA (global_load_lds_dwordx4, 1 KB per
    // wave instruction, no VGPRs): a wave requests ITS four rows of the next tile as soon as it has read this tile's, the
    // request stays in flight for the whole tile (the barriers of this kernel wait on LDS traffic only, and no other
    // load is left in the loop), and the compiler's vmcnt wait sits in front of the reads of `xraw` one tile later.
    {
      constexpr int NI4 = KCM / 4;
      const int q = lane >> 4, c = lane & 15;
      const int rr = wave * 4 + q;
      const int row = row0 + rr;
      f32x4 xv[NI4];
      if (PF) {
        // Rows landed? hipcc puts its own vmcnt wait only in front of the FIRST tile's reads of xraw, not on the loop's
        // back edge, so the wait is ours. It has to be vmcnt(0): loads and stores do not retire in order relative to each
        // other (a wait that allowed the previous tile's youngest stores to stay in flight read stale rows), so the
        // stores a wave issues last in a tile -- gi -- are held back and issued right AFTER this wait (below), which
        // leaves them a whole tile to retire; the mid-tile stores (xhat, masks, statistics) are half a tile old here.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NI4; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xraw + rr * XP + 4 * c + 64 * i);
      } else {
        const float* xr = a.x + (int64_t)(row < a.R ? row : a.R - 1) * D;
#pragma unroll
        for (int i = 0; i < NI4; ++i) xv[i] = load4c<VEC>(xr, 4 * c + 64 * i, D);
      }
      flush_gi();   // the previous tile's gi rows leave now
      OPE_STAMP();
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NI4; ++i) {
        xv[i] = mask4(xv[i], 4 * c + 64 * i, D);
        s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      }
      const float mean = a.no_fn ? 0.f : row16_sum(s) / (float)D;
      if (PF) {   // xraw rows of this wave are in registers now: refill them with the next tile's
        const int nxt = tile + (int)gridDim.x;
        if (nxt < ntiles) request_rows(nxt);
      }
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NI4; ++i) {
        const int k = 4 * c + 64 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = (k + r < D) ? xv[i][r] - mean : 0.f;
          xv[i][r] = d;
          sq = fmaf(d, d, sq);
        }
      }
      const float rstd = a.no_fn ? 1.0f : 1.0f / sqrtf(row16_sum(sq) / (float)D + OPE_LN_EPS);
#pragma unroll
      for (int i = 0; i < NI4; ++i) {
        const int k = 4 * c + 64 * i;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(&fnp[0][k]), bt = *reinterpret_cast<const f32x4*>(&fnp[1][k]);   // 0 beyond D
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaf(xv[i][r] * rstd, gm[r], bt[r]);   // exactly 0 beyond D
        *reinterpret_cast<f32x4*>(xn + rr * Dp + k) = o;
      }
      if (SAVE && c == 0 && row < a.R) {
        a.mu0[row] = mean;
        a.rstd0[row] = rstd;
      }
    }
    OPE_STAMP();   // phase 0 done (before barrier)
    lds_barrier();
    const int row = row0 + j;
    const bool valid = row < a.R;

    OPE_STAMP();   // after barrier
    // ---- fc1 from registers ----
    f32x4 acc = lnp4(0);
#pragma unroll
    for (int c = 0; c < KCM; ++c) {
      // keep the LDS operand reads at most 4 chunks ahead of their MFMAs (hoisting all KCM of them costs 4*KCM registers)
      if ((c & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xn + j * Dp + 16 * c + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma16(w1[c][r], xv[r], acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    OPE_STAMP();   // fc1 done
    f32x4 xh, act;
    float rs, mu;
    uint32_t bits;
    auto save_row = [&](uint64_t* mask, float* rstd_out, float* mu_out) {
      uint32_t b = bits;
      b |= __shfl_xor((int)b, 16, 64);
      b |= __shfl_xor((int)b, 32, 64);
      if (g == 0 && valid) {
        if (!a.tanh_act) reinterpret_cast<uint16_t*>(mask + row)[wave] = (uint16_t)b;
        if (wave == 0) {
          rstd_out[row] = rs;
          if (mu_out) mu_out[row] = mu;
          if (a.tanh_act) reinterpret_cast<float*>(mask + row)[0] = mu;      // no ReLU bits to keep: the adjoint needs the row mean instead
        }
      }
    };
    relu_ln(acc, lnp4(1), lnp4(2), xh, act, rs, mu, bits);
    if (SAVE) {
      if (valid) *reinterpret_cast<f32x4*>(a.xhat1 + (int64_t)row * OPE_H + 16 * wave + 4 * g) = xh;
      save_row(a.mask1, a.rstd1, a.mu1);
    }
    *reinterpret_cast<f32x4*>(actb + j * kActPitch + 16 * wave + 4 * g) = act;
    lds_barrier();

    OPE_STAMP();   // LN1 + saves + actb + barrier
    // ---- fc2 ----
    acc = lnp4(3);
    if (!WREG) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) w2[ft] = *reinterpret_cast<const f32x4*>(th + a.L.fc2_w + (int64_t)(16 * wave + j) * OPE_H + 16 * ft + 4 * g);
    }
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(actb + j * kActPitch + 16 * ft + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma16(w2[ft][r], xv[r], acc);
    }
    relu_ln(acc, lnp4(4), lnp4(5), xh, act, rs, mu, bits);   // its first barrier also fences the fc2 reads of actb
    if (SAVE) {
      if (valid) *reinterpret_cast<f32x4*>(a.xhat2 + (int64_t)row * OPE_H + 16 * wave + 4 * g) = xh;
      save_row(a.mask2, a.rstd2, nullptr);
    }
    OPE_STAMP();   // fc2 + LN2 + saves
    if (a.a2_out) {   // MLP nets: the trunk output feeds the head directly
      if (valid) *reinterpret_cast<f32x4*>(a.a2_out + (int64_t)row * OPE_H + 16 * wave + 4 * g) = act;
      if (!a.head_out) continue;
      *reinterpret_cast<f32x4*>(actb + j * kActPitch + 16 * wave + 4 * g) = act;
      lds_barrier();
      if (wave == 0) {   // fused small Linear head (<= 16 outputs = one MFMA tile)
        const int hd = a.head_dim;
        f32x4 ho;
#pragma unroll
        for (int r = 0; r < 4; ++r) ho[r] = (4 * g + r < hd) ? th[a.L.q_b + 4 * g + r] : 0.f;
        const float* __restrict__ Wh = th + a.L.q_w + (int64_t)(j < hd ? j : hd - 1) * OPE_H + 4 * g;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(Wh + 16 * ft);
          const f32x4 xv = *reinterpret_cast<const f32x4*>(actb + j * kActPitch + 16 * ft + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) ho = mfma16(wv[r], xv[r], ho);
        }
        if (valid)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * g + r < hd) a.head_out[(int64_t)row * hd + 4 * g + r] = ho[r];
      }
      continue;   // the next tile's first barrier orders wave 0's actb reads before anyone overwrites actb
    }
    *reinterpret_cast<f32x4*>(actb + j * kActPitch + 16 * wave + 4 * g) = act;
    lds_barrier();

    // ---- gi = W_ih a2 + b_ih : 3 of the 12 output tiles ----
    if (!WREG) {
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
          w3[u][ft] = *reinterpret_cast<const f32x4*>(th + a.L.wih + (int64_t)(16 * (3 * wave + u) + j) * OPE_H + 16 * ft + 4 * g);
    }
    f32x4 o[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) o[u] = *reinterpret_cast<const f32x4*>(&bihs[16 * (3 * wave + u) + 4 * g]);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(actb + j * kActPitch + 16 * ft + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 3; ++u) o[u] = mfma16(w3[u][ft][r], xv[r], o[u]);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) go[u] = o[u];
    go_row = valid ? row : -1;
    OPE_STAMP();   // wih done (gi stores deferred to the next tile's phase 0)
  }
  flush_gi();
#undef OPE_STAMP
}

template <int VEC, bool SAVE>
static int launch3(const TrunkFwdArgs& a, hipStream_t st) {
  const int KC = (a.D + 15) >> 4;
  const int ntiles = ope_cdiv(a.R, 16);
  static const int cap = getenv("OPE_TRUNK3_BLOCKS") ? atoi(getenv("OPE_TRUNK3_BLOCKS")) : 512;   // persistent: two workgroups per CU
  const int blocks = ntiles < cap ? ntiles : cap;
  kprof_work(2.0 * a.R * ((double)a.D * OPE_H + OPE_H * OPE_H + (a.gi ? 3.0 * OPE_H * OPE_H : 0.0) + (a.head_out ? (double)OPE_H * a.head_dim : 0.0)));
  if (KC <= 8) OPE_LAUNCH((trunk_fwd3_kernel<VEC, 8, SAVE>), dim3(blocks), dim3(256), 0, st, a);
  else if (KC <= 16) OPE_LAUNCH((trunk_fwd3_kernel<VEC, 16, SAVE>), dim3(blocks), dim3(256), 0, st, a);
  else if (KC <= 24) OPE_LAUNCH((trunk_fwd3_kernel<VEC, 24, SAVE>), dim3(blocks), dim3(256), 0, st, a);
  else OPE_LAUNCH((trunk_fwd3_kernel<VEC, 32, SAVE>), dim3(blocks), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("trunk_fwd3", VEC, KC <= 8 ? 8 : (KC <= 16 ? 16 : (KC <= 24 ? 24 : 32)));
  return OPE_OK;
}

int launch_trunk_fwd3(const TrunkFwdArgs& a, bool save, hipStream_t st) {
  if (a.R < 1 || a.D < 1 || a.D > 512) return OPE_EINVAL;
  const int vec = ope_vec_of(a.D);
  if (save) {
    if (vec == 4) return launch3<4, true>(a, st);
    if (vec == 2) return launch3<2, true>(a, st);
    return launch3<1, true>(a, st);
  }
  if (vec == 4) return launch3<4, false>(a, st);
  if (vec == 2) return launch3<2, false>(a, st);
  return launch3<1, false>(a, st);
}

}  // namespace ope
