// GRU scans, one wave per row ("gru1"): the forward recurrence and its BPTT with the WHOLE W_hh held by one wave.
//
// Replaces (reference): nn.GRU inside RNNLayer.forward, offpolicy/algorithms/utils/rnn.py:19-23, stepped over the
// T+1 entries of an episode by AgentQFunction.forward (qmix/algorithm/agent_q_function.py:34-67), and the autograd of it.
//
// Used when a launch has more rows than SIMDs (throughput regime; ope_gru4.hip covers the latency regime).
// Here a workgroup is one row = two waves with disjoint jobs:
//   wave 0 (compute): lane f owns hidden feature f and keeps rows f, 64+f, 128+f of W_hh (forward) or column f of W_hh
//                     (backward) in 192 VGPRs. Per step: publish the 64-vector to broadcast (h, or the three gate
//                     adjoints) to LDS, read it back with broadcast ds_read_b128 (same wave: no barrier), 96
//                     v_pk_fma_f32 on independent partial sums, gates. It never touches global memory, so no vmcnt
//                     wait sits on the serial chain.
//   wave 1 (memory):  per 8-step chunk: hands the next chunk's inputs (loaded one chunk ago with compiler-invisible
//                     asm loads) to the compute wave through an LDS ring, starts the loads of the chunk after it, and
//                     writes the previous chunk's results (h and the saved gates / dgi and dghn) from an LDS ring to
//                     HBM with coalesced 256-byte stores.
// The two waves meet at ONE workgroup barrier per chunk. Summation order is fixed (four partial sums per gate, combined
// pairwise), so results are deterministic; they differ from ope_gru4.hip's by rounding only.
//   r = sigma(gi_r + gh_r), z = sigma(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h' = (1-z) n + z h      (nn.GRU)
#include <stdlib.h>

#include "ope_agent.h"

namespace ope {
namespace {

constexpr int kC = 8;   // steps per chunk
constexpr int kPF = 4;  // LDS broadcast reads in flight

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}
__device__ __forceinline__ float hsum4(f32x2 a, f32x2 b) { return (a[0] + a[1]) + (b[0] + b[1]); }

__global__ void __launch_bounds__(128) gru_fwd1_kernel(GruFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float hs[OPE_H];
  __shared__ __attribute__((aligned(16))) float gis[2][kC][3][OPE_H];
  __shared__ __attribute__((aligned(16))) float outs[2][kC][5][OPE_H];
  const int lane = threadIdx.x & 63;
  const bool mem = threadIdx.x >= 64;
  const int rid = blockIdx.x;
  const int net = rid / a.NB;
  const int row = rid - net * a.NB;
  const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;
  const bool save = (net == 0) && (a.rg != nullptr);
  const int nchunks = (a.L + kC - 1) / kC;

  if (mem) {
    const float* __restrict__ gi = net == 0 ? a.gi0 : a.gi1;
    float* __restrict__ hout = net == 0 ? a.h0out : a.h1out;
    const int64_t stride_t = (int64_t)a.NB * (3 * OPE_H);
    const float* gp = gi + (int64_t)row * (3 * OPE_H) + lane;
    float pre[kC][3];
    auto load_chunk = [&](int c) {
#pragma unroll
      for (int s = 0; s < kC; ++s) {
        const float* p = gp + (int64_t)min(c * kC + s, a.L - 1) * stride_t;
        gload_async(pre[s][0], p);
        gload_async(pre[s][1], p + OPE_H);
        gload_async(pre[s][2], p + 2 * OPE_H);
      }
    };
    auto publish = [&](int buf) {
      OPE_GWAIT24(pre);
#pragma unroll
      for (int s = 0; s < kC; ++s) {
        gis[buf][s][0][lane] = pre[s][0];
        gis[buf][s][1][lane] = pre[s][1];
        gis[buf][s][2][lane] = pre[s][2];
      }
    };
    auto store_chunk = [&](int c) {
      const int ns = min(kC, a.L - c * kC);
      for (int s = 0; s < ns; ++s) {
        const int64_t o = ((int64_t)(c * kC + s) * a.NB + row) * OPE_H + lane;
        hout[o] = outs[c & 1][s][0][lane];
        if (save) {
          a.rg[o] = outs[c & 1][s][1][lane];
          a.zg[o] = outs[c & 1][s][2][lane];
          a.ng[o] = outs[c & 1][s][3][lane];
          a.ghn[o] = outs[c & 1][s][4][lane];
        }
      }
    };
    load_chunk(0);
    publish(0);
    if (nchunks > 1) load_chunk(1);
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) {
        publish((c + 1) & 1);                    // ring slot last read during chunk c-1
        if (c + 2 < nchunks) load_chunk(c + 2);
      }
      if (c > 0) store_chunk(c - 1);
      lds_barrier();
    }
    store_chunk(nchunks - 1);
    return;
  }

  // ---- compute wave
  f32x2 wr[OPE_H / 2], wz[OPE_H / 2], wn[OPE_H / 2];
  {
    const float* w = th + a.whh_off;
#pragma unroll
    for (int k = 0; k < OPE_H / 4; ++k) {
      const f32x4 vr = *reinterpret_cast<const f32x4*>(w + (int64_t)lane * OPE_H + 4 * k);
      const f32x4 vz = *reinterpret_cast<const f32x4*>(w + (int64_t)(OPE_H + lane) * OPE_H + 4 * k);
      const f32x4 vn = *reinterpret_cast<const f32x4*>(w + (int64_t)(2 * OPE_H + lane) * OPE_H + 4 * k);
      wr[2 * k] = f32x2{vr[0], vr[1]}; wr[2 * k + 1] = f32x2{vr[2], vr[3]};
      wz[2 * k] = f32x2{vz[0], vz[1]}; wz[2 * k + 1] = f32x2{vz[2], vz[3]};
      wn[2 * k] = f32x2{vn[0], vn[1]}; wn[2 * k + 1] = f32x2{vn[2], vn[3]};
    }
  }
  const float br = th[a.bhh_off + lane], bz = th[a.bhh_off + OPE_H + lane], bn = th[a.bhh_off + 2 * OPE_H + lane];
  const float* hin = net == 0 ? a.hinit : a.hinit1;
  float h = hin ? hin[(int64_t)row * OPE_H + lane] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const int ns = min(kC, a.L - c * kC);
    for (int s = 0; s < ns; ++s) {
      hs[lane] = h;
      const float gir = gis[buf][s][0][lane], giz = gis[buf][s][1][lane], gin = gis[buf][s][2][lane];
      __builtin_amdgcn_wave_barrier();
      f32x2 ar0 = {0.f, 0.f}, ar1 = {0.f, 0.f}, az0 = {0.f, 0.f}, az1 = {0.f, 0.f}, an0 = {0.f, 0.f}, an1 = {0.f, 0.f};
      f32x4 hq[kPF];   // broadcast reads kept kPF deep in flight (LDS latency ~ 4 x 6 packed FMAs)
#pragma unroll
      for (int v = 0; v < kPF; ++v) hq[v] = *reinterpret_cast<const f32x4*>(hs + 4 * v);
#pragma unroll
      for (int v = 0; v < OPE_H / 4; ++v) {
        const f32x4 hv = hq[v % kPF];
        if (v + kPF < OPE_H / 4) hq[v % kPF] = *reinterpret_cast<const f32x4*>(hs + 4 * (v + kPF));
        const f32x2 lo = {hv[0], hv[1]}, hi = {hv[2], hv[3]};
        ar0 = __builtin_elementwise_fma(wr[2 * v], lo, ar0);
        az0 = __builtin_elementwise_fma(wz[2 * v], lo, az0);
        an0 = __builtin_elementwise_fma(wn[2 * v], lo, an0);
        ar1 = __builtin_elementwise_fma(wr[2 * v + 1], hi, ar1);
        az1 = __builtin_elementwise_fma(wz[2 * v + 1], hi, az1);
        an1 = __builtin_elementwise_fma(wn[2 * v + 1], hi, an1);
      }
      const float ar = br + hsum4(ar0, ar1), az = bz + hsum4(az0, az1), an = bn + hsum4(an0, an1);
      const float r = sigm(gir + ar);
      const float z = sigm(giz + az);
      const float n = tanh_(gin + r * an);
      h = (1.0f - z) * n + z * h;
      outs[buf][s][0][lane] = h;
      if (save) {
        outs[buf][s][1][lane] = r;
        outs[buf][s][2][lane] = z;
        outs[buf][s][3][lane] = n;
        outs[buf][s][4][lane] = an;
      }
    }
    lds_barrier();
  }
}

// BPTT, t = T-1 .. t_lo:
//   dh_{t-1}[k] = dh_t[k] z[k] + sum_i ( W_hr[i][k] dr_pre[i] + W_hz[i][k] dz_pre[i] + W_hn[i][k] dghn[i] )
__global__ void __launch_bounds__(128) gru_bwd1_kernel(GruBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float ds[3][OPE_H];
  __shared__ __attribute__((aligned(16))) float sav[2][kC][6][OPE_H];
  __shared__ __attribute__((aligned(16))) float outs[2][kC][4][OPE_H];
  const int lane = threadIdx.x & 63;
  const bool mem = threadIdx.x >= 64;
  const int row = blockIdx.x;
  const int64_t NB = a.NB;
  const int nsteps = a.T - a.t_lo;
  const int nchunks = (nsteps + kC - 1) / kC;

  if (mem) {
    float preA[kC][3], preB[kC][3];   // r,z,n | ghn,dh_out,h_prev
    auto load_chunk = [&](int c) {
#pragma unroll
      for (int s = 0; s < kC; ++s) {
        const int t = max(a.T - 1 - (c * kC + s), a.t_lo);
        const int64_t o = ((int64_t)t * NB + row) * OPE_H + lane;
        gload_async(preA[s][0], a.rg + o);
        gload_async(preA[s][1], a.zg + o);
        gload_async(preA[s][2], a.ng + o);
        gload_async(preB[s][0], a.ghn + o);
        gload_async(preB[s][1], a.dh_out + o);
        gload_async(preB[s][2], a.h + (t > 0 ? o - NB * OPE_H : o));
      }
    };
    auto publish = [&](int c) {
      OPE_GWAIT24(preA);
      OPE_GWAIT24(preB);
      const int buf = c & 1;
#pragma unroll
      for (int s = 0; s < kC; ++s) {
        const int t = a.T - 1 - (c * kC + s);
        sav[buf][s][0][lane] = preA[s][0];
        sav[buf][s][1][lane] = preA[s][1];
        sav[buf][s][2][lane] = preA[s][2];
        sav[buf][s][3][lane] = preB[s][0];
        sav[buf][s][4][lane] = preB[s][1];
        sav[buf][s][5][lane] = t > 0 ? preB[s][2] : 0.f;   // h_{-1} = 0
      }
    };
    auto store_chunk = [&](int c) {
      const int ns = min(kC, nsteps - c * kC);
      for (int s = 0; s < ns; ++s) {
        const int t = a.T - 1 - (c * kC + s);
        float* gout = a.dgi + ((int64_t)t * NB + row) * (3 * OPE_H) + lane;
        gout[0] = outs[c & 1][s][0][lane];
        gout[OPE_H] = outs[c & 1][s][1][lane];
        gout[2 * OPE_H] = outs[c & 1][s][2][lane];
        a.dghn[((int64_t)t * NB + row) * OPE_H + lane] = outs[c & 1][s][3][lane];
      }
    };
    load_chunk(0);
    publish(0);
    if (nchunks > 1) load_chunk(1);
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) {
        publish(c + 1);
        if (c + 2 < nchunks) load_chunk(c + 2);
      }
      if (c > 0) store_chunk(c - 1);
      lds_barrier();
    }
    store_chunk(nchunks - 1);
    return;
  }

  // ---- compute wave: column `lane` of W_hh
  f32x2 wr[OPE_H / 2], wz[OPE_H / 2], wn[OPE_H / 2];
  {
    const float* w = a.theta + a.whh_off + lane;
#pragma unroll
    for (int i = 0; i < OPE_H / 2; ++i) {
      wr[i] = f32x2{w[(int64_t)(2 * i) * OPE_H], w[(int64_t)(2 * i + 1) * OPE_H]};
      wz[i] = f32x2{w[(int64_t)(OPE_H + 2 * i) * OPE_H], w[(int64_t)(OPE_H + 2 * i + 1) * OPE_H]};
      wn[i] = f32x2{w[(int64_t)(2 * OPE_H + 2 * i) * OPE_H], w[(int64_t)(2 * OPE_H + 2 * i + 1) * OPE_H]};
    }
  }
  float dh = a.dh_in ? a.dh_in[(int64_t)row * OPE_H + lane] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const int ns = min(kC, nsteps - c * kC);
    for (int s = 0; s < ns; ++s) {
      const float r = sav[buf][s][0][lane], z = sav[buf][s][1][lane], n = sav[buf][s][2][lane];
      const float gn = sav[buf][s][3][lane], dho = sav[buf][s][4][lane], hp = sav[buf][s][5][lane];
      const float dht = dh + dho;
      const float dn = dht * (1.0f - z);
      const float dzg = dht * (hp - n);
      const float dn_pre = dn * (1.0f - n * n);
      const float dz_pre = dzg * z * (1.0f - z);
      const float dr_pre = dn_pre * gn * r * (1.0f - r);
      const float dgn = dn_pre * r;
      ds[0][lane] = dr_pre;
      ds[1][lane] = dz_pre;
      ds[2][lane] = dgn;
      __builtin_amdgcn_wave_barrier();
      outs[buf][s][0][lane] = dr_pre;
      outs[buf][s][1][lane] = dz_pre;
      outs[buf][s][2][lane] = dn_pre;
      outs[buf][s][3][lane] = dgn;
      f32x2 c0 = {0.f, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c3 = {0.f, 0.f}, c4 = {0.f, 0.f}, c5 = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < OPE_H / 4; ++v) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(&ds[0][4 * v]);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(&ds[1][4 * v]);
        const f32x4 nv = *reinterpret_cast<const f32x4*>(&ds[2][4 * v]);
        c0 = __builtin_elementwise_fma(wr[2 * v], f32x2{rv[0], rv[1]}, c0);
        c1 = __builtin_elementwise_fma(wr[2 * v + 1], f32x2{rv[2], rv[3]}, c1);
        c2 = __builtin_elementwise_fma(wz[2 * v], f32x2{zv[0], zv[1]}, c2);
        c3 = __builtin_elementwise_fma(wz[2 * v + 1], f32x2{zv[2], zv[3]}, c3);
        c4 = __builtin_elementwise_fma(wn[2 * v], f32x2{nv[0], nv[1]}, c4);
        c5 = __builtin_elementwise_fma(wn[2 * v + 1], f32x2{nv[2], nv[3]}, c5);
      }
      dh = dht * z + ((hsum4(c0, c1) + hsum4(c2, c3)) + hsum4(c4, c5));
    }
    lds_barrier();
  }
  if (a.dh_carry) a.dh_carry[(int64_t)row * OPE_H + lane] = dh;
}

}  // namespace

int launch_gru_fwd1(const GruFwdArgs& a, hipStream_t st) {
  kprof_work(2.0 * a.nets * a.NB * (double)a.L * 3.0 * OPE_H * OPE_H);
  OPE_LAUNCH(gru_fwd1_kernel, dim3(a.nets * a.NB), dim3(128), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("gru_fwd1");
  return OPE_OK;
}

int launch_gru_bwd1(const GruBwdArgs& a, hipStream_t st) {
  kprof_work(2.0 * a.NB * (double)(a.T - a.t_lo) * 3.0 * OPE_H * OPE_H);
  OPE_LAUNCH(gru_bwd1_kernel, dim3(a.NB), dim3(128), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("gru_bwd1");
  return OPE_OK;
}

}  // namespace ope
