// live_plan_kernel: the rows of a padded recurrent batch that carry a loss term, found on the device from dones_env at the start of a step.
//
// Reference (what makes a row dead): offpolicy/algorithms/qmix/qmix.py:161 builds bad_transitions_mask = [0; dones_env[:T-1]], :166 multiplies
// the Bellman error by 1 - mask, :184-186 divide by (1 - mask).sum(), :198 masks Q_tot before its mean; priorities (:177-181) see |error| = 0.
// The runners stop writing an episode at termination and leave the rest of the T slots at their dones_env = 1 defaults
// (runner/rnn/smac_runner.py:65-71,107-130, utils/rec_buffer.py:139-141): on real data most of a batch is such padding.
// Index spaces, ranking and the record formats: LivePlan, ope_common.h. Nothing here is host-visible: the launchers size their grids for
// the padded batch and the kernels read the live counts from `hdr`.
//
// Every workgroup rebuilds the small tables in LDS (dones_env is T * B floats: 19 KB at 3s5z, B = 32) and writes its slice of the row maps;
// workgroup 0 also writes the tables themselves. No float arithmetic, integer atomics in LDS only (max): deterministic.
#include "ope_live.h"

namespace ope {
namespace {

struct LiveW {      // writable views of the plan region
  int* hdr; int* len; int* perm; int* cum; int* nn; int* tbrec; int* tbsrc; int* srcrow; int* prevrow;
};
struct LiveOffs { int64_t len, perm, cum, nn, tbrec, tbsrc, srcrow, prevrow, total; };
LiveOffs live_offs(int T, int N, int B) {
  auto r4 = [](int64_t x) { return (x + 3) & ~(int64_t)3; };
  LiveOffs o;
  int64_t p = 16;
  o.len = p; p += r4(B);
  o.perm = p; p += r4(B);
  o.cum = p; p += r4(T + 2);
  o.nn = p; p += r4(T + 2);
  o.tbrec = p; p += 8 * (int64_t)T * B;
  o.tbsrc = p; p += r4((int64_t)T * B);
  o.srcrow = p; p += r4((int64_t)(T + 1) * N * B);
  o.prevrow = p; p += r4((int64_t)(T + 1) * N * B);
  o.total = p;
  return o;
}

constexpr int kLiveThreads = 256;

__global__ void __launch_bounds__(kLiveThreads) live_plan_kernel(LiveArgs a, LiveW w) {
  __shared__ int last_s[kLiveMaxB], len_s[kLiveMaxB], inv_s[kLiveMaxB], lsort_s[kLiveMaxB];
  __shared__ int nn_s[kLiveMaxT + 2], cum_s[kLiveMaxT + 2];
  __shared__ int wsum_s[kLiveThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.T, B = a.B, N = a.N, NB = N * B, TB = T * B;
  for (int b = tid; b < B; b += kLiveThreads) last_s[b] = -1;
  __syncthreads();
  // last t' with dones_env[t', b] != 1 (the reference's mask is 1 - dones_env: exactly zero only where the flag is exactly one)
  for (int i = tid; i < TB; i += kLiveThreads)
    if (a.dones_env[i] != 1.0f) atomicMax(&last_s[i % B], i / B);
  __syncthreads();
  for (int b = tid; b < B; b += kLiveThreads) len_s[b] = last_s[b] < 0 ? 1 : last_s[b] + 2;
  __syncthreads();
  // rank by length, longest first, ties in batch order
  for (int b = tid; b < B; b += kLiveThreads) {
    const int l = len_s[b];
    int r = 0;
    for (int c = 0; c < B; ++c) {
      const int lc = len_s[c];
      r += (lc > l || (lc == l && c < b)) ? 1 : 0;
    }
    inv_s[b] = r;
    lsort_s[r] = l;
    if (blockIdx.x == 0) { w.perm[r] = b; w.len[r] = l; }
  }
  __syncthreads();
  for (int t = tid; t < kLiveMaxT + 2; t += kLiveThreads) {
    int c = 0;
    if (t <= T)
      for (int j = 0; j < B; ++j) c += lsort_s[j] > t ? 1 : 0;
    nn_s[t] = c;
  }
  __syncthreads();
  {   // exclusive prefix sums over t: four entries per thread, a wave scan, the four waves' totals through LDS
    const int base = 4 * tid;
    const int v0 = nn_s[base], v1 = nn_s[base + 1], v2 = nn_s[base + 2], v3 = nn_s[base + 3];
    const int tot = (v0 + v1) + (v2 + v3);
    int x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum_s[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wave; ++q) woff += wsum_s[q];
    const int excl = woff + x - tot;
    cum_s[base] = excl; cum_s[base + 1] = excl + v0; cum_s[base + 2] = excl + v0 + v1; cum_s[base + 3] = excl + v0 + v1 + v2;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int t = tid; t <= T + 1; t += kLiveThreads) { w.cum[t] = cum_s[t]; w.nn[t] = nn_s[t]; }
    if (tid == 0) {
      const int RL = N * cum_s[T + 1], R1L = N * cum_s[T], TBL = cum_s[T];
      w.hdr[0] = RL; w.hdr[1] = R1L; w.hdr[2] = TBL; w.hdr[3] = lsort_s[0]; w.hdr[4] = T; w.hdr[5] = N; w.hdr[6] = B; w.hdr[7] = 0;
      long long* acc = reinterpret_cast<long long*>(w.hdr + kLiveAccOff);
      acc[0] += RL; acc[1] += R1L; acc[2] += TBL; acc[3] += 1;
    }
  }
  const int gtid = blockIdx.x * kLiveThreads + tid, gstride = gridDim.x * kLiveThreads;
  for (int i = gtid; i < TB; i += gstride) a.err_abs[i] = 0.f;
  for (int i = gtid; i < a.n_loss_part; i += gstride) a.loss_part[i] = 0.f;
  // packed agent rows: one thread per BATCH row (t, agent, b)
  const int R = (T + 1) * NB;
  for (int s = gtid; s < R; s += gstride) {
    const int t = s / NB, rem = s - t * NB, ag = rem / B, b = rem - ag * B;
    if (t < len_s[b]) {
      const int j = inv_s[b];
      const int p = N * cum_s[t] + ag * nn_s[t] + j;
      w.srcrow[p] = s;
      w.prevrow[p] = t > 0 ? N * cum_s[t - 1] + ag * nn_s[t - 1] + j : -1;
    }
  }
  // packed (t, b) rows
  for (int s = gtid; s < TB; s += gstride) {
    const int t = s / B, b = s - t * B;
    if (t < len_s[b]) {      // (t < T here: min(len_b, T) rows)
      const int j = inv_s[b];
      const int q = cum_s[t] + j;
      int4 r0, r1;
      r0.x = t; r0.y = b; r0.z = N * cum_s[t] + j; r0.w = nn_s[t];
      r1.x = N * cum_s[t + 1] + j; r1.y = nn_s[t + 1]; r1.z = j < nn_s[t + 1] ? 1 : 0; r1.w = j;
      reinterpret_cast<int4*>(w.tbrec)[2 * q] = r0;
      reinterpret_cast<int4*>(w.tbrec)[2 * q + 1] = r1;
      w.tbsrc[q] = s;
    }
  }
}

}  // namespace

bool live_plan_shape_ok(int T, int N, int B) { return T >= 1 && T <= kLiveMaxT && B >= 1 && B <= kLiveMaxB && N >= 1; }

int64_t live_plan_ints(int T, int N, int B) { return live_offs(T, N, B).total; }

LivePlan live_plan_view(const int* base, int T, int N, int B) {
  const LiveOffs o = live_offs(T, N, B);
  LivePlan p;
  p.hdr = base; p.len = base + o.len; p.perm = base + o.perm; p.cum = base + o.cum; p.nn = base + o.nn;
  p.tbrec = base + o.tbrec; p.tbsrc = base + o.tbsrc; p.srcrow = base + o.srcrow; p.prevrow = base + o.prevrow;
  return p;
}

int launch_live_plan(const LiveArgs& a, hipStream_t st) {
  if (!live_plan_shape_ok(a.T, a.N, a.B) || !a.dones_env || !a.plan || !a.err_abs || !a.loss_part) return OPE_EINVAL;
  const LiveOffs o = live_offs(a.T, a.N, a.B);
  LiveW w;
  w.hdr = a.plan; w.len = a.plan + o.len; w.perm = a.plan + o.perm; w.cum = a.plan + o.cum; w.nn = a.plan + o.nn;
  w.tbrec = a.plan + o.tbrec; w.tbsrc = a.plan + o.tbsrc; w.srcrow = a.plan + o.srcrow; w.prevrow = a.plan + o.prevrow;
  const int64_t R = (int64_t)(a.T + 1) * a.N * a.B;
  int blocks = ope_cdiv(R, 4 * kLiveThreads);
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  OPE_LAUNCH(live_plan_kernel, dim3(blocks), dim3(kLiveThreads), 0, st, a, w);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("live_plan");
  return OPE_OK;
}

}  // namespace ope
