// The live-plan computation itself (LivePlan, ope_common.h), as a device function two kernels share: live_plan_kernel (ope_live.hip: from the
// batch's own dones_env) and the gather's rider workgroups (ope_store.hip: from the store's dones_env through the sampled slots, inside the
// launch that copies the batch -- off the step's critical path). See ope_live.hip for the reference lines.
#pragma once
#include "ope_live.h"

namespace ope {

struct LiveW {      // writable views of the plan region
  int* hdr; int* len; int* perm; int* cum; int* nn; int* tbrec; int* tbsrc; int* srcrow; int* prevrow;
};
struct LiveOffs { int64_t len, perm, cum, nn, tbrec, tbsrc, srcrow, prevrow, total; };
static inline LiveOffs live_offs(int T, int N, int B) {
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
static inline LiveW live_views(int* base, int T, int N, int B) {
  const LiveOffs o = live_offs(T, N, B);
  LiveW w;
  w.hdr = base; w.len = base + o.len; w.perm = base + o.perm; w.cum = base + o.cum; w.nn = base + o.nn;
  w.tbrec = base + o.tbrec; w.tbsrc = base + o.tbsrc; w.srcrow = base + o.srcrow; w.prevrow = base + o.prevrow;
  return w;
}

constexpr int kLiveThreads = 256;
// LDS ints the body needs for a batch of B episodes of T steps: four B-tables, the two time tables padded to a multiple of 4 * kLiveThreads
// entries... of which only TP = round-up(T + 2, 4) are ever non-zero; the scan walks 4 entries per thread over kLiveThreads threads, so
// TPmax = 4 * kLiveThreads = 1024 entries bound episode_length (kLiveMaxT).
__host__ __device__ static inline int live_tp(int T) { return (T + 2 + 3) & ~3; }
__host__ __device__ static inline int live_lds_ints(int T, int B) { return 5 * B + 2 * live_tp(T) + 8; }      // (+ B: the accessor's own table)

// Dones: key(b) -> what locates episode b's flags (a load for the store: its slot), at(key, t) -> the flag, both branch-free so that a round's
// U keys and then its U flags are each ONE batch of independent loads (a bounds-checked load per element compiled to 2 U dependent round
// trips: 28 us); BMAJOR: walk the (t, b) pairs episode by episode (the store keeps an episode's T flags contiguous) instead of step by step.
// `wg` of `nwg` workgroups of kLiveThreads threads: every one builds the small tables in `lds` (live_lds_ints ints), workgroup 0 writes them
// out, all write their slice of the row maps and of the zero-filled regions.
// U: flags in flight per thread and round of the first phase (the riders of a gather launch take 20: one round at 3s5z, B = 32 -- a second
// round's loads would queue behind the copy that saturates the memory system beside them).
template <bool BMAJOR, int U, class Dones>
__device__ __forceinline__ void live_plan_body(const LiveW& w, float* err_abs, float* loss_part, int n_loss_part, int T, int N, int B, const Dones& dones,
                                               int* lds, int wg, int nwg) {
  const int TP = live_tp(T);
  int* const last_s = lds;
  int* const len_s = last_s + B;
  int* const inv_s = len_s + B;
  int* const lsort_s = inv_s + B;
  int* const nn_s = lsort_s + B;
  int* const cum_s = nn_s + TP;
  int* const wsum_s = cum_s + TP;
  int* const keys_s = wsum_s + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NB = N * B, TB = T * B;
  for (int b = tid; b < B; b += kLiveThreads) last_s[b] = -1;
  dones.prepare(keys_s, B);      // (the store: its slots -> LDS by wave-uniform scalar loads, a path the copy beside a rider does not load)
  __syncthreads();
  // last t' with dones_env[t', b] != 1 (the reference's mask is 1 - dones_env: exactly zero only where the flag is exactly one)
  // (eight independent loads in flight per thread and round: a load -> compare -> atomic loop is one memory round trip per element -- 19 of
  // them at 3s5z, B = 32: 10 of the first version's 12 us)
  for (int base = 0; base < TB; base += U * kLiveThreads) {
    float v[U];
    int tt[U], bb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(base + u * kLiveThreads + tid, TB - 1);
      if (BMAJOR) { bb[u] = i / T; tt[u] = i - bb[u] * T; }
      else { tt[u] = i / B; bb[u] = i - tt[u] * B; }
    }
    int key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) key[u] = dones.key(keys_s, bb[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = dones.at(key[u], tt[u]);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + u * kLiveThreads + tid < TB && v[u] != 1.0f) atomicMax(&last_s[bb[u]], tt[u]);
  }
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
    if (wg == 0) { w.perm[r] = b; w.len[r] = l; }
  }
  __syncthreads();
  for (int t = tid; t < TP; t += kLiveThreads) {
    int c = 0;
    if (t <= T)
      for (int j = 0; j < B; ++j) c += lsort_s[j] > t ? 1 : 0;
    nn_s[t] = c;
  }
  __syncthreads();
  {   // exclusive prefix sums over t: four entries per thread, a wave scan, the four waves' totals through LDS
    const int base = 4 * tid;
    const bool in = base < TP;      // (TP is a multiple of 4)
    const int v0 = in ? nn_s[base] : 0, v1 = in ? nn_s[base + 1] : 0, v2 = in ? nn_s[base + 2] : 0, v3 = in ? nn_s[base + 3] : 0;
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
    if (in) { cum_s[base] = excl; cum_s[base + 1] = excl + v0; cum_s[base + 2] = excl + v0 + v1; cum_s[base + 3] = excl + v0 + v1 + v2; }
  }
  __syncthreads();
  if (wg == 0) {
    for (int t = tid; t <= T + 1; t += kLiveThreads) { w.cum[t] = cum_s[t]; w.nn[t] = nn_s[t]; }
    if (tid == 0) {
      const int RL = N * cum_s[T + 1], R1L = N * cum_s[T], TBL = cum_s[T];
      w.hdr[0] = RL; w.hdr[1] = R1L; w.hdr[2] = TBL; w.hdr[3] = lsort_s[0]; w.hdr[4] = T; w.hdr[5] = N; w.hdr[6] = B; w.hdr[7] = 0;
      long long* acc = reinterpret_cast<long long*>(w.hdr + kLiveAccOff);
      acc[0] += RL; acc[1] += R1L; acc[2] += TBL; acc[3] += 1;
    }
  }
  const int gtid = wg * kLiveThreads + tid, gstride = nwg * kLiveThreads;
  if (err_abs)
    for (int i = gtid; i < TB; i += gstride) err_abs[i] = 0.f;
  if (loss_part)
    for (int i = gtid; i < n_loss_part; i += gstride) loss_part[i] = 0.f;
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

// workgroups that share the row maps of a plan: ~4 batch rows per thread, at most 64
static inline int live_plan_blocks(int T, int N, int B) {
  const int64_t R = (int64_t)(T + 1) * N * B;
  const int blocks = (int)((R + 4 * kLiveThreads - 1) / (4 * kLiveThreads));
  return blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
}

}  // namespace ope
