// Device-resident proportional prioritisation (sum / min segment trees) for the prioritized replay buffers.
//   SumSegmentTree / MinSegmentTree                 offpolicy/utils/segment_tree.py:18-165
//   PrioritizedRecReplayBuffer insert / _sample_proportional / sample / update_priorities
//                                                   offpolicy/utils/rec_buffer.py:262-324  (mlp_buffer.py:280-330 likewise)
// Trees are float64 arrays of 2*capacity nodes (node i has children 2i, 2i+1; leaves at [capacity, 2*capacity)), the sum
// tree followed by the min tree, followed by one double: the running maximum priority. Keeping them on the device removes
// the per-update host round trip of the prioritized configurations (priorities -> host tree -> indices -> device).
// Batches are <= 1024 entries per call: one workgroup, one thread per entry, tree levels separated by barriers so that a
// node is recomputed only after everything below it has landed (deterministic; duplicate indices: the last entry wins,
// like numpy fancy assignment).
#include <math.h>

#include "ope_common.h"

namespace {

struct PerView {
  double* sum; double* mn; double* maxp;
};
__host__ __device__ inline PerView per_view(void* trees, int cap) {
  double* t = (double*)trees;
  return PerView{t, t + 2 * (int64_t)cap, t + 4 * (int64_t)cap};
}

__global__ void per_init_kernel(void* trees, int cap) {
  const PerView v = per_view(trees, cap);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * (int64_t)cap; i += (int64_t)gridDim.x * blockDim.x) {
    v.sum[i] = 0.0;
    v.mn[i] = INFINITY;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) v.maxp[0] = 1.0;   // max_priorities starts at 1.0 (rec_buffer.py:259)
}

// leaves[idx[i]] = value_i ** alpha with value_i = prio[i] (update_priorities) or the running max priority (insert);
// then the touched paths are recomputed level by level; the running max is raised to max(prio) (update only).
__global__ void __launch_bounds__(1024) per_set_kernel(void* trees, int cap, const int64_t* __restrict__ idx, const float* __restrict__ prio,
                                                        double alpha, int n) {
  const PerView v = per_view(trees, cap);
  __shared__ double red[1024];
  const int i = threadIdx.x;
  int64_t pos = 0;
  double pr = 0.0;
  // The host path asserts priorities > 0 and 0 <= idx < len (rec_buffer.py:306-324); device data cannot be asserted on
  // without a sync, so the kernel makes bad input harmless instead: an index outside [0, capacity) is skipped (nothing is
  // written, its thread keeps recomputing the path of leaf 0, which is idempotent), a NaN / non-positive / infinite priority
  // from a diverged step is replaced by the running maximum (what a freshly inserted sample gets), so neither can poison
  // the sum/min trees.
  bool on = i < n;
  const bool in_range = on && idx[i] >= 0 && idx[i] < cap;
  if (on) {
    pos = (in_range ? idx[i] : 0) + cap;
    pr = prio ? (double)prio[i] : v.maxp[0];
    if (!(pr > 0.0) || isinf(pr)) pr = v.maxp[0];
    bool last = in_range;                   // duplicates: only the last occurrence writes
    for (int j = i + 1; j < n; ++j) last = last && (idx[j] != idx[i]);
    if (last) {
      const double val = pow(pr, alpha);
      v.sum[pos] = val;
      v.mn[pos] = val;
    }
  }
  red[i] = in_range && prio ? pr : 0.0;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (i < o) red[i] = fmax(red[i], red[i + o]);
    __syncthreads();
  }
  if (i == 0 && prio) v.maxp[0] = fmax(v.maxp[0], red[0]);
  for (int c = cap; c > 1; c >>= 1) {       // log2(cap) levels
    pos >>= 1;
    if (on) {
      v.sum[pos] = v.sum[2 * pos] + v.sum[2 * pos + 1];
      v.mn[pos] = fmin(v.mn[2 * pos], v.mn[2 * pos + 1]);
    }
    __syncthreads();
  }
}

// sum over the leaves [0, end) (half-open), bottom-up (segment_tree.py reduce)
__device__ double per_prefix_total(const double* sum, int cap, int end) {
  double res = 0.0;
  int64_t lo = cap, hi = (int64_t)end + cap;
  while (lo < hi) {
    if (lo & 1) res += sum[lo++];
    if (hi & 1) res += sum[--hi];
    lo >>= 1;
    hi >>= 1;
  }
  return res;
}

// idx[i] = find_prefixsum_idx(mass01[i] * sum(leaves[0, filled-1)))   (rec_buffer.py:272-276: the last filled leaf is
// left out of the mass, an upstream quirk kept as is); weights[i] = (p_i * filled)^-beta / max_w, max_w = (p_min * filled)^-beta
// filled_dev / beta_dev (optional): the two values that change between the replays of a captured graph, read from HBM instead
// of the launch arguments (filled clamped to [2, capacity]).
__global__ void per_sample_kernel(const void* trees, int cap, int filled, const double* __restrict__ mass01, double beta, int n,
                                  int64_t* __restrict__ idx_out, float* __restrict__ w_out, const int32_t* __restrict__ filled_dev,
                                  const double* __restrict__ beta_dev) {
  const PerView v = per_view(const_cast<void*>(trees), cap);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (filled_dev) filled = min(max(filled_dev[0], 2), cap);
  if (beta_dev) beta = beta_dev[0];
  const double total = per_prefix_total(v.sum, cap, filled - 1);
  double p = mass01[i] * total;
  int64_t node = 1;
  while (node < cap) {
    const int64_t left = 2 * node;
    const double lv = v.sum[left];
    if (lv <= p) { p -= lv; node = left + 1; } else { node = left; }
  }
  const int64_t leaf = node - cap;
  idx_out[i] = leaf;
  if (w_out) {
    const double s = v.sum[1];
    const double max_w = pow(v.mn[1] / s * (double)filled, -beta);
    w_out[i] = (float)(pow(v.sum[node] / s * (double)filled, -beta) / max_w);
  }
}

bool pow2(int c) { return c > 0 && (c & (c - 1)) == 0; }

}  // namespace

extern "C" int64_t ope_per_tree_bytes(int32_t capacity) {
  if (!pow2(capacity)) return OPE_EINVAL;
  return (4 * (int64_t)capacity + 2) * (int64_t)sizeof(double);
}

extern "C" int ope_per_tree_init(void* trees, int32_t capacity, void* stream) {
  (void)hipGetLastError();
  if (!trees || !pow2(capacity)) return OPE_EINVAL;
  OPE_LAUNCH(per_init_kernel, dim3(ope_cdiv(2 * (int64_t)capacity, 256) < 1024 ? ope_cdiv(2 * (int64_t)capacity, 256) : 1024), dim3(256), 0,
                     (hipStream_t)stream, trees, capacity);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_per_tree_set(void* trees, int32_t capacity, const int64_t* idx, const float* priorities, double alpha, int32_t n,
                                void* stream) {
  (void)hipGetLastError();
  if (!trees || !pow2(capacity) || !idx || n < 1) return OPE_EINVAL;
  for (int done = 0; done < n; done += 1024) {      // later chunks overwrite earlier ones: "last wins" across chunks too
    const int m = n - done < 1024 ? n - done : 1024;
    OPE_LAUNCH(per_set_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, trees, capacity, idx + done,
                       priorities ? priorities + done : nullptr, alpha, m);
    OPE_CHECK_LAUNCH();
  }
  return OPE_OK;
}

extern "C" int ope_per_tree_sample(const void* trees, int32_t capacity, int32_t filled, const double* mass01, double beta, int32_t n,
                                   int64_t* idx_out, float* weights_out, void* stream) {
  (void)hipGetLastError();
  if (!trees || !pow2(capacity) || filled < 2 || filled > capacity || !mass01 || !idx_out || n < 1) return OPE_EINVAL;
  OPE_LAUNCH(per_sample_kernel, dim3(ope_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, trees, capacity, filled, mass01, beta, n,
                     idx_out, weights_out, (const int32_t*)nullptr, (const double*)nullptr);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_per_tree_sample_dev(const void* trees, int32_t capacity, const int32_t* filled_dev, const double* mass01, const double* beta_dev,
                                       int32_t n, int64_t* idx_out, float* weights_out, void* stream) {
  (void)hipGetLastError();
  if (!trees || !pow2(capacity) || !filled_dev || !beta_dev || !mass01 || !idx_out || n < 1) return OPE_EINVAL;
  OPE_LAUNCH(per_sample_kernel, dim3(ope_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, trees, capacity, 2, mass01, 0.0, n, idx_out,
                     weights_out, filled_dev, beta_dev);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}
