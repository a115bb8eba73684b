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
// workgroup 0 also writes the tables themselves. No float arithmetic, integer atomics in LDS only (max): deterministic. The computation is
// live_plan_body (ope_live_dev.h), shared with the rider workgroups of the gather launch (ope_store.hip), which build the same plan from the
// store's flags while the batch is being copied.
#include "ope_live_dev.h"

namespace ope {
namespace {

struct BatchDones {
  const float* p; int B;
  __device__ __forceinline__ void prepare(int*, int) const {}
  __device__ __forceinline__ int key(const int*, int b) const { return b; }
  __device__ __forceinline__ float at(int b, int t) const { return p[t * B + b]; }
};

__global__ void __launch_bounds__(kLiveThreads) live_plan_kernel(LiveArgs a, LiveW w) {
  extern __shared__ __attribute__((aligned(16))) int live_lds[];
  BatchDones d{a.dones_env, a.B};
  live_plan_body<false, 8>(w, a.err_abs, a.loss_part, a.n_loss_part, a.T, a.N, a.B, d, live_lds, (int)blockIdx.x, (int)gridDim.x);
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
  const LiveW w = live_views(a.plan, a.T, a.N, a.B);
  OPE_LAUNCH(live_plan_kernel, dim3(live_plan_blocks(a.T, a.N, a.B)), dim3(kLiveThreads), 4 * live_lds_ints(a.T, a.B), st, a, w);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("live_plan");
  return OPE_OK;
}

}  // namespace ope
