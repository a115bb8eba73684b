// clip_grad_norm_ + Adam + Polyak over flat vectors.
//   torch.nn.utils.clip_grad_norm_(params, max_norm)   qmix.py:192   (coef = min(1, max/(norm+1e-6)))
//   torch.optim.Adam(lr, betas, eps, weight_decay)     qmix.py:71-72,193 ; MADDPGPolicy.py:53-55
//   soft_update / hard_update                          offpolicy/utils/util.py:123-145
// Two launches: (1) per-block partial sums of grad^2; (2) every block re-adds the partials in the same fixed
// order (deterministic), derives the clip coefficient and applies the update to its slice. Pure HBM streaming.
#include "ope_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 4;
constexpr int kPerBlock = kBlock * kPerThread;

__device__ __forceinline__ float block_sum(float v, float* sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int q = 0; q < kBlock / 64; ++q) t += sm[q];
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(kBlock) sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sm[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kPerBlock + threadIdx.x * kPerThread;
  float s = 0.f;
  if (base + 3 < n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + base);
    s = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  } else {
    for (int q = 0; q < kPerThread; ++q)
      if (base + q < n) s = fmaf(g[base + q], g[base + q], s);
  }
  const float t = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// beta^t for an integer t by repeated squaring in double (exact to a few ulp of double; a software pow() is thousands of
// instructions on the one thread every block waits for)
__device__ __forceinline__ double ipow(double b, int t) {
  double r = 1.0;
  while (t > 0) {
    if (t & 1) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

struct AdamK {
  float lr_t;        // lr / (1 - beta1^t)
  float inv_sqrt_bc2;  // 1 / sqrt(1 - beta2^t)
  float beta1, beta2, eps, max_norm, wd, tau, qden;
  int do_polyak;
  int nblocks;
  long long tail;   // index of [loss_sum, mask_count, qtot_sum] in g
  float lr;                 // with a device step counter the bias corrections are formed in the kernel
  long long skip_begin, skip_end;   // elements Adam leaves alone (grad-less tensors); Polyak still applies
  int* step_counter;        // [0] steps taken so far, [1] ticket of the blocks that are done with the current step
};

__global__ void __launch_bounds__(kBlock) adam_kernel(AdamK c, int64_t n, float* __restrict__ theta, float* __restrict__ tgt,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const float* __restrict__ g, const float* __restrict__ part,
                                                       float* __restrict__ stats) {
  __shared__ float sm[kBlock / 64];
  __shared__ float s_bc[2];
  __shared__ int s_t;
  if (c.step_counter) {
    if (threadIdx.x == 0) {
      s_t = c.step_counter[0] + 1;       // every block reads the count before it takes its ticket below
      s_bc[0] = (float)((double)c.lr / (1.0 - ipow((double)c.beta1, s_t)));
      s_bc[1] = (float)(1.0 / sqrt(1.0 - ipow((double)c.beta2, s_t)));
    }
    __syncthreads();
    c.lr_t = s_bc[0];
    c.inv_sqrt_bc2 = s_bc[1];
  }
  // This thread's four elements (n, every tensor start and the skip range are multiples of 4: all in or all out together), as
  // 16-byte accesses requested BEFORE the norm's block reduction: the update of 10^4..10^5 parameters is one memory round trip
  // plus a reduction, and an element-at-a-time loop with early exits serialised it into several.
  const int64_t base = (int64_t)blockIdx.x * kPerBlock + threadIdx.x * kPerThread;
  const bool live = base < n;
  const bool skip = live && base >= c.skip_begin && base < c.skip_end;      // grad-less tensor: torch's Adam does not touch it
  const int64_t lb = live ? base : 0;
  const f32x4 th4 = *reinterpret_cast<const f32x4*>(theta + lb);
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + lb);
  const f32x4 m4 = *reinterpret_cast<const f32x4*>(m + lb);
  const f32x4 v4 = *reinterpret_cast<const f32x4*>(v + lb);
  f32x4 tg4 = {0.f, 0.f, 0.f, 0.f};
  if (c.do_polyak) tg4 = *reinterpret_cast<const f32x4*>(tgt + lb);
  float s = 0.f;
  // (four partials requested per pass before the first add: a load / wait / add loop is one memory round trip per ~kBlock partials)
  for (int q0 = threadIdx.x; q0 < c.nblocks; q0 += 4 * kBlock) {
    float pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pv[u] = q0 + u * kBlock < c.nblocks ? part[q0 + u * kBlock] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += pv[u];
  }
  // fixed-order: each thread adds a fixed subset, then the same tree everywhere -> identical in all blocks
  const float tot = block_sum(s, sm);
  const float cnt = g[c.tail + 1];
  const float inv = 1.0f / cnt;
  const float norm = sqrtf(tot) * inv;
  const float coef = fminf(1.0f, c.max_norm / (norm + 1e-6f));
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats) {
    stats[0] = g[c.tail] * inv;     // loss
    stats[1] = norm;                // pre-clip global L2 norm
    stats[2] = g[c.tail + 2] / c.qden;   // mean of Q_tot*(1-mask) over ALL T*B steps
    stats[3] = cnt;
  }
  const float scale = coef * inv;
  if (live) {
    f32x4 out = th4;
    if (!skip) {
      f32x4 mo, vo;
#pragma unroll
      for (int q = 0; q < kPerThread; ++q) {
        float gi = g4[q] * scale;
        if (c.wd != 0.f) gi = fmaf(c.wd, th4[q], gi);
        mo[q] = c.beta1 * m4[q] + (1.0f - c.beta1) * gi;
        vo[q] = c.beta2 * v4[q] + (1.0f - c.beta2) * gi * gi;
        const float den = sqrtf(vo[q]) * c.inv_sqrt_bc2 + c.eps;
        out[q] = th4[q] - c.lr_t * (mo[q] / den);
      }
      *reinterpret_cast<f32x4*>(m + base) = mo;
      *reinterpret_cast<f32x4*>(v + base) = vo;
      *reinterpret_cast<f32x4*>(theta + base) = out;
    }
    if (c.do_polyak) {
#pragma unroll
      for (int q = 0; q < kPerThread; ++q) tg4[q] = tg4[q] * (1.0f - c.tau) + out[q] * c.tau;
      *reinterpret_cast<f32x4*>(tgt + base) = tg4;
    }
  }
  // Device step counter: the LAST block to finish publishes t and resets the ticket, so the update needs no separate
  // "count += 1" launch (and a captured graph of it replays with an advancing bias correction).
  if (c.step_counter && threadIdx.x == 0) {
    if (atomicAdd(&c.step_counter[1], 1) == (int)gridDim.x - 1) {
      c.step_counter[1] = 0;
      c.step_counter[0] = s_t;
    }
  }
}

__global__ void polyak_kernel(int64_t n, const float* __restrict__ theta, float* __restrict__ tgt, float tau) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  tgt[i] = (tau >= 1.0f) ? theta[i] : tgt[i] * (1.0f - tau) + theta[i] * tau;
}

}  // namespace

extern "C" int64_t ope_adam_scratch_floats(int64_t n) { return n < 1 ? OPE_EINVAL : ope_cdiv(n, kPerBlock) + 4; }

extern "C" int ope_adam_step(const ope_adam_cfg* cfg, int64_t n, float* theta, float* theta_tgt, float* adam_m, float* adam_v,
                             const float* grad, float* scratch, float* stats_out, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (!cfg || n < 1 || (n & 3) || !theta || !adam_m || !adam_v || !grad || !scratch) return OPE_EINVAL;
  if (cfg->do_polyak && !theta_tgt) return OPE_EINVAL;
  if ((cfg->skip_begin & 3) || (cfg->skip_end & 3)) return OPE_EINVAL;      // tensors start on multiples of 4 floats
  if (!cfg->step_counter && cfg->step < 1) return OPE_EINVAL;
  const int nb = ope_cdiv(n, kPerBlock);
  hipStream_t st = (hipStream_t)stream;
  const bool have_parts = cfg->sumsq_partials != nullptr && cfg->n_sumsq_partials > 0;
  if (!have_parts) {
    ope::kprof_work(0.0, 4.0 * (double)n);
    OPE_LAUNCH(sumsq_kernel, dim3(nb), dim3(kBlock), 0, st, grad, n, scratch);
    OPE_CHECK_LAUNCH();
  }
  AdamK c;
  const double tstep = cfg->step_counter ? 1.0 : (double)cfg->step;   // placeholder when the kernel reads the device counter
  const double bc1 = 1.0 - pow((double)cfg->beta1, tstep);
  const double bc2 = 1.0 - pow((double)cfg->beta2, tstep);
  c.lr_t = (float)((double)cfg->lr / bc1);
  c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  c.beta1 = cfg->beta1; c.beta2 = cfg->beta2; c.eps = cfg->eps; c.max_norm = cfg->max_grad_norm;
  c.wd = cfg->weight_decay; c.tau = cfg->tau; c.qden = cfg->qtot_denominator; c.do_polyak = cfg->do_polyak;
  c.nblocks = have_parts ? cfg->n_sumsq_partials : nb;
  c.lr = cfg->lr; c.step_counter = cfg->step_counter;
  c.skip_begin = cfg->skip_begin; c.skip_end = cfg->skip_end;
  c.tail = cfg->tail_offset > 0 ? cfg->tail_offset : n;
  ope::kprof_work(0.0, 4.0 * (double)n * (cfg->do_polyak ? 9.0 : 7.0));      // theta, m, v (+ target) read and written, grad read
  OPE_LAUNCH(adam_kernel, dim3(nb), dim3(kBlock), 0, st, c, n, theta, theta_tgt, adam_m, adam_v, grad,
                     have_parts ? cfg->sumsq_partials : scratch, stats_out);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_polyak(int64_t n, const float* theta, float* theta_tgt, float tau, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (n < 1 || !theta || !theta_tgt) return OPE_EINVAL;
  ope::kprof_work(0.0, 4.0 * 3.0 * (double)n);
  OPE_LAUNCH(polyak_kernel, dim3(ope_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, theta, theta_tgt, tau);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}
