// Replay store kernels: episode-major device rings, ring-write insert and batch gather.
//
// Replaces (reference): RecPolicyBuffer.insert's ring write  offpolicy/utils/rec_buffer.py:167-185
//                       RecPolicyBuffer.sample_inds           offpolicy/utils/rec_buffer.py:206-238
//
// Every field is a dense [episode][t][agent][dim] array (agent axis = 1 for share_obs / dones_env). One (episode,
// t) "item" is `chunk = NA*DD` contiguous floats in the store. The gather writes the batch as [t][agent][b][dim]:
// the memory behind the reference's [N, T(+1), B, dim] transpose view and, flattened, the [T(+1), N*B, dim]
// row-stacked tensor the trainer consumes (row = agent*B + b, qmix.py:108-109).
//
// HBM-bound byte movers: no arithmetic, 16-byte accesses whenever dim % 4 == 0, ~16 KB in flight per workgroup,
// grid >> 256 workgroups. Algorithmic bytes per gather = 2 * B * episode_bytes (read + write).
#include <stdlib.h>

#include "ope_common.h"

namespace {

constexpr int kFields = 7;
constexpr int kBlock = 256;
constexpr int kTargetFloats = 6144;  // ~24 KB of payload per workgroup

struct FieldDesc {
  const float* src;
  float* dst;
  int TT;              // time entries (T or T+1)
  int NA;              // agent axis (1 if none)
  int DD;              // innermost dim
  int rows_per_block;  // destination rows (segments of DD floats) per workgroup
  int block_begin;     // first blockIdx.x of this field
};
struct CopyArgs {
  FieldDesc f[kFields];
  int n_episodes;    // B (gather) or n_insert (insert)
  int total_blocks;
  int xcd_swizzle;   // G > 1: runs of G consecutive logical blocks share an XCD (and its L2)
};

// A "row" is the DD contiguous floats of one (episode, t, agent). Rows are numbered in DESTINATION order, so a block
// owns one contiguous destination range (whole 128-byte lines when the range is a multiple of B rows: for the
// gather, [t][a][0..B) is exactly B*DD floats) and pulls its rows from wherever they live in the source:
//   GATHER  dst row r = (t*NA + a)*E + b   <- src row (idx[b]*TT + t)*NA + a     (store -> batch)
//   INSERT  dst row r = (idx[e]*TT + t)*NA + a, numbered r = (e*TT + t)*NA + a  <- src row (t*E + e)*NA + a
template <bool GATHER>
__device__ __forceinline__ void row_ptrs(const FieldDesc& F, const int64_t* __restrict__ idx, int E, int r,
                                         const float*& sp, float*& dp) {
  if (GATHER) {
    const int ta = r / E, b = r - ta * E;
    const int t = ta / F.NA, a = ta - t * F.NA;
    sp = F.src + (((int64_t)idx[b] * F.TT + t) * F.NA + a) * F.DD;
    dp = F.dst + (int64_t)r * F.DD;
  } else {
    const int et = r / F.NA, a = r - et * F.NA;
    const int e = et / F.TT, t = et - e * F.TT;
    sp = F.src + (((int64_t)t * E + e) * F.NA + a) * F.DD;
    dp = F.dst + (((int64_t)idx[e] * F.TT + t) * F.NA + a) * F.DD;
  }
}

template <bool GATHER, int VEC>
__device__ __forceinline__ void copy_field(const FieldDesc& F, const int64_t* __restrict__ idx, int E, int blk) {
  constexpr int UNROLL = 8;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_rows = E * F.TT * F.NA;
  const int r0 = blk * F.rows_per_block;
  const int nr = min(F.rows_per_block, n_rows - r0);
  const int pieces = F.DD / VEC;          // VEC-wide pieces per row
  if (pieces < 32) {
    // short rows (acts, rewards, dones ...): flat (row, piece) space, one piece per thread per iteration
    const int total = nr * pieces;
    for (int x = threadIdx.x; x < total; x += kBlock) {
      const int rl = x / pieces, pc = x - rl * pieces;
      const float* sp; float* dp;
      row_ptrs<GATHER>(F, idx, E, r0 + rl, sp, dp);
      *reinterpret_cast<vec_t*>(dp + pc * VEC) = *reinterpret_cast<const vec_t*>(sp + pc * VEC);
    }
    return;
  }
  // long rows: a wave moves UNROLL rows at a time, 64 lanes striding over the pieces of each; all UNROLL loads of a
  // lane are issued before its first store
  for (int s0 = wave * UNROLL; s0 < nr; s0 += 4 * UNROLL) {
    const float* sp[UNROLL]; float* dp[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) row_ptrs<GATHER>(F, idx, E, r0 + min(s0 + u, nr - 1), sp[u], dp[u]);
    for (int pc = lane; pc < pieces; pc += 64) {
      vec_t v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const vec_t*>(sp[u] + pc * VEC);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (s0 + u < nr) *reinterpret_cast<vec_t*>(dp[u] + pc * VEC) = v[u];
    }
  }
}

template <bool GATHER>
__global__ void __launch_bounds__(kBlock) episode_copy_kernel(CopyArgs args, const int64_t* __restrict__ idx) {
  // Hardware deals workgroups round-robin over the 8 XCDs, each with a private L2. Neighbouring destination ranges read
  // neighbouring (line-sharing) source rows of the same episodes, so runs of G consecutive logical blocks are remapped onto
  // ONE XCD (the G blocks of a run are dispatched 8 apart, i.e. close in time), runs interleaved over the XCDs so that
  // every XCD still sees every field.
  int bid = blockIdx.x;
  const int G = args.xcd_swizzle;
  if (G > 1 && bid < (args.total_blocks / (8 * G)) * (8 * G)) {
    const int super = bid / (8 * G), w = bid - super * (8 * G);
    bid = (super * 8 + (w & 7)) * G + (w >> 3);
  }
  int f = 0;
#pragma unroll
  for (int i = 1; i < kFields; ++i)
    if (bid >= args.f[i].block_begin) f = i;
  const FieldDesc& F = args.f[f];
  const int blk = bid - F.block_begin;
  const int vec = (F.DD % 4 == 0) ? 4 : ((F.DD % 2 == 0) ? 2 : 1);
  if (vec == 4)
    copy_field<GATHER, 4>(F, idx, args.n_episodes, blk);
  else if (vec == 2)
    copy_field<GATHER, 2>(F, idx, args.n_episodes, blk);
  else
    copy_field<GATHER, 1>(F, idx, args.n_episodes, blk);
}

int build_args(const ope_dims* d, const ope_fields* src, const ope_fields* dst, int E, bool gather, CopyArgs* out) {
  if (!d || !src || !dst) return OPE_EINVAL;
  const int T = d->episode_length, N = d->n_agents, A = d->act_dim, D = d->obs_dim, S = d->state_dim;
  if (T < 1 || N < 1 || A < 1 || D < 1 || S < 1 || E < 1) return OPE_EINVAL;
  static const int target = getenv("OPE_GATHER_FLOATS") ? atoi(getenv("OPE_GATHER_FLOATS")) : kTargetFloats;
  const float* s[kFields] = {src->obs, src->share_obs, src->acts, src->rewards, src->dones, src->dones_env, src->avail_acts};
  float* t[kFields] = {dst->obs, dst->share_obs, dst->acts, dst->rewards, dst->dones, dst->dones_env, dst->avail_acts};
  const int TT[kFields] = {T + 1, T + 1, T, T, T, T, T + 1};
  const int NA[kFields] = {N, 1, N, N, N, 1, N};
  const int DD[kFields] = {D, S, A, 1, 1, 1, A};
  int blocks = 0;
  for (int i = 0; i < kFields; ++i) {
    FieldDesc& F = out->f[i];
    F.src = s[i];
    F.dst = t[i];
    F.TT = TT[i];
    F.NA = NA[i];
    F.DD = DD[i];
    F.block_begin = blocks;
    // whole groups of E rows (gather: one (t, agent) over all episodes = one contiguous, line-aligned output range)
    const int unit = gather ? E : NA[i];
    // short rows go through the flat one-piece-per-thread path: keep those blocks to ~1 piece per thread so that they
    // are not the tail of the launch
    const int vecw = (DD[i] % 4 == 0) ? 4 : ((DD[i] % 2 == 0) ? 2 : 1);
    const int tgt = (DD[i] / vecw < 32) ? kBlock * vecw : target;
    int groups = tgt / (unit * DD[i]);
    if (groups < 1) groups = 1;
    F.rows_per_block = groups * unit;
    if (s[i] == nullptr || t[i] == nullptr) continue;  // field not stored (e.g. no avail_acts): zero blocks
    blocks += ope_cdiv((int64_t)E * TT[i] * NA[i], F.rows_per_block);
  }
  // fields with no blocks must not capture any blockIdx: give them the begin of the next one
  for (int i = kFields - 1; i >= 0; --i)
    if (s[i] == nullptr || t[i] == nullptr) out->f[i].block_begin = (i + 1 < kFields) ? out->f[i + 1].block_begin : blocks;
  out->n_episodes = E;
  out->total_blocks = blocks;
  const char* sw = getenv("OPE_GATHER_XCD");
  out->xcd_swizzle = sw ? atoi(sw) : 8;   // G = 8: -21 % HBM read traffic at equal or better time
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Reward normalisation statistics (rec_buffer.py:209-222, mlp_buffer.py:229-231). Two launches, fixed summation order:
// kStatBlocks blocks accumulate (count, sum, sum of squares) in double over strided slices of the reward ring, one
// block folds the partials and writes {mean, population std, count}.
// ---------------------------------------------------------------------------------------------------------
constexpr int kStatBlocks = 512;

__global__ void __launch_bounds__(256) reward_stats_partial_kernel(const float* __restrict__ rewards, const float* __restrict__ dones_env,
                                                                   int64_t n, int T, int N, double* __restrict__ part) {
  __shared__ double red[256][3];
  double c = 0.0, s = 0.0, q = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    bool ok = true;
    if (dones_env) {   // step t of episode e counts unless the PREVIOUS step ended the episode
      const int64_t et = i / N;
      const int t = (int)(et % T);
      ok = (t == 0) || (dones_env[et - 1] != 1.0f);
    }
    if (ok) {
      const double r = (double)rewards[i];
      c += 1.0; s += r; q += r * r;
    }
  }
  red[threadIdx.x][0] = c; red[threadIdx.x][1] = s; red[threadIdx.x][2] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[threadIdx.x][k] += red[threadIdx.x + o][k];
    __syncthreads();
  }
  if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x];
}

__global__ void __launch_bounds__(64) reward_stats_final_kernel(const double* __restrict__ part, int nblocks, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double c = 0.0, s = 0.0, q = 0.0;
  for (int b = 0; b < nblocks; ++b) { c += part[b * 3]; s += part[b * 3 + 1]; q += part[b * 3 + 2]; }
  const double mean = s / c;
  double var = q / c - mean * mean;
  if (var < 0.0) var = 0.0;
  out[0] = (float)mean; out[1] = (float)sqrt(var); out[2] = (float)c; out[3] = 0.f;
}

__global__ void reward_normalize_kernel(float* __restrict__ r, int64_t n, const float* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = (r[i] - stats[0]) / stats[1];
}

}  // namespace

extern "C" int64_t ope_reward_stats_scratch_bytes(void) { return (int64_t)kStatBlocks * 3 * sizeof(double); }

extern "C" int ope_store_reward_stats(const ope_dims* dims, int32_t filled, const float* rewards, const float* dones_env,
                                      void* scratch, float* stats_out, void* stream) {
  (void)hipGetLastError();
  if (!dims || filled < 1 || !rewards || !scratch || !stats_out || dims->episode_length < 1 || dims->n_agents < 1) return OPE_EINVAL;
  const int64_t n = (int64_t)filled * dims->episode_length * dims->n_agents;
  int blocks = (int)((n + 255) / 256);
  if (blocks > kStatBlocks) blocks = kStatBlocks;
  hipLaunchKernelGGL(reward_stats_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rewards, dones_env, n,
                     dims->episode_length, dims->n_agents, (double*)scratch);
  OPE_CHECK_LAUNCH();
  hipLaunchKernelGGL(reward_stats_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)scratch, blocks, stats_out);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_reward_normalize(float* rewards, int64_t n, const float* stats, void* stream) {
  (void)hipGetLastError();
  if (!rewards || n < 1 || !stats) return OPE_EINVAL;
  hipLaunchKernelGGL(reward_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, n, stats);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int64_t ope_episode_bytes(const ope_dims* d) {
  if (!d) return OPE_EINVAL;
  const int64_t T = d->episode_length, N = d->n_agents, A = d->act_dim, D = d->obs_dim, S = d->state_dim;
  return 4 * ((T + 1) * N * D + (T + 1) * S + T * N * A + (T + 1) * N * A + T * N + T * N + T);
}

extern "C" int ope_store_gather(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds,
                                int32_t batch, const ope_fields* out, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (capacity < 1 || !inds) return OPE_EINVAL;
  CopyArgs args;
  int rc = build_args(dims, store, out, batch, true, &args);
  if (rc != OPE_OK) return rc;
  // with a hole in the middle (missing field) the "last begin <= bid" scan still works because holes alias the next begin
  hipLaunchKernelGGL(episode_copy_kernel<true>, dim3(args.total_blocks), dim3(kBlock), 0, (hipStream_t)stream, args, inds);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_insert(const ope_dims* dims, int32_t capacity, const ope_fields* store, const ope_fields* staged,
                                const int64_t* slots, int32_t n_insert, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (capacity < 1 || !slots) return OPE_EINVAL;
  CopyArgs args;
  int rc = build_args(dims, staged, store, n_insert, false, &args);
  if (rc != OPE_OK) return rc;
  hipLaunchKernelGGL(episode_copy_kernel<false>, dim3(args.total_blocks), dim3(kBlock), 0, (hipStream_t)stream, args, slots);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}
