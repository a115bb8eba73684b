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
#include "ope_common.h"

namespace {

constexpr int kFields = 7;
constexpr int kBlock = 256;
constexpr int kFloatsPerBlock = 16384;  // 64 KB of payload per workgroup

struct FieldDesc {
  const float* src;
  float* dst;
  int TT;           // time entries (T or T+1)
  int NA;           // agent axis (1 if none)
  int DD;           // innermost dim
  int items_per_block;
  int block_begin;  // first blockIdx.x of this field
};
struct CopyArgs {
  FieldDesc f[kFields];
  int n_episodes;    // B (gather) or n_insert (insert)
  int total_blocks;
};

// GATHER: src = store[inds[b]][t][a][d]          dst = out[t][a][b][d]
// INSERT: src = staged[t][e][a][d]               dst = store[slots[e]][t][a][d]
//
// Work unit = one "segment": the DD contiguous floats of one (episode, t, agent). Segments of a block are dealt to its
// 4 waves; a wave moves a segment with its 64 lanes striding over VEC-wide pieces, UNROLL segments at a time so that
// several independent 16-byte loads are in flight per lane before the first store. All index arithmetic is per
// segment (wave-uniform), none per element.
template <bool GATHER, int VEC>
__device__ __forceinline__ void copy_field(const FieldDesc& F, const int64_t* __restrict__ idx, int E, int blk) {
  constexpr int UNROLL = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_items = E * F.TT;
  const int item0 = blk * F.items_per_block;
  const int n_here = min(F.items_per_block, n_items - item0);
  const int nseg = n_here * F.NA;
  const int pieces = F.DD / VEC;          // VEC-wide pieces per segment
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  auto seg_ptrs = [&](int sg, const float*& sp, float*& dp) {
    const int il = sg / F.NA, a = sg - il * F.NA;
    const int item = item0 + il;
    const int b = item / F.TT, t = item - b * F.TT;
    if (GATHER) {
      sp = F.src + (((int64_t)idx[b] * F.TT + t) * F.NA + a) * F.DD;
      dp = F.dst + (((int64_t)t * F.NA + a) * E + b) * F.DD;
    } else {
      sp = F.src + (((int64_t)t * E + b) * F.NA + a) * F.DD;
      dp = F.dst + (((int64_t)idx[b] * F.TT + t) * F.NA + a) * F.DD;
    }
  };
  if (pieces <= 16) {
    // short segments (acts, rewards, dones ...): one lane per piece, 64/pieces... keep it simple: lane-per-piece over a
    // flattened (segment, piece) space of this block
    const int total = nseg * pieces;
    for (int x = threadIdx.x; x < total; x += kBlock) {
      const int sg = x / pieces, pc = x - sg * pieces;
      const float* sp; float* dp;
      seg_ptrs(sg, sp, dp);
      *reinterpret_cast<vec_t*>(dp + pc * VEC) = *reinterpret_cast<const vec_t*>(sp + pc * VEC);
    }
    return;
  }
  for (int s0 = wave * UNROLL; s0 < nseg; s0 += 4 * UNROLL) {
    const float* sp[UNROLL]; float* dp[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) seg_ptrs(min(s0 + u, nseg - 1), sp[u], dp[u]);
    for (int pc = lane; pc < pieces; pc += 64) {
      vec_t v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const vec_t*>(sp[u] + pc * VEC);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (s0 + u < nseg) *reinterpret_cast<vec_t*>(dp[u] + pc * VEC) = v[u];
    }
  }
}

template <bool GATHER>
__global__ void __launch_bounds__(kBlock) episode_copy_kernel(CopyArgs args, const int64_t* __restrict__ idx) {
  const int bid = blockIdx.x;
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

int build_args(const ope_dims* d, const ope_fields* src, const ope_fields* dst, int E, CopyArgs* out) {
  if (!d || !src || !dst) return OPE_EINVAL;
  const int T = d->episode_length, N = d->n_agents, A = d->act_dim, D = d->obs_dim, S = d->state_dim;
  if (T < 1 || N < 1 || A < 1 || D < 1 || S < 1 || E < 1) return OPE_EINVAL;
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
    const int chunk = NA[i] * DD[i];
    F.items_per_block = kFloatsPerBlock / chunk > 0 ? kFloatsPerBlock / chunk : 1;
    if (s[i] == nullptr || t[i] == nullptr) {  // field not stored (e.g. no avail_acts): zero blocks
      F.items_per_block = 1;
      continue;
    }
    blocks += ope_cdiv((int64_t)E * TT[i], F.items_per_block);
  }
  // fields with no blocks must not capture any blockIdx: give them the begin of the next one
  for (int i = kFields - 1; i >= 0; --i)
    if (s[i] == nullptr || t[i] == nullptr) out->f[i].block_begin = (i + 1 < kFields) ? out->f[i + 1].block_begin : blocks;
  out->n_episodes = E;
  out->total_blocks = blocks;
  return OPE_OK;
}

}  // namespace

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
  int rc = build_args(dims, store, out, batch, &args);
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
  int rc = build_args(dims, staged, store, n_insert, &args);
  if (rc != OPE_OK) return rc;
  hipLaunchKernelGGL(episode_copy_kernel<false>, dim3(args.total_blocks), dim3(kBlock), 0, (hipStream_t)stream, args, slots);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}
