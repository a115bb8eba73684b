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

#include <hip/hip_ext.h>

#include <algorithm>

#include "ope_common.h"
#include "ope_live_dev.h"
#include "ope_rng.h"

namespace {

constexpr int kFields = 8;
constexpr int kBlock = 256;
constexpr int kTargetFloats = 6144;  // insert (row path): ~24 KB of payload per workgroup
constexpr int kStepBytes = 16384;    // gather: a workgroup reads about this many contiguous bytes of ONE episode (round 6, 3s5z B = 32, same box: 8 KB 23.7 us / 22.3 copy-only,
                                     // 12 KB 20.6 / 20.3, 16 KB 21.1 / 20.3; round 2 had measured 8 KB best on the kernel of that time)
constexpr int kTileMax = 7168;       // short-row tiles staged in LDS: at most 28 KB (+ padding)

// tuning knobs (defaults = measured best; ope_set_gather_params / OPE_GATHER_* for A/B runs)
struct GatherTune {
  int floats = kStepBytes / 4;  // gather: contiguous floats of one episode per workgroup (whole time steps)
  int xcd = 8;                 // > 1: the B workgroups that write one [t][agent][0..B) range run on ONE XCD (one L2)
  int unroll = 8;              // 16-byte loads in flight per thread: 4, 8 or 16
  int nt = 0;                  // bit 0: non-temporal loads from the store, bit 1: non-temporal stores of the batch
  int small = 1;               // 1: short rows (dim < 128 floats) go through the LDS-transposing tile path
  int tile = 2048;             // floats per short-row tile (<= kTileMax)
  bool env_read = false;
};
GatherTune g_tune;

void read_env_once() {
  if (g_tune.env_read) return;
  g_tune.env_read = true;
  const char* e;
  if ((e = getenv("OPE_GATHER_FLOATS"))) g_tune.floats = atoi(e);
  if ((e = getenv("OPE_GATHER_XCD"))) g_tune.xcd = atoi(e);
  if ((e = getenv("OPE_GATHER_UNROLL"))) g_tune.unroll = atoi(e);
  if ((e = getenv("OPE_GATHER_NT"))) g_tune.nt = atoi(e);
  if ((e = getenv("OPE_GATHER_SMALL"))) g_tune.small = atoi(e);
  if ((e = getenv("OPE_GATHER_TILE"))) g_tune.tile = atoi(e);
  if (g_tune.tile < 256) g_tune.tile = 256;
  if (g_tune.tile > kTileMax) g_tune.tile = kTileMax;
  if (g_tune.floats < 64) g_tune.floats = 64;
  if (g_tune.floats > 65536) g_tune.floats = 65536;
}

}  // namespace
static const GatherTune& g_tune_defaults() { return g_tune; }
namespace {

struct FieldDesc {
  const float* src;
  float* dst;
  int TT;              // time entries (T or T+1)
  int NA;              // agent axis (1 if none)
  int DD;              // innermost dim
  int rows_per_block;  // insert (row path): destination rows per workgroup; gather tiles: (t, agent) rows per tile;
                       // gather steps: time steps per workgroup
  int block_begin;     // first blockIdx.x of this field
  int block_end;       // one past its last blockIdx.x (the range may be padded to a multiple of the XCD run)
  int block_count;     // workgroups of this field that have work
  int tiled;           // gather: 1 = LDS-transposing tile path, 0 = episode-contiguous step path
};
struct CopyArgs {
  FieldDesc f[kFields];
  int n_episodes;    // B (gather) or n_insert (insert)
  int capacity;      // ring slots: every index must lie in [0, capacity)
  int total_blocks;
  int xcd_swizzle;
  int unroll;
  int nt;
  int* bad_index;    // device flag: set to 1 if an index was out of range (the offending rows are skipped)
  int lds_bytes;     // dynamic LDS the tile fields need
  int64_t* idx_out;  // gather, optional: the episode indices as used, for consumers that read rows of the store themselves (ope_obs_ref)
  // gather, optional (ope_store_gather_attach_live): `live_blocks` rider workgroups IN FRONT of the copy's build the live-row plan of this
  // batch (LivePlan, ope_common.h) from the STORE's dones_env through the sampled slots -- the plan of the training step that follows, off
  // its critical path (as a launch of its own it costs the step ~9 us)
  int live_blocks, live_T, live_N;
  ope::LiveW live_w;
  // ... and (ope_live_target.copy_live_only) the step path leaves the time entries at and behind len_b of the sampled episode unwritten: the
  // rows the live-row step never reads (every (t, b) there is multiplied by a zero mask in the loss, qmix.py:161-166)
  int live_skip;
};
// the store's termination flags of the sampled episodes: [capacity][T][1]
template <class IDX>
struct StoreDones {
  const float* ring; const IDX* idx; int capacity, T;
  __device__ __forceinline__ void prepare(int* keys, int B) const {      // one wave-uniform index per iteration: scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    for (int b = wave; b < B; b += nw) {
      const int64_t slot = (*idx)[b];
      if ((threadIdx.x & 63) == 0) keys[b] = (slot < 0 || slot >= capacity) ? -1 : (int)slot;
    }
  }
  __device__ __forceinline__ int key(const int* keys, int b) const { return keys[b]; }
  __device__ __forceinline__ float at(int slot, int t) const {      // (an index out of range: the copy raises the flag; here the episode counts as over)
    const float x = ring[(int64_t)(slot < 0 ? 0 : slot) * T + t];
    return slot < 0 ? 1.0f : x;
  }
};

// Where the episode indices come from: device memory (device-resident index tensors: prioritized sampling on the device,
// HIP-graph replays) or the kernel-argument block itself (host index arrays: no upload, no copy-engine -> compute
// dependency in front of the launch, and the per-workgroup index becomes a scalar load).
constexpr int kMaxArgIdx = 512;
struct DevIdx {
  const int64_t* p;
  __device__ __forceinline__ int64_t operator[](int i) const { return p[i]; }
};
struct ArgIdx {
  int32_t v[kMaxArgIdx];
  __device__ __forceinline__ int64_t operator[](int i) const { return (int64_t)v[i]; }
};

// Uniform sampling with replacement drawn where the data lives (np.random.choice(filled, batch) of the reference's sample(),
// rec_buffer.py:86 / mlp_buffer.py:74): index i of the batch = Philox4x32-10(seed; i, stream 2, *counter) scaled to [0, filled).
// Every workgroup that needs index i recomputes it (40 integer instructions) instead of reading an index tensor, so there is no
// index upload, no host round trip per step, and a captured graph replays with fresh indices as the device counter advances.
struct RngIdx {
  uint64_t seed;
  const int32_t* counter;
  uint32_t filled;
  const int32_t* filled_dev;   // optional DEVICE count of filled slots, read at run time (a captured graph sees later inserts)
  uint32_t capacity;
  int64_t* out;      // optional: the drawn indices, for the caller (priorities, inspection)
  __device__ __forceinline__ int64_t operator[](int i) const {
    const ope::Philox4 x = ope::philox4x32_10((uint32_t)i, 0u, (uint32_t)(counter ? counter[0] : 0), 2u, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t f = filled;
    if (filled_dev) {
      const int32_t fd = filled_dev[0];
      f = fd < 1 ? 1u : ((uint32_t)fd > capacity ? capacity : (uint32_t)fd);
    }
    const int64_t v = (int64_t)(((uint64_t)x.v[0] * (uint64_t)f) >> 32);
    if (out) out[i] = v;
    return v;
  }
};

template <class IDX>
__device__ __forceinline__ int64_t checked_index(const IDX& idx, int i, int capacity, int* bad) {
  const int64_t v = idx[i];
  if (v < 0 || v >= capacity) {
    if (bad) *bad = 1;
    return -1;
  }
  return v;
}

// A "row" is the DD contiguous floats of one (episode, t, agent). Rows are numbered in DESTINATION order, so a block
// owns one contiguous destination range (whole 128-byte lines when the range is a multiple of B rows: for the
// gather, [t][a][0..B) is exactly B*DD floats) and pulls its rows from wherever they live in the source:
//   GATHER  dst row r = (t*NA + a)*E + b   <- src row (idx[b]*TT + t)*NA + a     (store -> batch)
//   INSERT  dst row r = (idx[e]*TT + t)*NA + a, numbered r = (e*TT + t)*NA + a  <- src row (t*E + e)*NA + a
// Returns false (row skipped) if the episode index is out of range.
template <bool GATHER, class IDX>
__device__ __forceinline__ bool row_ptrs(const FieldDesc& F, const CopyArgs& A, const IDX& idx, int E, int r,
                                         const float*& sp, float*& dp) {
  if (GATHER) {
    const int ta = r / E, b = r - ta * E;
    const int t = ta / F.NA, a = ta - t * F.NA;
    const int64_t e = checked_index(idx, b, A.capacity, A.bad_index);
    sp = F.src + ((((e < 0 ? 0 : e)) * F.TT + t) * F.NA + a) * F.DD;
    dp = F.dst + (int64_t)r * F.DD;
    return e >= 0;
  } else {
    const int et = r / F.NA, a = r - et * F.NA;
    const int e = et / F.TT, t = et - e * F.TT;
    const int64_t slot = checked_index(idx, e, A.capacity, A.bad_index);
    sp = F.src + (((int64_t)t * E + e) * F.NA + a) * F.DD;
    dp = F.dst + ((((slot < 0 ? 0 : slot)) * F.TT + t) * F.NA + a) * F.DD;
    return slot >= 0;
  }
}

template <int VEC, bool NT>
__device__ __forceinline__ float __attribute__((ext_vector_type(VEC))) ld_vec(const float* p) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p));
  return *reinterpret_cast<const vec_t*>(p);
}
template <int VEC, bool NT>
__device__ __forceinline__ void st_vec(float* p, float __attribute__((ext_vector_type(VEC))) v) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(p));
  else *reinterpret_cast<vec_t*>(p) = v;
}

// long rows: a wave moves UNROLL rows at a time, 64 lanes striding over the pieces of each; all UNROLL loads of a lane
// are issued before its first store
template <bool GATHER, int VEC, int UNROLL, bool NTL, bool NTS, class IDX>
__device__ __forceinline__ void copy_rows(const FieldDesc& F, const CopyArgs& A, const IDX& idx, int E, int blk) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_rows = E * F.TT * F.NA;
  const int r0 = blk * F.rows_per_block;
  const int nr = min(F.rows_per_block, n_rows - r0);
  const int pieces = F.DD / VEC;          // VEC-wide pieces per row
  if (pieces < 32) {
    // short rows without a tile path (odd sizes): flat (row, piece) space, one piece per thread per iteration
    const int total = nr * pieces;
    for (int x = threadIdx.x; x < total; x += kBlock) {
      const int rl = x / pieces, pc = x - rl * pieces;
      const float* sp; float* dp;
      if (row_ptrs<GATHER>(F, A, idx, E, r0 + rl, sp, dp)) st_vec<VEC, NTS>(dp + pc * VEC, ld_vec<VEC, NTL>(sp + pc * VEC));
    }
    return;
  }
  for (int s0 = wave * UNROLL; s0 < nr; s0 += 4 * UNROLL) {
    const float* sp[UNROLL]; float* dp[UNROLL]; bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) ok[u] = row_ptrs<GATHER>(F, A, idx, E, r0 + min(s0 + u, nr - 1), sp[u], dp[u]) && (s0 + u < nr);
    for (int pc = lane; pc < pieces; pc += 64) {
      vec_t v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = ld_vec<VEC, NTL>(sp[u] + pc * VEC);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (ok[u]) st_vec<VEC, NTS>(dp[u] + pc * VEC, v[u]);
    }
  }
}

// Short rows (acts, avail_acts: A floats; rewards, dones, dones_env: 1 float): a (t, agent) row of ONE episode is a few
// bytes, but the KR consecutive rows [ta0, ta0+KR) of an episode are one contiguous run of L = KR*DD floats in the store,
// and the same rows of all E episodes are one contiguous run of KR*E*DD floats in the batch ([ta][b][dim]). So a
// workgroup moves one such tile through LDS: E coalesced runs in (16-byte loads when the run is 16-byte aligned), LDS as
// [b][L] with an odd row stride (conflict-free for the transposing reads), one contiguous run out in 16-byte stores
// (gather only: the staged side of an insert is [t][episode][agent][dim], not [t][agent][episode][dim], and inserts are
// off the training step -- they keep the row path).
// x / d for 0 <= x < 2^20 via a float reciprocal: (x + 0.5) / d is at least 0.5/d away from an integer and the float
// error is below (x/d) * 2^-22, so the truncation is exact (checked exhaustively for x < 2^20 over d = 1..600 and a set of
// larger d). A runtime integer division costs ~40 instructions; every use here keeps x below 2^17.
__device__ __forceinline__ int fast_div(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }

template <class IDX>
__device__ __forceinline__ void gather_tile(const FieldDesc& F, const CopyArgs& A, const IDX& idx, int E, int blk, float* lds) {
  const int R = F.TT * F.NA;                   // (t, agent) rows per episode
  const int ta0 = blk * F.rows_per_block;
  const int KR = min(F.rows_per_block, R - ta0);
  const int DD = F.DD;
  const int L = KR * DD;                       // floats per episode in this tile
  const int LS = L | 1;                        // odd LDS row stride
  const int tid = threadIdx.x;
  const float* ring = F.src;                   // E runs of L contiguous floats, run e at ring[(slot_e * R + ta0) * DD]
  float* flat = F.dst + (int64_t)ta0 * E * DD;  // ONE run of KR*E*DD floats, element (k, e, j) at (k*E + e)*DD + j
  const bool ring_al = (((int64_t)R * DD) % 4 == 0) && (((int64_t)ta0 * DD) % 4 == 0) && ((reinterpret_cast<uintptr_t>(ring) & 15) == 0);
  const bool flat_al = (((int64_t)ta0 * E * DD) % 4 == 0) && ((reinterpret_cast<uintptr_t>(F.dst) & 15) == 0);
  const int total = E * L;
  const float inv_dd = 1.0f / (float)DD, inv_e = 1.0f / (float)E;
  // phase 1: ring -> LDS[e][x]
  const int L4 = ring_al ? (L >> 2) : 0;       // whole float4 pieces per run
  if (L4 > 0) {
    const float inv_l4 = 1.0f / (float)L4;
    for (int q = tid; q < E * L4; q += kBlock) {
      const int e = fast_div(q, inv_l4), x4 = q - e * L4;
      const int64_t slot = checked_index(idx, e, A.capacity, A.bad_index);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (slot >= 0) v = *reinterpret_cast<const f32x4*>(ring + (slot * R + ta0) * DD + 4 * x4);
      float* d = lds + e * LS + 4 * x4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  }
  const int rem0 = 4 * L4, rem = L - rem0;     // scalar tail of every run (everything if unaligned)
  if (rem > 0) {
    const float inv_rem = 1.0f / (float)rem;
    for (int q = tid; q < E * rem; q += kBlock) {
      const int e = fast_div(q, inv_rem), x = rem0 + (q - e * rem);
      const int64_t slot = checked_index(idx, e, A.capacity, A.bad_index);
      lds[e * LS + x] = slot >= 0 ? ring[(slot * R + ta0) * DD + x] : 0.f;
    }
  }
  __syncthreads();
  // phase 2: LDS -> flat, 4 consecutive output floats per thread
  const int T4 = flat_al ? (total >> 2) : 0;
  for (int q = tid; q < T4; q += kBlock) {
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      const int ke = fast_div(o, inv_dd), j = o - ke * DD;
      const int k = fast_div(ke, inv_e), e = ke - k * E;
      v[r] = lds[e * LS + k * DD + j];
    }
    *reinterpret_cast<f32x4*>(flat + 4 * q) = v;
  }
  for (int o = 4 * T4 + tid; o < total; o += kBlock) {
    const int ke = fast_div(o, inv_dd), j = o - ke * DD;
    const int k = fast_div(ke, inv_e), e = ke - k * E;
    flat[o] = lds[e * LS + k * DD + j];
  }
}

// Gather, long rows (and short rows when the tile path is off): a workgroup = (KT consecutive time steps, ONE sampled
// episode b). Its source is ONE contiguous, line-aligned run of KT*NA*DD floats of episode idx[b] (an episode-major store
// makes every time step of an episode contiguous over agents and features), read with every lane busy; its destination is
// KT*NA rows of DD floats at [t][agent][b]. Workgroups are numbered (time-block major, b minor), and the XCD remap in the
// kernel puts the B workgroups of one time block -- whose destination rows are neighbours in memory and share their
// boundary cache lines -- on the same XCD, so the partial lines meet in one L2. Measured against the round-1 mapping
// (workgroup = one contiguous destination range, 1008-byte source rows from B different episodes): 3s5z obs 17.2 -> 14.5 us
// (5.4 TB/s, 96 % of a plain contiguous copy of the same bytes), MMM2 obs (8-byte vectors) 183 -> 130 us
// (tools/microbench_gather.hip, profiles/r02_microbench_gather_*.txt).
template <int VEC, int UNROLL, bool NTL, bool NTS, bool SKIP, class IDX>
__device__ __forceinline__ void gather_steps(const FieldDesc& F, const CopyArgs& A, const IDX& idx, int E, int blk, int EXTRA = 0) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int KT = F.rows_per_block;
  const int tb = blk / E, b = blk - tb * E;
  const int t0 = tb * KT;
  int nt = min(KT, F.TT - t0);
  const int64_t e = checked_index(idx, b, A.capacity, A.bad_index);
  if (e < 0) return;
  if (SKIP) {
    // len_b exactly as the plan computes it (live_plan_body: 2 + the last t with dones_env[t, b] != 1, 1 if there is none), from the flags at
    // and behind t0 - 1 only: ONE batch of independent loads per wave in front of the copy's (every wave for itself: no barrier)
    const float* fl = A.f[5].src + e * A.live_T;
    const int lane = threadIdx.x & 63;
    int last = -1;      // (anything before t0 - 1 changes nothing: the block is live then anyway)
    for (int base = max(t0 - 1, 0); base < A.live_T; base += 64) {
      const int t = base + lane;
      const unsigned long long m = __ballot(t < A.live_T && fl[t] != 1.0f);
      if (m) last = base + 63 - __builtin_clzll(m);
    }
    // (EXTRA = 1, the state: the chain kernels evaluate the target mixer of the last live (t, b) on state entry t + 1 = len_b and multiply
    // the result by 1 - dones_env[t] = 0 -- the value must be the store's (finite), not whatever the destination held)
    nt = min(nt, max(last + 2, 1) + EXTRA - t0);
    if (nt <= 0) return;
  }
  const int pieces = F.DD / VEC, total = nt * F.NA * pieces;
  const float* sbase = F.src + ((int64_t)e * F.TT + t0) * F.NA * F.DD;
  float* dbase = F.dst + ((int64_t)t0 * F.NA * E + b) * F.DD;      // row (k = (t - t0)*NA + a) at dbase + k*E*DD
  const float inv_p = 1.0f / (float)pieces;
  const int64_t row_stride = (int64_t)E * F.DD;
  for (int q0 = threadIdx.x; q0 < total; q0 += kBlock * UNROLL) {
    vec_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = ld_vec<VEC, NTL>(sbase + (int64_t)min(q0 + kBlock * u, total - 1) * VEC);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int q = q0 + kBlock * u;
      if (q < total) {
        const int row = fast_div(q, inv_p), pc = q - row * pieces;
        st_vec<VEC, NTS>(dbase + row * row_stride + pc * VEC, v[u]);
      }
    }
  }
}

template <bool GATHER, int UNROLL, bool NTL, bool NTS, class IDX>
__device__ __forceinline__ void copy_dispatch(const FieldDesc& F, const CopyArgs& A, const IDX& idx, int blk) {
  const int vec = (F.DD % 4 == 0) ? 4 : ((F.DD % 2 == 0) ? 2 : 1);
  if (vec == 4)
    copy_rows<GATHER, 4, UNROLL, NTL, NTS>(F, A, idx, A.n_episodes, blk);
  else if (vec == 2)
    copy_rows<GATHER, 2, UNROLL, NTL, NTS>(F, A, idx, A.n_episodes, blk);
  else
    copy_rows<GATHER, 1, UNROLL, NTL, NTS>(F, A, idx, A.n_episodes, blk);
}

template <bool GATHER, int UNROLL, int NT, class IDX>
__global__ void __launch_bounds__(kBlock) episode_copy_kernel(CopyArgs args, IDX idx) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // sized by the launch: what the tile fields need, else nothing
  // Hardware deals workgroups round-robin over the 8 XCDs, each with a private L2. Neighbouring destination ranges read
  // neighbouring (line-sharing) source rows of the same episodes, so runs of G consecutive logical blocks are remapped onto
  // ONE XCD (the G blocks of a run are dispatched 8 apart, i.e. close in time), runs interleaved over the XCDs so that
  // every XCD still sees every field.
  int bid = blockIdx.x;
  if (GATHER && args.live_blocks > 0) {
    // riders: the launch's FIRST workgroups (dispatched at once: their two batches of loads are out before the copy saturates the memory
    // system); a multiple of 8, so the copy's workgroups keep their XCD (blockIdx.x % 8)
    if (bid < args.live_blocks) {
      StoreDones<IDX> dn{args.f[5].src, &idx, args.capacity, args.live_T};
      ope::live_plan_body<true, 20>(args.live_w, nullptr, nullptr, 0, args.live_T, args.live_N, args.n_episodes, dn, reinterpret_cast<int*>(lds), bid, args.live_blocks);
      return;
    }
    bid -= args.live_blocks;
  }
  if (GATHER && args.idx_out && bid == 0)
    for (int i = threadIdx.x; i < args.n_episodes; i += kBlock) args.idx_out[i] = idx[i];
  const int G = args.xcd_swizzle;     // gather: B (runs of B consecutive logical workgroups on one XCD); insert: the knob
  if (G > 1 && bid < (args.total_blocks / (8 * G)) * (8 * G)) {
    const int super = bid / (8 * G), w = bid - super * (8 * G);
    bid = (super * 8 + (w & 7)) * G + (w >> 3);
  }
  int f = 0;
#pragma unroll
  for (int i = 0; i < kFields; ++i)
    if (bid >= args.f[i].block_begin && bid < args.f[i].block_end) f = i;
  const FieldDesc& F = args.f[f];
  const int blk = bid - F.block_begin;
  if (GATHER) {
    if (blk >= F.block_count) return;            // padding that keeps the fields' XCD runs aligned
    if (F.tiled) {
      gather_tile(F, args, idx, args.n_episodes, blk, lds);
      return;
    }
    const int vec = (F.DD % 4 == 0) ? 4 : ((F.DD % 2 == 0) ? 2 : 1);
    if (args.live_skip && f != 5 && f != 7) {      // (the flags themselves and the transition buffers' extra field are copied whole)
      const int extra = f == 1 ? 1 : 0;
      if (vec == 4) gather_steps<4, UNROLL, (NT & 1) != 0, (NT & 2) != 0, true>(F, args, idx, args.n_episodes, blk, extra);
      else if (vec == 2) gather_steps<2, UNROLL, (NT & 1) != 0, (NT & 2) != 0, true>(F, args, idx, args.n_episodes, blk, extra);
      else gather_steps<1, UNROLL, (NT & 1) != 0, (NT & 2) != 0, true>(F, args, idx, args.n_episodes, blk, extra);
      return;
    }
    if (vec == 4) gather_steps<4, UNROLL, (NT & 1) != 0, (NT & 2) != 0, false>(F, args, idx, args.n_episodes, blk);
    else if (vec == 2) gather_steps<2, UNROLL, (NT & 1) != 0, (NT & 2) != 0, false>(F, args, idx, args.n_episodes, blk);
    else gather_steps<1, UNROLL, (NT & 1) != 0, (NT & 2) != 0, false>(F, args, idx, args.n_episodes, blk);
    return;
  }
  copy_dispatch<GATHER, UNROLL, (NT & 1) != 0, (NT & 2) != 0>(F, args, idx, blk);
}

// Per-dispatch timing of the gather (bench.py's roofline leg): when enabled, every gather is launched with
// hipExtLaunchKernel's start / stop events, which time the DISPATCH itself (what rocprofv3's kernel trace reports) instead of
// the interval between two event markers around it (that interval carries two command-processor boundaries, ~4.7 us).
// ope_store_gather_attach_live: the target of the NEXT gather launched from this thread (consumed by it, whatever its outcome)
thread_local ope_live_target g_live_next;
thread_local bool g_live_pending = false;
constexpr int kProfRing = 512;
struct GatherProf {
  bool on = false;
  hipEvent_t ev[kProfRing][2];
  bool made = false;
  int n = 0;           // pairs recorded since the last read (capped at kProfRing)
  int stride = 1;      // every stride-th gather launch carries the events (the pair costs the step ~3.5 us: a 1 % tax on a timed region that times all of them)
  unsigned seen = 0;   // gather launches since profiling was switched on
};
GatherProf g_prof;
using ope::g_kprof_on;
using ope::kprof_work;


template <bool GATHER, class IDX>
void launch_copy(const CopyArgs& args0, const IDX& idx, hipStream_t st) {
  CopyArgs args = args0;
  args.live_blocks = 0;
  args.live_skip = 0;
  size_t lds = (size_t)args.lds_bytes;
  if (GATHER && g_live_pending) {
    g_live_pending = false;
    const ope_live_target& t = g_live_next;
    const FieldDesc& de = args.f[5];      // the store's dones_env ring
    if (t.plan && de.src && de.NA == 1 && de.DD == 1 && de.TT == t.episode_length && args.n_episodes == t.batch && ope::live_plan_shape_ok(de.TT, t.n_agents, t.batch)) {
      args.live_blocks = (ope::live_plan_blocks(de.TT, t.n_agents, t.batch) + 7) & ~7;
      args.live_T = de.TT; args.live_N = t.n_agents;
      args.live_w = ope::live_views(t.plan, de.TT, t.n_agents, t.batch);
      lds = std::max(lds, (size_t)4 * ope::live_lds_ints(de.TT, t.batch));
      args.live_skip = t.copy_live_only ? 1 : 0;
    }
  }
  const dim3 grid(args.total_blocks + args.live_blocks), block(kBlock);
  if (GATHER && g_prof.on && g_prof.n < kProfRing && args.unroll == 8 && args.nt == 0 && (g_prof.seen++ % (unsigned)g_prof.stride) == 0) {
    hipEvent_t e0 = g_prof.ev[g_prof.n][0], e1 = g_prof.ev[g_prof.n][1];
    ++g_prof.n;
    hipExtLaunchKernelGGL((episode_copy_kernel<GATHER, 8, 0, IDX>), grid, block, (uint32_t)lds, st, e0, e1, 0, args, idx);
    return;
  }
  if (g_kprof_on) {     // every byte of the E episodes once in, once out
    double b = 0;
    for (int q = 0; q < kFields; ++q)
      if (args.f[q].src && args.f[q].dst) b += 4.0 * args.f[q].TT * (double)args.f[q].NA * args.f[q].DD;
    kprof_work(0.0, 2.0 * b * args.n_episodes);
  }
  // one instantiation per variant: a single kernel holding all of them would be allocated the registers of the largest
  if (args.unroll == 4) OPE_LAUNCH((episode_copy_kernel<GATHER, 4, 0, IDX>), grid, block, lds, st, args, idx);
  else if (args.unroll == 16) OPE_LAUNCH((episode_copy_kernel<GATHER, 16, 0, IDX>), grid, block, lds, st, args, idx);
  else if (args.nt == 1) OPE_LAUNCH((episode_copy_kernel<GATHER, 8, 1, IDX>), grid, block, lds, st, args, idx);
  else if (args.nt == 2) OPE_LAUNCH((episode_copy_kernel<GATHER, 8, 2, IDX>), grid, block, lds, st, args, idx);
  else if (args.nt == 3) OPE_LAUNCH((episode_copy_kernel<GATHER, 8, 3, IDX>), grid, block, lds, st, args, idx);
  else OPE_LAUNCH((episode_copy_kernel<GATHER, 8, 0, IDX>), grid, block, lds, st, args, idx);
}

int build_args(const ope_dims* d, const ope_fields* src, const ope_fields* dst, int E, int capacity, bool gather, CopyArgs* out,
               const ope_gather_tune* over = nullptr) {
  if (!d || !src || !dst) return OPE_EINVAL;
  const int T = d->episode_length, N = d->n_agents, A = d->act_dim, D = d->obs_dim, S = d->state_dim;
  if (T < 1 || N < 1 || A < 1 || D < 1 || S < 1 || E < 1) return OPE_EINVAL;
  read_env_once();
  // the process defaults (environment, ope_set_gather_params), overlaid with the caller's per-call knobs: two stores in one process need not share them
  GatherTune g_tune = ::g_tune_defaults();
  if (over) {
    if (over->tile_floats > 0) g_tune.tile = over->tile_floats < 256 ? 256 : (over->tile_floats > kTileMax ? kTileMax : over->tile_floats);
    if (over->floats_per_block > 0) g_tune.floats = over->floats_per_block < 64 ? 64 : (over->floats_per_block > 65536 ? 65536 : over->floats_per_block);
    if (over->xcd_run > 0) g_tune.xcd = over->xcd_run;
    if (over->unroll == 4 || over->unroll == 8 || over->unroll == 16) g_tune.unroll = over->unroll;
    if (over->nontemporal > 0) g_tune.nt = (over->nontemporal - 1) & 3;
    if (over->small_tiles > 0) g_tune.small = over->small_tiles == 1 ? 1 : 0;
  }
  // a launch that stops at each episode's termination (ope_live_target.copy_live_only) pays one dependent round trip for the flags in front of a
  // workgroup's copy: four times the bytes per workgroup amortise it (3s5z, B = 32: 26.8 us at 2 048 floats, 20.2 at 4 096, 19.2 at 6 144, 19.9 at 8 192, 21.9 at
  // 16 384; the whole batch: 23.6 / 21.5 / 23.6 / 23.4)
  if (gather && g_live_pending && g_live_next.copy_live_only && !(over && over->floats_per_block > 0) && !getenv("OPE_GATHER_FLOATS")) g_tune.floats = 6144;
  const float* s[kFields] = {src->obs, src->share_obs, src->acts, src->rewards, src->dones, src->dones_env, src->avail_acts,
                             src->valid_transition};
  float* t[kFields] = {dst->obs, dst->share_obs, dst->acts, dst->rewards, dst->dones, dst->dones_env, dst->avail_acts,
                       dst->valid_transition};
  const int TT[kFields] = {T + 1, T + 1, T, T, T, T, T + 1, T};
  const int NA[kFields] = {N, 1, N, N, N, 1, N, N};
  const int DD[kFields] = {D, S, A, 1, 1, 1, A, 1};
  int blocks = 0, lds_bytes = 0;
  const int G = gather ? ((g_tune.xcd > 1) ? E : 1) : g_tune.xcd;
  // gather: short-row (tile) fields first -- their workgroups run two dependent phases and must not be the tail of the
  // launch -- then the step-path fields in size order; every field's range is padded to a multiple of the XCD run G
  int order[kFields] = {0, 1, 2, 3, 4, 5, 6, 7};
  if (gather) {
    int n = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < kFields; ++i) {
        const int vecw = (DD[i] % 4 == 0) ? 4 : ((DD[i] % 2 == 0) ? 2 : 1);
        const bool tile = g_tune.small && DD[i] / vecw < 32 && E <= 512 && (int64_t)E * DD[i] <= g_tune.tile;
        if ((pass == 0) == tile) order[n++] = i;
      }
  }
  for (int oi = 0; oi < kFields; ++oi) {
    const int i = order[oi];
    FieldDesc& F = out->f[i];
    F.src = s[i];
    F.dst = t[i];
    F.TT = TT[i];
    F.NA = NA[i];
    F.DD = DD[i];
    F.block_begin = blocks;
    F.block_end = blocks;
    F.block_count = 0;
    F.tiled = 0;
    const int vecw = (DD[i] % 4 == 0) ? 4 : ((DD[i] % 2 == 0) ? 2 : 1);
    const bool present = s[i] != nullptr && t[i] != nullptr;   // field not stored (e.g. no avail_acts): zero blocks
    if (gather) {
      if (g_tune.small && DD[i] / vecw < 32 && E <= 512 && (int64_t)E * DD[i] <= g_tune.tile) {
        // short rows: LDS-transposing tiles of KR (t, agent) rows x all E episodes
        int KR = g_tune.tile / (E * DD[i]);
        if (KR > 256) KR = 256;
        // keep the tile starts 16-byte aligned on both sides whenever the dimensions allow: KR*DD % 4 == 0
        while (KR > 1 && ((int64_t)KR * DD[i]) % 4 != 0) --KR;
        F.tiled = 1;
        F.rows_per_block = KR;
        if (present) {
          F.block_count = ope_cdiv((int64_t)TT[i] * NA[i], KR);
          const int need = E * ((KR * DD[i]) | 1) * (int)sizeof(float);
          if (need > lds_bytes) lds_bytes = need;
        }
      } else {
        // whole time steps of one episode, about g_tune.floats contiguous floats per workgroup
        const int step = NA[i] * DD[i];
        int KT = (g_tune.floats + step / 2) / step;
        if (KT < 1) KT = 1;
        if (KT > TT[i]) KT = TT[i];
        F.rows_per_block = KT;
        if (present) F.block_count = ope_cdiv(TT[i], KT) * E;
      }
      blocks += ope_cdiv(F.block_count, G) * G;
      F.block_end = blocks;
      continue;
    }
    // insert: whole groups of NA rows
    const int unit = NA[i];
    // short rows go through the flat one-piece-per-thread path: keep those blocks to ~1 piece per thread so that they
    // are not the tail of the launch
    const int tgt = (DD[i] / vecw < 32) ? kBlock * vecw : kTargetFloats;
    int groups = tgt / (unit * DD[i]);
    if (groups < 1) groups = 1;
    F.rows_per_block = groups * unit;
    F.block_end = blocks;
    if (!present) continue;
    F.block_count = ope_cdiv((int64_t)E * TT[i] * NA[i], F.rows_per_block);
    blocks += F.block_count;
    F.block_end = blocks;
  }
  out->n_episodes = E;
  out->capacity = capacity;
  out->total_blocks = blocks;
  out->xcd_swizzle = G;
  out->unroll = g_tune.unroll;
  out->nt = g_tune.nt;
  out->bad_index = nullptr;
  out->idx_out = nullptr;
  out->lds_bytes = (lds_bytes + 15) & ~15;
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
  OPE_LAUNCH(reward_stats_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rewards, dones_env, n,
                     dims->episode_length, dims->n_agents, (double*)scratch);
  OPE_CHECK_LAUNCH();
  OPE_LAUNCH(reward_stats_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)scratch, blocks, stats_out);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_reward_normalize(float* rewards, int64_t n, const float* stats, void* stream) {
  (void)hipGetLastError();
  if (!rewards || n < 1 || !stats) return OPE_EINVAL;
  OPE_LAUNCH(reward_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, n, stats);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int64_t ope_episode_bytes(const ope_dims* d) {
  if (!d) return OPE_EINVAL;
  const int64_t T = d->episode_length, N = d->n_agents, A = d->act_dim, D = d->obs_dim, S = d->state_dim;
  return 4 * ((T + 1) * N * D + (T + 1) * S + T * N * A + (T + 1) * N * A + T * N + T * N + T);
}

extern "C" int ope_store_gather_profile(int32_t enable) {
  if (enable && !g_prof.made) {
    for (int i = 0; i < kProfRing; ++i)
      for (int k = 0; k < 2; ++k)
        if (hipEventCreate(&g_prof.ev[i][k]) != hipSuccess) return OPE_EHIP;
    g_prof.made = true;
  }
  g_prof.on = enable != 0;
  g_prof.stride = enable > 1 ? enable : 1;      // enable = N > 1: every N-th gather launch is timed
  g_prof.seen = 0;
  g_prof.n = 0;
  return OPE_OK;
}

extern "C" int ope_store_gather_profile_read(float* ms_out_host, int32_t max_n) {
  if (!ms_out_host || max_n < 0) return OPE_EINVAL;
  const int n = g_prof.n < max_n ? g_prof.n : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess) return OPE_EHIP;
    if (hipEventElapsedTime(&ms_out_host[i], g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) return OPE_EHIP;
  }
  g_prof.n = 0;
  return n;
}

extern "C" void ope_set_gather_params(int floats_per_block, int xcd_run, int unroll, int nontemporal, int small_tiles, int tile_floats) {
  read_env_once();
  if (tile_floats > 0) g_tune.tile = tile_floats < 256 ? 256 : (tile_floats > kTileMax ? kTileMax : tile_floats);
  if (floats_per_block > 0) g_tune.floats = floats_per_block < 64 ? 64 : (floats_per_block > 65536 ? 65536 : floats_per_block);
  if (xcd_run >= 0) g_tune.xcd = xcd_run;
  if (unroll == 4 || unroll == 8 || unroll == 16) g_tune.unroll = unroll;
  if (nontemporal >= 0) g_tune.nt = nontemporal & 3;
  if (small_tiles >= 0) g_tune.small = small_tiles ? 1 : 0;
}

// The live-row plan of a batch that has not been gathered yet (ope.h): the same computation as live_plan_kernel (ope_live.hip), on the
// store's flags of the sampled slots.
template <class IDX>
__global__ void __launch_bounds__(ope::kLiveThreads) store_live_plan_kernel(ope::LiveW w, const float* __restrict__ ring, IDX idx, int capacity, int T, int N, int B) {
  extern __shared__ __attribute__((aligned(16))) int plan_lds[];
  StoreDones<IDX> dn{ring, &idx, capacity, T};
  ope::live_plan_body<true, 20>(w, nullptr, nullptr, 0, T, N, B, dn, plan_lds, (int)blockIdx.x, (int)gridDim.x);
}

extern "C" int ope_store_gather_attach_live(const ope_live_target* target) {
  if (!target) { g_live_pending = false; return OPE_OK; }
  if (!target->plan || target->n_agents < 1 || target->batch < 1 || target->episode_length < 1) return OPE_EINVAL;
  g_live_next = *target;
  g_live_pending = true;
  return OPE_OK;
}

extern "C" int ope_store_live_plan(int32_t capacity, int32_t episode_length, const float* store_dones_env, const int64_t* inds_dev,
                                   const int64_t* inds_host, const ope_live_target* target, void* stream) {
  (void)hipGetLastError();
  if (capacity < 1 || !store_dones_env || !target || !target->plan || (!inds_dev) == (!inds_host)) return OPE_EINVAL;
  const int T = episode_length, N = target->n_agents, B = target->batch;
  if (T != target->episode_length || !ope::live_plan_shape_ok(T, N, B)) return OPE_EINVAL;
  const ope::LiveW w = ope::live_views(target->plan, T, N, B);
  const dim3 grid(ope::live_plan_blocks(T, N, B)), block(ope::kLiveThreads);
  const size_t lds = (size_t)4 * ope::live_lds_ints(T, B);
  if (inds_host) {
    if (B > kMaxArgIdx) return OPE_EINVAL;
    ArgIdx ai;
    for (int i = 0; i < B; ++i) {
      if (inds_host[i] < 0 || inds_host[i] >= capacity) return OPE_EINVAL;
      ai.v[i] = (int32_t)inds_host[i];
    }
    OPE_LAUNCH((store_live_plan_kernel<ArgIdx>), grid, block, lds, (hipStream_t)stream, w, store_dones_env, ai, capacity, T, N, B);
  } else {
    OPE_LAUNCH((store_live_plan_kernel<DevIdx>), grid, block, lds, (hipStream_t)stream, w, store_dones_env, DevIdx{inds_dev}, capacity, T, N, B);
  }
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_gather(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds,
                                int32_t batch, const ope_fields* out, int32_t* bad_index_flag, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (capacity < 1 || !inds) return OPE_EINVAL;
  CopyArgs args;
  int rc = build_args(dims, store, out, batch, capacity, true, &args);
  if (rc != OPE_OK) return rc;
  args.bad_index = bad_index_flag;
  launch_copy<true>(args, DevIdx{inds}, (hipStream_t)stream);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_gather_sampled(const ope_dims* dims, int32_t capacity, int32_t filled, const int32_t* filled_dev,
                                        const ope_fields* store, uint64_t seed, const int32_t* counter, int32_t batch,
                                        const ope_fields* out, int64_t* inds_out, void* stream) {
  (void)hipGetLastError();
  if (capacity < 1 || batch < 1) return OPE_EINVAL;
  if (!filled_dev && (filled < 1 || filled > capacity)) return OPE_EINVAL;
  CopyArgs args;
  int rc = build_args(dims, store, out, batch, capacity, true, &args);
  if (rc != OPE_OK) return rc;
  launch_copy<true>(args, RngIdx{seed, counter, (uint32_t)(filled > 0 ? filled : 1), filled_dev, (uint32_t)capacity, inds_out}, (hipStream_t)stream);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_gather_host_inds(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds_host,
                                          int32_t batch, const ope_fields* out, void* stream) {
  (void)hipGetLastError();
  if (capacity < 1 || !inds_host || batch < 1 || batch > kMaxArgIdx) return OPE_EINVAL;
  ArgIdx ai;
  for (int i = 0; i < batch; ++i) {
    if (inds_host[i] < 0 || inds_host[i] >= capacity) return OPE_EINVAL;   // host data: rejected here, like numpy's IndexError
    ai.v[i] = (int32_t)inds_host[i];
  }
  CopyArgs args;
  int rc = build_args(dims, store, out, batch, capacity, true, &args);
  if (rc != OPE_OK) return rc;
  launch_copy<true>(args, ai, (hipStream_t)stream);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_gather_tuned(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds_dev,
                                      const int64_t* inds_host, int32_t batch, const ope_fields* out, int32_t* bad_index_flag,
                                      const ope_gather_tune* tune, void* stream) {
  (void)hipGetLastError();
  if (capacity < 1 || batch < 1 || (!inds_dev) == (!inds_host)) return OPE_EINVAL;      // exactly one index source
  CopyArgs args;
  if (inds_host) {
    if (batch > kMaxArgIdx) return OPE_EINVAL;
    ArgIdx ai;
    for (int i = 0; i < batch; ++i) {
      if (inds_host[i] < 0 || inds_host[i] >= capacity) return OPE_EINVAL;
      ai.v[i] = (int32_t)inds_host[i];
    }
    int rc = build_args(dims, store, out, batch, capacity, true, &args, tune);
    if (rc != OPE_OK) return rc;
    launch_copy<true>(args, ai, (hipStream_t)stream);
  } else {
    int rc = build_args(dims, store, out, batch, capacity, true, &args, tune);
    if (rc != OPE_OK) return rc;
    args.bad_index = bad_index_flag;
    launch_copy<true>(args, DevIdx{inds_dev}, (hipStream_t)stream);
  }
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_gather_ref(const ope_dims* dims, int32_t capacity, const ope_fields* store, const int64_t* inds_dev,
                                    const int64_t* inds_host, int32_t batch, const ope_fields* out, int64_t* inds_out, int32_t* bad_index_flag,
                                    const ope_gather_tune* tune, void* stream) {
  (void)hipGetLastError();
  if (capacity < 1 || batch < 1 || (!inds_dev) == (!inds_host) || !inds_out) return OPE_EINVAL;
  CopyArgs args;
  if (inds_host) {
    if (batch > kMaxArgIdx) return OPE_EINVAL;
    ArgIdx ai;
    for (int i = 0; i < batch; ++i) {
      if (inds_host[i] < 0 || inds_host[i] >= capacity) return OPE_EINVAL;
      ai.v[i] = (int32_t)inds_host[i];
    }
    int rc = build_args(dims, store, out, batch, capacity, true, &args, tune);
    if (rc != OPE_OK) return rc;
    if (args.total_blocks < 1) return OPE_EINVAL;      // (nothing to copy at all: the caller wants at least one field)
    args.idx_out = inds_out;
    launch_copy<true>(args, ai, (hipStream_t)stream);
  } else {
    int rc = build_args(dims, store, out, batch, capacity, true, &args, tune);
    if (rc != OPE_OK) return rc;
    if (args.total_blocks < 1) return OPE_EINVAL;
    args.bad_index = bad_index_flag;
    args.idx_out = inds_out == inds_dev ? nullptr : inds_out;
    launch_copy<true>(args, DevIdx{inds_dev}, (hipStream_t)stream);
  }
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_store_insert(const ope_dims* dims, int32_t capacity, const ope_fields* store, const ope_fields* staged,
                                const int64_t* slots, int32_t n_insert, int32_t* bad_index_flag, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (capacity < 1 || !slots) return OPE_EINVAL;
  CopyArgs args;
  int rc = build_args(dims, staged, store, n_insert, capacity, false, &args);
  if (rc != OPE_OK) return rc;
  args.bad_index = bad_index_flag;
  launch_copy<false>(args, DevIdx{slots}, (hipStream_t)stream);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}
