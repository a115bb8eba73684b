// Shared device helpers for the ope kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/ope.h"

#define OPE_H 64          // GRU / MLP hidden width (reference default hidden_size, config.py:63)
#define OPE_MIX 32        // mixer_hidden_dim (config.py:148)
#define OPE_HYP 64        // hypernet_hidden_dim (config.py:150)
#define OPE_LN_EPS 1e-5f  // nn.LayerNorm default eps

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define OPE_CHECK_LAUNCH()                          \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH; \
  } while (0)

// Launch log (ope_last_launches, ope.h): every launcher of the training path notes the kernel variant it ACTUALLY launched, so a test that
// pins a kernel family (ope_qmix_cfg.trunk_path / mixer_path / scan_family) can assert what ran instead of trusting the request. Host-side
// only: one bounded string append per launch into a thread-local buffer that the step entry points clear (defined in ope_api.hip).
namespace ope {
void note_launch(const char* name, int p0 = -1, int p1 = -1);
void clear_launch_log();
}

// Every kernel launch of the library goes through OPE_LAUNCH: plain hipLaunchKernelGGL, or -- while ope_kernel_profile is on (ope_prof.hip)
// -- the same launch with a start / stop event pair attached to the dispatch, so per-kernel durations can be read back in-process.
namespace ope {
extern bool g_kprof_on;
bool kprof_events(const void* fn, hipEvent_t* e0, hipEvent_t* e1);
void kprof_work(double flop, double bytes = 0);     // algorithmic work of the NEXT launch (only recorded while profiling is on)
void kprof_rows(int kind);                          // ... which runs on live rows: 1 agent rows, 2 those with t < T, 3 (t, b) rows (ope.h)
}
#define OPE_LAUNCH(kernel, grid, block, lds, st, ...)                                                   \
  do {                                                                                                  \
    hipEvent_t ope_e0_, ope_e1_;                                                                        \
    if (ope::g_kprof_on && ope::kprof_events((const void*)(kernel), &ope_e0_, &ope_e1_))                \
      hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), st, ope_e0_, ope_e1_, 0, __VA_ARGS__); \
    else                                                                                                \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                    \
  } while (0)

static inline int ope_round4(int64_t x) { return (int)((x + 3) & ~(int64_t)3); }
static inline int ope_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------------
// f32 MFMA 16x16x4 (exact fp32 FMA chain; 157 TF/s peak on MI355X).  D[i][j] += sum_{kk<4} A[i][kk] * B[kk][j]
//   lane l supplies  a = A[i = l&15][kk = l>>4],  b = B[kk = l>>4][j = l&15]
//   lane l holds     D[i = 4*(l>>4) + r][j = l&15]  in acc[r], r = 0..3
//
// "Transposed chain" convention used by every row-parallel kernel here: a wave owns 16 data rows (j = l&15),
// weights are the A operand (i = output feature), activations the B operand. With g = l>>4 and a K-chunk of 16
// features [16c, 16c+16), lane (j,g) reads the 4 consecutive features 16c+4g..+3 of its row (one float4) and of
// weight row i (one float4); MFMA step r consumes element r, i.e. feature 16c+4g+r -- a permutation of the
// chunk, identical on both operands. The result lane (j,g) holds output features 16*it + 4g + r of row j, which is
// again "4 consecutive features at 16*it + 4g": layers chain with no shuffles and no LDS.
// ---------------------------------------------------------------------------------------------------------
#ifndef OPE_MFMA_EMULATE
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
#else
// Debug fallback (same lane contract, built from shuffles) to separate layout bugs from MFMA semantics.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  const int l = threadIdx.x & 63;
  const int j = l & 15, g = l >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = c[r];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float av = __shfl(a, (4 * g + r) + 16 * kk, 64);
      float bv = __shfl(b, j + 16 * kk, 64);
      s = fmaf(av, bv, s);
    }
    c[r] = s;
  }
  return c;
}
#endif

// 4 consecutive floats p[k..k+3] of a row of logical length K, zero beyond K.
// VEC = widest aligned access the row stride allows: 4 (K%4==0), 2 (K%2==0) or 1.
template <int VEC>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, int k, int K) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC == 4) {
    if (k < K) v = *reinterpret_cast<const f32x4*>(p + k);
  } else if (VEC == 2) {
    if (k < K) {
      f32x2 a = *reinterpret_cast<const f32x2*>(p + k);
      v[0] = a[0];
      v[1] = a[1];
    }
    if (k + 2 < K) {
      f32x2 b = *reinterpret_cast<const f32x2*>(p + k + 2);
      v[2] = b[0];
      v[3] = b[1];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (k + r < K) v[r] = p[k + r];
  }
  return v;
}

// Branch-free variant for hot loops: ALWAYS loads (addresses clamped into [0, K)), never zero-fills. Elements whose
// index is >= K come back as finite duplicates of in-range data; the caller zeroes the OTHER operand of the
// product (mask4) so they contribute nothing. No divergent control flow -> loads can be hoisted and pipelined.
template <int VEC>
__device__ __forceinline__ f32x4 load4c(const float* __restrict__ p, int k, int K) {
  f32x4 v;
  if (VEC == 4) {
    const int kk = min(k, K - 4);  // K % 4 == 0, K >= 4
    v = *reinterpret_cast<const f32x4*>(p + kk);
  } else if (VEC == 2) {
    const int k0 = min(k, K - 2), k1 = min(k + 2, K - 2);
    const f32x2 a = *reinterpret_cast<const f32x2*>(p + k0);
    const f32x2 b = *reinterpret_cast<const f32x2*>(p + k1);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = p[min(k + r, K - 1)];
  }
  return v;
}
// zero the elements of v whose index k+r is >= K (selects, no branches)
__device__ __forceinline__ f32x4 mask4(f32x4 v, int k, int K) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (k + r < K) ? v[r] : 0.f;
  return v;
}

template <int VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, int k, int K, f32x4 v) {
  if (VEC == 4) {
    if (k < K) *reinterpret_cast<f32x4*>(p + k) = v;
  } else if (VEC == 2) {
    if (k < K) {
      f32x2 a = {v[0], v[1]};
      *reinterpret_cast<f32x2*>(p + k) = a;
    }
    if (k + 2 < K) {
      f32x2 b = {v[2], v[3]};
      *reinterpret_cast<f32x2*>(p + k + 2) = b;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (k + r < K) p[k + r] = v[r];
  }
}

static inline int ope_vec_of(int K) { return (K % 4 == 0) ? 4 : ((K % 2 == 0) ? 2 : 1); }

// Reductions over the 16 lanes of a DPP row (lanes 16q .. 16q+15), result in every lane of the row; VALU only (no LDS
// pipe, unlike __shfl_xor = ds_bpermute): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.
#define OPE_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false))
#define OPE_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, false)
__device__ __forceinline__ float row16_sum(float v) {
  v += OPE_DPP_F(v, 0xB1);
  v += OPE_DPP_F(v, 0x4E);
  v += OPE_DPP_F(v, 0x141);
  v += OPE_DPP_F(v, 0x140);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, OPE_DPP_F(v, 0xB1));
  v = fmaxf(v, OPE_DPP_F(v, 0x4E));
  v = fmaxf(v, OPE_DPP_F(v, 0x141));
  v = fmaxf(v, OPE_DPP_F(v, 0x140));
  return v;
}
__device__ __forceinline__ int row16_min_i(int v) {
  v = min(v, OPE_DPP_I(v, 0xB1));
  v = min(v, OPE_DPP_I(v, 0x4E));
  v = min(v, OPE_DPP_I(v, 0x141));
  v = min(v, OPE_DPP_I(v, 0x140));
  return v;
}

// Sum over the 4 lanes (g = 0..3) that share a data row j in the transposed-chain layout (lanes j, j+16, j+32, j+48), result in
// all four. gfx950's VALU lane swaps instead of two ds_bpermute round trips through the LDS pipe: v_permlane16_swap pairs row g
// with row g ^ 1, v_permlane32_swap half with half; with both operands = x the two results are the pair's values, so each step
// is the same commutative pair sum __shfl_xor gave (bitwise identical results, identical in the 4 lanes).
__device__ __forceinline__ float rowsum4(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Workgroup barrier that orders LDS traffic only. __syncthreads() also emits s_waitcnt vmcnt(0), which would drain a
// wave's in-flight global prefetches / stores at every step of the scan kernels.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Global load whose completion the COMPILER does not track (inline asm): used for prefetches that must stay in flight
// across a loop. hipcc's waitcnt pass drains every compiler-visible load before entering a loop that contains other
// memory traffic; an asm load is invisible to it, so the wait is ours: gwait*() before the first use.
__device__ __forceinline__ void gload_async(float& dst, const float* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// The same with a uniform base (an SGPR pair), a 32-bit per-lane byte offset and an immediate: no per-lane 64-bit address arithmetic.
template <int IMM>
__device__ __forceinline__ void gload_async_s(float& dst, const float* base, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
// s_waitcnt vmcnt(0) that names 24 destination registers as read-write so they stay put until the data has landed
#define OPE_GWAIT24(a)                                                                                                   \
  asm volatile("s_waitcnt vmcnt(0)"                                                                                      \
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[2][0]), \
                 "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[3][2]), "+v"(a[4][0]), "+v"(a[4][1]), \
                 "+v"(a[4][2]), "+v"(a[5][0]), "+v"(a[5][1]), "+v"(a[5][2]), "+v"(a[6][0]), "+v"(a[6][1]), "+v"(a[6][2]), \
                 "+v"(a[7][0]), "+v"(a[7][1]), "+v"(a[7][2])                                                              \
               :                                                                                                         \
               : "memory")

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Observation rows read IN PLACE from the episode-major store (ope_obs_ref, ope.h; "lazy batch"): batch row r = tn * B + b (tn = t * N + n)
// of the [T+1][N][B][D] batch is store row ep[b] * TTN + tn of the [capacity][T+1][N][D] ring, TTN = (T+1) N. The consumers (trunk_fwd4,
// wgrad) keep the sampled episode slots ep[0 .. B) in LDS (B <= kObsRefMaxB) and walk the rows EPISODE-major -- 16 (trunk tile) or 4
// (wgrad fetch) consecutive tn of one b, which are contiguous in the store -- rather than in batch order (16 / 4 different episodes,
// i.e. as many different pages per access: measured 5 us slower in either kernel at 3s5z).
constexpr int kObsRefMaxB = 512;
constexpr int kObsRefMaxRows = 1 << 20;     // div_small is exact below this
struct ObsRef {
  const int64_t* inds;   // DEVICE int64[B] sampled episode slots (null: rows come from a gathered batch)
  int cap, B, TTN;
};
// x / d for 0 <= x < 2^20 via a float reciprocal: (x + 0.5) / d is at least 0.5 / d away from an integer and the float error is below
// (x / d) * 2^-22, so the truncation is exact (ope_store.hip uses the same form; checked exhaustively there for d <= 600).
__device__ __forceinline__ int div_small(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }
__device__ __forceinline__ int obs_ref_slot(const ObsRef& o, int64_t v) {       // out-of-range slots are clamped here; the gather of the other fields raises the flag
  return v < 0 ? 0 : (v >= o.cap ? o.cap - 1 : (int)v);
}

// Live-row plan of one batch (ope_live.hip; built on the device from dones_env at the start of every step, never read by the host).
// The reference pads every sampled episode to T steps; every (t, b) with dones_env[t-1, b] = 1 is multiplied by 1 - bad_transitions_mask = 0
// (qmix.py:161-166), excluded from the loss normaliser (:184-186) and from Q_tot's mean (:198): no contribution to any output or gradient.
// Episode b needs the agent-network rows t < len_b, len_b = 2 + (last t' with dones_env[t', b] != 1) (1 when there is none; T + 1 for an
// episode that never ends), and the (t, b) rows t < min(len_b, T). Episodes are ranked by len_b, longest first (stable): the live episodes
// of a time step are then a PREFIX j < n[t] of the ranking, and rows are packed time-major:
//   agent row  (t, a, j) -> N * cum[t] + a * n[t] + j        RL = N * cum[T + 1] rows, the t < T ones are the first R1L = N * cum[T]
//   (t, b) row (t, j)    -> cum[t] + j                        TBL = cum[T] rows
// with cum[t] = n[0] + ... + n[t - 1]. Every intermediate of the step lives in that packed index space; only the batch itself (the
// reference's layout, untouched) is read through `srcrow` / `tbsrc` / the records.
struct LivePlan {
  const int* hdr;       // [16]: RL, R1L, TBL, max len, 0 ...        (null: no plan -- every padded row is computed)
  const int* len;       // [B]   len of the episode ranked j
  const int* perm;      // [B]   its batch column b
  const int* cum;       // [T+2]
  const int* nn;        // [T+2] n[t] (n[T+1] = 0)
  const int* tbrec;     // [T*B][8] per packed (t, b) row: t, b, agent-0 row at t, n[t], agent-0 row at t+1, n[t+1], t+1 < len_b, 0
  const int* tbsrc;     // [T*B] packed (t, b) row -> t * B + b
  const int* srcrow;    // [(T+1)*N*B] packed agent row -> batch row (t * N + a) * B + b
  const int* prevrow;   // [(T+1)*N*B] packed agent row -> packed row of (t - 1, a, j), -1 at t = 0
};
constexpr int kLiveMaxB = 256, kLiveMaxT = 1022;      // what the plan kernel's LDS tables hold

// Flat-parameter offsets (floats) of the agent q-network and the QMixer. Order = reference named_parameters()
// (SURVEY.md Appendix D); every tensor starts on a multiple of 4 floats.
struct AgentLayout {
  int fn_w, fn_b, fc1_w, fc1_b, ln1_w, ln1_b, fch_w, fch_b, lnh_w, lnh_b, fc2_w, fc2_b, ln2_w, ln2_b;
  int wih, whh, bih, bhh, lno_w, lno_b, q_w, q_b;
  int end;
  int fc2b_w, fc2b_b, ln2b_w, ln2b_b;   // layer_N = 2: the second hidden block rnn.mlp.fc2.1.{0,2} (mlp.py:14-28); -1 otherwise
  int layer_N;
};
struct MixerLayout {
  int w1a_w, w1a_b, w1b_w, w1b_b, w2a_w, w2a_b, w2b_w, w2b_b, b1_w, b1_b, b2a_w, b2a_b, b2b_w, b2b_b;
  int end;
  int one_layer;     // hypernet_layers = 1 (q_mixer.py:39-44): hyper_w1 / hyper_w2 are single Linear layers from the state -- w1a_* = hyper_w1
                     // [N*32][S], w2a_* = hyper_w2 [32][S]; w1b_* / w2b_* do not exist (-1)
};

static inline AgentLayout ope_agent_layout(int D, int A, int base, int layer_N = 1) {
  AgentLayout L;
  int o = base;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  L.fn_w = take(D); L.fn_b = take(D);
  L.fc1_w = take(OPE_H * D); L.fc1_b = take(OPE_H); L.ln1_w = take(OPE_H); L.ln1_b = take(OPE_H);
  L.fch_w = take(OPE_H * OPE_H); L.fch_b = take(OPE_H); L.lnh_w = take(OPE_H); L.lnh_b = take(OPE_H);
  L.fc2_w = take(OPE_H * OPE_H); L.fc2_b = take(OPE_H); L.ln2_w = take(OPE_H); L.ln2_b = take(OPE_H);
  L.fc2b_w = L.fc2b_b = L.ln2b_w = L.ln2b_b = -1;
  L.layer_N = layer_N == 2 ? 2 : 1;
  if (layer_N == 2) { L.fc2b_w = take(OPE_H * OPE_H); L.fc2b_b = take(OPE_H); L.ln2b_w = take(OPE_H); L.ln2b_b = take(OPE_H); }
  L.wih = take(3 * OPE_H * OPE_H); L.whh = take(3 * OPE_H * OPE_H); L.bih = take(3 * OPE_H); L.bhh = take(3 * OPE_H);
  L.lno_w = take(OPE_H); L.lno_b = take(OPE_H);
  L.q_w = take(A * OPE_H); L.q_b = take(A);
  L.end = o;
  return L;
}

// MLP (non-recurrent) agent network of the mqmix / maddpg families: same trunk, head fed directly by LN2; the GRU and
// rnn.norm tensors do not exist (offsets -1).
static inline AgentLayout ope_agent_layout_mlp(int D, int A, int base) {
  AgentLayout L;
  int o = base;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  L.fn_w = take(D); L.fn_b = take(D);
  L.fc1_w = take(OPE_H * D); L.fc1_b = take(OPE_H); L.ln1_w = take(OPE_H); L.ln1_b = take(OPE_H);
  L.fch_w = take(OPE_H * OPE_H); L.fch_b = take(OPE_H); L.lnh_w = take(OPE_H); L.lnh_b = take(OPE_H);
  L.fc2_w = take(OPE_H * OPE_H); L.fc2_b = take(OPE_H); L.ln2_w = take(OPE_H); L.ln2_b = take(OPE_H);
  L.wih = L.whh = L.bih = L.bhh = L.lno_w = L.lno_b = -1;
  L.fc2b_w = L.fc2b_b = L.ln2b_w = L.ln2b_b = -1;
  L.layer_N = 1;
  L.q_w = take(A * OPE_H); L.q_b = take(A);
  L.end = o;
  return L;
}

static inline MixerLayout ope_mixer_layout(int N, int S, int base) {
  MixerLayout L;
  int o = base;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  L.w1a_w = take(OPE_HYP * S); L.w1a_b = take(OPE_HYP);
  L.w1b_w = take(N * OPE_MIX * OPE_HYP); L.w1b_b = take(N * OPE_MIX);
  L.w2a_w = take(OPE_HYP * S); L.w2a_b = take(OPE_HYP);
  L.w2b_w = take(OPE_MIX * OPE_HYP); L.w2b_b = take(OPE_MIX);
  L.b1_w = take(OPE_MIX * S); L.b1_b = take(OPE_MIX);
  L.b2a_w = take(OPE_HYP * S); L.b2a_b = take(OPE_HYP);
  L.b2b_w = take(OPE_HYP); L.b2b_b = take(1);
  L.end = o;
  L.one_layer = 0;
  return L;
}
// hypernet_layers = 1: named_parameters() order hyper_w1.{weight,bias}, hyper_w2.{weight,bias}, hyper_b1.*, hyper_b2.0.*, hyper_b2.2.*
static inline MixerLayout ope_mixer_layout1(int N, int S, int base) {
  MixerLayout L;
  int o = base;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  L.w1a_w = take(N * OPE_MIX * S); L.w1a_b = take(N * OPE_MIX);
  L.w2a_w = take(OPE_MIX * S); L.w2a_b = take(OPE_MIX);
  L.w1b_w = L.w1b_b = L.w2b_w = L.w2b_b = -1;
  L.b1_w = take(OPE_MIX * S); L.b1_b = take(OPE_MIX);
  L.b2a_w = take(OPE_HYP * S); L.b2a_b = take(OPE_HYP);
  L.b2b_w = take(OPE_HYP); L.b2b_b = take(1);
  L.end = o;
  L.one_layer = 1;
  return L;
}
