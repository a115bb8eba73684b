// First hyper-network layers of the QMIX mixer for WIDE centralized states, as one throughput-bound GEMM.
//   QMixer.forward, hyper_w1[0] / hyper_w2[0] / hyper_b1 / hyper_b2[0]   offpolicy/algorithms/qmix/algorithm/q_mixer.py:39-66, 79-93
//   the state the reference's own SMAC launch script feeds it: --use_global_all_local_state (scripts/train_smac_qmix.sh:17) appends
//   every agent's observation to the global state, S = state + N * obs (StarCraft2_Env.py:1314-1315): 2 232 at 3s5z, 240 at 3m.
//
// At S = 216 the four first layers (S -> 64, 64, 64, 32: 224 output features) are a latency chain per 16-row tile and live in
// mixer_fwd3 / mixer_fwd2 (ope_mixer.hip). At S = 2 232 they are [T*B = 4 800 rows x 2 232] . [2 232 x 224] per net, 9.6 GFLOP for the
// live + target pair -- 96 % of the mixer's work and no longer a chain: a GEMM, built here the way the f32 matrix pipe wants it.
//
//   C[net][m][f] = sum_k X[m + net * B][k] * W_net[f][k]        m < T*B,  f < 224,  k < S
//
// * f32 MFMA 16x16x4 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the f32 vector rate: the matrix pipe buys operand reuse, not rate;
//   there is no xf32 on gfx950). "Transposed chain" convention of ope_common.h: weights are the A operand, state rows the B operand,
//   so lane (j, g) ends up with features 16 it + 4 g .. + 3 of row j -- the same fragment the mixer's second stage consumes.
// * Block tile 128 rows x 224 features (one net), K staged 32 columns at a time through LDS: (128 + 224) x 128 B per stage
//   = 0.0123 staged bytes per MFMA-byte-equivalent -- between the 128 x 128 tile (0.0156) and the 256 x 256 one. 8 waves as 4 (rows)
//   x 2 (features): a wave owns 32 x 112 = 2 x 7 accumulator tiles (56 VGPRs) and per 16 columns reads 2 + 7 fragments
//   (ds_read_b128, conflict-free at a 40-float row pitch: see the bank note below) for 56 MFMAs.
// * Staging through REGISTERS (global_load_dwordx4 -> ds_write_b128), not LDS-DMA: at the f32 MFMA rate a stage is 3 584 MFMA
//   cycles per wave against 6 loads + 6 LDS stores, so the staging instructions are free, and register staging allows the padded
//   LDS rows, any row alignment (S % 4 != 0 falls back to 8- / 4-byte loads) and exact zero-fill of the K tail.
//   Loads of stage s + 1 are issued before the MFMAs of stage s and written to the other LDS buffer after them: one
//   workgroup barrier per stage, two waves per SIMD to cover it.
// * Stream-K over a persistent grid of min(256, units) workgroups (one per CU: 112.6 KB of LDS): the (tile, K-stage) units -- 76 tiles
//   x 70 stages at 3s5z -- are cut into equal contiguous shares, so every CU carries the same number of MFMAs whatever the tile
//   count (76 tiles on 256 CUs would otherwise be 30 % busy, and no uniform split-K comes out even). A workgroup writes the partial
//   accumulators of each (tile, K-range) segment it covered into its own slab; the consumer (mixer_fwd2's second stage, ope_mixer.hip)
//   adds a tile's 3-4 slabs in ascending K order on top of the bias: fixed order, bitwise reproducible, no atomics, and the
//   "reduction" costs one pass over 30 MB that the consumer had to read anyway.
//
// Bank note (MI355X_MICROARCH.md, LDS): ds_read_b128 is served in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32; a group is conflict-free when its 16 16-byte slots differ mod 16. With rows j = lane & 15 at a pitch of
// 10 slots (40 floats) and K-quarter g = lane >> 4 one slot apart, a group's slots are 10 j + g over j in {0-3, 12-15} and
// 10 j + g + 1 over j in {4-11}: the eight even and the eight odd residues. (A pitch of 36 floats, the usual "+ 4", is 2-way.)
#include <stdlib.h>

#include "ope_mixer.h"

namespace ope {

namespace {

constexpr int BM = kWideBM, BN = kWideBN, BK = 32;
constexpr int PITCH = 40;                         // floats per staged row (32 + 8 pad)
constexpr int STAGE_ROWS = BM + BN;               // 352
constexpr int STAGE_FLOATS = STAGE_ROWS * PITCH;  // 14 080 floats = 56 320 B
constexpr int PIECES = STAGE_ROWS * (BK / 4);     // 2 816 16-byte pieces per stage
constexpr int NLD = (PIECES + 511) / 512;         // 6 per thread (the last one partly)

}  // namespace

template <int VEC>
__global__ void __launch_bounds__(512, 2) mixer_wide_gemm_kernel(MixerFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int wm = wave & 3, wn = wave >> 2;         // wave tile: rows 32 wm .. + 31, features 112 wn .. + 111
  const WidePlan P = wide_plan(a.TB, a.S);
  const int S = a.S;
  const int wg = blockIdx.x;
  // diagnostics (ope_qmix_cfg.debug; tools/wide_phases.py): [workgroup][8] = s_memtime at start / first segment staged / end of the
  // last stage loop / end, then the same four points on the constant 100 MHz wall clock (-> effective shader clock)
  long long* dbg = a.dbg ? a.dbg + (int64_t)wg * 8 : nullptr;
  auto stamp = [&](int k) {
    if (dbg && tid == 0) { dbg[k] = __builtin_amdgcn_s_memtime(); dbg[4 + k] = wall_clock64(); }
  };
  const int ex = a.k_stagger;      // experiment mask (OPE_WIDE_EXP, timing only -- results are wrong): 1 no global loads in the loop,
                                   // 2 no MFMAs, 4 no LDS deposits in the loop
  stamp(0);
  int u = wide_bound(P, wg);
  const int u_end = wide_bound(P, wg + 1);
  const int first_tile = u / P.nst;

  // fragment read offsets inside a stage buffer (floats): X rows first, then the 224 weight rows
  int xoff[2], woff[7];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) xoff[mi] = (32 * wm + 16 * mi + j) * PITCH + 4 * g;
#pragma unroll
  for (int ni = 0; ni < 7; ++ni) woff[ni] = (BM + 112 * wn + 16 * ni + j) * PITCH + 4 * g;

  while (u < u_end) {
    const int tile = u / P.nst;
    const int st0 = u - tile * P.nst;
    const int st1 = min(P.nst, st0 + (u_end - u));       // stages [st0, st1) of this tile
    const int net = tile & 1, rb = tile >> 1;
    const int m0 = rb * BM;
    const float* __restrict__ th = net == 0 ? a.theta0 : a.theta1;

    // this thread's staging pieces: piece p = tid + 512 i -> staged row (tid >> 3) + 64 i, 16-byte column piece tid & 7 (512 is a
    // multiple of 8: the same column piece for all six). i = 0, 1: state rows; i = 2 .. 5: row tid >> 3 of hyper_w1.0, hyper_w2.0,
    // hyper_b2.0 and (threads 0-255 only: 32 rows) hyper_b1
    const int prow = tid >> 3, pc4 = 4 * (tid & 7);
    const float* rowp[NLD];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + prow + 64 * i;
      const int mm = m < a.TB ? m : a.TB - 1;              // rows past T*B: a valid row's data, never stored by the consumer
      rowp[i] = a.share + ((int64_t)mm + (int64_t)net * a.B) * S;
    }
    rowp[2] = th + a.L.w1a_w + (int64_t)prow * S;
    rowp[3] = th + a.L.w2a_w + (int64_t)prow * S;
    rowp[4] = th + a.L.b2a_w + (int64_t)prow * S;
    rowp[5] = th + a.L.b1_w + (int64_t)(prow & 31) * S;
    const bool last_ok = tid < 256;                        // piece 5 exists for staged rows 320 .. 351 only
    const int dst0 = prow * PITCH + pc4;                   // + 64 i * PITCH
    f32x4 stg[NLD];
    auto fetch = [&](int st) {
      const int k = BK * st + pc4;
#pragma unroll
      for (int i = 0; i < NLD; ++i) stg[i] = load4c<VEC>(rowp[i], k, S);      // clamped into the row; masked in deposit()
    };
    auto deposit = [&](int st, float* buf) {
      const int k = BK * st + pc4;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const f32x4 v = mask4(stg[i], k, S);               // exact zeros beyond S, in BOTH operands
        if (i < NLD - 1 || last_ok) *reinterpret_cast<f32x4*>(buf + dst0 + 64 * i * PITCH) = v;
      }
    };

    f32x4 acc[2][7];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 7; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Stage loop: request stage s + 1, read this stage's 18 fragments, 112 MFMAs, deposit stage s + 1 into the other buffer, barrier.
    // Measured at 3s5z / S = 2 232 (tools/wide_phases.py; cycles per stage of a workgroup's loop; 7 168 = the two waves of a SIMD issuing
    // their 224 MFMAs back to back): 9.6 k this form (97.5 us kernel); without the loop's global loads 8.4 k, without MFMAs 3.2 k,
    // without both 1.6 k -- the deposits (ds_write_b128: 13 cycles of the VGPR -> LDS path each) + masks + fragment reads + barrier
    // are not hidden behind the other wave of the SIMD, which reaches them at the same time. Two variants that were SLOWER:
    //  * fragment reads of the next stage's first half hoisted across the barrier, deposit in mid-stage (MFMAs queued on both sides of
    //    the barrier): 10.1 k (105 us) -- the loads requested at the top of a stage are not back half a stage later under MFMA load;
    //  * LDS-DMA staging (global_load_lds_dwordx4, three buffers, requests two stages ahead, source-column XOR swizzle for
    //    conflict-free unpadded rows): 11.9 k (125 us) -- each DMA instruction costs the issuing wave 100-185 cycles beside MFMAs
    //    (MI355X_MICROARCH.md) and all eight waves pay it together.
    // What would get to ~7.4 k: the two waves of a SIMD in opposite phases (one issues the stage's MFMAs while the other does all of the
    // stage's memory work, two barriers per stage) -- next.
    fetch(st0);
    deposit(st0, lds);                                     // (the previous segment's readers left behind its last barrier)
    lds_barrier();
    if (tile == first_tile) stamp(1);
    for (int st = st0; st < st1; ++st) {
      const float* cur = lds + ((st - st0) & 1) * STAGE_FLOATS;
      float* nxt = lds + (((st - st0) & 1) ^ 1) * STAGE_FLOATS;
      const bool more = st + 1 < st1;
      if (more && !(ex & 1)) fetch(st + 1);                // in flight behind this stage's MFMAs
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 xf[2], wf[7];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) xf[mi] = *reinterpret_cast<const f32x4*>(cur + xoff[mi] + 16 * c);
#pragma unroll
        for (int ni = 0; ni < 7; ++ni) wf[ni] = *reinterpret_cast<const f32x4*>(cur + woff[ni] + 16 * c);
        if (!(ex & 2)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < 7; ++ni)
#pragma unroll
              for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = mfma16(wf[ni][r], xf[mi][r], acc[mi][ni]);
        }
      }
      if (more && !(ex & 4)) deposit(st + 1, nxt);
      lds_barrier();
    }
    stamp(2);

    // partial accumulators of this (tile, K-range) segment -> this workgroup's slab: [BM][BN] row-major, feature-contiguous 16-byte
    // pieces per lane (lane (j, g) of accumulator tile (mi, ni) = features 112 wn + 16 ni + 4 g .. + 3 of row 32 wm + 16 mi + j)
    float* __restrict__ slab = a.wide_slab + ((int64_t)wg * P.maxseg + (tile - first_tile)) * (int64_t)(BM * BN);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 7; ++ni)
        *reinterpret_cast<f32x4*>(slab + (32 * wm + 16 * mi + j) * BN + 112 * wn + 16 * ni + 4 * g) = acc[mi][ni];
    u += st1 - st0;
  }
  stamp(3);
}

int launch_mixer_wide_gemm(const MixerFwdArgs& a0, hipStream_t st) {
  if (a0.TB < 1 || a0.S < 1 || !a0.wide_slab) return OPE_EINVAL;
  const WidePlan P = wide_plan(a0.TB, a0.S);
  const int vec = ope_vec_of(a0.S);
  // timing experiments only (skips loads / MFMAs / deposits: WRONG results): honoured only for a call that also carries the debug
  // stamps (ope_qmix_cfg.debug / ope_set_debug, i.e. tools/wide_phases.py), never for a plain training call
#ifdef OPE_EXPERIMENTS      // (libope_exp.so only: the release library never skips loads / MFMAs / deposits)
  static const int ex_env = getenv("OPE_WIDE_EXP") ? atoi(getenv("OPE_WIDE_EXP")) : 0;
  const int ex = a0.dbg ? ex_env : 0;
#else
  const int ex = 0;
#endif
  MixerFwdArgs a = a0;
  a.k_stagger = ex;
  kprof_work(2.0 * 2.0 * a.TB * (double)a.S * (3.0 * OPE_HYP + OPE_MIX));
  if (vec == 4) OPE_LAUNCH(mixer_wide_gemm_kernel<4>, dim3(P.nwg), dim3(512), 0, st, a);
  else if (vec == 2) OPE_LAUNCH(mixer_wide_gemm_kernel<2>, dim3(P.nwg), dim3(512), 0, st, a);
  else OPE_LAUNCH(mixer_wide_gemm_kernel<1>, dim3(P.nwg), dim3(512), 0, st, a);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  note_launch("mixer_wide_gemm", vec);
  return OPE_OK;
}

}  // namespace ope
