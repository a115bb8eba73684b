// C-ABI entry points that orchestrate the kernels: parameter layout, workspace carving, the QMIX/VDN
// loss-and-gradient step (QMix.train_policy_on_batch, qmix.py:77-190) and the stand-alone agent forward.
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "ope_live.h"
#include "ope_mixer.h"
#include "ope_wgrad.h"
#include "ope_workspace.h"

using namespace ope;

namespace {

int g_debug = 0;
constexpr int kMaxChunksCfg = 8;

bool cfg_ok(const ope_qmix_cfg* c) {
  if (!c) return false;
  const ope_dims& d = c->dims;
  if (c->mlp && d.episode_length != 1) return false;   // MLP (transition) mode = one-step "episodes"
  if (c->phase < 0 || c->phase > 3) return false;
  if (c->mixer_path < 0 || c->mixer_path > 3 || c->time_chunks < 0 || c->time_chunks > kMaxChunksCfg) return false;
  if (c->trunk_path != 0 && c->trunk_path != 3 && c->trunk_path != 4) return false;
  if (c->chain_path < 0 || c->chain_path > 2) return false;
  if (c->hypernet_layers < 0 || c->hypernet_layers > 2) return false;
  if (c->wgrad_path < 0 || c->wgrad_path > 2 || c->live_rows < 0 || c->live_rows > 4) return false;
  if (d.layer_N < 0 || d.layer_N > 2) return false;
  if (d.flags & ~(OPE_DIMS_NO_FEATURE_NORM | OPE_DIMS_TANH | OPE_DIMS_MASK_TARGET_MAX)) return false;
  if ((d.flags & OPE_DIMS_TANH) && (c->phase != 0 || d.layer_N == 2 || d.obs_dim > 384 || c->trunk_path == 4)) return false;   // tanh: trunk_fwd3 / trunk_bwd3
  if (d.layer_N == 2 && (c->mlp || c->phase != 0 || c->time_chunks > 1)) return false;      // a second hidden block: whole steps of recurrent nets
  if (c->hypernet_layers == 1 && (c->mlp || c->phase != 0 || c->mixer_path == 3 || c->chain_path == 1)) return false;   // one-layer hyper-networks: the fused chain only
  return d.n_agents >= 1 && d.act_dim >= 1 && d.obs_dim >= 1 && d.obs_dim <= 512 && d.state_dim >= 1 &&
         d.episode_length >= 1 && c->batch >= 1 && d.n_agents <= 64 && d.act_dim <= 200;
}

struct Raw {  // offsets inside one split slab, agent region then mixer region
  int P1, s1, P2, s2, P3, s3, WHH, shh, E, sq, agent_end;
  int P2b, s2b;      // layer_N = 2: the second hidden block's weight gradient (against xhat2) and bias gradient
  int mixer_size;
};

constexpr int kMaxChunks = 8;
constexpr int kSideEvents = 16;

// One non-blocking side stream + events per device, created on first use and kept for the life of the process.
struct SidePool {
  hipStream_t s;
  hipEvent_t ev[kSideEvents];
  hipEvent_t scan_done[kMaxChunks], bptt_done[kMaxChunks];
};
SidePool* side_pool() {
  static SidePool pools[16];
  static bool made[16] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!made[dev]) {
    SidePool& P = pools[dev];
    if (hipStreamCreateWithFlags(&P.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (int i = 0; i < kSideEvents; ++i)
      if (hipEventCreateWithFlags(&P.ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i < kMaxChunks; ++i)
      if (hipEventCreateWithFlags(&P.scan_done[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&P.bptt_done[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    made[dev] = true;
  }
  return &pools[dev];
}

struct Plan {
  int T, N, A, D, S, B, NB, A4, NM, mlp;
  int64_t R, R1, TB;
  AgentLayout AL;
  MixerLayout ML;   // offsets in the full theta (base = AL.end)
  int64_t P;        // padded parameter count
  Raw raw;
  int ns_agent, ns_mixer;
  int chunks;                   // time chunks of the two-stream schedule (1 = whole episodes on one stream)
  int tb[kMaxChunks + 1];       // chunk c covers time steps [tb[c], tb[c+1]) of the T+1 forward steps
  int ns_chunk[kMaxChunks];     // weight-gradient K-splits of chunk c
  int n_loss_tiles;
  Workspace ws;
  // region offsets
  int64_t mu0, rstd0, xhat1, rstd1, mask1, xhat2, rstd2, mask2, gi, h, rg, zg, ng, ghn, xhat_o, rstd_o, act_idx,
      agent_q, agent_nq, gi_t, h_t, qtot, nqtot, hw1, hw2, hb2, v1, hpre, v2, loss_part, err_abs, dqtot, d_agent_q,
      d_b1, d_v2, d_v1, d_hw1, d_hw2, d_hb2, dh_out, dqoh, dgi, dghn, dz1, dz2, thetaT, mixT, raw_agent, raw_mixer,
      rsum, q_all, loss_tot, ln_zero, ln_one, dh_carry, dbg, gsq_part, mix_slab, raw2, live, live1;
  int n_gsq;
  bool wide;
  bool chain;                   // the (t, b)-row chain runs as mixer_hyp + qchain (ope_chain.hip) instead of head_fwd / mixer_fwd / mixer_bwd / head_bwd
  bool chain_can;               // ... could (shape, phase, schedule)
  bool hyp1;                    // one-layer hyper-networks (hypernet_layers = 1)
  int layerN;                   // hidden blocks behind fc1 (1 | 2)
  int64_t a2, a2_t, xhat3, rstd3, mask3, dz3, da2;
  int64_t hb1, hw1_t, hw2_t, hb2_t, hb1_t, v1_t, v2_t;
};

thread_local char g_launch_log[2048];
thread_local int g_launch_len = 0, g_launch_n = 0;

int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

void make_plan(const ope_qmix_cfg* c, Plan* p) {
  const ope_dims& d = c->dims;
  p->T = d.episode_length; p->N = d.n_agents; p->A = d.act_dim; p->D = d.obs_dim; p->S = d.state_dim; p->B = c->batch;
  p->NB = p->N * p->B; p->A4 = ope_round4(p->A); p->NM = p->N * OPE_MIX;
  p->R = (int64_t)(p->T + 1) * p->NB; p->R1 = (int64_t)p->T * p->NB; p->TB = (int64_t)p->T * p->B;
  p->mlp = c->mlp;
  p->layerN = d.layer_N == 2 ? 2 : 1;
  p->AL = c->mlp ? ope_agent_layout_mlp(p->D, p->A, 0) : ope_agent_layout(p->D, p->A, 0, p->layerN);
  if (c->vdn) {
    memset(&p->ML, 0, sizeof(p->ML));
    p->ML.end = p->AL.end;
    p->P = p->AL.end;
  } else {
    p->ML = c->hypernet_layers == 1 ? ope_mixer_layout1(p->N, p->S, p->AL.end) : ope_mixer_layout(p->N, p->S, p->AL.end);
    p->P = p->ML.end;
  }
  p->hyp1 = !c->vdn && c->hypernet_layers == 1;
  Raw& w = p->raw;
  int o = 0;
  auto take = [&](int n) { int r = o; o += ope_round4(n); return r; };
  w.P1 = take(OPE_H * p->D); w.s1 = take(OPE_H);
  w.P2 = take(OPE_H * OPE_H); w.s2 = take(OPE_H);
  w.P2b = w.s2b = -1;
  if (p->layerN == 2) { w.P2b = take(OPE_H * OPE_H); w.s2b = take(OPE_H); }
  w.P3 = take(3 * OPE_H * OPE_H); w.s3 = take(3 * OPE_H);
  w.WHH = take(3 * OPE_H * OPE_H); w.shh = take(3 * OPE_H);
  w.E = take(p->A * OPE_H); w.sq = take(p->A4);
  w.agent_end = o;
  w.mixer_size = c->vdn ? 0 : (p->ML.end - p->AL.end);
  // K-splits: every wave reduces the same number of rows whatever the problem, so workgroups are equally long (multiples of 4:
  // the four waves of a workgroup hold consecutive splits and pre-reduce them). Measured on MI355X (tools/_ab sweeps of
  // OPE_WGRAD_ROWS at 3s5z, 3s5z_gall and MMM2, profiles/r03q_wgrad_rows.txt): the launch takes
  //     ceil(workgroups / CUs) x rows per split x ~0.1 us
  // whatever the number of workgroups resident on a CU -- a CU reduces its workgroups' rows at a fixed rate (~58 % of its matrix
  // pipes), so what matters is the most loaded CU. 160 rows gave the 3s5z launch 896 workgroups = 4 per CU on half the chip
  // (65.6 us); 192 rows = 754 workgroups = 3 per CU (60.7 us). The row count minimising that product (plus ~40 rows of
  // per-workgroup prologue / reduction) is searched here among the counts that keep at least three workgroups on the busiest CU
  // (small problems that cannot: 160 rows); OPE_WGRAD_ROWS pins it instead (read once).
  static const int rows_env = getenv("OPE_WGRAD_ROWS") ? atoi(getenv("OPE_WGRAD_ROWS")) : 0;
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n > 1 ? n : 256; }();
  auto splits_for = [](int64_t K, int cap, int rows) {
    int s = (int)((K + rows - 1) / rows);
    s = s >= 4 ? ((s + 3) / 4) * 4 : s;
    return clampi(s, 1, cap);
  };
  // 64 x 64 output tiles of the agent problems (fc1, fc2, W_ih, W_hh in two pieces, q head) and of the mixer problems
  // (three S-wide first layers of 64 rows, the NM x 64 and 32 x 64 second layers, the 32 x S state bias, the scalar head)
  const int agent_tiles = ope_cdiv(p->D, 64) + 1 + (p->layerN == 2 ? 1 : 0) + (c->mlp ? 0 : 3 + 2 + 1) + ope_cdiv(p->A, 64);
  const int mixer_tiles = c->vdn ? 0 : (p->hyp1 ? ope_cdiv(p->S, 64) * (ope_cdiv(p->NM, 64) + 3) + 1 : 4 * ope_cdiv(p->S, 64) + ope_cdiv(p->NM, 64) + 2);
  int rows_per_split = rows_env > 0 ? rows_env : 160;
  if (rows_env <= 0) {
    int64_t best = -1;
    for (int r = 96; r <= 1024; r += 4) {
      const int na = splits_for(p->R1, 256, r), nm = splits_for(p->TB, 64, r);
      const int64_t wgs = (int64_t)agent_tiles * ope_cdiv(na, 4) + (int64_t)mixer_tiles * ope_cdiv(nm, 4);
      const int64_t len = std::max(ope_cdiv(p->R1, na), c->vdn ? 0 : ope_cdiv(p->TB, nm));     // rows a wave really reduces
      const int64_t per_cu = ope_cdiv(wgs, cus);
      if (per_cu < 3) continue;      // the rate above was measured with 2-4 workgroups per CU; fewer leave load latency uncovered
      const int64_t cost = per_cu * (len + 40);
      if (best < 0 || cost < best) { best = cost; rows_per_split = r; }
    }
  }
  p->ns_agent = splits_for(p->R1, 256, rows_per_split);
  p->ns_mixer = splits_for(p->TB, 64, rows_per_split);
  p->n_loss_tiles = ope_cdiv(p->TB, 16);
  // time chunks: boundaries on multiples of 8 steps (the scans prefetch in 8-step groups); short episodes stay whole
  {
    // Measured on MI355X (3s5z, B = 32): at this size every kernel is a latency-bound chain with ~1 wave per SIMD, so
    // cutting the time axis (C > 1) makes each piece barely faster while adding launches and ~7 us cross-stream hand-offs
    // (0.61 -> 0.73 ms at C = 2, 1.0 ms at C = 4), and kernels run side by side slow each other down about as much as
    // the overlap saves (BPTT beside a weight-gradient launch: 80 -> 150 us). Default: whole episodes, one stream; the
    // chunked two-stream schedule stays available for larger batches (OPE_CHUNKS).
    static const int env_chunks = getenv("OPE_CHUNKS") ? atoi(getenv("OPE_CHUNKS")) : 1;    // process default, read once
    int C = c->time_chunks > 0 ? c->time_chunks : env_chunks;
    C = clampi(C, 1, kMaxChunks);
    const int L = p->T + 1;
    if (c->mlp || c->phase) C = 1;
    while (C > 1 && L / C < 16) --C;
    p->chunks = C;
    for (int q = 0; q <= C; ++q) p->tb[q] = q == C ? L : ((int)((int64_t)q * L / C) / 8) * 8;
    for (int q = 0; q < C; ++q) {
      const int hi = p->tb[q + 1] < p->T ? p->tb[q + 1] : p->T;
      const int64_t rows = (int64_t)(hi - p->tb[q]) * p->NB;
      (void)rows;
      p->ns_chunk[q] = p->ns_agent;   // every chunk keeps the full split count: shorter K per wave, same number of waves
    }
  }

  Workspace& W = p->ws;
  const int64_t R = p->R, R1 = p->R1, TB = p->TB;
  p->mu0 = W.add("mu0", R); p->rstd0 = W.add("rstd0", R);
  p->xhat1 = W.add("xhat1", R * OPE_H); p->rstd1 = W.add("rstd1", R); p->mask1 = W.add("mask1", 2 * R);
  p->xhat2 = W.add("xhat2", R * OPE_H); p->rstd2 = W.add("rstd2", R); p->mask2 = W.add("mask2", 2 * R);
  p->gi = W.add("gi", R * 3 * OPE_H); p->h = W.add("h", R * OPE_H);
  p->rg = W.add("rg", R * OPE_H); p->zg = W.add("zg", R * OPE_H); p->ng = W.add("ng", R * OPE_H); p->ghn = W.add("ghn", R * OPE_H);
  p->xhat_o = W.add("xhat_o", R * OPE_H); p->rstd_o = W.add("rstd_o", R);
  p->act_idx = W.add("act_idx", R1);
  p->agent_q = W.add("agent_q", TB * p->N); p->agent_nq = W.add("agent_nq", TB * p->N);
  p->gi_t = W.add("gi_t", R * 3 * OPE_H); p->h_t = W.add("h_t", R * OPE_H);
  p->qtot = W.add("qtot", TB); p->nqtot = W.add("nqtot", TB);
  p->hw1 = W.add("hw1", TB * OPE_HYP); p->hw2 = W.add("hw2", TB * OPE_HYP); p->hb2 = W.add("hb2", TB * OPE_HYP);
  p->v1 = W.add("v1", TB * p->NM); p->hpre = W.add("hpre", TB * OPE_MIX); p->v2 = W.add("v2", TB * OPE_MIX);
  p->loss_part = W.add("loss_part", (int64_t)p->n_loss_tiles * 4);
  p->loss_tot = W.add("loss_tot", 4);
  p->ln_zero = W.add("ln_zero", R);   // mu = 0 / rstd = 1 vectors: "no LayerNorm" operands of the wgrad kernel
  p->ln_one = W.add("ln_one", R);
  // per-workgroup partial sums of grad^2 written by the finalize launch (zero beyond the launch's workgroups: the region
  // is zero-filled once at workspace_init); upper bound of finalize_blocks()
  p->n_gsq = ope_cdiv((int64_t)p->P + OPE_GRAD_TAIL, 256) + 1 + ope_cdiv((int64_t)2 * (p->D + (p->layerN == 2 ? 4 : 3) * OPE_H) * 64, 256) + 4;
  // ... or of launch_wgrad2_fin's live workgroups: 4 rows each of at most 4 tiles + column sums + column partials per unit, + tail + zero blocks
  p->n_gsq = std::max(p->n_gsq, kMaxW2Units * ((4 * 64 + 4 + 8 + 3) / 4) + 1 + (int)ope_cdiv((int64_t)p->P, 2048) + 1);
  p->gsq_part = W.add("gsq_part", p->n_gsq);
  p->err_abs = W.add("err_abs", TB); p->dqtot = W.add("dqtot", 4 * TB); p->d_agent_q = W.add("d_agent_q", TB * p->N);
  p->d_b1 = W.add("d_b1", TB * OPE_MIX); p->d_v2 = W.add("d_v2", TB * OPE_MIX); p->d_v1 = W.add("d_v1", TB * p->NM);
  p->d_hw1 = W.add("d_hw1", TB * OPE_HYP); p->d_hw2 = W.add("d_hw2", TB * OPE_HYP); p->d_hb2 = W.add("d_hb2", TB * OPE_HYP);
  p->dh_out = W.add("dh_out", R1 * OPE_H); p->dqoh = W.add("dqoh", R1 * p->A4);
  p->dgi = W.add("dgi", R1 * 3 * OPE_H); p->dghn = W.add("dghn", R1 * OPE_H);
  p->dz1 = W.add("dz1", R1 * OPE_H); p->dz2 = W.add("dz2", R1 * OPE_H);
  p->thetaT = W.add("thetaT", OPE_H * 3 * OPE_H + 2 * OPE_H * OPE_H);      // W_ih^T, fc2.0^T (, fc2.1^T)
  p->mixT = W.add("mixT", (int64_t)OPE_HYP * p->NM + OPE_HYP * OPE_MIX);
  p->dh_carry = W.add("dh_carry", (int64_t)p->NB * OPE_H);
  p->raw_agent = W.add("raw_agent", (int64_t)p->ns_agent * p->chunks * w.agent_end);
  p->raw_mixer = W.add("raw_mixer", (int64_t)p->ns_mixer * (w.mixer_size > 0 ? w.mixer_size : 4));
  p->rsum = W.add("rsum", (int64_t)w.agent_end + w.mixer_size + 4);
  // one slab per workgroup of the register-blocked weight-gradient launch -- only where a step of this configuration can take it (w2_shape_can)
  p->raw2 = (p->chunks == 1 && p->D % 2 == 0 && p->D <= 1024 && (c->vdn || (p->S % 2 == 0 && p->S <= 1024)) && p->NM <= 1024)
                ? W.add("raw2", (int64_t)w2_max_workgroups() * kW2Slab) : -1;
  p->q_all = W.add("q_all", R * p->A);
  p->dbg = W.add("dbg", 2 * 16 * 4096 + 2 * 16 * 2400);   // per-wave s_memtime stamps (ope_set_debug)
  // wide-state mixer (ope_mixer_wide.hip): stream-K partial sums of the first hyper-layers
  // fused (t, b)-row chain: whole steps of one shared recurrent policy on one stream. ope_qmix_cfg.chain_path: 0 by shape (process default
  // OPE_CHAIN = 1 | 0, read once), 1 the four separate kernels, 2 the fused pair (OPE_EINVAL from the step when the configuration cannot run
  // it). Its first-hyper-layer kernel streams the weights per 16-row wave (fine up to MMM2's S = 322); for the really wide states
  // (--use_global_all_local_state: S = 2 232) "by shape" keeps the stream-K GEMM + the separate kernels.
  static const int chain_env = getenv("OPE_CHAIN") ? atoi(getenv("OPE_CHAIN")) : 1;
  constexpr int kChainAutoS = 512;
  p->chain_can = c->phase == 0 && !c->mlp && p->chunks == 1 && c->mixer_path != 3 && qchain_shape_ok(p->N, p->A);
  // (more than 8 agents = two per wave of the chain kernel, without the register prefetch: measured 0.6091 vs 0.6048 ms at QMIX-MMM2 -- not
  // picked by shape)
  p->chain = p->chain_can && (c->chain_path == 2 || p->hyp1 || (c->chain_path == 0 && chain_env != 0 && p->N <= 8 && (c->vdn || p->S <= kChainAutoS)));
  p->wide = !c->vdn && !p->chain && c->phase != 1 && c->phase != 3 && (c->mixer_path == 3 || (c->mixer_path == 0 && p->S > kWideAutoS));
  p->mix_slab = p->wide ? W.add("mix_slab", wide_slab_floats((int)p->TB, p->S)) : -1;
  p->hb1 = W.add("hb1", TB * OPE_MIX);
  p->hw1_t = W.add("hw1_t", TB * OPE_HYP); p->hw2_t = W.add("hw2_t", TB * OPE_HYP); p->hb2_t = W.add("hb2_t", TB * OPE_HYP);
  p->hb1_t = W.add("hb1_t", TB * OPE_MIX);
  p->v1_t = W.add("v1_t", TB * p->NM); p->v2_t = W.add("v2_t", TB * OPE_MIX);
  // live-row plan (ope_live.hip): int32 tables, built on the device at the start of every whole recurrent step that runs on packed rows
  p->live = (!c->mlp && c->phase == 0 && live_plan_shape_ok(p->T, p->N, p->B)) ? W.add("live_plan", live_plan_ints(p->T, p->N, p->B)) : -1;
  p->live1 = p->live >= 0 ? W.add("live_plan1", live_plan_ints(p->T, p->N, p->B)) : -1;      // (a second one: plans built ahead of the step, ope_store_live_plan)
  if (p->layerN == 2) {      // second hidden block: the trunk's output of both nets, the block's saves and adjoints
    p->a2 = W.add("a2", R * OPE_H); p->a2_t = W.add("a2_t", R * OPE_H);
    p->xhat3 = W.add("xhat3", R * OPE_H); p->rstd3 = W.add("rstd3", R); p->mask3 = W.add("mask3", 2 * R);
    p->dz3 = W.add("dz3", R1 * OPE_H); p->da2 = W.add("da2", R1 * OPE_H);
  }
}

}  // namespace

namespace ope {
void note_launch(const char* name, int p0, int p1) {
  char item[96];
  int n = p0 < 0 ? snprintf(item, sizeof(item), "%s", name) : (p1 < 0 ? snprintf(item, sizeof(item), "%s<%d>", name, p0) : snprintf(item, sizeof(item), "%s<%d,%d>", name, p0, p1));
  if (n < 0) return;
  if (n >= (int)sizeof(item)) n = (int)sizeof(item) - 1;
  ++g_launch_n;
  if (g_launch_len + n + 2 >= (int)sizeof(g_launch_log)) return;     // bounded: later names are counted, not kept
  if (g_launch_len) g_launch_log[g_launch_len++] = ';';
  memcpy(g_launch_log + g_launch_len, item, n);
  g_launch_len += n;
  g_launch_log[g_launch_len] = 0;
}
void clear_launch_log() { g_launch_len = 0; g_launch_n = 0; g_launch_log[0] = 0; }
}  // namespace ope

extern "C" int ope_last_launches(char* out, int32_t cap) {
  if (out && cap > 0) {
    const int n = g_launch_len < cap - 1 ? g_launch_len : cap - 1;
    memcpy(out, g_launch_log, n);
    out[n] = 0;
  }
  return g_launch_n;
}

extern "C" int ope_version(void) { return OPE_VERSION; }
extern "C" int64_t ope_abi_sizeof(const char* name) {
  if (!name) return -1;
#define OPE_SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T)
  OPE_SZ(ope_dims); OPE_SZ(ope_fields); OPE_SZ(ope_gather_tune); OPE_SZ(ope_obs_ref); OPE_SZ(ope_qmix_cfg); OPE_SZ(ope_adam_cfg);
  OPE_SZ(ope_ddpg_cfg); OPE_SZ(ope_mlp_batch); OPE_SZ(ope_ddpg_opt); OPE_SZ(ope_rddpg_cfg); OPE_SZ(ope_allreduce_ctx); OPE_SZ(ope_live_target);
#undef OPE_SZ
  return -1;
}
extern "C" const char* ope_strerror(int code) {
  switch (code) {
    case OPE_OK: return "ok";
    case OPE_EINVAL: return "invalid argument or unsupported dimension";
    case OPE_ELAUNCH: return "HIP kernel launch failed";
    case OPE_ENOSPC: return "workspace too small";
    case OPE_EHIP: return "HIP runtime call failed (allocation / IPC)";
    default: return "unknown error";
  }
}
extern "C" void ope_set_debug(int on) { g_debug = on; }
// process default of "fold the finalize step into the wgrad2 slab sum" (launch_wgrad2_fin): 1 | 0; OPE_W2_FIN sets the initial value
static int g_w2_fin = getenv("OPE_W2_FIN") ? atoi(getenv("OPE_W2_FIN")) : 1;
extern "C" void ope_set_w2_fin(int on) { g_w2_fin = on ? 1 : 0; }
extern "C" void ope_set_scan_kernel(int family, int waves_per_row) {
  ope::g_scan_family = (family == 1 || family == 4) ? family : 0;
  ope::g_scan_waves = (waves_per_row == 2 || waves_per_row == 4) ? waves_per_row : 0;
}

extern "C" int64_t ope_qmix_param_layout(const ope_qmix_cfg* cfg, int64_t* offsets, int64_t* sizes) {
  if (!cfg_ok(cfg)) return OPE_EINVAL;
  const int D = cfg->dims.obs_dim, A = cfg->dims.act_dim, N = cfg->dims.n_agents, S = cfg->dims.state_dim;
  const AgentLayout L = cfg->mlp ? ope_agent_layout_mlp(D, A, 0) : ope_agent_layout(D, A, 0, cfg->dims.layer_N);
  int na = 0;
  auto put = [&](int off, int size) {
    if (offsets) offsets[na] = off;
    if (sizes) sizes[na] = size;
    ++na;
  };
  put(L.fn_w, D); put(L.fn_b, D); put(L.fc1_w, OPE_H * D); put(L.fc1_b, OPE_H); put(L.ln1_w, OPE_H); put(L.ln1_b, OPE_H);
  put(L.fch_w, OPE_H * OPE_H); put(L.fch_b, OPE_H); put(L.lnh_w, OPE_H); put(L.lnh_b, OPE_H);
  put(L.fc2_w, OPE_H * OPE_H); put(L.fc2_b, OPE_H); put(L.ln2_w, OPE_H); put(L.ln2_b, OPE_H);
  if (L.layer_N == 2) { put(L.fc2b_w, OPE_H * OPE_H); put(L.fc2b_b, OPE_H); put(L.ln2b_w, OPE_H); put(L.ln2b_b, OPE_H); }
  if (!cfg->mlp) {
    put(L.wih, 3 * OPE_H * OPE_H); put(L.whh, 3 * OPE_H * OPE_H); put(L.bih, 3 * OPE_H); put(L.bhh, 3 * OPE_H);
    put(L.lno_w, OPE_H); put(L.lno_b, OPE_H);
  }
  put(L.q_w, A * OPE_H); put(L.q_b, A);
  if (cfg->vdn) return L.end;
  if (cfg->hypernet_layers == 1) {     // 10 tensors: hyper_w1.{weight,bias}, hyper_w2.*, hyper_b1.*, hyper_b2.0.*, hyper_b2.2.*
    const MixerLayout M1 = ope_mixer_layout1(N, S, L.end);
    put(M1.w1a_w, N * OPE_MIX * S); put(M1.w1a_b, N * OPE_MIX); put(M1.w2a_w, OPE_MIX * S); put(M1.w2a_b, OPE_MIX);
    put(M1.b1_w, OPE_MIX * S); put(M1.b1_b, OPE_MIX); put(M1.b2a_w, OPE_HYP * S); put(M1.b2a_b, OPE_HYP); put(M1.b2b_w, OPE_HYP); put(M1.b2b_b, 1);
    return M1.end;
  }
  const MixerLayout M = ope_mixer_layout(N, S, L.end);
  const int mo[OPE_QMIX_NPARAM_MIXER] = {M.w1a_w, M.w1a_b, M.w1b_w, M.w1b_b, M.w2a_w, M.w2a_b, M.w2b_w, M.w2b_b,
                                         M.b1_w, M.b1_b, M.b2a_w, M.b2a_b, M.b2b_w, M.b2b_b};
  const int ms[OPE_QMIX_NPARAM_MIXER] = {OPE_HYP * S, OPE_HYP, N * OPE_MIX * OPE_HYP, N * OPE_MIX, OPE_HYP * S, OPE_HYP,
                                         OPE_MIX * OPE_HYP, OPE_MIX, OPE_MIX * S, OPE_MIX, OPE_HYP * S, OPE_HYP, OPE_HYP, 1};
  for (int i = 0; i < OPE_QMIX_NPARAM_MIXER; ++i) put(mo[i], ms[i]);
  return M.end;
}

extern "C" int64_t ope_qmix_workspace_bytes(const ope_qmix_cfg* cfg) {
  if (!cfg_ok(cfg)) return OPE_EINVAL;
  Plan p;
  make_plan(cfg, &p);
  return p.ws.total * (int64_t)sizeof(float);
}

extern "C" int64_t ope_qmix_workspace_find(const ope_qmix_cfg* cfg, const char* name, int64_t* n_floats) {
  if (!cfg_ok(cfg) || !name) return OPE_EINVAL;
  Plan p;
  make_plan(cfg, &p);
  const int64_t off = p.ws.find(name, n_floats);
  return off < 0 ? -1 : off * (int64_t)sizeof(float);
}

extern "C" int ope_qmix_workspace_init(const ope_qmix_cfg* cfg, void* workspace, int64_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!cfg_ok(cfg) || !workspace) return OPE_EINVAL;
  Plan p;
  make_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  float* W = (float*)workspace;
  int rc;
  if ((rc = launch_fill(W + p.ln_zero, p.R, 0.f, (hipStream_t)stream))) return rc;
  if ((rc = launch_fill(W + p.gsq_part, p.n_gsq, 0.f, (hipStream_t)stream))) return rc;
  if (p.live >= 0 && (rc = launch_fill(W + p.live, 16, 0.f, (hipStream_t)stream))) return rc;      // (header + the executed-row accumulators)
  if (p.live1 >= 0 && (rc = launch_fill(W + p.live1, 16, 0.f, (hipStream_t)stream))) return rc;
  return launch_fill(W + p.ln_one, p.R, 1.f, (hipStream_t)stream);
}

// Live rows (LivePlan, ope_common.h; ope_qmix_cfg.live_rows): whole steps of one shared recurrent policy on the kernels that know packed
// rows -- trunk_fwd4 (pair), gru_fwd4 / gru_bwd4, mixer_hyp + qchain, trunk_bwd4, wgrad2. This is the part of the decision that depends on the
// configuration alone (the step adds: observations gathered, no debug outputs, a weight-gradient table wgrad2 can plan).
// Process default of "by shape": OPE_LIVE_ROWS = 1 | 0 (read once).
static bool w2_shape_can(const ope_qmix_cfg* cfg, const Plan& p) {
  return p.chunks == 1 && p.D % 2 == 0 && p.D <= 1024 && (cfg->vdn || (p.S % 2 == 0 && p.S <= 1024)) && p.NM <= 1024;      // (rows 8-byte aligned at least)
}
static bool w2_wanted(const ope_qmix_cfg* cfg) {
  static const int w2_env = getenv("OPE_WGRAD2") ? atoi(getenv("OPE_WGRAD2")) : 1;
  return cfg->wgrad_path == 2 || (cfg->wgrad_path == 0 && w2_env);
}
static bool live_cfg_ok(const ope_qmix_cfg* cfg, const Plan& p) {
  static const int live_env = getenv("OPE_LIVE_ROWS") ? atoi(getenv("OPE_LIVE_ROWS")) : 1;
  auto scan4 = [&](int64_t rows) { const int want = (cfg->scan_family == 1 || cfg->scan_family == 4) ? cfg->scan_family : g_scan_family; return (want ? want : (rows <= kGru4MaxRows ? 4 : 1)) == 4; };
  const bool tanh_on = (cfg->dims.flags & OPE_DIMS_TANH) != 0;
  return p.live >= 0 && p.chain && cfg->phase == 0 && !p.mlp && p.chunks == 1 && p.layerN == 1 && !tanh_on && !cfg->debug && w2_shape_can(cfg, p) && w2_wanted(cfg) &&
         trunk4_pair_can(p.D, p.R, cfg->trunk_path, tanh_on) && trunk_bwd4_can(p.R1, cfg->trunk_path, tanh_on) && scan4(2 * (int64_t)p.NB) && scan4(p.NB) &&
         (int64_t)p.R * 1024 < ((int64_t)1 << 32) &&      // (row byte offsets are 32-bit in the scan kernels)
         (cfg->live_rows >= 2 || (cfg->live_rows == 0 && live_env));
}
extern "C" int ope_qmix_live_rows_ok(const ope_qmix_cfg* cfg) {
  if (!cfg_ok(cfg)) return 0;
  Plan p;
  make_plan(cfg, &p);
  return live_cfg_ok(cfg, p) ? 1 : 0;
}
extern "C" int ope_qmix_live_target(const ope_qmix_cfg* cfg, void* workspace, int64_t workspace_bytes, int32_t which, ope_live_target* out) {
  if (!cfg_ok(cfg) || !workspace || !out || which < 0 || which > 1) return OPE_EINVAL;
  Plan p;
  make_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  if (p.live < 0) return OPE_EINVAL;
  float* W = (float*)workspace;
  out->plan = reinterpret_cast<int32_t*>(W + (which ? p.live1 : p.live)); out->err_abs = W + p.err_abs; out->loss_part = W + p.loss_part; out->n_loss_part = p.n_loss_tiles * 4;
  out->n_agents = p.N; out->episode_length = p.T; out->batch = p.B; out->copy_live_only = 0;
  return OPE_OK;
}
extern "C" int ope_qmix_live_plan(const ope_qmix_cfg* cfg, const float* dones_env, void* workspace, int64_t workspace_bytes, void* stream) {
  (void)hipGetLastError();
  if (!cfg_ok(cfg) || !dones_env || !workspace) return OPE_EINVAL;
  Plan p;
  make_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  if (p.live < 0) return OPE_EINVAL;
  float* W = (float*)workspace;
  LiveArgs la;
  la.T = p.T; la.N = p.N; la.B = p.B; la.dones_env = dones_env; la.plan = reinterpret_cast<int*>(W + p.live);
  la.err_abs = W + p.err_abs; la.loss_part = W + p.loss_part; la.n_loss_part = p.n_loss_tiles * 4;
  return launch_live_plan(la, (hipStream_t)stream);
}

// ope_qmix_obs_ref_ok: the static part of "can this configuration read observation rows from the store" (the launchers check the rest)
static bool obs_ref_cfg_ok(const ope_qmix_cfg* cfg) {
  if (!cfg_ok(cfg) || cfg->mlp || cfg->phase != 0 || cfg->dims.layer_N == 2 || (cfg->dims.flags & OPE_DIMS_TANH)) return false;
  const ope_dims& d = cfg->dims;
  const int KC = (d.obs_dim + 15) >> 4;
  const int64_t R = (int64_t)(d.episode_length + 1) * d.n_agents * cfg->batch;
  static const int t4 = getenv("OPE_TRUNK4") ? atoi(getenv("OPE_TRUNK4")) : 1;
  return d.obs_dim % 4 == 0 && (KC == 4 || KC == 8 || KC == 12 || KC == 16) && d.state_dim % 4 == 0 && cfg->batch <= kObsRefMaxB &&
         R < kObsRefMaxRows && (cfg->trunk_path == 4 || (cfg->trunk_path == 0 && t4 && R >= trunk4_pair_min_rows() && cfg->time_chunks <= 1));
}
extern "C" int ope_qmix_obs_ref_ok(const ope_qmix_cfg* cfg) { return obs_ref_cfg_ok(cfg) ? 1 : 0; }

static int qmix_step(const ope_qmix_cfg* cfg, const ope_fields* batch, const ope_obs_ref* oref, const float* theta, const float* theta_tgt,
                     const float* per_weights, void* workspace, int64_t workspace_bytes, float* grad, float* td_abs_stats, void* stream);

// ope_qmix_signal_event: an event the NEXT step launched from this thread records on its stream in front of one of its launches
static thread_local hipEvent_t g_signal_ev = nullptr;
static thread_local int g_signal_at = 0;
extern "C" int ope_qmix_signal_event(void* event, int32_t at) {
  if (event && (at < 1 || at > 5)) return OPE_EINVAL;
  g_signal_ev = (hipEvent_t)event;
  g_signal_at = event ? at : 0;
  return OPE_OK;
}
static int step_signal(int point, hipStream_t st) {
  if (g_signal_at != point || !g_signal_ev) return OPE_OK;
  const hipEvent_t ev = g_signal_ev;
  g_signal_ev = nullptr; g_signal_at = 0;
  return hipEventRecord(ev, st) == hipSuccess ? OPE_OK : OPE_ELAUNCH;
}

extern "C" int ope_qmix_loss_and_grad(const ope_qmix_cfg* cfg, const ope_fields* batch, const float* theta,
                                      const float* theta_tgt, const float* per_weights, void* workspace,
                                      int64_t workspace_bytes, float* grad, float* td_abs_stats, void* stream) {
  return qmix_step(cfg, batch, nullptr, theta, theta_tgt, per_weights, workspace, workspace_bytes, grad, td_abs_stats, stream);
}

extern "C" int ope_qmix_loss_and_grad_ref(const ope_qmix_cfg* cfg, const ope_fields* batch, const ope_obs_ref* obs, const float* theta,
                                          const float* theta_tgt, const float* per_weights, void* workspace, int64_t workspace_bytes,
                                          float* grad, float* td_abs_stats, void* stream) {
  if (!obs || !obs->store_obs || !obs->inds || obs->capacity < 1 || !obs_ref_cfg_ok(cfg)) return OPE_EINVAL;
  return qmix_step(cfg, batch, obs, theta, theta_tgt, per_weights, workspace, workspace_bytes, grad, td_abs_stats, stream);
}

static int qmix_step(const ope_qmix_cfg* cfg, const ope_fields* batch, const ope_obs_ref* oref, const float* theta, const float* theta_tgt,
                     const float* per_weights, void* workspace, int64_t workspace_bytes, float* grad, float* td_abs_stats, void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  clear_launch_log();
  if (!cfg_ok(cfg) || !batch || !theta || !theta_tgt || !workspace || !grad) return OPE_EINVAL;
  // Phased calls (several policies under one mixer, see ope.h): 1 = agent networks forward only (leaves "agent_q" /
  // "agent_nq" in the workspace), 2 = mixer + TD loss + mixer gradients only (reads "agent_q" / "agent_nq" the caller
  // assembled, leaves "d_agent_q"), 3 = agent networks backward only (reads "d_agent_q"). 0 = the whole step.
  const int phase = cfg->phase;
  const int dbg_on = g_debug | cfg->debug;      // per-call request or the process default
  const bool do_fwd = phase == 0 || phase == 1, do_mix = phase == 0 || phase == 2, do_bwd = phase == 0 || phase == 3;
  if ((do_fwd || do_bwd) && ((!batch->obs && !oref) || !batch->acts)) return OPE_EINVAL;
  ObsRef ref;
  memset(&ref, 0, sizeof(ref));
  const float* const obs_rows = oref ? oref->store_obs : batch->obs;      // the batch's [T+1][N][B][D] rows, or the store's ring read through `ref`
  if (oref) {
    ref.inds = oref->inds; ref.cap = oref->capacity; ref.B = cfg->batch; ref.TTN = (cfg->dims.episode_length + 1) * cfg->dims.n_agents;
  }
  if (do_mix && (!batch->share_obs || !batch->rewards || !batch->dones_env)) return OPE_EINVAL;
  if (phase == 2 && cfg->vdn == 0 && cfg->dims.n_agents < 1) return OPE_EINVAL;
  if ((phase == 1 || phase == 3) && !cfg->vdn) return OPE_EINVAL;   // per-policy parts carry no mixer parameters
  if (do_mix && cfg->use_per && !per_weights) return OPE_EINVAL;
  Plan p;
  make_plan(cfg, &p);
  if (workspace_bytes < p.ws.total * (int64_t)sizeof(float)) return OPE_ENOSPC;
  if (cfg->chain_path == 2 && !p.chain) return OPE_EINVAL;      // the fused chain was asked for explicitly and cannot run this configuration
  if (p.hyp1 && !p.chain) return OPE_EINVAL;                    // one-layer hyper-networks exist on the fused chain only
  hipStream_t st = (hipStream_t)stream;
  float* W = (float*)workspace;
  int rc;

  // Time-chunked, two-stream schedule (recurrent nets). The GRU scans are latency-bound chains that occupy a fraction
  // of the machine; the trunk / head / weight-gradient kernels are row-parallel over time. The episode is cut into C
  // time chunks: the scan of chunk c runs on a side stream while the main stream computes the trunks of chunk c+1 (and
  // the heads of chunk c-1); in the backward pass the BPTT of chunk c overlaps the trunk adjoint and the weight-gradient
  // K-slices of chunk c+1. Results are identical to the unchunked schedule (same kernels, same per-row arithmetic; the
  // weight-gradient K-splits are summed in a fixed order either way).
  const int C = p.chunks;
  const bool use_side = C > 1;
  SidePool* sp = nullptr;
  hipStream_t side = st;
  if (use_side) {
    sp = side_pool();
    if (!sp) return OPE_ELAUNCH;
    side = sp->s;
  }
  int ev_i = 0;
  auto sync_to = [&](hipStream_t from, hipStream_t to) -> int {   // `to` waits for everything issued so far on `from`
    if (from == to) return OPE_OK;
    hipEvent_t e = sp->ev[ev_i++ % kSideEvents];
    if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) return OPE_ELAUNCH;
    return OPE_OK;
  };
  if ((rc = sync_to(st, side))) return rc;   // fork: the side stream starts after the caller's prior work (the gather)

  // transposed copies of the matrices the backward chains read column-wise: carried as extra workgroups by the head launch (separate
  // kernels, MFMA head in one piece) or by the first-hyper-layer launch (fused chain), else one launch of their own
  Transp4 tr;
  memset(&tr, 0, sizeof(tr));
  {
    int nt = 0, tot = 0;
    auto add = [&](const float* src, int rows, int cols, float* dst) {
      tr.src[nt] = src; tr.dst[nt] = dst; tr.rows[nt] = rows; tr.cols[nt] = cols; tr.begin[nt] = tot; tot += rows * cols; ++nt;
    };
    if (!p.mlp && phase != 2) add(theta + p.AL.wih, 3 * OPE_H, OPE_H, W + p.thetaT);
    if (phase != 2) add(theta + p.AL.fc2_w, OPE_H, OPE_H, W + p.thetaT + OPE_H * 3 * OPE_H);
    if (p.layerN == 2) add(theta + p.AL.fc2b_w, OPE_H, OPE_H, W + p.thetaT + OPE_H * 3 * OPE_H + OPE_H * OPE_H);
    if (!cfg->vdn && phase != 1 && !p.hyp1) {
      add(theta + p.ML.w1b_w, p.NM, OPE_HYP, W + p.mixT);
      add(theta + p.ML.w2b_w, OPE_MIX, OPE_HYP, W + p.mixT + (int64_t)OPE_HYP * p.NM);
    }
    tr.n = nt; tr.total = tot;
  }
  TdArgs td;
  td.B = p.B; td.N = p.N; td.gamma = cfg->gamma; td.use_huber = cfg->use_huber; td.huber_delta = cfg->huber_delta;
  td.rewards = batch->rewards; td.dones_env = batch->dones_env; td.per_weights = cfg->use_per ? per_weights : nullptr;

  // ---- decisions that must be taken before the first launch (nothing may fail between two kernels of a step) -------------------------
  // the register-blocked weight-gradient launch (ope_wgrad2.hip) takes whole steps on one stream: no time chunks, no rows read in place
  const bool w2_can = !use_side && !oref && p.raw2 >= 0 && w2_shape_can(cfg, p);
  if (cfg->wgrad_path == 2 && !w2_can) return OPE_EINVAL;
  bool want_w2 = w2_can && w2_wanted(cfg);
  bool merge_hh = want_w2;
  // live rows: any configuration live_cfg_ok refuses computes every padded row as before (live_rows = 2 then returns OPE_EINVAL)
  bool live = want_w2 && !oref && !dbg_on && batch->obs && live_cfg_ok(cfg, p);
  LivePlan lp;
  memset(&lp, 0, sizeof(lp));
  if (live) lp = live_plan_view(reinterpret_cast<const int*>(W + (cfg->live_rows == 4 ? p.live1 : p.live)), p.T, p.N, p.B);
  const Raw& rw = p.raw;
  // weight-gradient problem tables: mixer problems (K = T*B) go first, on the main stream, while the side stream runs the
  // BPTT of the last chunk; the agent problems are cut into the same time chunks (each chunk = its own K-splits/slabs)
  auto prob = [&](WgTable& wt, const float* A, int lda, int M, const float* Bm, int ldb, int N, int K, int out_off, int ldc, int s_off,
                  int nsplit, int64_t base, int64_t stride) -> WgProb& {
    WgProb& q = wt.p[wt.n++];
    q.A = A; q.lda = lda; q.M = M; q.B = Bm; q.ldb = ldb; q.N = N; q.K = K; q.b_shift = 0; q.ln_mu = W + p.ln_zero; q.ln_rstd = W + p.ln_one;
    q.out_off = out_off; q.ldc = ldc; q.s_off = s_off; q.nsplit = nsplit; q.raw_base = base; q.raw_stride = stride;
    q.rs_base = base == p.raw_mixer ? rw.agent_end : 0;      // (where launch_split_reduce puts the two regions in `rsum`)
    if (live) {      // packed rows: the live count from the plan's header; operands that are batch fields get their row map below
      q.K_dev = lp.hdr + (base == p.raw_mixer ? 2 : 1);
      q.b_map = lp.srcrow; q.map_on = 0;
    }
    return q;
  };
  auto batch_rows = [&](WgProb& q, const int* map) { if (live) { q.b_map = map; q.map_on = 1; } };
  auto add_mixer_problems = [&](WgTable& wt) {
    const MixerLayout& M = p.ML;
    const int mbase = p.AL.end;
    const int64_t mb_ = p.raw_mixer, ms = rw.mixer_size;
    const int TBk = (int)p.TB;
    const float* S0 = batch->share_obs;  // rows 0..TB-1 are states at t < T
    if (p.hyp1) {      // every hyper-network layer reads the state: grad W = (pre-activation adjoint)^T S
      batch_rows(prob(wt, W + p.d_v1, p.NM, p.NM, S0, p.S, p.S, TBk, M.w1a_w - mbase, p.S, M.w1a_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
      batch_rows(prob(wt, W + p.d_v2, OPE_MIX, OPE_MIX, S0, p.S, p.S, TBk, M.w2a_w - mbase, p.S, M.w2a_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
      batch_rows(prob(wt, W + p.d_b1, OPE_MIX, OPE_MIX, S0, p.S, p.S, TBk, M.b1_w - mbase, p.S, M.b1_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
      batch_rows(prob(wt, W + p.d_hb2, OPE_HYP, OPE_HYP, S0, p.S, p.S, TBk, M.b2a_w - mbase, p.S, M.b2a_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
      prob(wt, W + p.dqtot, 4, 1, W + p.hb2, OPE_HYP, OPE_HYP, TBk, M.b2b_w - mbase, OPE_HYP, M.b2b_b - mbase, p.ns_mixer, mb_, ms);
      return;
    }
    batch_rows(prob(wt, W + p.d_hw1, OPE_HYP, OPE_HYP, S0, p.S, p.S, TBk, M.w1a_w - mbase, p.S, M.w1a_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
    prob(wt, W + p.d_v1, p.NM, p.NM, W + p.hw1, OPE_HYP, OPE_HYP, TBk, M.w1b_w - mbase, OPE_HYP, M.w1b_b - mbase, p.ns_mixer, mb_, ms);
    batch_rows(prob(wt, W + p.d_hw2, OPE_HYP, OPE_HYP, S0, p.S, p.S, TBk, M.w2a_w - mbase, p.S, M.w2a_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
    prob(wt, W + p.d_v2, OPE_MIX, OPE_MIX, W + p.hw2, OPE_HYP, OPE_HYP, TBk, M.w2b_w - mbase, OPE_HYP, M.w2b_b - mbase, p.ns_mixer, mb_, ms);
    batch_rows(prob(wt, W + p.d_b1, OPE_MIX, OPE_MIX, S0, p.S, p.S, TBk, M.b1_w - mbase, p.S, M.b1_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
    batch_rows(prob(wt, W + p.d_hb2, OPE_HYP, OPE_HYP, S0, p.S, p.S, TBk, M.b2a_w - mbase, p.S, M.b2a_b - mbase, p.ns_mixer, mb_, ms), lp.tbsrc);
    prob(wt, W + p.dqtot, 4, 1, W + p.hb2, OPE_HYP, OPE_HYP, TBk, M.b2b_w - mbase, OPE_HYP, M.b2b_b - mbase, p.ns_mixer, mb_, ms);
  };
  // agent problems over the rows [r0, r0 + K1) (whole time steps), slabs starting at `slab0`
  // the register-blocked weight-gradient launch (ope_wgrad2.hip) takes whole steps on one stream: no time chunks, no rows read in place
  auto add_agent_problems = [&](WgTable& wt, int64_t r0, int K1, int nsplit, int slab0) {
    const int64_t ab = p.raw_agent + (int64_t)slab0 * rw.agent_end, as = rw.agent_end;
    {
      WgProb& q = prob(wt, W + p.dz1 + r0 * OPE_H, OPE_H, OPE_H, oref ? obs_rows : obs_rows + r0 * p.D, p.D, p.D, K1, rw.P1, p.D, rw.s1, nsplit, ab, as);
      q.ln_mu = W + p.mu0 + r0; q.ln_rstd = W + p.rstd0 + r0; q.ln_on = 1;
      if (oref) { q.ref_row1 = (int)r0 + 1; wt.ref = ref; }
      batch_rows(q, lp.srcrow);      // (packed rows: the observation row of packed row k is batch row srcrow[k])
    }
    prob(wt, W + p.dz2 + r0 * OPE_H, OPE_H, OPE_H, W + p.xhat1 + r0 * OPE_H, OPE_H, OPE_H, K1, rw.P2, OPE_H, rw.s2, nsplit, ab, as);
    if (!p.mlp) {
      const float* dgi = W + p.dgi + r0 * 3 * OPE_H;
      if (p.layerN == 2) prob(wt, W + p.dz3 + r0 * OPE_H, OPE_H, OPE_H, W + p.xhat2 + r0 * OPE_H, OPE_H, OPE_H, K1, rw.P2b, OPE_H, rw.s2b, nsplit, ab, as);
      prob(wt, dgi, 3 * OPE_H, 3 * OPE_H, (p.layerN == 2 ? W + p.xhat3 : W + p.xhat2) + r0 * OPE_H, OPE_H, OPE_H, K1, rw.P3, OPE_H, rw.s3, nsplit, ab, as);
      // h_{t-1}: the first chunk shifts inside the kernel (rows of t = 0 see zeros), later chunks start one step back
      const float* hprev = r0 > 0 ? W + p.h + (r0 - p.NB) * OPE_H : W + p.h;
      const int shift = r0 > 0 ? 0 : p.NB;
      if (merge_hh) {      // register-blocked launch: one problem, dgi's r / z panels and dghn as the third panel, h_{t-1} read once for all three
        WgProb& q = prob(wt, dgi, 3 * OPE_H, 3 * OPE_H, hprev, OPE_H, OPE_H, K1, rw.WHH, OPE_H, rw.shh, nsplit, ab, as);
        q.b_shift = shift; q.A2 = W + p.dghn + r0 * OPE_H; q.lda2 = OPE_H; q.a2_from = 2;
        if (live) { q.b_shift = 0; q.b_map = lp.prevrow; q.map_on = 1; }      // h_{t-1} of packed row k sits at packed row prevrow[k] (none at t = 0)
      } else {
        prob(wt, dgi, 3 * OPE_H, 2 * OPE_H, hprev, OPE_H, OPE_H, K1, rw.WHH, OPE_H, rw.shh, nsplit, ab, as).b_shift = shift;
        prob(wt, W + p.dghn + r0 * OPE_H, OPE_H, OPE_H, hprev, OPE_H, OPE_H, K1, rw.WHH + 2 * OPE_H * OPE_H, OPE_H, rw.shh + 2 * OPE_H, nsplit, ab,
             as).b_shift = shift;
      }
    }
    // q head: fed by rnn.norm (recurrent) or directly by the trunk's LN2 (MLP) -- both "Linear after LayerNorm"
    prob(wt, W + p.dqoh + r0 * p.A4, p.A4, p.A, (p.mlp ? W + p.xhat2 : W + p.xhat_o) + r0 * OPE_H, OPE_H, OPE_H, K1, rw.E, OPE_H, rw.sq, nsplit, ab, as);
  };
  // whole step in one launch (no time chunks, no side stream): the problem table of the register-blocked form is built -- and its plan found or
  // rebuilt -- HERE, so that a table it cannot take (too many units, an operand that is only 4-byte aligned) falls back to the one-tile-per-wave
  // launch before anything runs ("by shape"), or fails before anything runs (wgrad_path = 2 / live rows asked for explicitly)
  WgTable wt2;
  memset(&wt2, 0, sizeof(wt2));
  static thread_local int w2_key[kMaxWgProbs * 10 + 1];
  static thread_local W2Table w2;
  auto plan_w2 = [&]() -> bool {
    memset(&wt2, 0, sizeof(wt2));
    if (do_bwd) add_agent_problems(wt2, 0, (int)p.R1, p.ns_chunk[0], 0);
    if (!cfg->vdn && do_mix) add_mixer_problems(wt2);
    if (wt2.n < 1) return true;
    if (wg_finish(&wt2) || !w2_ok(wt2)) return false;
    // (units and workgroup shares depend on the problems' shapes only: planned once per configuration -- the greedy passes cost ~15 us
    // of host time, which a short step like 3m's cannot hide; the pointers are refreshed on every call)
    int key[kMaxWgProbs * 10 + 1];
    memset(key, 0, sizeof(key));
    key[0] = wt2.n;
    for (int q = 0; q < wt2.n; ++q) {
      const WgProb& P = wt2.p[q];
      int same = q;      // first problem that walks the same A rows (what the XCD pairing looks at)
      for (int r = q - 1; r >= 0; --r)
        if (wt2.p[r].A == P.A) same = r;
      const int al = (((uintptr_t)P.A & 15) ? 2 : 0) | (((uintptr_t)P.B & 15) ? 4 : 0) | ((P.A2 && ((uintptr_t)P.A2 & 15)) ? 8 : 0);      // (the load width follows the operands' alignment)
      const int f[10] = {P.M, P.N, P.K, P.lda, P.ldb, P.b_shift, (P.s_off >= 0 ? 1 : 0) | al, same, P.A2 ? P.a2_from + 1 : 0, P.lda2};
      memcpy(key + 1 + 10 * q, f, sizeof(f));
    }
    if (memcmp(w2_key, key, sizeof(key)) != 0) {
      if (w2_build(wt2, &w2)) { w2_key[0] = -1; return false; }
      memcpy(w2_key, key, sizeof(key));
    }
    for (int q = 0; q < wt2.n; ++q) w2.p[q] = wt2.p[q];
    return true;
  };
  if (want_w2 && !plan_w2()) {
    if (cfg->wgrad_path == 2 || cfg->live_rows >= 2) return OPE_EINVAL;
    want_w2 = merge_hh = live = false;      // "by shape": the one-tile-per-wave launch on every padded row
    memset(&lp, 0, sizeof(lp));
  }
  if (cfg->live_rows >= 2 && !live) return OPE_EINVAL;
  if (live && cfg->live_rows >= 3) {      // the plan was built ahead of the step (ope_store_live_plan): only the per-row error array is cleared here
    if (td_abs_stats && (rc = launch_fill(W + p.err_abs, p.TB, 0.f, st))) return rc;      // (the chain kernel writes the live entries; td_stats reads all of them)
  } else if (live) {
    LiveArgs la;
    la.T = p.T; la.N = p.N; la.B = p.B; la.dones_env = batch->dones_env; la.plan = reinterpret_cast<int*>(W + p.live);
    la.err_abs = W + p.err_abs; la.loss_part = W + p.loss_part; la.n_loss_part = p.n_loss_tiles * 4;
    if ((rc = launch_live_plan(la, st))) return rc;
  }

  bool hyp_on_side = false, hyp_late = false;
  HypFirstArgs hyp_args;
  memset(&hyp_args, 0, sizeof(hyp_args));
  // ---- forward ----
  for (int c = 0; c < C && do_fwd; ++c) {
    const int64_t r0 = (int64_t)p.tb[c] * p.NB, rows = (int64_t)(p.tb[c + 1] - p.tb[c]) * p.NB;
    TrunkFwdArgs tf;
    memset(&tf, 0, sizeof(tf));
    tf.x = oref ? obs_rows : obs_rows + r0 * p.D; tf.R = (int)rows; tf.D = p.D; tf.theta = theta; tf.L = p.AL;
    tf.ref = ref; tf.ref_row0 = (int)r0;
    tf.no_fn = (cfg->dims.flags & OPE_DIMS_NO_FEATURE_NORM) ? 1 : 0;
    tf.tanh_act = (cfg->dims.flags & OPE_DIMS_TANH) ? 1 : 0;
    tf.lp = lp;
    tf.gi = p.mlp ? nullptr : W + p.gi + r0 * 3 * OPE_H; tf.a2_out = p.mlp ? W + p.h + r0 * OPE_H : nullptr;
    if (p.layerN == 2) { tf.gi = nullptr; tf.a2_out = W + p.a2 + r0 * OPE_H; }      // the trunk stops at the first block's output; ope_block.hip continues
    tf.mu0 = W + p.mu0 + r0; tf.rstd0 = W + p.rstd0 + r0;
    tf.xhat1 = W + p.xhat1 + r0 * OPE_H; tf.rstd1 = W + p.rstd1 + r0; tf.mask1 = (uint64_t*)(W + p.mask1) + r0;
    tf.xhat2 = W + p.xhat2 + r0 * OPE_H; tf.rstd2 = W + p.rstd2 + r0; tf.mask2 = (uint64_t*)(W + p.mask2) + r0;
    tf.dbg = dbg_on ? (long long*)(W + p.dbg) + 16 * 2400 : nullptr;
    TrunkFwdArgs tt = tf;
    tt.dbg = nullptr;
    tt.theta = theta_tgt; tt.gi = p.mlp ? nullptr : W + p.gi_t + r0 * 3 * OPE_H; tt.a2_out = p.mlp ? W + p.h_t + r0 * OPE_H : nullptr;
    if (p.layerN == 2) { tt.gi = nullptr; tt.a2_out = W + p.a2_t + r0 * OPE_H; }
    if ((rc = launch_trunk_fwd_pair(tf, tt, cfg->trunk_path, st))) return rc;    // one launch for both nets where the shape allows it (ope_trunk4.hip)
    if (p.layerN == 2) {
      BlockFwdArgs bf;
      memset(&bf, 0, sizeof(bf));
      bf.R = (int)rows; bf.x = W + p.a2 + r0 * OPE_H; bf.theta = theta; bf.L = p.AL; bf.gi = W + p.gi + r0 * 3 * OPE_H;
      bf.xhat3 = W + p.xhat3 + r0 * OPE_H; bf.rstd3 = W + p.rstd3 + r0; bf.mask3 = (uint64_t*)(W + p.mask3) + r0;
      if ((rc = launch_block_fwd(bf, st))) return rc;
      BlockFwdArgs bt;
      memset(&bt, 0, sizeof(bt));
      bt.R = (int)rows; bt.x = W + p.a2_t + r0 * OPE_H; bt.theta = theta_tgt; bt.L = p.AL; bt.gi = W + p.gi_t + r0 * 3 * OPE_H;
      if ((rc = launch_block_fwd(bt, st))) return rc;
    }
    if (p.mlp) continue;
    if (p.chain) {
      // The mixers' first hyper-layers need the centralized state only: launched here, in front of the scan (whose launch leaves the
      // matrix pipes idle), with the weight transposes of the backward kernels as passengers. VDN: only the transposes.
      if (cfg->vdn) {
        if ((rc = launch_transpose4(tr, st))) return rc;
      } else {
        HypFirstArgs hy;
        memset(&hy, 0, sizeof(hy));
        hy.TB = (int)p.TB; hy.B = p.B; hy.S = p.S; hy.theta0 = theta; hy.theta1 = theta_tgt; hy.share = batch->share_obs;
        hyp_tiles_for(p.ML, p.N, p.S, &hy);
        hy.out[0][HYP_HW1] = W + p.hw1; hy.out[0][HYP_HW2] = W + p.hw2; hy.out[0][HYP_HB2] = W + p.hb2; hy.out[0][HYP_HB1] = W + p.hb1;
        hy.out[1][HYP_HW1] = W + p.hw1_t; hy.out[1][HYP_HW2] = W + p.hw2_t; hy.out[1][HYP_HB2] = W + p.hb2_t; hy.out[1][HYP_HB1] = W + p.hb1_t;
        hy.out[0][HYP_V1] = W + p.v1; hy.out[0][HYP_V2] = W + p.v2; hy.out[1][HYP_V1] = W + p.v1_t; hy.out[1][HYP_V2] = W + p.v2_t;
        hy.side = tr;
        hy.lp = lp;
        // OPE_CHAIN_SIDE = 1: on the side stream, concurrent with the scan (fork behind the trunk launch, join in front of the chain kernel);
        // 2: the same, but launched BEHIND the scan (below), so that the scan's workgroups are resident first and the GEMM's workgroups
        // take the slots that are left (the scan waves run at s_setprio 3)
        static const int side_env = getenv("OPE_CHAIN_SIDE") ? atoi(getenv("OPE_CHAIN_SIDE")) : 0;
        if (side_env) {
          if (!sp && !(sp = side_pool())) return OPE_ELAUNCH;
          if (hipEventRecord(sp->ev[0], st) != hipSuccess || hipStreamWaitEvent(sp->s, sp->ev[0], 0) != hipSuccess) return OPE_ELAUNCH;
          if (side_env == 2) {
            hyp_late = true;
            hyp_args = hy;
          } else {
            if ((rc = launch_mixer_hyp(hy, sp->s))) return rc;
            if (hipEventRecord(sp->ev[1], sp->s) != hipSuccess) return OPE_ELAUNCH;
          }
          hyp_on_side = true;
        } else if ((rc = launch_mixer_hyp(hy, st))) return rc;
      }
    }
    hipStream_t scan_st = side;
    if ((rc = sync_to(st, side))) return rc;
    GruFwdArgs gf;   // live + target in one launch
    memset(&gf, 0, sizeof(gf));
    gf.nets = 2; gf.NB = p.NB; gf.L = p.tb[c + 1] - p.tb[c]; gf.theta0 = theta; gf.theta1 = theta_tgt;
    gf.gi0 = W + p.gi + r0 * 3 * OPE_H; gf.gi1 = W + p.gi_t + r0 * 3 * OPE_H;
    gf.h0out = W + p.h + r0 * OPE_H; gf.h1out = W + p.h_t + r0 * OPE_H;
    if (c > 0) { gf.hinit = W + p.h + (r0 - p.NB) * OPE_H; gf.hinit1 = W + p.h_t + (r0 - p.NB) * OPE_H; }
    gf.whh_off = p.AL.whh; gf.bhh_off = p.AL.bhh; gf.family = cfg->scan_family; gf.waves = cfg->scan_waves;
    gf.rg = W + p.rg + r0 * OPE_H; gf.zg = W + p.zg + r0 * OPE_H; gf.ng = W + p.ng + r0 * OPE_H; gf.ghn = W + p.ghn + r0 * OPE_H;
    gf.dbg = dbg_on ? (long long*)(W + p.dbg) + 71168 : nullptr;
    gf.lp = lp; gf.B = p.B; gf.N = p.N;
    if (c == 0 && (rc = step_signal(1, st))) return rc;
    if ((rc = launch_gru_fwd(gf, scan_st))) return rc;
    if (hyp_late) {
      if ((rc = launch_mixer_hyp(hyp_args, sp->s))) return rc;
      if (hipEventRecord(sp->ev[1], sp->s) != hipSuccess) return OPE_ELAUNCH;
      hyp_late = false;
    }
    if (C > 1 && hipEventRecord(sp->scan_done[c], side) != hipSuccess) return OPE_ELAUNCH;
  }
  const bool ride = C == 1 && p.A <= 32 && do_fwd;       // launch_head_fwd(mode 0) picks head_fwd_mfma for A <= 32
  if (!ride && !p.chain && tr.n > 0 && phase != 3)
    if ((rc = launch_transpose4(tr, st))) return rc;
  if (p.chain) {
    ChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.TB = (int)p.TB; ca.B = p.B; ca.N = p.N; ca.T = p.T; ca.A = p.A; ca.NB = p.NB; ca.vdn = cfg->vdn; ca.double_q = cfg->use_double_q;
    ca.theta0 = theta; ca.theta1 = theta_tgt; ca.AL = p.AL; ca.ML = p.ML; ca.mixT = W + p.mixT;
    ca.h0 = W + p.h; ca.h1 = W + p.h_t; ca.acts = batch->acts; ca.avail = batch->avail_acts;
    ca.hw1[0] = W + p.hw1; ca.hw2[0] = W + p.hw2; ca.hb2[0] = W + p.hb2; ca.hb1[0] = W + p.hb1;
    ca.hw1[1] = W + p.hw1_t; ca.hw2[1] = W + p.hw2_t; ca.hb2[1] = W + p.hb2_t; ca.hb1[1] = W + p.hb1_t;
    ca.v1x[0] = W + p.v1; ca.v1x[1] = W + p.v1_t; ca.v2x[0] = W + p.v2; ca.v2x[1] = W + p.v2_t;
    ca.td = td;
    ca.mask_target_max = (cfg->dims.flags & OPE_DIMS_MASK_TARGET_MAX) ? 1 : 0;
    ca.lp = lp;
    ca.xhat_o = W + p.xhat_o; ca.rstd_o = W + p.rstd_o; ca.act_idx = (int*)(W + p.act_idx);
    ca.loss_part = W + p.loss_part; ca.err_abs = W + p.err_abs; ca.dqtot = W + p.dqtot;
    ca.d_v1 = W + p.d_v1; ca.d_v2 = W + p.d_v2; ca.d_b1 = W + p.d_b1; ca.d_hw1 = W + p.d_hw1; ca.d_hw2 = W + p.d_hw2; ca.d_hb2 = W + p.d_hb2;
    ca.dh_out = W + p.dh_out; ca.dqoh = W + p.dqoh;
    if (dbg_on) {      // what the separate kernels leave in the workspace anyway: tests and tools read them
      ca.q_all = W + p.q_all; ca.agent_q = W + p.agent_q; ca.agent_nq = W + p.agent_nq; ca.qtot = W + p.qtot; ca.nqtot = W + p.nqtot;
      if (!p.hyp1) { ca.v1 = W + p.v1; ca.v2 = W + p.v2; }       // (one-layer hyper-networks: the first-layer kernel wrote them there already)
      ca.hpre = W + p.hpre; ca.d_agent_q = W + p.d_agent_q;
      ca.dbg = (long long*)(W + p.dbg);
    }
    if (hyp_on_side && hipStreamWaitEvent(st, sp->ev[1], 0) != hipSuccess) return OPE_ELAUNCH;
    if ((rc = step_signal(2, st))) return rc;
    if ((rc = launch_qchain(ca, st))) return rc;
    if ((rc = step_signal(3, st))) return rc;
  }
  for (int c = 0; c < C && do_fwd && !p.chain; ++c) {   // heads of chunk c as soon as its scan is done
    if (C > 1 && hipStreamWaitEvent(st, sp->scan_done[c], 0) != hipSuccess) return OPE_ELAUNCH;
    HeadFwdArgs hf;
    memset(&hf, 0, sizeof(hf));
    hf.r_begin = (int64_t)p.tb[c] * p.NB; hf.R = (int64_t)p.tb[c + 1] * p.NB;
    hf.NB = p.NB; hf.B = p.B; hf.N = p.N; hf.T = p.T; hf.A = p.A; hf.theta0 = theta; hf.theta1 = theta_tgt; hf.L = p.AL;
    hf.h0 = W + p.h; hf.h1 = W + p.h_t; hf.acts = batch->acts; hf.avail = batch->avail_acts; hf.double_q = cfg->use_double_q;
    hf.no_ln = p.mlp; hf.target_mask_avail = p.mlp || (cfg->dims.flags & OPE_DIMS_MASK_TARGET_MAX) != 0;
    hf.agent_q = W + p.agent_q; hf.agent_nq = W + p.agent_nq; hf.act_idx = (int*)(W + p.act_idx);
    hf.xhat_o = W + p.xhat_o; hf.rstd_o = W + p.rstd_o; hf.q_out = nullptr; hf.q_all = dbg_on ? W + p.q_all : nullptr;
    if (ride) hf.side = tr;
    if ((rc = launch_head_fwd(hf, 0, st))) return rc;
  }
  if (phase == 1) return OPE_OK;

  // ---- mixer forward + TD + mixer backward ----
  if (!do_mix || p.chain) {
  } else if (cfg->vdn) {
    VdnArgs va;
    va.TB = (int)p.TB; va.N = p.N; va.td = td; va.agent_q = W + p.agent_q; va.agent_nq = W + p.agent_nq;
    va.loss_part = W + p.loss_part; va.err_abs = W + p.err_abs; va.d_agent_q = W + p.d_agent_q;
    if ((rc = launch_vdn(va, st))) return rc;
  } else {
    MixerFwdArgs mf;
    mf.TB = (int)p.TB; mf.B = p.B; mf.N = p.N; mf.S = p.S; mf.theta0 = theta; mf.theta1 = theta_tgt; mf.L = p.ML;
    mf.share = batch->share_obs; mf.agent_q = W + p.agent_q; mf.agent_nq = W + p.agent_nq; mf.qtot = W + p.qtot; mf.nqtot = W + p.nqtot;
    mf.hw1 = W + p.hw1; mf.hw2 = W + p.hw2; mf.hb2 = W + p.hb2; mf.v1 = W + p.v1; mf.hpre = W + p.hpre; mf.v2 = W + p.v2;
    mf.dbg = dbg_on ? (long long*)(W + p.dbg) : nullptr;
    mf.k_stagger = 1; mf.path = cfg->mixer_path; mf.wide_slab = p.wide ? W + p.mix_slab : nullptr;
    if ((rc = launch_mixer_fwd(mf, st))) return rc;
    MixerBwdArgs mb;
    mb.TB = (int)p.TB; mb.N = p.N; mb.theta = theta; mb.thetaT = W + p.mixT; mb.L = p.ML; mb.td = td;
    mb.qtot = W + p.qtot; mb.nqtot = W + p.nqtot; mb.agent_q = W + p.agent_q;
    mb.hw1 = W + p.hw1; mb.hw2 = W + p.hw2; mb.hb2 = W + p.hb2; mb.v1 = W + p.v1; mb.hpre = W + p.hpre; mb.v2 = W + p.v2;
    mb.loss_part = W + p.loss_part; mb.err_abs = W + p.err_abs; mb.dqtot = W + p.dqtot; mb.d_agent_q = W + p.d_agent_q;
    mb.d_b1 = W + p.d_b1; mb.d_v2 = W + p.d_v2; mb.d_v1 = W + p.d_v1; mb.d_hw1 = W + p.d_hw1; mb.d_hw2 = W + p.d_hw2; mb.d_hb2 = W + p.d_hb2;
    if ((rc = launch_mixer_bwd(mb, st))) return rc;
  }
  if (td_abs_stats && do_mix)
    if ((rc = launch_td_stats(W + p.err_abs, p.T, p.B, td_abs_stats, st))) return rc;

  // ---- agent backward ----
  HeadBwdArgs hb;
  hb.R = p.R1; hb.NB = p.NB; hb.B = p.B; hb.N = p.N; hb.A = p.A; hb.theta = theta; hb.L = p.AL;
  hb.no_ln = p.mlp;
  hb.xhat_o = W + p.xhat_o; hb.rstd_o = W + p.rstd_o; hb.act_idx = (const int*)(W + p.act_idx); hb.d_agent_q = W + p.d_agent_q;
  hb.dh_out = W + p.dh_out; hb.dqoh = W + p.dqoh;
  if (do_bwd && !p.chain)
    if ((rc = launch_head_bwd(hb, st))) return rc;
  if ((rc = sync_to(st, side))) return rc;

  int agent_slabs = 0, mixer_slabs = 0;
  bool reduced = false;
  bool w2_pending = false;      // the weight-gradient launch summed its own slabs into `rsum`
  if (use_side && !cfg->vdn) {   // main stream, beside the BPTT of the last chunk on the side stream
    WgTable wm;
    memset(&wm, 0, sizeof(wm));
    add_mixer_problems(wm);
    if ((rc = wg_finish(&wm))) return rc;
    if ((rc = launch_wgrad(wm, W, st))) return rc;
    mixer_slabs = wg_slabs(wm, p.ns_mixer);
  }
  for (int c = C - 1; c >= 0; --c) {   // BPTT, last chunk first, on the side stream
    const int lo = p.tb[c], hi = p.tb[c + 1] < p.T ? p.tb[c + 1] : p.T;
    if (p.mlp || lo >= hi || !do_bwd) continue;
    GruBwdArgs gb;
    memset(&gb, 0, sizeof(gb));
    gb.NB = p.NB; gb.T = hi; gb.t_lo = lo; gb.theta = theta; gb.whh_off = p.AL.whh; gb.h = W + p.h;
    gb.rg = W + p.rg; gb.zg = W + p.zg; gb.ng = W + p.ng; gb.ghn = W + p.ghn; gb.dh_out = W + p.dh_out; gb.dgi = W + p.dgi; gb.dghn = W + p.dghn;
    gb.dh_in = hi < p.T ? W + p.dh_carry : nullptr;
    gb.dh_carry = lo > 0 ? W + p.dh_carry : nullptr;
    gb.dbg = dbg_on ? (long long*)(W + p.dbg) + 87552 : nullptr;
    gb.family = cfg->scan_family; gb.waves = cfg->scan_waves;
    gb.lp = lp; gb.B = p.B; gb.N = p.N;
    if ((rc = launch_gru_bwd(gb, side))) return rc;
    if (use_side && hipEventRecord(sp->bptt_done[c], side) != hipSuccess) return OPE_ELAUNCH;
  }
  for (int c = C - 1; c >= 0; --c) {   // trunk adjoint + weight-gradient K-slices of chunk c behind its BPTT
    const int lo = p.tb[c], hi = p.tb[c + 1] < p.T ? p.tb[c + 1] : p.T;
    if (lo >= hi) continue;
    if (use_side && !p.mlp && hipStreamWaitEvent(st, sp->bptt_done[c], 0) != hipSuccess) return OPE_ELAUNCH;
    const int64_t r0 = (int64_t)lo * p.NB;
    const int K1 = (hi - lo) * p.NB;
    TrunkBwdArgs tb;
    memset(&tb, 0, sizeof(tb));
    tb.R = K1; tb.theta = theta; tb.thetaT = W + p.thetaT; tb.L = p.AL;
    tb.tanh_act = (cfg->dims.flags & OPE_DIMS_TANH) ? 1 : 0;
    tb.R_dev = live ? lp.hdr + 1 : nullptr;
    tb.dgi = p.mlp ? nullptr : W + p.dgi + r0 * 3 * OPE_H; tb.da2_in = p.mlp ? W + p.dh_out + r0 * OPE_H : nullptr;
    if (p.layerN == 2 && do_bwd) {      // the second block's adjoint first: dgi -> dz3 (its weight gradient), da2 (what the trunk adjoint continues from)
      BlockBwdArgs bb;
      memset(&bb, 0, sizeof(bb));
      bb.R = K1; bb.dgi = W + p.dgi + r0 * 3 * OPE_H; bb.theta = theta; bb.L = p.AL;
      bb.wihT = W + p.thetaT; bb.fc2bT = W + p.thetaT + OPE_H * 3 * OPE_H + OPE_H * OPE_H;
      bb.xhat3 = W + p.xhat3 + r0 * OPE_H; bb.rstd3 = W + p.rstd3 + r0; bb.mask3 = (const uint64_t*)(W + p.mask3) + r0;
      bb.dz3 = W + p.dz3 + r0 * OPE_H; bb.da2 = W + p.da2 + r0 * OPE_H;
      if ((rc = launch_block_bwd(bb, st))) return rc;
      tb.dgi = nullptr; tb.da2_in = W + p.da2 + r0 * OPE_H;
    }
    tb.xhat1 = W + p.xhat1 + r0 * OPE_H; tb.rstd1 = W + p.rstd1 + r0; tb.mask1 = (const uint64_t*)(W + p.mask1) + r0;
    tb.xhat2 = W + p.xhat2 + r0 * OPE_H; tb.rstd2 = W + p.rstd2 + r0; tb.mask2 = (const uint64_t*)(W + p.mask2) + r0;
    tb.dz1 = W + p.dz1 + r0 * OPE_H; tb.dz2 = W + p.dz2 + r0 * OPE_H;
    if (do_bwd)
      if ((rc = launch_trunk_bwd_path(tb, cfg->trunk_path, st))) return rc;
    WgTable wt;
    memset(&wt, 0, sizeof(wt));
    if (want_w2) {      // (C = 1: this is the only pass; the table and its plan were made before the first launch)
      if (wt2.n > 0) {
        if ((rc = step_signal(4, st))) return rc;
        w2_pending = true;      // (launched below, once the segment table says whether the finalize step can be folded in)
        reduced = true;
      }
      continue;
    }
    if (do_bwd) add_agent_problems(wt, r0, K1, p.ns_chunk[c], agent_slabs);
    if (!use_side && !cfg->vdn && do_mix) add_mixer_problems(wt);
    if (wt.n > 0) {
      if ((rc = wg_finish(&wt))) return rc;
      if ((rc = launch_wgrad(wt, W, st))) return rc;
    }
    if (do_bwd) agent_slabs += wg_slabs(wt, p.ns_chunk[c]);
    if (!use_side && !cfg->vdn && do_mix) mixer_slabs = wg_slabs(wt, p.ns_mixer);
  }
  {
    SplitRed sr;
    sr.raw0 = W + p.raw_agent; sr.n0 = rw.agent_end; sr.ns0 = agent_slabs;
    sr.raw1 = W + p.raw_mixer; sr.n1 = cfg->vdn ? 0 : rw.mixer_size; sr.ns1 = mixer_slabs;
    sr.rsum = W + p.rsum;
    if (!(phase == 2 && cfg->vdn) && !reduced)       // (a VDN mixing part has no parameters: only the loss tail follows)
      if ((rc = launch_split_reduce(sr, st))) return rc;
  }

  // ---- finalize into the flat gradient ----
  FinTable ft;
  memset(&ft, 0, sizeof(ft));
  int k = 0;
  const AgentLayout& L = p.AL;
  const int srcE = rw.E, srcSq = rw.sq;
  auto seg = [&](int begin, int size, int kind, int src, int src_s, int M, int K, int w, int gamma, int beta) {
    FinSeg& s = ft.seg[k++];
    s.begin = begin; s.size = size; s.kind = kind; s.src = src; s.src_s = src_s; s.M = M; s.K = K; s.w = w; s.gamma = gamma; s.beta = beta;
  };
  if (phase == 2) {
    seg(0, L.end, FIN_SKIP, 0, 0, 0, 0, 0, 0, 0);   // the agent block belongs to the per-policy backward calls
  } else {
    if (cfg->dims.flags & OPE_DIMS_NO_FEATURE_NORM) {      // the constant ones / zeros in the feature_norm slots are not parameters: zero gradient
      seg(L.fn_w, p.D, FIN_ZERO, 0, 0, 0, 0, 0, 0, 0);
      seg(L.fn_b, p.D, FIN_ZERO, 0, 0, 0, 0, 0, 0, 0);
    } else {
      seg(L.fn_w, p.D, FIN_LNLIN_G, rw.P1, rw.s1, OPE_H, p.D, L.fc1_w, 0, 0);
      seg(L.fn_b, p.D, FIN_LNLIN_B, rw.P1, rw.s1, OPE_H, p.D, L.fc1_w, 0, 0);
    }
    seg(L.fc1_w, OPE_H * p.D, FIN_LNLIN_W, rw.P1, rw.s1, OPE_H, p.D, L.fc1_w, L.fn_w, L.fn_b);
    seg(L.fc1_b, OPE_H, FIN_COPY, rw.s1, 0, 0, 0, 0, 0, 0);
    seg(L.ln1_w, OPE_H, FIN_LNLIN_G, rw.P2, rw.s2, OPE_H, OPE_H, L.fc2_w, 0, 0);
    seg(L.ln1_b, OPE_H, FIN_LNLIN_B, rw.P2, rw.s2, OPE_H, OPE_H, L.fc2_w, 0, 0);
    seg(L.fch_w, 0, FIN_ZERO, 0, 0, 0, 0, 0, 0, 0);  // fc_h.*: registered, never used (mlp.py:21-23) -> zero gradient
    seg(L.fc2_w, OPE_H * OPE_H, FIN_LNLIN_W, rw.P2, rw.s2, OPE_H, OPE_H, L.fc2_w, L.ln1_w, L.ln1_b);
    seg(L.fc2_b, OPE_H, FIN_COPY, rw.s2, 0, 0, 0, 0, 0, 0);
    if (!p.mlp && p.layerN == 2) {      // LN2 feeds the second block, whose LayerNorm feeds W_ih
      seg(L.ln2_w, OPE_H, FIN_LNLIN_G, rw.P2b, rw.s2b, OPE_H, OPE_H, L.fc2b_w, 0, 0);
      seg(L.ln2_b, OPE_H, FIN_LNLIN_B, rw.P2b, rw.s2b, OPE_H, OPE_H, L.fc2b_w, 0, 0);
      seg(L.fc2b_w, OPE_H * OPE_H, FIN_LNLIN_W, rw.P2b, rw.s2b, OPE_H, OPE_H, L.fc2b_w, L.ln2_w, L.ln2_b);
      seg(L.fc2b_b, OPE_H, FIN_COPY, rw.s2b, 0, 0, 0, 0, 0, 0);
      seg(L.ln2b_w, OPE_H, FIN_LNLIN_G, rw.P3, rw.s3, 3 * OPE_H, OPE_H, L.wih, 0, 0);
      seg(L.ln2b_b, OPE_H, FIN_LNLIN_B, rw.P3, rw.s3, 3 * OPE_H, OPE_H, L.wih, 0, 0);
      seg(L.wih, 3 * OPE_H * OPE_H, FIN_LNLIN_W, rw.P3, rw.s3, 3 * OPE_H, OPE_H, L.wih, L.ln2b_w, L.ln2b_b);
    } else if (!p.mlp) {
      seg(L.ln2_w, OPE_H, FIN_LNLIN_G, rw.P3, rw.s3, 3 * OPE_H, OPE_H, L.wih, 0, 0);
      seg(L.ln2_b, OPE_H, FIN_LNLIN_B, rw.P3, rw.s3, 3 * OPE_H, OPE_H, L.wih, 0, 0);
      seg(L.wih, 3 * OPE_H * OPE_H, FIN_LNLIN_W, rw.P3, rw.s3, 3 * OPE_H, OPE_H, L.wih, L.ln2_w, L.ln2_b);
    }
    if (!p.mlp) {
      seg(L.whh, 3 * OPE_H * OPE_H, FIN_COPY, rw.WHH, 0, 0, 0, 0, 0, 0);
      seg(L.bih, 3 * OPE_H, FIN_COPY, rw.s3, 0, 0, 0, 0, 0, 0);
      seg(L.bhh, 3 * OPE_H, FIN_COPY, rw.shh, 0, 0, 0, 0, 0, 0);
      seg(L.lno_w, OPE_H, FIN_LNLIN_G, srcE, srcSq, p.A, OPE_H, L.q_w, 0, 0);
      seg(L.lno_b, OPE_H, FIN_LNLIN_B, srcE, srcSq, p.A, OPE_H, L.q_w, 0, 0);
      seg(L.q_w, p.A * OPE_H, FIN_LNLIN_W, srcE, srcSq, p.A, OPE_H, L.q_w, L.lno_w, L.lno_b);
    } else {   // MLP nets: LN2 feeds the q head
      seg(L.ln2_w, OPE_H, FIN_LNLIN_G, srcE, srcSq, p.A, OPE_H, L.q_w, 0, 0);
      seg(L.ln2_b, OPE_H, FIN_LNLIN_B, srcE, srcSq, p.A, OPE_H, L.q_w, 0, 0);
      seg(L.q_w, p.A * OPE_H, FIN_LNLIN_W, srcE, srcSq, p.A, OPE_H, L.q_w, L.ln2_w, L.ln2_b);
    }
    seg(L.q_b, p.A, FIN_COPY, srcSq, 0, 0, 0, 0, 0, 0);
  }
  if (!cfg->vdn && do_mix && p.hyp1) {
    const MixerLayout& M = p.ML;
    const int mo1[10] = {M.w1a_w, M.w1a_b, M.w2a_w, M.w2a_b, M.b1_w, M.b1_b, M.b2a_w, M.b2a_b, M.b2b_w, M.b2b_b};
    const int ms1[10] = {p.NM * p.S, p.NM, OPE_MIX * p.S, OPE_MIX, OPE_MIX * p.S, OPE_MIX, OPE_HYP * p.S, OPE_HYP, OPE_HYP, 1};
    for (int q = 0; q < 10; ++q) seg(mo1[q], ms1[q], FIN_COPY, rw.agent_end + (mo1[q] - p.AL.end), 0, 0, 0, 0, 0, 0);
  } else if (!cfg->vdn && do_mix) {  // mixer gradients: raw mixer slab has the same relative layout as the parameters
    const MixerLayout& M = p.ML;
    const int mo[OPE_QMIX_NPARAM_MIXER] = {M.w1a_w, M.w1a_b, M.w1b_w, M.w1b_b, M.w2a_w, M.w2a_b, M.w2b_w, M.w2b_b,
                                           M.b1_w, M.b1_b, M.b2a_w, M.b2a_b, M.b2b_w, M.b2b_b};
    const int ms[OPE_QMIX_NPARAM_MIXER] = {OPE_HYP * p.S, OPE_HYP, p.NM * OPE_HYP, p.NM, OPE_HYP * p.S, OPE_HYP,
                                           OPE_MIX * OPE_HYP, OPE_MIX, OPE_MIX * p.S, OPE_MIX, OPE_HYP * p.S, OPE_HYP, OPE_HYP, 1};
    for (int q = 0; q < OPE_QMIX_NPARAM_MIXER; ++q)
      seg(mo[q], ms[q], FIN_COPY, rw.agent_end + (mo[q] - p.AL.end), 0, 0, 0, 0, 0, 0);
  }
  if (do_mix) seg((int)p.P, OPE_GRAD_TAIL, FIN_TAIL, 0, 0, 0, 0, 0, 0, 0);
  ft.n = k;
  ft.total = p.P + (do_mix ? OPE_GRAD_TAIL : 0);
  if (w2_pending) {
    // wgrad2 with the finalize step folded in (ope_wgrad2.hip: launch_wgrad2_fin): the slabs of a LayerNorm-fed Linear hold dW = C gamma + s (x) beta
    // and the column partials of dgamma / dbeta, and ONE launch sums every slab straight into the flat gradient (+ loss tail, zero ranges, the
    // clip norm's partial sums of squares) -- instead of w2_reduce -> rsum -> finalize. Falls back when a segment has no producer in the table.
    FinMisc misc;
    if (g_w2_fin && phase == 0 && do_mix && w2_attach_fin(&w2, ft, &misc) && w2_fin_blocks(w2, misc) <= p.n_gsq) {
      misc.loss_part = W + p.loss_part; misc.n_loss_tiles = p.n_loss_tiles; misc.n_gsq_total = p.n_gsq;
      if ((rc = launch_wgrad2_fin(w2, W + p.raw2, theta, grad, W + p.gsq_part, misc, st))) return rc;
      if ((rc = step_signal(5, st))) return rc;
      if (g_signal_ev) {
        const int at = g_signal_at;
        if ((rc = step_signal(at, st))) return rc;
      }
      return OPE_OK;
    }
    if ((rc = launch_wgrad2(w2, W + p.raw2, W + p.rsum, st))) return rc;
    if ((rc = step_signal(5, st))) return rc;
  }
  if (finalize_blocks(ft) > p.n_gsq) return OPE_ENOSPC;
  // (n_loss_tiles < 0: no loss tail -- an agent-backward part writes its parameter block only)
  if ((rc = launch_finalize(ft, W + p.rsum, theta, W + p.loss_part, do_mix ? p.n_loss_tiles : -1, grad, st, W + p.gsq_part, p.n_gsq))) return rc;
  if (g_signal_ev) {      // (a point this step's path does not pass: the event still fires, behind the step's last launch)
    const int at = g_signal_at;
    if ((rc = step_signal(at, st))) return rc;
  }
  return OPE_OK;
}

// ---------------------------------------------------------------------------------------------------------
extern "C" int64_t ope_agent_forward_workspace_bytes(const ope_dims* d, int32_t seq_len, int32_t rows) {
  if (!d || seq_len < 1 || rows < 1) return OPE_EINVAL;
  const int64_t R = (int64_t)seq_len * rows;
  return (R * 3 * OPE_H + (d->layer_N == 2 ? R * OPE_H : 0) + 64) * (int64_t)sizeof(float);      // gi (+ the first block's output with layer_N = 2)
}

extern "C" int ope_agent_forward(const ope_dims* d, int32_t seq_len, int32_t rows, const float* obs, const float* h0,
                                 const float* theta, void* workspace, int64_t workspace_bytes, float* q_out, float* h_out,
                                 void* stream) {
  (void)hipGetLastError();  // drop stale errors from the caller's own HIP use
  if (!d || seq_len < 1 || rows < 1 || !obs || !theta || !workspace || !q_out || !h_out) return OPE_EINVAL;
  if (d->obs_dim < 1 || d->obs_dim > 512 || d->act_dim < 1) return OPE_EINVAL;
  if (workspace_bytes < ope_agent_forward_workspace_bytes(d, seq_len, rows)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = (int64_t)seq_len * rows;
  if (d->layer_N < 0 || d->layer_N > 2) return OPE_EINVAL;
  const AgentLayout L = ope_agent_layout(d->obs_dim, d->act_dim, 0, d->layer_N);
  float* gi = (float*)workspace;
  int rc;
  clear_launch_log();
  TrunkFwdArgs tf;
  memset(&tf, 0, sizeof(tf));
  tf.x = obs; tf.R = (int)R; tf.D = d->obs_dim; tf.theta = theta; tf.L = L; tf.gi = gi;
  tf.no_fn = (d->flags & OPE_DIMS_NO_FEATURE_NORM) ? 1 : 0;
  tf.tanh_act = (d->flags & OPE_DIMS_TANH) ? 1 : 0;
  if (tf.tanh_act && (L.layer_N == 2 || d->obs_dim > 384)) return OPE_EINVAL;
  if (L.layer_N == 2) { tf.gi = nullptr; tf.a2_out = gi + R * 3 * OPE_H; }      // the trunk stops at the first block; ope_block.hip continues
  if ((rc = launch_trunk_fwd(tf, false, st))) return rc;
  if (L.layer_N == 2) {
    BlockFwdArgs bf;
    memset(&bf, 0, sizeof(bf));
    bf.R = (int)R; bf.x = gi + R * 3 * OPE_H; bf.theta = theta; bf.L = L; bf.gi = gi;
    if ((rc = launch_block_fwd(bf, st))) return rc;
  }
  GruFwdArgs gf;
  memset(&gf, 0, sizeof(gf));
  gf.nets = 1; gf.NB = rows; gf.L = seq_len; gf.theta0 = theta; gf.theta1 = theta; gf.gi0 = gi; gf.gi1 = gi; gf.h0out = h_out; gf.h1out = h_out;
  gf.hinit = h0; gf.whh_off = L.whh; gf.bhh_off = L.bhh;
  if ((rc = launch_gru_fwd(gf, st))) return rc;
  HeadFwdArgs hf;
  memset(&hf, 0, sizeof(hf));
  hf.R = R; hf.NB = rows; hf.B = rows; hf.N = 1; hf.T = seq_len; hf.A = d->act_dim; hf.theta0 = theta; hf.theta1 = theta; hf.L = L;
  hf.h0 = h_out; hf.q_out = q_out;
  return launch_head_fwd(hf, 1, st);
}

// ---------------------------------------------------------------------------------------------------------
extern "C" int64_t ope_agent_forward_mlp_workspace_bytes(const ope_dims* d, int32_t rows) {
  if (!d || rows < 1) return OPE_EINVAL;
  return ((int64_t)rows * OPE_H + 64) * (int64_t)sizeof(float);
}

extern "C" int ope_agent_forward_mlp(const ope_dims* d, int32_t rows, const float* obs, const float* theta, void* workspace,
                                     int64_t workspace_bytes, float* q_out, void* stream) {
  (void)hipGetLastError();
  if (!d || rows < 1 || !obs || !theta || !workspace || !q_out) return OPE_EINVAL;
  if (d->obs_dim < 1 || d->obs_dim > 512 || d->act_dim < 1 || (d->flags & ~(OPE_DIMS_NO_FEATURE_NORM | OPE_DIMS_TANH | OPE_DIMS_MASK_TARGET_MAX))) return OPE_EINVAL;
  if ((d->flags & OPE_DIMS_TANH) && d->obs_dim > 384) return OPE_EINVAL;      // (only trunk_fwd3 carries the activation)
  if (workspace_bytes < ope_agent_forward_mlp_workspace_bytes(d, rows)) return OPE_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  const AgentLayout L = ope_agent_layout_mlp(d->obs_dim, d->act_dim, 0);
  float* a2 = (float*)workspace;
  int rc;
  TrunkFwdArgs tf;
  memset(&tf, 0, sizeof(tf));
  tf.x = obs; tf.R = rows; tf.D = d->obs_dim; tf.theta = theta; tf.L = L; tf.a2_out = a2;
  tf.no_fn = (d->flags & OPE_DIMS_NO_FEATURE_NORM) ? 1 : 0;
  tf.tanh_act = (d->flags & OPE_DIMS_TANH) ? 1 : 0;
  if ((rc = launch_trunk_fwd(tf, false, st))) return rc;
  HeadFwdArgs hf;
  memset(&hf, 0, sizeof(hf));
  hf.R = rows; hf.NB = rows; hf.B = rows; hf.N = 1; hf.T = 1; hf.A = d->act_dim; hf.theta0 = theta; hf.theta1 = theta; hf.L = L;
  hf.h0 = a2; hf.q_out = q_out; hf.no_ln = 1;
  return launch_head_fwd(hf, 1, st);
}
