// Problem / segment tables for the batched weight-gradient kernel and the gradient finalize step.
#pragma once
#include "ope_common.h"

namespace ope {

struct WgProb {
  const float* A; int lda; int M;      // A[k][m], m < M
  const float* B; int ldb; int N;      // B[k][n], n < N
  int K;                               // rows reduced over
  int b_shift;                         // B row used for k is (k - b_shift); zero contribution if negative
  const float* ln_mu; const float* ln_rstd;  // if set: B element -> (B - mu[row]) * rstd[row]
  int out_off; int ldc;                // offset (inside one split slab) / leading dim of C
  int s_off;                           // slab offset of colsum(A)[M], or -1
  int nsplit;                          // K-splits of this problem
  int64_t raw_base; int64_t raw_stride;  // split q writes to raw[raw_base + q*raw_stride + ...]
  int mt, nt, kchunk, wave_begin;      // filled by wg_finish
  int ref_row1;                        // > 0: B is the store's obs ring and reduction row k is batch row ref_row1 - 1 + k (WgTable.ref); 0: plain
  int ln_on;                           // 1: ln_mu / ln_rstd are a real LayerNorm (0: the zeros / ones vectors of a plain problem; wgrad2 skips the loads)
  int rs_base;                         // wgrad2: offset of this problem's region in the reduced vector `rsum` (out_off / s_off are relative to it)
  const float* A2; int lda2; int a2_from;  // wgrad2: 64-column panels >= a2_from of A come from A2 (column 64 * (panel - a2_from)); null: one matrix
  // wgrad2 on the packed rows of a live plan (LivePlan, ope_common.h; the MAP instantiation runs when any problem of the table sets K_dev):
  const int* K_dev;                    // rows reduced over, read on the device (<= K; the units' row chunks are re-derived from it in the kernel)
  const int* b_map; int map_on;        // map_on: B's row for reduction row k is b_map[k], no contribution where that is negative (b_shift must
                                       // be 0); LayerNorm statistics stay indexed by k. map_on = 0: b_map still points at >= K readable ints
  // wgrad2 with the finalize step folded in (launch_wgrad2_fin; filled by w2_attach_fin from the step's FinTable): where this problem's results
  // go in the FLAT GRADIENT (= the parameter vector's layout), and whether its Linear is fed by a LayerNorm
  int fin_kind;                        // 0: C is the gradient as it is; 1: dW = C gamma + s (x) beta, dgamma[n] = sum_m W[m][n] C[m][n], dbeta[n] = sum_m s[m] W[m][n]
  int g_off;                           // flat-gradient (and parameter) offset of W [M][ldc], or -1: C is not a parameter's gradient
  int gs_off;                          // ... of the vector colsum(A) is the gradient of (the Linear's bias), or -1
  int gam_off, bet_off;                // fin_kind 1: parameter offsets of the LayerNorm's weight / bias = where dgamma / dbeta go
};
constexpr int kMaxWgProbs = 16;
struct WgTable {
  WgProb p[kMaxWgProbs];
  int n, total_waves, wg_reduce;
  int wbegin[kMaxWgProbs];             // p[q].wave_begin again, contiguous (INT_MAX beyond n): one burst of scalar loads finds a wave's problem
  ObsRef ref;                          // where the problems with ref_row1 > 0 find their B rows (observations left in the store)
};
int wg_finish(WgTable* tb);
int launch_wgrad(const WgTable& tb, float* raw, hipStream_t st);
int wg_slabs(const WgTable& tb, int nsplit);
// ---- register-blocked form (ope_wgrad2.hip): units of up to four tiles per wave, one slab per workgroup ----
constexpr int kW2Slab = 4 * 4096 + 4 * 64 + 8 * 64;      // floats a workgroup writes: four 64 x 64 tiles + four 64-float column sums + (fin) the column partials of dgamma / dbeta, four panels each
constexpr int kW2CpOff = 4 * 4096 + 4 * 64;
constexpr int kMaxW2Units = 40;
struct W2Unit {
  int prob;                            // index into W2Table.p
  short mp0, np0, pa, pb;              // first m / n panel (64 columns each) and panels per side, pa * pb <= 4
  int wg_begin, nwg, kchunk;           // workgroups [wg_begin, wg_begin + nwg); wave c of the unit reduces rows [c * kchunk, (c + 1) * kchunk)
  int red_rows;                        // 64-float rows the slab-sum launch adds for this unit (tile rows + column-sum vectors)
  int fin_rows, fin_blk;               // launch_wgrad2_fin: rows incl. the dgamma / dbeta partial vectors; index of the unit's first workgroup among the launch's live ones
};
struct W2Table {
  WgProb p[kMaxWgProbs];
  W2Unit u[kMaxW2Units];
  int ubegin[kMaxW2Units];             // u[q].wg_begin again, contiguous (INT_MAX beyond nu): what a workgroup searches
  int np, nu, total_wg, red_blocks;
  int vec;                             // 4 | 2: widest load every operand row of the table allows (picks the kernel and its ring depth)
  int fin;                             // 1: every problem carries its flat-gradient placement (w2_attach_fin succeeded): launch_wgrad2_fin may run
  int fin_blocks_x, fin_live_blocks;   // grid.x of the fin launch / its workgroups that write something (= sum-of-squares partials it leaves)
};
struct FinTable;
// zero ranges of the flat gradient nobody's result lands in (grad-less tensors, padding), and where the loss tail goes
constexpr int kMaxFinZero = 24;
struct FinMisc { int zbegin[kMaxFinZero], zlen[kMaxFinZero], zcum[kMaxFinZero + 1]; int nz; int tail_off; int n_loss_tiles; const float* loss_part; int n_gsq_total; };
// Fills the problems' placements and `misc` from the step's segment table; false (fin = 0) if some segment has no producer among the problems or a
// LayerNorm-fed Linear is split over several row blocks: the caller then runs launch_wgrad2 + launch_finalize as before.
bool w2_attach_fin(W2Table* w, const FinTable& ft, FinMisc* misc);
int w2_fin_blocks(const W2Table& w, const FinMisc& misc);
int launch_wgrad2_fin(const W2Table& w, float* raw, const float* theta, float* grad, float* gsq_part, const FinMisc& misc, hipStream_t st);
bool w2_ok(const WgTable& tb);
int w2_max_workgroups();
int w2_build(const WgTable& tb, W2Table* out);
int launch_wgrad2(const W2Table& w, float* raw, float* rsum, hipStream_t st);
struct SplitRed { const float* raw0; int64_t n0; int ns0; const float* raw1; int64_t n1; int ns1; float* rsum; };
int launch_split_reduce(const SplitRed& a, hipStream_t st);
constexpr int kMaxTransp = 6;
struct Transp4 { const float* src[kMaxTransp]; float* dst[kMaxTransp]; int rows[kMaxTransp], cols[kMaxTransp], begin[kMaxTransp]; int n, total; };
int launch_transpose4(const Transp4& a, hipStream_t st);
// dst[c][r] = src[r][c] for up to 4 small matrices; `i` = flat element index over all of them. Shared by the stand-alone
// kernel and by kernels that carry the transposes as extra workgroups (a weight transpose costs nothing beside a launch that
// leaves CU slots free, but ~5 us + a kernel boundary on its own in a serial stream).
__device__ __forceinline__ void transpose4_element(const Transp4& a, int i) {
  if (i >= a.total) return;
  int m = 0;
#pragma unroll
  for (int q = 1; q < kMaxTransp; ++q)
    if (q < a.n && i >= a.begin[q]) m = q;
  const int j = i - a.begin[m];
  const int rows = a.rows[m], cols = a.cols[m];
  const int c = j / rows, r = j - c * rows;
  a.dst[m][j] = a.src[m][(int64_t)r * cols + c];
}

enum FinKind { FIN_ZERO = 0, FIN_COPY = 1, FIN_LNLIN_W = 2, FIN_LNLIN_G = 3, FIN_LNLIN_B = 4, FIN_TAIL = 5, FIN_SKIP = 6 };
struct FinSeg {
  int begin;   // first flat-gradient index of this segment (segments are sorted, padded to x4)
  int size;    // valid elements (rest of the slot up to the next begin is zero)
  int kind;
  int src;     // rsum offset of P / of the plain gradient
  int src_s;   // rsum offset of the column-sum vector s
  int M, K;    // shape of the Linear weight [M][K] the segment belongs to
  int w;       // theta offset of that weight
  int gamma, beta;  // theta offsets of the LayerNorm feeding it
};
constexpr int kMaxFinSegs = 48;
struct FinTable {
  FinSeg seg[kMaxFinSegs];
  int begin[kMaxFinSegs];   // seg[q].begin again, INT_MAX beyond n: filled by launch_finalize (one scalar load burst finds a segment)
  int n;
  int64_t total;  // length of the flat gradient including the tail
  // column-reduction workgroup b -> segment (low 6 bits) and 16-column block of it: filled by launch_finalize when they fit (ncb > 0); a
  // workgroup otherwise walks the segment table itself (a serial chain of scalar loads: 3 us at 40 segments)
  int ncb;
  unsigned short cb[128];
};
constexpr int kMaxColBlocks = 128;
int launch_fill(float* p, int64_t n, float v, hipStream_t st);
// workgroups of a finalize launch = number of floats `gsq_part` must hold
int finalize_blocks(const FinTable& ft);
int launch_finalize(const FinTable& ft, const float* rsum, const float* theta, const float* loss_part, int n_loss_tiles,
                    float* grad, hipStream_t st, float* gsq_part = nullptr, int n_gsq_total = 0);

}  // namespace ope
