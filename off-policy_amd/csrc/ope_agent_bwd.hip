// Agent q-network backward kernels. The reference gets these from torch autograd over AgentQFunction
// (loss.backward(), qmix.py:191); here every op's adjoint is written out:
//
//   head_bwd  : d agent_q -> dq at the chosen action -> LN backward -> dh_out[t]             (ope_head.hip)
//   gru_bwd   : BPTT over t = T-1..0                                                           (ope_gru4.hip / ope_gru1.hip)
//   trunk_bwd : dgi -> W_ih^T -> LN2 bwd -> ReLU -> fc2^T -> LN1 bwd -> ReLU -> dz1           (ope_trunk_bwd3.hip)
// This file holds their launch entry points and the weight transposes the backward chains read.
//
// Weight gradients are K-reductions over all rows and are done by ope_wgrad.hip from the per-row adjoints
// stored here (dz1, dz2, dgi, dghn, dqoh).
#include <stdlib.h>

#include "ope_agent.h"

namespace ope {

int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  return launch_head_bwd_rows(a, st);   // ope_head.hip
}

// ---------------------------------------------------------------------------------------------------------
// gru_bwd: BPTT over t = T-1 .. t_lo; kernels in ope_gru4.hip / ope_gru1.hip, chosen like launch_gru_fwd.
// ---------------------------------------------------------------------------------------------------------
int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st) {
  if (a.NB < 1 || a.T < 1 || a.t_lo < 0 || a.t_lo >= a.T) return OPE_EINVAL;
  const int want = (a.family == 1 || a.family == 4) ? a.family : g_scan_family;
  const int kind = want ? want : (a.NB <= kGru4MaxRows ? 4 : 1);
  return kind == 4 ? launch_gru_bwd4(a, st) : launch_gru_bwd1(a, st);
}

int launch_trunk_bwd(const TrunkBwdArgs& a, hipStream_t st) {
  if (a.R < 1) return OPE_EINVAL;
  return launch_trunk_bwd3(a, st);   // ope_trunk_bwd3.hip
}

// ---------------------------------------------------------------------------------------------------------
// Transposed copies of the matrices the backward chains read column-wise: dst[c][r] = src[r][c].
// ---------------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int c = i / rows, r = i - c * rows;  // dst index i = c*rows + r  (coalesced writes)
  dst[i] = src[(int64_t)r * cols + c];
}

int launch_transpose(const float* src, int rows, int cols, float* dst, hipStream_t st) {
  OPE_LAUNCH(transpose_kernel, dim3(ope_cdiv((int64_t)rows * cols, 256)), dim3(256), 0, st, src, rows, cols, dst);
  if (hipGetLastError() != hipSuccess) return OPE_ELAUNCH;
  return OPE_OK;
}

int launch_transpose_weights(const float* theta, const AgentLayout& L, float* thetaT, hipStream_t st) {
  int rc = launch_transpose(theta + L.wih, 3 * OPE_H, OPE_H, thetaT, st);
  if (rc) return rc;
  return launch_transpose(theta + L.fc2_w, OPE_H, OPE_H, thetaT + OPE_H * 3 * OPE_H, st);
}

}  // namespace ope
