// Argument blocks of the MADDPG / MATD3 kernels.
#pragma once
#include "ope_mixer.h"
#include "ope_wgrad.h"
#include "ope_workspace.h"

namespace ope {

struct CriticTdArgs {
  int B, K, K4;
  float gamma; int use_huber; float huber_delta; float per_eps;
  const float* q; const float* q_tgt;        // [B][K4]
  const float* rewards;                      // [N][B][1]: agent 0's slice is read
  const float* dones_env;                    // [B][1]
  const float* per_weights;                  // [B] or null
  float* dq;                                 // [B][K4]  d loss_sum / d Q_k
  float* prio_out;                           // [B] or null
  float* loss_part;                          // [tiles][4]
};

struct InGradArgs {
  int R, D;
  const float* dz1;                          // [R][64]
  const float* fc1_w; const float* gamma;    // [64][D], [D]
  const float* x; const float* mu0; const float* rstd0;
  float* dx;                                 // [R][D]
};

}  // namespace ope
