// Live-row plan of a recurrent Q-learning step (LivePlan, ope_common.h): which rows of the padded batch carry a loss term at all.
#pragma once
#include "ope_common.h"

namespace ope {

struct LiveArgs {
  int T, N, B;
  const float* dones_env;      // [T][B][1]
  int* plan;                   // live_plan_ints() ints
  float* err_abs;              // [T*B] zero-filled here (the chain kernel writes the live entries only; td_stats reads all of them)
  float* loss_part; int n_loss_part;   // zero-filled here (tiles past the live ones never run)
};
// region size (ints) and the views into it
int64_t live_plan_ints(int T, int N, int B);
LivePlan live_plan_view(const int* base, int T, int N, int B);
bool live_plan_shape_ok(int T, int N, int B);
int launch_live_plan(const LiveArgs& a, hipStream_t st);
// hdr[8 .. 15] as four int64: sums over the steps so far of RL, R1L, TBL and the number of steps (what bench.py reports as executed rows)
constexpr int kLiveAccOff = 8;

}  // namespace ope
