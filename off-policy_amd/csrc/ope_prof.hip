// In-process per-kernel timing (ope_kernel_profile / ope_kernel_profile_read, ope.h): while enabled, every kernel launch of the library
// (OPE_LAUNCH, ope_common.h) goes through hipExtLaunchKernel with a start / stop event pair attached to the DISPATCH -- the quantity
// rocprofv3's kernel trace reports -- so bench.py can print a per-kernel roofline table measured in the same run, on the box it runs on
// (SURVEY.md section 8(d) "Measurement"; the committed rocprofv3 summaries under profiles/ must agree). No reference counterpart.
// Not for use during HIP-graph capture (event-carrying launches cannot be captured): callers enable it around eager steps only.
#include <cxxabi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ope_common.h"

namespace ope {

bool g_kprof_on = false;

namespace {
constexpr int kMaxRec = 16384;
struct KRec { const void* fn; hipEvent_t e0, e1; double flop, bytes; int rows; };
KRec* g_rec = nullptr;
int g_made = 0, g_n = 0;
double g_next_flop = 0, g_next_bytes = 0;
int g_next_rows = 0;
}  // namespace

// A launch on the live rows of a step runs fewer rows than the padded batch its launcher knows: `kind` says which of the plan's counts
// scales the stated work (ope.h, ope_kernel_profile_read); the reader of the table applies it (the counts live on the device).
void kprof_rows(int kind) {
  if (g_kprof_on) g_next_rows = kind;
}

// The launcher of a kernel states the ALGORITHMIC work of the launch it is about to make (GEMM-shaped FLOP: 2 x MACs, LayerNorm / gates /
// elementwise excluded; bytes for the bandwidth-bound kernels: what must be read + written once): attached to the next
// OPE_LAUNCH while profiling is on, summed per kernel by ope_kernel_profile_read. A kernel's roofline fraction is then work / measured time.
void kprof_work(double flop, double bytes) {
  if (!g_kprof_on) return;
  g_next_flop = flop;
  g_next_bytes = bytes;
}

bool kprof_events(const void* fn, hipEvent_t* e0, hipEvent_t* e1) {
  if (!g_kprof_on || g_n >= g_made) return false;
  g_rec[g_n].fn = fn;
  g_rec[g_n].flop = g_next_flop;
  g_rec[g_n].bytes = g_next_bytes;
  g_rec[g_n].rows = g_next_rows;
  g_next_flop = g_next_bytes = 0;
  g_next_rows = 0;
  *e0 = g_rec[g_n].e0;
  *e1 = g_rec[g_n].e1;
  ++g_n;
  return true;
}

}  // namespace ope

using namespace ope;

extern "C" int ope_kernel_profile(int32_t enable, int32_t max_launches) {
  if (enable) {
    int want = max_launches > 0 ? max_launches : 4096;
    if (want > kMaxRec) want = kMaxRec;
    if (!g_rec) {
      g_rec = (KRec*)calloc(kMaxRec, sizeof(KRec));
      if (!g_rec) return OPE_EHIP;
    }
    for (; g_made < want; ++g_made)
      if (hipEventCreate(&g_rec[g_made].e0) != hipSuccess || hipEventCreate(&g_rec[g_made].e1) != hipSuccess) return OPE_EHIP;
  }
  g_kprof_on = enable != 0;
  g_n = 0;
  return OPE_OK;
}

extern "C" int ope_kernel_profile_read(char* out, int32_t cap) {
  if (!out || cap < 2) return OPE_EINVAL;
  out[0] = 0;
  const int n = g_n;
  g_n = 0;
  if (n == 0) return 0;
  struct Agg { const void* fn; int calls; double total, mn, mx, flop, bytes; int rows; };
  Agg* agg = (Agg*)calloc(n, sizeof(Agg));
  if (!agg) return OPE_EHIP;
  int na = 0;
  if (hipEventSynchronize(g_rec[n - 1].e1) != hipSuccess) { free(agg); return OPE_EHIP; }
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_rec[i].e1) != hipSuccess || hipEventElapsedTime(&ms, g_rec[i].e0, g_rec[i].e1) != hipSuccess) { free(agg); return OPE_EHIP; }
    int k = 0;
    while (k < na && agg[k].fn != g_rec[i].fn) ++k;
    if (k == na) { agg[na].fn = g_rec[i].fn; agg[na].calls = 0; agg[na].total = 0; agg[na].mn = 1e30; agg[na].mx = 0; agg[na].rows = g_rec[i].rows; ++na; }
    agg[k].calls += 1; agg[k].total += ms; agg[k].flop += g_rec[i].flop; agg[k].bytes += g_rec[i].bytes;
    if (ms < agg[k].mn) agg[k].mn = ms;
    if (ms > agg[k].mx) agg[k].mx = ms;
  }
  int len = 0;
  for (int k = 0; k < na; ++k) {     // launch order of first appearance; one line per kernel: name \t calls \t total_ms \t min_ms \t max_ms \t flop \t bytes
    const char* mangled = hipKernelNameRefByPtr(agg[k].fn, nullptr);
    int status = -1;
    char* dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
    const char* name = (status == 0 && dem) ? dem : (mangled ? mangled : "?");
    char line[768];
    int m = snprintf(line, sizeof(line), "%.600s\t%d\t%.6f\t%.6f\t%.6f\t%.0f\t%.0f\t%d\n", name, agg[k].calls, agg[k].total, agg[k].mn, agg[k].mx, agg[k].flop,
                     agg[k].bytes, agg[k].rows);
    if (dem) free(dem);
    if (m < 0 || len + m + 1 >= cap) break;
    memcpy(out + len, line, m);
    len += m;
    out[len] = 0;
  }
  free(agg);
  return na;
}
