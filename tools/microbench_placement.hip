// Where does the dispatcher put the workgroups of a launch shaped like the scans' (512 workgroups of 384 threads, ~8 KB of LDS, 256 CUs)?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pl tools/microbench_placement.hip && /tmp/pl
// Every workgroup records (XCC_ID, HW_ID) and its start time, then spins ~30 us so that the whole grid is resident at once. The host
// prints, for each workgroup b, which other workgroups share its CU -- the scans want to know whether b and b + 256 do (then pairing a long
// episode with a short one is a permutation of the block index) or b and b + 8, or nothing regular.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>

struct Rec { unsigned xcc, hwid; long long t0; };

__global__ void __launch_bounds__(384) probe(Rec* r, long long spin) {
  __shared__ float pad[2048];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    r[blockIdx.x].xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // HW_REG_XCC_ID
    r[blockIdx.x].hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
    r[blockIdx.x].t0 = t0;
  }
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (pad[(threadIdx.x + 1) % 384] < 0) r[blockIdx.x].t0 = 0;
}

int main() {
  const int nb = 512;
  Rec* d;
  hipMalloc(&d, nb * sizeof(Rec));
  std::vector<Rec> h(nb);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(nb), dim3(384), 0, 0, d, 3000LL);      // 100 MHz clock: 30 us
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, nb * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
      const unsigned x = h[b].xcc & 0xF, se = (h[b].hwid >> 13) & 7, sh = (h[b].hwid >> 12) & 1, c = (h[b].hwid >> 8) & 0xF;
      cu[(x << 12) | (se << 8) | (sh << 4) | c].push_back(b);
    }
    int hist[8] = {0}, d256 = 0, d8 = 0, dother = 0, xcd_rr = 0;
    for (int b = 0; b < nb; ++b) xcd_rr += ((h[b].xcc & 0xF) == (unsigned)(b % 8));
    for (auto& kv : cu) {
      hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
      if (kv.second.size() == 2) {
        const int df = kv.second[1] - kv.second[0];
        if (df == 256) ++d256; else if (df == 8) ++d8; else ++dother;
      }
    }
    printf("rep %d: %zu distinct CUs; workgroups per CU histogram 1:%d 2:%d 3:%d 4:%d; pairs with block distance 256: %d, 8: %d, other: %d; block %% 8 == XCC for %d of %d\n",
           rep, cu.size(), hist[1], hist[2], hist[3], hist[4], d256, d8, dother, xcd_rr, nb);
    if (rep == 2) {
      int shown = 0;
      for (auto& kv : cu) {
        if (shown++ >= 24) break;
        printf("  cu %05x:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
      }
    }
  }
  return 0;
}
