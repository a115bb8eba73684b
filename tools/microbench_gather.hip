// Design-space micro-benchmark for the replay gather on gfx950: ONE long-row field (obs: [episode][TT][N][D] ->
// [TT][N][B][D]) moved by several kernel designs, next to a plain contiguous copy of the same bytes under the same
// launch geometry. Standalone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbg tools/microbench_gather.hip && /tmp/mbg [D=252] [B=32] [TT=151] [N=8] [episodes=1024]
// Every variant is launched 64 times back to back over 16 different random index sets (624 MB of distinct source episodes at
// the 3s5z size: more than the 256 MiB Infinity Cache), timed with HIP events; per-launch time = total / 64 (includes the
// ~1.5 us kernel boundary, identical for all variants). Output is checked against variant 0 once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
  const float* src; float* dst; const int64_t* idx;
  int TT, N, D, B;        // rows: (t, a, b), row length D floats
  int rows;               // TT*N*B
  int rpb;                // rows per block
};

template <int VEC> struct V { typedef float t __attribute__((ext_vector_type(VEC))); };

__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// ---- K0: round-1 mapping. block = rpb consecutive dst rows; wave takes 8 rows per pass, lane = 16-byte piece ----------
template <int VEC, int UN, bool NTS>
__global__ void __launch_bounds__(256) k_rows(P p) {
  typedef typename V<VEC>::t vt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * p.rpb, nr = min(p.rpb, p.rows - r0), pieces = p.D / VEC;
  for (int s0 = wave * UN; s0 < nr; s0 += 4 * UN) {
    const float* sp[UN]; float* dp[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = r0 + min(s0 + u, nr - 1);
      const int ta = r / p.B, b = r - ta * p.B;
      sp[u] = p.src + ((int64_t)p.idx[b] * p.TT * p.N + ta) * p.D;
      dp[u] = p.dst + (int64_t)r * p.D;
    }
    for (int pc = lane; pc < pieces; pc += 64) {
      vt v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const vt*>(sp[u] + pc * VEC);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (s0 + u < nr) {
          if (NTS) __builtin_nontemporal_store(v[u], reinterpret_cast<vt*>(dp[u] + pc * VEC));
          else *reinterpret_cast<vt*>(dp[u] + pc * VEC) = v[u];
        }
    }
  }
}

// ---- K1: same, but the B episode bases sit in LDS (one index load per block instead of one dependent load per row) ----
template <int VEC, int UN>
__global__ void __launch_bounds__(256) k_rows_lds(P p) {
  typedef typename V<VEC>::t vt;
  __shared__ int64_t base[512];
  for (int b = threadIdx.x; b < p.B; b += 256) base[b] = p.idx[b] * (int64_t)p.TT * p.N;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * p.rpb, nr = min(p.rpb, p.rows - r0), pieces = p.D / VEC;
  const float invB = 1.0f / (float)p.B;
  for (int s0 = wave * UN; s0 < nr; s0 += 4 * UN) {
    const float* sp[UN]; float* dp[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = r0 + min(s0 + u, nr - 1);
      const int ta = fdiv(r, invB), b = r - ta * p.B;
      sp[u] = p.src + (base[b] + ta) * p.D;
      dp[u] = p.dst + (int64_t)r * p.D;
    }
    for (int pc = lane; pc < pieces; pc += 64) {
      vt v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const vt*>(sp[u] + pc * VEC);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (s0 + u < nr) *reinterpret_cast<vt*>(dp[u] + pc * VEC) = v[u];
    }
  }
}

// ---- K2: flat destination: the block's dst range is one contiguous array of pieces; thread -> piece q, all 64 lanes busy ----
template <int VEC, int UN>
__global__ void __launch_bounds__(256) k_flat(P p) {
  typedef typename V<VEC>::t vt;
  __shared__ int64_t base[512];
  for (int b = threadIdx.x; b < p.B; b += 256) base[b] = p.idx[b] * (int64_t)p.TT * p.N;
  __syncthreads();
  const int r0 = blockIdx.x * p.rpb, nr = min(p.rpb, p.rows - r0), pieces = p.D / VEC;
  const int total = nr * pieces;
  const float invP = 1.0f / (float)pieces, invB = 1.0f / (float)p.B;
  float* dbase = p.dst + (int64_t)r0 * p.D;
  for (int q0 = threadIdx.x; q0 < total; q0 += 256 * UN) {
    vt v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = min(q0 + 256 * u, total - 1);
      const int rl = fdiv(q, invP), pc = q - rl * pieces;
      const int r = r0 + rl, ta = fdiv(r, invB), b = r - ta * p.B;
      v[u] = *reinterpret_cast<const vt*>(p.src + (base[b] + ta) * p.D + pc * VEC);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (q0 + 256 * u < total) *reinterpret_cast<vt*>(dbase + (int64_t)(q0 + 256 * u) * VEC) = v[u];
  }
}

// ---- K3: persistent grid-stride over row groups with the next group's loads issued before this group's stores ----------
template <int VEC>
__global__ void __launch_bounds__(256) k_persist(P p) {
  typedef typename V<VEC>::t vt;
  constexpr int UN = 8;
  __shared__ int64_t base[512];
  for (int b = threadIdx.x; b < p.B; b += 256) base[b] = p.idx[b] * (int64_t)p.TT * p.N;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pieces = p.D / VEC;
  const float invB = 1.0f / (float)p.B;
  // unit of work: 8 rows handled by one wave; units numbered over the whole field; wave w of block g takes units (g*4 + w) + k*grid*4
  const int n_units = (p.rows + UN - 1) / UN;
  const int stride = gridDim.x * 4;
  int unit = blockIdx.x * 4 + wave;
  if (pieces > 64) return;   // single-pass rows only in this variant
  vt cur[UN], nxt[UN];
  auto load = [&](int un, vt (&v)[UN]) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = min(un * UN + u, p.rows - 1);
      const int ta = fdiv(r, invB), b = r - ta * p.B;
      if (lane < pieces) v[u] = *reinterpret_cast<const vt*>(p.src + (base[b] + ta) * p.D + lane * VEC);
    }
  };
  if (unit < n_units) load(unit, cur);
  while (unit < n_units) {
    const int nu = unit + stride;
    if (nu < n_units) load(nu, nxt);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = unit * UN + u;
      if (r < p.rows && lane < pieces) *reinterpret_cast<vt*>(p.dst + (int64_t)r * p.D + lane * VEC) = cur[u];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) cur[u] = nxt[u];
    unit = nu;
  }
}

// ---- K4: episode-contiguous reads: block = (b, t-range): reads KT*N*D contiguous floats of ONE episode, writes KT*N rows ----
template <int VEC, int UN>
__global__ void __launch_bounds__(256) k_epi(P p, int KT) {
  typedef typename V<VEC>::t vt;
  const int tblocks = (p.TT + KT - 1) / KT;
  const int b = blockIdx.x / tblocks, tb = blockIdx.x - b * tblocks;
  const int t0 = tb * KT, nt = min(KT, p.TT - t0);
  const int pieces = p.D / VEC, total = nt * p.N * pieces;
  const float* sbase = p.src + ((int64_t)p.idx[b] * p.TT + t0) * p.N * p.D;
  const float invP = 1.0f / (float)pieces;
  for (int q0 = threadIdx.x; q0 < total; q0 += 256 * UN) {
    vt v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const vt*>(sbase + (int64_t)min(q0 + 256 * u, total - 1) * VEC);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = q0 + 256 * u;
      if (q < total) {
        const int row = fdiv(q, invP), pc = q - row * pieces;      // row = (t - t0)*N + a
        *reinterpret_cast<vt*>(p.dst + (((int64_t)t0 * p.N + row) * p.B + b) * p.D + pc * VEC) = v[u];
      }
    }
  }
}

// ---- K4x: K4 with (t-block major, b minor) block order + XCD run remap: the B blocks writing the same [t][a][0..B) dst range
//           (adjacent 1008-byte rows sharing their boundary lines) run on ONE XCD, i.e. meet in one L2 --------------------------
template <int VEC, int UN>
__global__ void __launch_bounds__(256) k_epi_x(P p, int KT, int G) {
  typedef typename V<VEC>::t vt;
  const int tblocks = (p.TT + KT - 1) / KT;
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if (G > 1 && bid < (nb / (8 * G)) * (8 * G)) {
    const int super = bid / (8 * G), w = bid - super * (8 * G);
    bid = (super * 8 + (w & 7)) * G + (w >> 3);
  }
  const int tb = bid / p.B, b = bid - tb * p.B;
  if (tb >= tblocks) return;
  const int t0 = tb * KT, nt = min(KT, p.TT - t0);
  const int pieces = p.D / VEC, total = nt * p.N * pieces;
  const float* sbase = p.src + ((int64_t)p.idx[b] * p.TT + t0) * p.N * p.D;
  const float invP = 1.0f / (float)pieces;
  for (int q0 = threadIdx.x; q0 < total; q0 += 256 * UN) {
    vt v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const vt*>(sbase + (int64_t)min(q0 + 256 * u, total - 1) * VEC);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = q0 + 256 * u;
      if (q < total) {
        const int row = fdiv(q, invP), pc = q - row * pieces;
        *reinterpret_cast<vt*>(p.dst + (((int64_t)t0 * p.N + row) * p.B + b) * p.D + pc * VEC) = v[u];
      }
    }
  }
}

// ---- K5: block = (t, KB consecutive b): reads KB contiguous N*D runs, writes N segments of KB*D contiguous floats -------
template <int VEC, int UN>
__global__ void __launch_bounds__(256) k_tb(P p, int KB) {
  typedef typename V<VEC>::t vt;
  const int bblocks = (p.B + KB - 1) / KB;
  const int t = blockIdx.x / bblocks, bb = blockIdx.x - t * bblocks;
  const int b0 = bb * KB, nb = min(KB, p.B - b0);
  const int pieces = p.D / VEC, ppe = p.N * pieces, total = nb * ppe;     // pieces per episode-step
  const float invE = 1.0f / (float)ppe, invP = 1.0f / (float)pieces;
  for (int q0 = threadIdx.x; q0 < total; q0 += 256 * UN) {
    vt v[UN];
    int db[UN], dq[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = min(q0 + 256 * u, total - 1);
      const int bl = fdiv(q, invE), x = q - bl * ppe;        // x = piece inside the N*D run: a*pieces + pc
      db[u] = b0 + bl; dq[u] = x;
      v[u] = *reinterpret_cast<const vt*>(p.src + ((int64_t)p.idx[db[u]] * p.TT + t) * p.N * p.D + (int64_t)x * VEC);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (q0 + 256 * u < total) {
        const int a = fdiv(dq[u], invP), pc = dq[u] - a * pieces;
        *reinterpret_cast<vt*>(p.dst + (((int64_t)t * p.N + a) * p.B + db[u]) * p.D + pc * VEC) = v[u];
      }
  }
}

// ---- K4u: K4 for rows whose length is 2 mod 4 floats: 16-byte loads from the (16-byte aligned) contiguous episode run,
//           each stored as two 8-byte halves (a 16-byte piece may straddle two rows; an 8-byte half cannot) -----------------
template <int UN>
__global__ void __launch_bounds__(256) k_epi_u(P p, int KT) {
  const int tblocks = (p.TT + KT - 1) / KT;
  const int b = blockIdx.x / tblocks, tb = blockIdx.x - b * tblocks;
  const int t0 = tb * KT, nt = min(KT, p.TT - t0);
  const int total4 = nt * p.N * p.D / 4;       // requires (N*D) % 4 == 0 and D % 2 == 0
  const float* sbase = p.src + ((int64_t)p.idx[b] * p.TT + t0) * p.N * p.D;
  const float invD = 1.0f / (float)p.D;
  for (int q0 = threadIdx.x; q0 < total4; q0 += 256 * UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const f32x4*>(sbase + (int64_t)min(q0 + 256 * u, total4 - 1) * 4);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = q0 + 256 * u;
      if (q < total4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f = 4 * q + 2 * h;                    // float index inside the run
          const int row = fdiv(f, invD), x = f - row * p.D;
          f32x2 w = {v[u][2 * h], v[u][2 * h + 1]};
          *reinterpret_cast<f32x2*>(p.dst + (((int64_t)t0 * p.N + row) * p.B + b) * p.D + x) = w;
        }
      }
    }
  }
}

// ---- KC: plain contiguous copy of the same number of bytes, 8 x 16 B per thread -----------------------------------------
__global__ void __launch_bounds__(256) k_copy(const f32x4* __restrict__ s, f32x4* __restrict__ d, int64_t n4) {
  const int64_t q0 = (int64_t)blockIdx.x * 2048 + threadIdx.x;
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = s[min(q0 + 256 * u, n4 - 1)];
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (q0 + 256 * u < n4) d[q0 + 256 * u] = v[u];
}

int main(int argc, char** argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 252, B = argc > 2 ? atoi(argv[2]) : 32, TT = argc > 3 ? atoi(argv[3]) : 151,
            N = argc > 4 ? atoi(argv[4]) : 8, EP = argc > 5 ? atoi(argv[5]) : 1024;
  const int64_t ep_floats = (int64_t)TT * N * D, out_floats = ep_floats * B;
  const int VEC = D % 4 == 0 ? 4 : (D % 2 == 0 ? 2 : 1);
  printf("obs-like field: D=%d (row %d B, vec %d) B=%d TT=%d N=%d, store %d episodes = %.2f GB, %.2f MB moved per launch (read+write)\n", D, 4 * D,
         VEC, B, TT, N, EP, EP * ep_floats * 4 / 1e9, 2.0 * out_floats * 4 / 1e6);
  float *src, *dst, *dst_ref;
  CK(hipMalloc(&src, EP * ep_floats * 4));
  CK(hipMalloc(&dst, out_floats * 4));
  CK(hipMalloc(&dst_ref, out_floats * 4));
  {   // fill the store with a cheap pattern on the host (once)
    std::vector<float> h(ep_floats);
    for (int e = 0; e < EP; ++e) {
      for (int64_t i = 0; i < ep_floats; ++i) h[i] = (float)((e * 131 + i * 7) % 100003);
      CK(hipMemcpy(src + e * ep_floats, h.data(), ep_floats * 4, hipMemcpyHostToDevice));
    }
  }
  const int NSETS = 16, LAUNCHES = 64;
  int64_t* idx;
  CK(hipMalloc(&idx, NSETS * B * 8));
  {
    std::vector<int64_t> h(NSETS * B);
    srand(1);
    for (auto& x : h) x = rand() % EP;
    CK(hipMemcpy(idx, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  P p{src, dst, idx, TT, N, D, B, TT * N * B, B};
  auto run = [&](const char* name, auto launch, bool check) {
    p.idx = idx; p.dst = dst;
    CK(hipMemset(dst, 0xff, out_floats * 4));
    launch(p);
    CK(hipDeviceSynchronize());
    if (check) {
      std::vector<float> a(out_floats), b(out_floats);
      CK(hipMemcpy(a.data(), dst, out_floats * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), dst_ref, out_floats * 4, hipMemcpyDeviceToHost));
      if (memcmp(a.data(), b.data(), out_floats * 4) != 0) { printf("  %-44s WRONG OUTPUT\n", name); return; }
    }
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int l = 0; l < LAUNCHES; ++l) { p.idx = idx + (l % NSETS) * B; launch(p); }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= LAUNCHES; best = ms < best ? ms : best; sum += ms;
    }
    printf("  %-44s mean %7.2f us  best %7.2f us  -> %6.0f GB/s (best)\n", name, 1e3 * sum / 5, 1e3 * best, 2.0 * out_floats * 4 / best / 1e6);
  };
  // reference output from K0
  {
    p.rpb = B; p.dst = dst_ref; p.idx = idx;
    if (VEC == 4) hipLaunchKernelGGL((k_rows<4, 8, false>), dim3((p.rows + p.rpb - 1) / p.rpb), dim3(256), 0, 0, p);
    else if (VEC == 2) hipLaunchKernelGGL((k_rows<2, 8, false>), dim3((p.rows + p.rpb - 1) / p.rpb), dim3(256), 0, 0, p);
    else hipLaunchKernelGGL((k_rows<1, 8, false>), dim3((p.rows + p.rpb - 1) / p.rpb), dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
  }
#define GRID(pp) dim3(((pp).rows + (pp).rpb - 1) / (pp).rpb)
#define VDISPATCH(KERN, ...)                                                                  \
  if (VEC == 4) hipLaunchKernelGGL((KERN<4, __VA_ARGS__>), GRID(q), dim3(256), 0, 0, q);      \
  else if (VEC == 2) hipLaunchKernelGGL((KERN<2, __VA_ARGS__>), GRID(q), dim3(256), 0, 0, q); \
  else hipLaunchKernelGGL((KERN<1, __VA_ARGS__>), GRID(q), dim3(256), 0, 0, q);
  for (int rpbm : {1, 2, 4}) {
    char nm[96];
    snprintf(nm, 96, "K0 rows8 (r01), %d rows/block", B * rpbm);
    run(nm, [&](P q) { q.rpb = B * rpbm; VDISPATCH(k_rows, 8, false) }, true);
  }
  run("K0 rows8 half group/block", [&](P q) { q.rpb = B / 2; VDISPATCH(k_rows, 8, false) }, true);
  run("K0 rows4", [&](P q) { q.rpb = B; VDISPATCH(k_rows, 4, false) }, true);
  run("K0 rows8 nt stores", [&](P q) { q.rpb = B; VDISPATCH(k_rows, 8, true) }, true);
  run("K1 rows8 + episode bases in LDS", [&](P q) { q.rpb = B; VDISPATCH(k_rows_lds, 8) }, true);
  run("K1 rows4 + episode bases in LDS", [&](P q) { q.rpb = B; VDISPATCH(k_rows_lds, 4) }, true);
  run("K1 rows8 + LDS bases, 2 groups/block", [&](P q) { q.rpb = 2 * B; VDISPATCH(k_rows_lds, 8) }, true);
  run("K2 flat pieces x8", [&](P q) { q.rpb = B; VDISPATCH(k_flat, 8) }, true);
  run("K2 flat pieces x4", [&](P q) { q.rpb = B; VDISPATCH(k_flat, 4) }, true);
  run("K2 flat pieces x4, half group/block", [&](P q) { q.rpb = B / 2; VDISPATCH(k_flat, 4) }, true);
  run("K2 flat pieces x8, 2 groups/block", [&](P q) { q.rpb = 2 * B; VDISPATCH(k_flat, 8) }, true);
  for (int g : {256, 512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, 96, "K3 persistent prefetch, grid %d", g);
    run(nm, [&](P q) {
      if (VEC == 4) hipLaunchKernelGGL((k_persist<4>), dim3(g), dim3(256), 0, 0, q);
      else if (VEC == 2) hipLaunchKernelGGL((k_persist<2>), dim3(g), dim3(256), 0, 0, q);
      else hipLaunchKernelGGL((k_persist<1>), dim3(g), dim3(256), 0, 0, q);
    }, D / VEC <= 64);
  }
  for (int kt : {1, 2, 4, 8}) {
    char nm[96];
    snprintf(nm, 96, "K4 episode-contiguous reads, %d steps/block", kt);
    run(nm, [&](P q) {
      const dim3 g(B * ((TT + kt - 1) / kt));
      if (VEC == 4) hipLaunchKernelGGL((k_epi<4, 8>), g, dim3(256), 0, 0, q, kt);
      else if (VEC == 2) hipLaunchKernelGGL((k_epi<2, 8>), g, dim3(256), 0, 0, q, kt);
      else hipLaunchKernelGGL((k_epi<1, 8>), g, dim3(256), 0, 0, q, kt);
    }, true);
  }
  for (int kt : {1, 2}) {
    char nm[96];
    snprintf(nm, 96, "K4x (t major, XCD run = B), %d steps/block", kt);
    run(nm, [&](P q) {
      const dim3 g(B * ((TT + kt - 1) / kt));
      if (VEC == 4) hipLaunchKernelGGL((k_epi_x<4, 8>), g, dim3(256), 0, 0, q, kt, B);
      else if (VEC == 2) hipLaunchKernelGGL((k_epi_x<2, 8>), g, dim3(256), 0, 0, q, kt, B);
      else hipLaunchKernelGGL((k_epi_x<1, 8>), g, dim3(256), 0, 0, q, kt, B);
    }, true);
    snprintf(nm, 96, "K4x (t major, no XCD remap), %d steps/block", kt);
    run(nm, [&](P q) {
      const dim3 g(B * ((TT + kt - 1) / kt));
      if (VEC == 4) hipLaunchKernelGGL((k_epi_x<4, 8>), g, dim3(256), 0, 0, q, kt, 1);
      else if (VEC == 2) hipLaunchKernelGGL((k_epi_x<2, 8>), g, dim3(256), 0, 0, q, kt, 1);
      else hipLaunchKernelGGL((k_epi_x<1, 8>), g, dim3(256), 0, 0, q, kt, 1);
    }, true);
  }
  for (int kb : {2, 4, 8}) {
    char nm[96];
    snprintf(nm, 96, "K5 block = (t, %d episodes)", kb);
    run(nm, [&](P q) {
      const dim3 g(TT * ((B + kb - 1) / kb));
      if (VEC == 4) hipLaunchKernelGGL((k_tb<4, 8>), g, dim3(256), 0, 0, q, kb);
      else if (VEC == 2) hipLaunchKernelGGL((k_tb<2, 8>), g, dim3(256), 0, 0, q, kb);
      else hipLaunchKernelGGL((k_tb<1, 8>), g, dim3(256), 0, 0, q, kb);
    }, true);
  }
  if (D % 2 == 0 && (N * D) % 4 == 0 && D % 4 != 0)
    for (int kt : {1, 2}) {
      char nm[96];
      snprintf(nm, 96, "K4u 16B loads + 2x8B stores, %d steps/block", kt);
      run(nm, [&](P q) { hipLaunchKernelGGL((k_epi_u<8>), dim3(B * ((TT + kt - 1) / kt)), dim3(256), 0, 0, q, kt); }, true);
    }
  {   // ceiling: contiguous copy of the same bytes (from a region of the store, rotating so that it is not cache resident)
    const int64_t n4 = out_floats / 4;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int l = 0; l < LAUNCHES; ++l) {
        const float* s = src + ((int64_t)(l % NSETS) * B % (EP - B)) * ep_floats;
        hipLaunchKernelGGL(k_copy, dim3((n4 + 2047) / 2048), dim3(256), 0, 0, (const f32x4*)s, (f32x4*)dst, n4);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= LAUNCHES; best = ms < best ? ms : best;
    }
    printf("  %-44s                  best %7.2f us  -> %6.0f GB/s (best)\n", "KC plain contiguous copy, 8x16B/thread", 1e3 * best, 2.0 * out_floats * 4 / best / 1e6);
  }
  return 0;
}
