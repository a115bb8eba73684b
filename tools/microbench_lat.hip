// Latency / issue-rate micro-benchmarks for the scan kernels' building blocks on gfx950 (one wave unless stated).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_lat.hip && /tmp/mb
// Every test runs `iters` repetitions of a dependent pattern between two clock64() reads (s_memtime) and two
// wall_clock64() reads (s_memrealtime, 100 MHz); the host prints shader cycles and ns per repetition.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Out { long long cyc, wall; float sink; };

#define BEGIN long long c0 = clock64(), w0 = wall_clock64();
#define END(v) long long c1 = clock64(), w1 = wall_clock64(); if (threadIdx.x == 0) { o->cyc = c1 - c0; o->wall = w1 - w0; } if (v == 123.456f) o->sink = v;

__global__ void t_fma_dep(Out* o, int iters, float x) {
  float a = x + threadIdx.x;
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(x));
  }
  END(a)
}

__global__ void t_pkfma_ind(Out* o, int iters, float x) {   // 6 independent accumulators, 96 per repetition
  f32x2 a0 = {x, x}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, w = {x, 1.0f};
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a0) : "v"(w));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a1) : "v"(w));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a2) : "v"(w));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a3) : "v"(w));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a4) : "v"(w));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a5) : "v"(w));
    }
  }
  float v = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a0[1] + a1[1] + a2[1] + a3[1] + a4[1] + a5[1];
  END(v)
}

__global__ void t_pkfma_dep(Out* o, int iters, float x) {   // 1 accumulator: dependent packed FMA latency, 32 per repetition
  f32x2 a0 = {x, x}, w = {x, 1.0f};
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a0) : "v"(w));
  }
  float v = a0[0] + a0[1];
  END(v)
}

__global__ void t_exp_dep(Out* o, int iters, float x) {   // 16 dependent v_exp_f32
  float a = x;
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  }
  END(a)
}

__global__ void t_exp_ind(Out* o, int iters, float x) {   // 16 v_exp_f32 on 4 independent registers
  float a = x, b = x + 1, c = x + 2, d = x + 3;
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(a));
      asm volatile("v_exp_f32 %0, %0" : "+v"(b));
      asm volatile("v_exp_f32 %0, %0" : "+v"(c));
      asm volatile("v_exp_f32 %0, %0" : "+v"(d));
    }
  }
  float v = a + b + c + d;
  END(v)
}

__global__ void t_rcp_dep(Out* o, int iters, float x) {
  float a = x;
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
  }
  END(a)
}

// LDS write -> broadcast ds_read_b128 -> use, dependent: one round trip per repetition
__global__ void t_lds_roundtrip(Out* o, int iters, float x) {
  __shared__ __attribute__((aligned(16))) float hs[64];
  float a = x + threadIdx.x;
  BEGIN
  for (int i = 0; i < iters; ++i) {
    hs[threadIdx.x & 63] = a;
    __builtin_amdgcn_wave_barrier();
    const f32x4 v = *reinterpret_cast<volatile f32x4*>(hs + 4 * (i & 15));
    a = v[0] + v[1] + v[2] + v[3];
    __builtin_amdgcn_wave_barrier();
  }
  END(a)
}

// LDS write -> per-lane ds_read_b32 (rotated lane) -> use
__global__ void t_lds_roundtrip_b32(Out* o, int iters, float x) {
  __shared__ float hs[64];
  float a = x + threadIdx.x;
  BEGIN
  for (int i = 0; i < iters; ++i) {
    hs[threadIdx.x & 63] = a;
    __builtin_amdgcn_wave_barrier();
    a = *reinterpret_cast<volatile float*>(hs + ((threadIdx.x + 1) & 63)) + 1.0f;
    __builtin_amdgcn_wave_barrier();
  }
  END(a)
}

// 16 broadcast ds_read_b128 back to back (throughput), then one wait
__global__ void t_lds_bcast16(Out* o, int iters, float x) {
  __shared__ __attribute__((aligned(16))) float hs[64];
  hs[threadIdx.x & 63] = x;
  __syncthreads();
  f32x4 acc = {0, 0, 0, 0};
  BEGIN
  for (int i = 0; i < iters; ++i) {
    f32x4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(0), "n"(16 * k));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(v[k]));
    acc += v[i & 15];
  }
  float v2 = acc[0] + acc[1] + acc[2] + acc[3];
  END(v2)
}

// same, 48 reads (the BPTT step)
__global__ void t_lds_bcast48(Out* o, int iters, float x) {
  __shared__ __attribute__((aligned(16))) float hs[192];
  for (int i = threadIdx.x; i < 192; i += blockDim.x) hs[i] = x;
  __syncthreads();
  f32x4 acc = {0, 0, 0, 0};
  BEGIN
  for (int i = 0; i < iters; ++i) {
    f32x4 v[16];
#pragma unroll
    for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(0), "n"(16 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(v[k]));
      acc += v[i & 15];
    }
  }
  float v2 = acc[0] + acc[1] + acc[2] + acc[3];
  END(v2)
}

// 96 packed FMAs with 16 broadcast reads interleaved, 4 deep (the forward step's FMA block)
__global__ void t_fwd_block(Out* o, int iters, float x) {
  __shared__ __attribute__((aligned(16))) float hs[64];
  hs[threadIdx.x & 63] = x;
  __syncthreads();
  f32x2 a0 = {x, x}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, w = {x, 1.0f};
  BEGIN
  for (int i = 0; i < iters; ++i) {
    f32x4 hq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hq[k] = *reinterpret_cast<volatile f32x4*>(hs + 4 * k);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const f32x4 hv = hq[v % 4];
      if (v + 4 < 16) hq[v % 4] = *reinterpret_cast<volatile f32x4*>(hs + 4 * (v + 4));
      const f32x2 lo = {hv[0], hv[1]}, hi = {hv[2], hv[3]};
      a0 = __builtin_elementwise_fma(w, lo, a0);
      a1 = __builtin_elementwise_fma(w, lo, a1);
      a2 = __builtin_elementwise_fma(w, lo, a2);
      a3 = __builtin_elementwise_fma(w, hi, a3);
      a4 = __builtin_elementwise_fma(w, hi, a4);
      a5 = __builtin_elementwise_fma(w, hi, a5);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
    }
  }
  float v = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a0[1] + a1[1] + a2[1] + a3[1] + a4[1] + a5[1];
  END(v)
}

// the gate chain of one GRU step (no LDS, no FMAs): latency of sigmoid, sigmoid, tanh, blend
__global__ void t_gates(Out* o, int iters, float x) {
  float h = x, ar = x * 0.5f, az = x * 0.25f, an = x * 0.125f;
  BEGIN
  for (int i = 0; i < iters; ++i) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __expf(-(ar + h)));
    const float z = __builtin_amdgcn_rcpf(1.0f + __expf(-(az + h)));
    const float pre = an + r * h;
    const float e = __expf(-2.0f * fabsf(pre));
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    const float n = copysignf(t, pre);
    h = (1.0f - z) * n + z * h;
    asm volatile("" : "+v"(h));
  }
  END(h)
}

// workgroup barrier cost with W waves (every wave just loops on the barrier)
__global__ void t_barrier(Out* o, int iters, float x) {
  float a = x;
  BEGIN
  for (int i = 0; i < iters; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  END(a)
}

// readlane broadcast: 64 v_readlane_b32 + 32 packed FMAs with SGPR-pair operands
__global__ void t_readlane64(Out* o, int iters, float x) {
  float h = x + threadIdx.x;
  float acc = 0.f;
  BEGIN
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const float s = __builtin_amdgcn_readlane(h, k);
      acc = fmaf(s, x, acc);
    }
    h = acc;
    asm volatile("" : "+v"(h));
  }
  END(h)
}

template <typename K>
void run(const char* name, K kern, int threads, int iters, int per, int blocks = 1) {
  Out* d; Out h;
  hipMalloc(&d, sizeof(Out));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.5f);
    hipDeviceSynchronize();
  }
  hipMemcpy(&h, d, sizeof(Out), hipMemcpyDeviceToHost);
  const double cyc = (double)h.cyc / iters, ns = (double)h.wall * 10.0 / iters;
  printf("%-28s thr=%3d blk=%4d : %9.1f clk/rep  %8.1f ns/rep  (%d ops/rep -> %.2f clk/op, %.2f ns/op; clk/ns=%.3f)\n", name, threads, blocks, cyc, ns, per,
         cyc / per, ns / per, cyc / ns);
  hipFree(d);
}

int main() {
  const int N = 20000;
  run("fma dep x32", t_fma_dep, 64, N, 32);
  run("pk_fma 6 indep x96", t_pkfma_ind, 64, N, 96);
  run("pk_fma dep x32", t_pkfma_dep, 64, N, 32);
  run("exp dep x16", t_exp_dep, 64, N, 16);
  run("exp indep x16", t_exp_ind, 64, N, 16);
  run("rcp dep x16", t_rcp_dep, 64, N, 16);
  run("lds roundtrip b128 bcast", t_lds_roundtrip, 64, N, 1);
  run("lds roundtrip b32", t_lds_roundtrip_b32, 64, N, 1);
  run("lds bcast b128 x16", t_lds_bcast16, 64, N, 16);
  run("lds bcast b128 x48", t_lds_bcast48, 64, N, 48);
  run("fwd FMA block (96+16rd)", t_fwd_block, 64, N, 96);
  run("gates chain", t_gates, 64, N, 1);
  run("barrier 2 waves", t_barrier, 128, N, 1);
  run("barrier 4 waves", t_barrier, 256, N, 1);
  run("readlane x64 + fma", t_readlane64, 64, N, 64);
  // the same with every CU busy (512 single-wave workgroups): clocks under load
  run("pk_fma 6 indep x96", t_pkfma_ind, 64, N, 96, 512);
  run("fwd FMA block (96+16rd)", t_fwd_block, 64, N, 96, 512);
  run("gates chain", t_gates, 64, N, 1, 512);
  run("lds roundtrip b128 bcast", t_lds_roundtrip, 64, N, 1, 512);
  return 0;
}
