// LDS read-pattern micro-benchmark (gfx950): cost of feeding the same 16 bytes to every lane of a wave.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mbl tools/microbench_lds.hip && /tmp/mbl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Out { long long cyc, wall; float sink; };

// MODE: 0 b128, 1 b64, 2 b32 ; address = (lane & mask) * stride_bytes
template <int MODE>
__global__ void t_read(Out* o, int iters, int mask, int stride) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) hs[i] = i;
  __syncthreads();
  const int addr = ((threadIdx.x & 63) & mask) * stride;
  float acc = 0.f;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      f32x4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(512 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    } else if (MODE == 1) {
      f32x2 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(512 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    } else {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(512 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    }
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
  if (acc == 123.456f) o->sink = acc;
}

// writes: 64 lanes, address = f(lane): MODE 0: lane*4 (conflict free b32), 1: replicated layout v*S + c*4 + i for 8 copies
__global__ void t_write8(Out* o, int iters, int S) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  const int f = threadIdx.x & 63;
  const int base = ((f >> 2) * S + (f & 3)) * 4;
  float h = f;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(h), "n"(16 * c) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
  if (hs[f] == 123.456f) o->sink = h;
}


#define LDS_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LDS_RD64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define FMA6(lo, hi)                                  \
  a0 = __builtin_elementwise_fma(w, lo, a0);          \
  a1 = __builtin_elementwise_fma(w, lo, a1);          \
  a2 = __builtin_elementwise_fma(w, lo, a2);          \
  a3 = __builtin_elementwise_fma(w, hi, a3);          \
  a4 = __builtin_elementwise_fma(w, hi, a4);          \
  a5 = __builtin_elementwise_fma(w, hi, a5);          \
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));

// forward-step FMA block: 16 broadcast b128 reads, 4 in flight, 96 packed FMAs
__global__ void t_fwd_block128(Out* o, int iters, int mask, int stride) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) hs[i] = 1e-3f * i;
  __syncthreads();
  const int addr = ((threadIdx.x & 63) & mask) * stride;
  f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, w = {0.5f, 0.25f};
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    f32x4 q0, q1, q2, q3;
    LDS_RD128(q0, addr, 0); LDS_RD128(q1, addr, 256); LDS_RD128(q2, addr, 512); LDS_RD128(q3, addr, 768);
#define STEP4(base, more)                                                                                       \
    LGKM(3); { f32x2 lo = {q0[0], q0[1]}, hi = {q0[2], q0[3]}; FMA6(lo, hi) } if (more) LDS_RD128(q0, addr, base + 1024); \
    LGKM(3); { f32x2 lo = {q1[0], q1[1]}, hi = {q1[2], q1[3]}; FMA6(lo, hi) } if (more) LDS_RD128(q1, addr, base + 1280); \
    LGKM(3); { f32x2 lo = {q2[0], q2[1]}, hi = {q2[2], q2[3]}; FMA6(lo, hi) } if (more) LDS_RD128(q2, addr, base + 1536); \
    LGKM(3); { f32x2 lo = {q3[0], q3[1]}, hi = {q3[2], q3[3]}; FMA6(lo, hi) } if (more) LDS_RD128(q3, addr, base + 1792);
    STEP4(0, true) STEP4(1024, true) STEP4(2048, true)
    LGKM(3); { f32x2 lo = {q0[0], q0[1]}, hi = {q0[2], q0[3]}; FMA6(lo, hi) }
    LGKM(2); { f32x2 lo = {q1[0], q1[1]}, hi = {q1[2], q1[3]}; FMA6(lo, hi) }
    LGKM(1); { f32x2 lo = {q2[0], q2[1]}, hi = {q2[2], q2[3]}; FMA6(lo, hi) }
    LGKM(0); { f32x2 lo = {q3[0], q3[1]}, hi = {q3[2], q3[3]}; FMA6(lo, hi) }
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
  float v = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a0[1] + a1[1] + a2[1] + a3[1] + a4[1] + a5[1];
  if (v == 123.456f) o->sink = v;
}

// same work fed by 32 ds_read_b64 (8 in flight)
__global__ void t_fwd_block64(Out* o, int iters, int mask, int stride) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) hs[i] = 1e-3f * i;
  __syncthreads();
  const int addr = ((threadIdx.x & 63) & mask) * stride;
  f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, w = {0.5f, 0.25f};
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    f32x2 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) LDS_RD64(p[k], addr, 128 * k);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      if (v < 12) { LGKM(6); } else if (v == 12) { LGKM(6); } else if (v == 13) { LGKM(4); } else if (v == 14) { LGKM(2); } else { LGKM(0); }
      const f32x2 lo = p[(2 * v) % 8], hi = p[(2 * v + 1) % 8];
      FMA6(lo, hi)
      if (v < 12) { LDS_RD64(p[(2 * v) % 8], addr, 128 * (2 * v + 8)); LDS_RD64(p[(2 * v + 1) % 8], addr, 128 * (2 * v + 9)); }
    }
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
  float v = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a0[1] + a1[1] + a2[1] + a3[1] + a4[1] + a5[1];
  if (v == 123.456f) o->sink = v;
}


// LDS pipe occupancy: W waves of one workgroup each issue 16 reads per repetition (MODE 0 b128, 1 b64, 2 b32)
template <int MODE>
__global__ void t_read_mw(Out* o, int iters, int mask, int stride) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) hs[i] = i;
  __syncthreads();
  const int addr = ((threadIdx.x & 63) & mask) * stride;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      f32x4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(512 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    } else if (MODE == 1) {
      f32x2 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(512 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    } else {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(512 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    }
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
}

// 16 ds_write_b32 per repetition, address = (lane >> shift) * 4: shift 0 distinct, shift 2 = four lanes per address
__global__ void t_write_same(Out* o, int iters, int shift, int unused) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  const int addr = ((threadIdx.x & 63) >> shift) * 4;
  float h = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(h), "n"(256 * c) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
  if (hs[threadIdx.x] == 123.456f) o->sink = h;
}

// one ds_write_b32 + s_waitcnt lgkmcnt(0): write completion latency
__global__ void t_write_lat(Out* o, int iters, int shift, int unused) {
  __shared__ __attribute__((aligned(16))) float hs[4096];
  const int addr = ((threadIdx.x & 63) >> shift) * 4;
  float h = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(h) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) { o->cyc = c1 - c0; }
  if (hs[threadIdx.x] == 123.456f) o->sink = h;
}

template <typename K, typename... A>
void runw(const char* name, K kern, int threads, int iters, int per, A... args) {
  Out* d; Out h;
  (void)hipMalloc(&d, sizeof(Out));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, iters, args...); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&h, d, sizeof(Out), hipMemcpyDeviceToHost);
  printf("%-36s waves=%2d : %8.1f clk/rep  %6.2f clk per wave-op  %6.2f clk per op (CU)\n", name, threads / 64, (double)h.cyc / iters, (double)h.cyc / iters / per,
         (double)h.cyc / iters / per / (threads / 64));
  (void)hipFree(d);
}

template <typename K, typename... A>
void run(const char* name, K kern, int iters, int per, A... args) {
  Out* d; Out h;
  (void)hipMalloc(&d, sizeof(Out));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, iters, args...); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&h, d, sizeof(Out), hipMemcpyDeviceToHost);
  printf("%-44s : %8.1f clk/rep  %6.2f clk/op\n", name, (double)h.cyc / iters, (double)h.cyc / iters / per);
  (void)hipFree(d);
}

int main() {
  const int N = 20000;
  run("b128 same address (mask 0)", t_read<0>, N, 16, 0, 16);
  run("b128 2 copies  (lane&1)*16", t_read<0>, N, 16, 1, 16);
  run("b128 4 copies  (lane&3)*16", t_read<0>, N, 16, 3, 16);
  run("b128 8 copies  (lane&7)*16", t_read<0>, N, 16, 7, 16);
  run("b128 16 copies (lane&15)*16", t_read<0>, N, 16, 15, 16);
  run("b128 distinct  lane*16", t_read<0>, N, 16, 63, 16);
  run("b128 8 copies stride 144 B", t_read<0>, N, 16, 7, 144);
  run("b64  same address", t_read<1>, N, 16, 0, 8);
  run("b64  16 copies (lane&15)*8", t_read<1>, N, 16, 15, 8);
  run("b64  distinct lane*8", t_read<1>, N, 16, 63, 8);
  run("b32  same address", t_read<2>, N, 16, 0, 4);
  run("b32  32 copies (lane&31)*4", t_read<2>, N, 16, 31, 4);
  run("b32  distinct lane*4", t_read<2>, N, 16, 63, 4);
  run("fwd block 96 pkfma + 16 b128 same addr", t_fwd_block128, N, 96, 0, 16);
  run("fwd block 96 pkfma + 16 b128 8 copies", t_fwd_block128, N, 96, 7, 16);
  run("fwd block 96 pkfma + 32 b64 same addr", t_fwd_block64, N, 96, 0, 8);
  run("fwd block 96 pkfma + 32 b64 16 copies", t_fwd_block64, N, 96, 15, 8);
  run("8 x ds_write_b32 replicated, S=36", t_write8, N, 8, 36);
  run("8 x ds_write_b32 replicated, S=32", t_write8, N, 8, 32);
  for (int w = 1; w <= 16; w *= 2) runw("b128 same address", t_read_mw<0>, 64 * w, N, 16, 0, 16);
  for (int w = 1; w <= 16; w *= 2) runw("b128 4 addresses (lane&3)*64", t_read_mw<0>, 64 * w, N, 16, 3, 64);
  for (int w = 4; w <= 16; w *= 2) runw("b128 distinct lane*16", t_read_mw<0>, 64 * w, N, 16, 63, 16);
  for (int w = 4; w <= 16; w *= 2) runw("b64 same address", t_read_mw<1>, 64 * w, N, 16, 0, 8);
  for (int w = 4; w <= 16; w *= 2) runw("b32 same address", t_read_mw<2>, 64 * w, N, 16, 0, 4);
  for (int w = 4; w <= 16; w *= 2) runw("b32 distinct", t_read_mw<2>, 64 * w, N, 16, 63, 4);
  runw("16 ds_write_b32 distinct", t_write_same, 64, N, 16, 0, 0);
  runw("16 ds_write_b32 4 lanes/address", t_write_same, 64, N, 16, 2, 0);
  runw("ds_write_b32 + wait (latency)", t_write_lat, 64, N, 1, 0, 0);
  runw("ds_write_b32 4 lanes/addr + wait", t_write_lat, 64, N, 1, 2, 0);
  return 0;
}
