// f32 products on the bf16 matrix path: accuracy and rate of the error-free three-way split on gfx950.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mb3 tools/microbench_bf16x3.hip && /tmp/mb3
//   x = x_h + x_m + x_l exactly (three bf16 pieces by truncation: 8 + 8 + 8 mantissa bits), x*y ~ the 6 partial products of weight
//   >= 2^-16 (hh, hm, mh, hl, lh, mm), accumulated in f32 by v_mfma_f32_16x16x32_bf16.
// Part 1 (one wave): C = A^T B with K = 38 656 reduction rows, three ways -- the f32 MFMA chain, the split form, f64 on the host.
// Part 2 (one wave, s_memtime): issue cycles per MFMA, four independent accumulators.
// Part 3 (whole chip, HIP events): sustained rate of both forms on random register data (the clock is power-limited under MFMA load).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hi16(float x) { return __builtin_bit_cast(unsigned, x) & 0xFFFF0000u; }
// pack the bf16 truncations of two floats (a -> low half, b -> high half)
__device__ __forceinline__ unsigned pack_hi(float a, float b) { return (__builtin_bit_cast(unsigned, a) >> 16) | hi16(b); }

struct Split8 { u32x4 h, m, l; };
__device__ __forceinline__ Split8 split8(const float* v) {
  Split8 s;
  float r1[8], r2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r1[i] = v[i] - __builtin_bit_cast(float, hi16(v[i]));
    r2[i] = r1[i] - __builtin_bit_cast(float, hi16(r1[i]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s.h[i] = pack_hi(v[2 * i], v[2 * i + 1]);
    s.m[i] = pack_hi(r1[2 * i], r1[2 * i + 1]);
    s.l[i] = pack_hi(r2[2 * i], r2[2 * i + 1]);
  }
  return s;
}
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Ak[K][16], Bk[K][16]; C[16][16] (row = A column index, col = B column index)
__global__ void acc_f32(const float* Ak, const float* Bk, float* C, int K) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int kb = 0; kb < K; kb += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ak[(kb + g) * 16 + i], Bk[(kb + g) * 16 + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = acc[r];
}
template <int NPROD>
__global__ void acc_split(const float* Ak, const float* Bk, float* C, int K) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int kb = 0; kb < K; kb += 32) {
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { a[r] = Ak[(kb + 8 * g + r) * 16 + i]; b[r] = Bk[(kb + 8 * g + r) * 16 + i]; }
    const Split8 sa = split8(a), sb = split8(b);
    if (NPROD >= 6) {
      acc = mfma_bf(sa.m, sb.m, acc);
      acc = mfma_bf(sa.h, sb.l, acc);
      acc = mfma_bf(sa.l, sb.h, acc);
    }
    if (NPROD >= 3) {
      acc = mfma_bf(sa.h, sb.m, acc);
      acc = mfma_bf(sa.m, sb.h, acc);
    }
    acc = mfma_bf(sa.h, sb.h, acc);
    if (NPROD >= 9) {
      acc = mfma_bf(sa.m, sb.l, acc);
      acc = mfma_bf(sa.l, sb.m, acc);
      acc = mfma_bf(sa.l, sb.l, acc);
    }
  }
  for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = acc[r];
}

struct Out { long long cyc; float sink; };
__global__ void t_bf_issue(Out* o, int iters, unsigned seed) {
  u32x4 a = {seed + threadIdx.x, seed * 3 + threadIdx.x, seed * 5, seed * 7}, b = {seed * 11, seed * 13 + threadIdx.x, seed * 17, seed * 19};
  a &= 0x3F803F80u; b &= 0x3F803F80u;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { c0 = mfma_bf(a, b, c0); c1 = mfma_bf(a, b, c1); c2 = mfma_bf(a, b, c2); c3 = mfma_bf(a, b, c3); }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) o->cyc = t1 - t0;
  if (c0[0] + c1[0] + c2[0] + c3[0] == 123.456f) o->sink = c0[1];
}
__global__ void t_bf16k_issue(Out* o, int iters, unsigned seed) {      // the K = 16 form
  s16x4 a = {(short)(0x3F80 + (threadIdx.x & 7)), (short)0x3F81, (short)0x3F82, (short)0x3F83}, b = a;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) o->cyc = t1 - t0;
  if (c0[0] + c1[0] + c2[0] + c3[0] == 123.456f) o->sink = c0[1] + seed;
}

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// whole chip, random register data refreshed every 32 MFMAs (a cheap xorshift per register keeps the operands toggling)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int FORM>   // 0: f32 16x16x4, 1: bf16 16x16x32, 2: f32 32x32x2 (two 16-register accumulators... eight here: 128 registers)
__global__ void __launch_bounds__(256) t_full(float* sink, int iters) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  u32x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = (hash(t * 8 + i) & 0x007F007Fu) | 0x3F803F80u; b[i] = (hash(t * 8 + 4 + i) & 0x807F807Fu) | 0x3F003F00u; }
  if (FORM == 2) {
    f32x16 d[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) d[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          d[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[(i + k) & 3]), __builtin_bit_cast(float, b[(i ^ k) & 3]), d[i], 0, 0, 0);
        a[k] = (a[k] ^ (a[k] << 3) ^ (unsigned)it) & 0x007F007Fu | 0x3F803F80u;
        b[k] = (b[k] ^ (b[k] >> 2) ^ (unsigned)it) & 0x807F807Fu | 0x3F003F00u;
      }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += d[i][e];
    if (s == 123.456f) sink[t] = s;
    return;
  }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (FORM == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a[(i + k) & 3]), __builtin_bit_cast(float, b[(i ^ k) & 3]), c[i], 0, 0, 0);
        else c[i] = mfma_bf(a, b, c[i]);
      }
      a[k] = (a[k] ^ (a[k] << 3) ^ (unsigned)it) & 0x007F007Fu | 0x3F803F80u;
      b[k] = (b[k] ^ (b[k] >> 2) ^ (unsigned)it) & 0x807F807Fu | 0x3F003F00u;
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (s == 123.456f) sink[t] = s;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static float nrand() { return (float)(sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand())); }

int main() {
  const int K = 38656;
  std::vector<float> A(K * 16), B(K * 16);
  srand(1);
  for (auto& v : A) v = nrand() * 0.37f;
  for (auto& v : B) v = nrand() * 2.1f + 0.05f;      // (a mean: partial sums that do not cancel to zero)
  std::vector<double> ref(256, 0.0), refabs(256, 0.0);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) { ref[i * 16 + j] += (double)A[k * 16 + i] * B[k * 16 + j]; refabs[i * 16 + j] += fabs((double)A[k * 16 + i] * B[k * 16 + j]); }
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 256 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  auto report = [&](const char* name) {
    std::vector<float> C(256);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
    double cmax = 0, emax = 0, erel = 0, esum = 0;
    for (int i = 0; i < 256; ++i) cmax = fmax(cmax, fabs(ref[i]));
    for (int i = 0; i < 256; ++i) { const double e = C[i] - ref[i]; emax = fmax(emax, fabs(e)); erel = fmax(erel, fabs(e) / refabs[i]); esum += e / refabs[i]; }
    printf("%-34s max|err| / max|C| = %.3e   max |err| / sum|terms| = %.3e   mean signed err / sum|terms| = %+.3e\n", name, emax / cmax, erel, esum / 256);
  };
  printf("Part 1: C = A^T B, K = %d, 16 x 16 outputs, against f64\n", K);
  acc_f32<<<1, 64>>>(dA, dB, dC, K); report("f32 MFMA chain (16x16x4)");
  acc_split<1><<<1, 64>>>(dA, dB, dC, K); report("bf16 hh only (plain bf16)");
  acc_split<3><<<1, 64>>>(dA, dB, dC, K); report("bf16 split, 3 products");
  acc_split<6><<<1, 64>>>(dA, dB, dC, K); report("bf16 split, 6 products");
  acc_split<9><<<1, 64>>>(dA, dB, dC, K); report("bf16 split, 9 products");

  Out* dO; CK(hipMalloc(&dO, sizeof(Out)));
  Out ho;
  printf("Part 2: one wave, 4 independent accumulators, cycles per MFMA\n");
  for (int rep = 0; rep < 2; ++rep) {
    t_bf_issue<<<1, 64>>>(dO, 1000, 12345u); CK(hipDeviceSynchronize()); CK(hipMemcpy(&ho, dO, sizeof(Out), hipMemcpyDeviceToHost));
    if (rep) printf("  v_mfma_f32_16x16x32_bf16: %.2f\n", ho.cyc / 32000.0);
    t_bf16k_issue<<<1, 64>>>(dO, 1000, 12345u); CK(hipDeviceSynchronize()); CK(hipMemcpy(&ho, dO, sizeof(Out), hipMemcpyDeviceToHost));
    if (rep) printf("  v_mfma_f32_16x16x16_bf16: %.2f\n", ho.cyc / 32000.0);
  }
  printf("Part 3: whole chip (256 workgroups x 4 waves, one wave per SIMD), 8 accumulators per wave, operands refreshed every 8 MFMAs\n");
  float* dS; CK(hipMalloc(&dS, 256 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int form = 0; form < 3; ++form) {
    const int iters = form == 0 ? 100 : (form == 1 ? 400 : 50);       // 32 MFMAs per iteration per wave
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      if (form == 0) t_full<0><<<256, 256>>>(dS, iters); else if (form == 1) t_full<1><<<256, 256>>>(dS, iters); else t_full<2><<<256, 256>>>(dS, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double n = 32.0 * iters, flop = n * 1024 * (form == 0 ? 2048.0 : (form == 1 ? 16384.0 : 4096.0));
      const int cyc = form == 0 ? 32 : (form == 1 ? 16 : 64);
      if (rep) printf("  %s: %.1f us for %d MFMAs per wave = %.1f ns each = %.1f TFLOP/s (at 1 MFMA per %d cycles: %.2f GHz)\n", form == 0 ? "f32 16x16x4 " : (form == 1 ? "bf16 16x16x32" : "f32 32x32x2 "),
                      ms * 1e3, (int)n, ms * 1e6 / n, flop / ms / 1e9, cyc, n * cyc / (ms * 1e6));
    }
  }
  return 0;
}
