// Counter-based uniform noise drawn inside the kernels that consume it (Philox4x32-10, Salmon et al. SC'11): no noise tensor,
// no generator launch in front of the update, and a captured HIP graph replays with fresh noise because part of the counter
// is a device-resident step count. Stands in for the reference's host-side torch.rand of the gumbel trick
// (offpolicy/utils/util.py:188-190 sample_gumbel): same distribution, not the same stream -- parity tests inject noise.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace ope {

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// u in [0, 1) for element `col` of row `row` of noise stream `stream` at step `step` (one Philox block per 4 columns)
__device__ __forceinline__ float device_uniform(uint64_t seed, int step, int stream, int64_t row, int col) {
  const Philox4 x = philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32) ^ ((uint32_t)(col >> 2) << 8), (uint32_t)step,
                                  (uint32_t)stream, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (float)(x.v[col & 3] >> 8) * (1.0f / 16777216.0f);
}

// the 4 uniforms of columns 4q .. 4q + 3 (one Philox block; element i is device_uniform(.., col = 4q + i))
__device__ __forceinline__ void device_uniform4(uint64_t seed, int step, int stream, int64_t row, int q, float (&u)[4]) {
  const Philox4 x = philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32) ^ ((uint32_t)q << 8), (uint32_t)step, (uint32_t)stream,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(x.v[i] >> 8) * (1.0f / 16777216.0f);
}

// What the kernels carry: a table pointer (caller-provided noise) or, when that is null, the generator's coordinates.
struct NoiseSrc {
  const float* u;          // [rows][A] uniform(0,1) or null
  uint64_t seed;           // used when u is null
  const int32_t* step;     // device step count (null = 0)
  int stream;
  __device__ __forceinline__ float at(int64_t row, int A, int col) const {
    return u ? u[row * A + col] : device_uniform(seed, step ? step[0] : 0, stream, row, col);
  }
  // columns 4q .. 4q + 3 of the row (0.5 beyond A)
  __device__ __forceinline__ void at4(int64_t row, int A, int q, float (&out)[4]) const {
    if (u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) out[i] = 4 * q + i < A ? u[row * A + 4 * q + i] : 0.5f;
    } else {
      device_uniform4(seed, step ? step[0] : 0, stream, row, q, out);
    }
  }
};

}  // namespace ope
