// One-shot all-reduce of the flat gradient vector over xGMI: push to every peer, flag, wait, fixed-order sum.
// See include/ope.h ("One-shot gradient all-reduce over xGMI") for the protocol and the buffer layout.
//
// Replaces (reference): nothing -- the reference trains on one device (its `average_gradients`,
// offpolicy/utils/util.py:148-153, is dead code). SURVEY.md section 8(e) defines the exchange: ONE sum of
// [grads | loss_sum | mask_count | qtot_sum | 0] between the backward pass and the optimizer.
//
// Why not a ring: 475 KB over 7 x 153 GB/s links is ~0.5 us of wire time per peer; a ring all-reduce pays 2(W-1)
// dependent hops of launch + fabric latency for it. Here every workgroup owns a 4 KB chunk of the vector: it writes the
// chunk into its rank's slot on all W exchange buffers (peer order staggered by rank so that the W ranks do not hit the
// same link at once), fences at system scope, raises flag[parity][rank][chunk] = epoch on every peer, polls its OWN
// buffer's flags of that chunk, and sums the W slots in rank order. No grid-wide sync: chunk c only depends on chunk c of
// the peers. Parity double-buffering makes the slots of epoch e safe from writers of epoch e+1 (a rank can only reach
// epoch e+2 after every peer has pushed e+1, i.e. finished reading e).
#include <string.h>

#include "ope_common.h"

namespace {

constexpr int kChunk = 1024;      // floats per workgroup: 256 threads x float4
constexpr int kBlock = 256;
constexpr int kDefaultTimeoutMs = 10000;   // ranks may be skewed by lazy initialisation or host work: wait up to 10 s

struct ArArgs {
  float* peer[OPE_AR_MAX_WORLD];
  int rank, world;
  int64_t max_floats;
  int chunks_max;
  uint32_t epoch;
  uint32_t* epoch_dev;                // if set: {epoch of the previous exchange, finished-workgroup ticket} on the device; epoch ignored
  unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
};
// next epoch of the cycle 1, 2, ..., 0xFFFFFFFE, 1, ... (never 0; parity = epoch & 1 keeps alternating over the even cycle)
__host__ __device__ __forceinline__ uint32_t next_epoch(uint32_t e) { return e % 0xFFFFFFFEu + 1u; }

__device__ __forceinline__ uint32_t* flag_ptr(float* base, int chunks_max, int world, int parity, int src, int chunk) {
  return reinterpret_cast<uint32_t*>(base) + ((int64_t)(parity * world + src) * chunks_max + chunk);
}
__device__ __forceinline__ float* slot_ptr(float* base, int chunks_max, int world, int64_t max_floats, int parity, int src) {
  return base + (int64_t)2 * world * chunks_max + (int64_t)(parity * world + src) * max_floats;
}

__global__ void __launch_bounds__(kBlock) allreduce_push_kernel(ArArgs a, float* __restrict__ flat, int64_t n, int32_t* status) {
  const int chunk = blockIdx.x, tid = threadIdx.x;
  // Device-held epoch (graph-capturable exchange: nothing of a launch changes between replays): every workgroup derives the epoch
  // from the value the previous exchange left, and the LAST workgroup to finish (ticket) stores it for the next one. No workgroup
  // can still have to read the old value by then: it would not have finished.
  const uint32_t epoch = a.epoch_dev ? next_epoch(__hip_atomic_load(a.epoch_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : a.epoch;
  auto finish = [&]() {
    if (a.epoch_dev && tid == 0) {
      if (atomicAdd(a.epoch_dev + 1, 1u) == gridDim.x - 1) {
        __hip_atomic_store(a.epoch_dev + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.epoch_dev, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  const int parity = epoch & 1u;
  const int64_t i = (int64_t)chunk * kChunk + 4 * tid;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i + 3 < n) {
    v = *reinterpret_cast<const f32x4*>(flat + i);
  } else {
    for (int r = 0; r < 4; ++r)
      if (i + r < n) v[r] = flat[i + r];
  }
  // push: own slot on every rank's buffer (own buffer included), staggered so that rank r starts with peer r+1
  for (int q = 1; q <= a.world; ++q) {
    const int dst = (a.rank + q) % a.world;
    *reinterpret_cast<f32x4*>(slot_ptr(a.peer[dst], a.chunks_max, a.world, a.max_floats, parity, a.rank) + i) = v;
  }
  // Release: every wave waits until ITS slot stores are acknowledged, the workgroup meets, and then ONE wave (the one that raises
  // the flags) executes the system-scope fence: the write-back it triggers acts on the whole L2, not on the issuing wave's lines, so
  // one per workgroup orders all four waves' stores before the flags. (Measured at world 1, 464 KB: a fence in every wave on both
  // sides made the launch 17.7 us, none 6.3 us -- the fences ARE the cost of this kernel; profiles/r03z_allreduce_fences.txt.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < 64) {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (hipcc may drop the wait after buffer_wbl2, MI355X_MICROARCH.md)
  }
  if (tid < a.world) {
    const int dst = (a.rank + 1 + tid) % a.world;
    __hip_atomic_store(flag_ptr(a.peer[dst], a.chunks_max, a.world, parity, a.rank, chunk), epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // wait for the W flags of this chunk in the OWN buffer (bounded)
  __shared__ int gave_up;
  if (tid == 0) gave_up = 0;
  __syncthreads();
  if (tid < a.world) {
    const uint32_t* f = flag_ptr(a.peer[a.rank], a.chunks_max, a.world, parity, tid, chunk);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > a.timeout_ticks) {
        atomicOr(status, 1);
        gave_up = 1;
        break;
      }
    }
  }
  // Acquire: the wave that saw the flags drops stale lines (L2 and this CU's vector cache are shared by the workgroup's waves)
  // before anybody reads the slots; the slot loads below are system-coherent loads on top of that.
  if (tid < 64) __threadfence_system();
  __syncthreads();
  if (gave_up) {
    // A peer's chunk never arrived: the slots hold a PREVIOUS epoch's data. Summing them would hand the optimizer a plausible
    // but wrong gradient on this rank only, and the replicas would drift apart silently. Poison the chunk instead: the clip
    // norm, the loss and every parameter of this rank go NaN on this very step, and through the next exchange on all ranks.
    const float qnan = __builtin_nanf("");
    for (int r = 0; r < 4; ++r)
      if (i + r < n) flat[i + r] = qnan;
    finish();
    return;
  }
  // The W slots are read with ONE 16-byte system-coherent load each, ALL issued before the first is waited for (asm: a volatile
  // C++ load gets an s_waitcnt vmcnt(0) of its own, and four system-scope dword loads per slot -- the first version -- were W
  // dependent round trips through uncached memory: what an 8-rank exchange spent most of its time in). The registers are
  // zero-filled first and tied read-write to the loads and to the wait, so hipcc neither copies nor reads them in between.
  // Summation stays in FIXED rank order: identical bits on every rank.
  f32x4 x[OPE_AR_MAX_WORLD];
#pragma unroll
  for (int q = 0; q < OPE_AR_MAX_WORLD; ++q) x[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < OPE_AR_MAX_WORLD; ++q)
    if (q < a.world) {
      const float* sp = slot_ptr(a.peer[a.rank], a.chunks_max, a.world, a.max_floats, parity, q) + i;
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "+v"(x[q]) : "v"(sp) : "memory");
    }
  static_assert(OPE_AR_MAX_WORLD == 16, "the wait below names sixteen registers");
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                 "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
               :
               : "memory");
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < OPE_AR_MAX_WORLD; ++q)
    if (q < a.world) s += x[q];
  if (i + 3 < n) {
    *reinterpret_cast<f32x4*>(flat + i) = s;
  } else {
    for (int r = 0; r < 4; ++r)
      if (i + r < n) flat[i + r] = s[r];
  }
  finish();
}

inline int64_t chunks_of(int64_t max_floats) { return (max_floats + kChunk - 1) / kChunk; }

}  // namespace

extern "C" int64_t ope_allreduce_buffer_bytes(int64_t max_floats, int32_t world) {
  if (max_floats < 1 || max_floats % kChunk != 0 || world < 1 || world > OPE_AR_MAX_WORLD) return OPE_EINVAL;
  return (int64_t)2 * world * (chunks_of(max_floats) + max_floats) * 4;
}

extern "C" int ope_allreduce_alloc(int64_t bytes, void** buf_out) {
  if (bytes < 1 || !buf_out) return OPE_EINVAL;
  void* p = nullptr;
  // fine-grained + uncached: peers' writes over xGMI and our own reads never sit in a non-coherent L2 line
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      return OPE_EHIP;
    }
  }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return OPE_EHIP;
  }
  *buf_out = p;
  return OPE_OK;
}

extern "C" int ope_allreduce_free(void* buf) {
  if (!buf) return OPE_EINVAL;
  return hipFree(buf) == hipSuccess ? OPE_OK : OPE_EHIP;
}

extern "C" int ope_allreduce_ipc_export(void* buf, void* handle_host) {
  static_assert(sizeof(hipIpcMemHandle_t) == OPE_AR_IPC_HANDLE_BYTES, "IPC handle size");
  if (!buf || !handle_host) return OPE_EINVAL;
  if (hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_host), buf) != hipSuccess) {
    (void)hipGetLastError();
    return OPE_EHIP;
  }
  return OPE_OK;
}

extern "C" int ope_allreduce_ipc_import(const void* handle_host, void** mapped_out) {
  if (!handle_host || !mapped_out) return OPE_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle_host, sizeof(h));
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
    (void)hipGetLastError();
    return OPE_EHIP;
  }
  *mapped_out = p;
  return OPE_OK;
}

extern "C" int ope_allreduce_enable_peer(int32_t peer_device) {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return OPE_EHIP;
  if (peer_device == cur) return OPE_OK;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, cur, peer_device) != hipSuccess || !can) {
    (void)hipGetLastError();
    return OPE_EHIP;
  }
  const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
  (void)hipGetLastError();
  return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? OPE_OK : OPE_EHIP;
}

extern "C" int ope_allreduce_ipc_close(void* mapped) {
  if (!mapped) return OPE_EINVAL;
  return hipIpcCloseMemHandle(mapped) == hipSuccess ? OPE_OK : OPE_EHIP;
}

static int allreduce_launch(const ope_allreduce_ctx* ctx, uint32_t epoch, uint32_t* epoch_dev, float* flat, int64_t n, int32_t* status, void* stream) {
  (void)hipGetLastError();
  if (!ctx || !flat || !status || (epoch == 0 && !epoch_dev) || ctx->world < 1 || ctx->world > OPE_AR_MAX_WORLD || ctx->rank < 0 ||
      ctx->rank >= ctx->world || n < 1 || n > ctx->max_floats || ctx->max_floats % kChunk != 0)
    return OPE_EINVAL;
  ArArgs a;
  for (int q = 0; q < ctx->world; ++q) {
    if (!ctx->peer[q]) return OPE_EINVAL;
    a.peer[q] = reinterpret_cast<float*>(ctx->peer[q]);
  }
  a.rank = ctx->rank; a.world = ctx->world; a.max_floats = ctx->max_floats; a.chunks_max = (int)chunks_of(ctx->max_floats);
  a.epoch = epoch; a.epoch_dev = epoch_dev;
  a.timeout_ticks = (unsigned long long)(ctx->timeout_ms > 0 ? ctx->timeout_ms : kDefaultTimeoutMs) * 100000ull;
  OPE_LAUNCH(allreduce_push_kernel, dim3(ope_cdiv(n, kChunk)), dim3(kBlock), 0, (hipStream_t)stream, a, flat, n, status);
  OPE_CHECK_LAUNCH();
  return OPE_OK;
}

extern "C" int ope_allreduce_flat(const ope_allreduce_ctx* ctx, uint32_t epoch, float* flat, int64_t n, int32_t* status, void* stream) {
  return allreduce_launch(ctx, epoch, nullptr, flat, n, status, stream);
}

extern "C" int ope_allreduce_flat_dev(const ope_allreduce_ctx* ctx, uint32_t* epoch_state_dev, float* flat, int64_t n, int32_t* status, void* stream) {
  if (!epoch_state_dev) return OPE_EINVAL;
  return allreduce_launch(ctx, 0, epoch_state_dev, flat, n, status, stream);
}
