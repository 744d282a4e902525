// advx_comm.h - device side of the peer all-reduce of the shared image gradient.
//
// Every rank owns ONE exchange segment (uncached device memory, exported through HIP IPC and
// mapped by all peers over xGMI):
//     [flags: kCommMaxRanks x uint32, one per peer][send: n floats][recv: n floats]
// all-reduce(sum) = barrier -> reduce -> barrier:
//   k_comm_barrier : one small workgroup; lane j stores the new epoch into peer j's flag word
//                    for this rank (release, system scope), then polls its own flag word for
//                    peer j (acquire, system scope) until it reaches the epoch - or until the
//                    wall-clock limit passes, in which case the sticky error word is set and
//                    the kernel EXITS (a lost peer can never hang the GPU).
//   k_comm_reduce  : rank r owns the r-th contiguous slice of the image; it reads that slice of
//                    every peer's send buffer, adds them IN RANK ORDER (so all ranks hold the
//                    same bits) and writes the sum into the recv buffer of every peer
//                    (two-shot: reduce-scatter by peer reads + all-gather by posted peer writes;
//                    each of the 7 xGMI links carries n/G floats each way).
// The epoch lives in device memory and is advanced by the barrier kernel itself, so the whole
// sequence has constant kernel arguments (graph-capturable, no host involvement per step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace advx {

constexpr int kCommMaxRanks = 16;
constexpr int kCommFlagBytes = 4096;   // flags + epoch + error, padded: send starts 4 KiB in

struct CommDev {
  int rank, world;
  uint32_t* flags[kCommMaxRanks];   // flags[j] = base of rank j's flag array (mapped here)
  const float* send[kCommMaxRanks];
  float* recv[kCommMaxRanks];
  uint32_t* epoch;                  // local: barrier counter
  uint32_t* error;                  // local: sticky, 1 = a barrier timed out
  unsigned long long timeout_ticks; // wall_clock64 ticks (100 MHz)
};

__global__ void __launch_bounds__(64) k_comm_barrier(CommDev c) {
  __shared__ uint32_t s_epoch;
  const int t = threadIdx.x;
  if (t == 0) {
    uint32_t e = *c.epoch + 1u;
    *c.epoch = e;
    s_epoch = e;
  }
  __syncthreads();
  const uint32_t e = s_epoch;
  if (t < c.world) {
    // everything this rank wrote before the barrier (kernel boundaries flushed it; the release
    // covers what a fused caller may add) becomes visible before the flag does
    __hip_atomic_store(c.flags[t] + c.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    const uint32_t* mine = c.flags[c.rank] + t;
    while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      if (wall_clock64() - t0 > c.timeout_ticks) {
        __hip_atomic_store(c.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
}

// n4 = number of float4 elements of the whole buffer; the slice of rank r is
// [r*per, min(n4, (r+1)*per)), per = ceil(n4 / world)
__global__ void __launch_bounds__(256) k_comm_reduce(CommDev c, long long n4) {
  const long long per = (n4 + c.world - 1) / c.world;
  const long long lo = per * c.rank;
  const long long hi = (lo + per < n4) ? lo + per : n4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long q = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; q < hi; q += stride) {
    float4 part[kCommMaxRanks];
#pragma unroll
    for (int r = 0; r < kCommMaxRanks; ++r)
      if (r < c.world) part[r] = reinterpret_cast<const float4*>(c.send[r])[q];
    float4 a = part[0];
#pragma unroll
    for (int r = 1; r < kCommMaxRanks; ++r)
      if (r < c.world) {
        a.x += part[r].x;
        a.y += part[r].y;
        a.z += part[r].z;
        a.w += part[r].w;
      }
#pragma unroll
    for (int r = 0; r < kCommMaxRanks; ++r)
      if (r < c.world) reinterpret_cast<float4*>(c.recv[r])[q] = a;
  }
}

}  // namespace advx
