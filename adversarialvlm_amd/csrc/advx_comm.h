// advx_comm.h - device side of the peer all-reduce of the shared image gradient.
//
// Every rank owns ONE exchange segment (uncached device memory, exported through HIP IPC and
// mapped by all peers over xGMI):
//     [flags A: 16 x uint32][flags B: 16 x uint32][error][send: n floats][recv: n floats]
// all-reduce(sum), two-shot, with the two rendezvous folded into the kernels that need them:
//   k_comm_reduce  : on entry block 0 tells every peer "my send buffer of exchange e is complete"
//                    (flag A = e) and every block waits until all peers have said so.  Rank r then owns the r-th contiguous slice of the image: it
//                    reads that slice of every peer's send buffer, adds them IN RANK ORDER (all
//                    ranks end up with the same bits) and posts the sum into the recv buffer of
//                    every peer (reduce-scatter by peer reads + all-gather by posted peer writes;
//                    each xGMI link carries n/G floats each way).  Every workgroup, once its
//                    stores have drained, counts itself in at every peer (flag B += 1).
//   comm_wait_b    : the consumer of recv (k_fused_update, or k_comm_wait for plain callers)
//                    waits on entry until every peer's flag B has reached the number of reduce
//                    workgroups launched so far.
// A wait that does not complete within the wall-clock limit sets the sticky error word and
// RETURNS: a lost peer costs a wrong step that advx_comm_status reports, never a hung device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace advx {

constexpr int kCommMaxRanks = 16;
constexpr int kCommFlagBytes = 4096;   // control words, padded: send starts 4 KiB into the segment
constexpr int kCommOffFlagsA = 0;
constexpr int kCommOffFlagsB = 256;
constexpr int kCommOffError = 512;

struct CommDev {
  int rank, world;
  char* base[kCommMaxRanks];        // base[j] = rank j's segment as mapped in this process
  long long send_off, recv_off;     // byte offsets of the payload buffers inside a segment
  unsigned long long timeout_ticks; // wall_clock64 ticks (100 MHz)
  // Both counters advance identically on every rank (an all-reduce is a collective call), so the
  // host passes them in: no device-side epoch word, one uncached round trip less per kernel.
  uint32_t epoch;                   // number of this exchange (1, 2, ...): value of flag A
  uint32_t posted;                  // reduce workgroups launched so far, this exchange included:
                                    // value flag B reaches once every slice of a peer has landed
};

// Visibility without fences (a fence is a cache-wide operation: 1323 workgroups issuing one each
// cost 80 us here).  The exchange segment is uncached device memory, and in addition EVERY access
// to a shared word or payload byte is a relaxed system-scope atomic (global_load/store ... sc0 sc1:
// write-through stores, cache-bypassing loads).  Ordering then needs only: a storing wave drains
// its stores (s_waitcnt vmcnt(0)) before the word that announces them is written, and a reading
// workgroup touches the payload only behind the barrier its polling wave joins after the match.
#define ADVX_RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

__device__ inline uint32_t* comm_word(const CommDev& c, int r, int off) {
  return reinterpret_cast<uint32_t*>(c.base[r] + off);
}

__device__ inline float4 comm_load4(const char* base, long long byte_off) {
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(base + byte_off);
  unsigned long long lo = __hip_atomic_load(p, ADVX_RLX_SYS), hi = __hip_atomic_load(p + 1, ADVX_RLX_SYS);
  return make_float4(__builtin_bit_cast(float, (uint32_t)lo), __builtin_bit_cast(float, (uint32_t)(lo >> 32)),
                     __builtin_bit_cast(float, (uint32_t)hi), __builtin_bit_cast(float, (uint32_t)(hi >> 32)));
}
__device__ inline void comm_store4(char* base, long long byte_off, float4 v) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(base + byte_off);
  unsigned long long lo = (unsigned long long)__builtin_bit_cast(uint32_t, v.x) |
                          ((unsigned long long)__builtin_bit_cast(uint32_t, v.y) << 32);
  unsigned long long hi = (unsigned long long)__builtin_bit_cast(uint32_t, v.z) |
                          ((unsigned long long)__builtin_bit_cast(uint32_t, v.w) << 32);
  __hip_atomic_store(p, lo, ADVX_RLX_SYS);
  __hip_atomic_store(p + 1, hi, ADVX_RLX_SYS);
}
__device__ inline float comm_load1(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), ADVX_RLX_SYS));
}

// threads 0..world-1 of the calling block publish `e` to every peer (slot = this rank).  The
// caller guarantees that whatever `e` announces has been drained (kernel boundary, or
// s_waitcnt vmcnt(0) + barrier + ticket).
__device__ inline void comm_signal(const CommDev& c, int flags_off, uint32_t e) {
  if ((int)threadIdx.x < c.world) __hip_atomic_store(comm_word(c, (int)threadIdx.x, flags_off) + c.rank, e, ADVX_RLX_SYS);
}

// the whole block waits until every peer's slot in THIS rank's flag array reached `e`
__device__ inline void comm_wait(const CommDev& c, int flags_off, uint32_t e) {
  if ((int)threadIdx.x < c.world) {
    const uint32_t* mine = comm_word(c, c.rank, flags_off) + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(mine, ADVX_RLX_SYS) - e) < 0) {
      if (wall_clock64() - t0 > c.timeout_ticks) {
        __hip_atomic_store(comm_word(c, c.rank, kCommOffError), 1u, ADVX_RLX_SYS);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

// consumer side: recv is complete once every peer's flag B - a count of the reduce workgroups
// that have drained their posted stores - has reached the number launched so far
__device__ inline void comm_wait_b(const CommDev& c) { comm_wait(c, kCommOffFlagsB, c.posted); }

__global__ void __launch_bounds__(64) k_comm_wait(CommDev c) { comm_wait_b(c); }

// n4 = number of float4 elements of the whole buffer; the slice of rank r is
// [r*per, min(n4, (r+1)*per)), per = ceil(n4 / world)
__global__ void __launch_bounds__(256) k_comm_reduce(CommDev c, long long n4) {
  if (blockIdx.x == 0) comm_signal(c, kCommOffFlagsA, c.epoch);   // kernel boundary before us: send is complete
  comm_wait(c, kCommOffFlagsA, c.epoch);
  const long long per = (n4 + c.world - 1) / c.world;
  const long long lo = per * c.rank;
  const long long hi = (lo + per < n4) ? lo + per : n4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long q = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; q < hi; q += stride) {
    float4 part[kCommMaxRanks];
#pragma unroll
    for (int r = 0; r < kCommMaxRanks; ++r)
      if (r < c.world) part[r] = comm_load4(c.base[r], c.send_off + q * 16);
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
      if (r < c.world) comm_store4(c.base[r], c.recv_off + q * 16, a);
  }
  // every wave drains its posted stores; behind the barrier the workgroup counts itself in at
  // every peer (memory-side atomic add, nothing returned, nothing waited for)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < c.world)
    __hip_atomic_fetch_add(comm_word(c, (int)threadIdx.x, kCommOffFlagsB) + c.rank, 1u, ADVX_RLX_SYS);
}

}  // namespace advx
