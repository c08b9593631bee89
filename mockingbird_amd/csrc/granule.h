// Tagged-granule hand-offs between the workgroups of a resident launch (wavernn_persist.h, wavernn_pipe.h,
// ppg_resident.h): an 8-byte {value, step tag} word written with ONE relaxed agent-scope store and read with one
// relaxed agent-scope load (global_store/load_dwordx2 sc1, MI355X_MICROARCH.md "R2 granule") -- no fences, the tag
// IS the flag.  Every spin has a wall-clock bail-out (wp_lost).
#pragma once
#include "common.h"

namespace mb {

constexpr unsigned long long WP_TIMEOUT_TICKS = 20000000ull;  // 0.2 s of the 100 MHz wall clock before a wait is declared lost
// every 1024th poll of a spin: start the wall clock at the first check, raise the abort word once the wait is older than
// WP_TIMEOUT_TICKS; true = the launch is aborting (this or another workgroup gave up), drain
__device__ __forceinline__ bool wp_lost(const int tries, unsigned long long& t0, int* abort_word) {
  const unsigned long long now = (unsigned long long)wall_clock64();
  if (tries == 1023) t0 = now;
  else if (now - t0 > WP_TIMEOUT_TICKS) atomicExch(abort_word, 1);
  return __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

__device__ __forceinline__ void wp_put(unsigned long long* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wp_put_u(unsigned long long* p, unsigned v, unsigned tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long wp_get(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait for one granule with TWO loads in flight: the next poll is on its way while the last one is looked at, so a fresh value is seen
// half a round trip sooner on average than with load / test / sleep (the chain edges of taco_front_kernel: ~0.2 us each).
// Returns the granule (stale if the wait was lost: the caller's launch is aborting anyway).
__device__ __forceinline__ unsigned long long wp_wait2(const unsigned long long* p, const unsigned tag, int* abort_word) {
  unsigned long long cur = wp_get(p), t0 = 0;
  for (int tries = 0;; ++tries) {
    const unsigned long long nxt = wp_get(p);
    if ((unsigned)(cur >> 32) == tag) return cur;
    if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) return cur;
    cur = nxt;
  }
}

}  // namespace mb
