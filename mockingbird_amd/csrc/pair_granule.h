// Operand pairs on tagged granules: the exchange format of the resident kernels that multiply on the fp16 matrix pipe
// (wavernn_pipe16.h, ppg_batch.h).  A value crosses workgroups as fp16 hi + fp16 lo with x = xh + 2^-11 xl (the residual stored
// scaled: conv1d.hip's scheme), an 8-byte granule carries TWO features {hi pair | lo pair} and is written with ONE relaxed
// agent-scope store; the step tag is the least significant bit of the first low half (tbit alternates per parity buffer, memory
// starts as 0).  See wavernn_pipe16.h's header for the derivation and the measurements behind each choice.
#pragma once
#include "granule.h"

namespace mb {

typedef _Float16 wh16;
typedef _Float16 wh16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wh16x8 __attribute__((ext_vector_type(8)));
typedef float wq_f2 __attribute__((ext_vector_type(2)));
typedef unsigned wq_u4 __attribute__((ext_vector_type(4)));

// ---- device side ----
// gate functions on the hardware exp2 / reciprocal (1 ulp each): this kernel has no bit-identical partner to keep (gru_scan.h's rule)
__device__ __forceinline__ float wq16_sigmoid(const float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float wq16_tanh(const float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
__device__ __forceinline__ unsigned wq16_tbit(const unsigned tag) { return ((tag + 1u) >> 1) & 1u; }
constexpr float WQ16_LO_SCALE = 2048.f, WQ16_LO_UNSCALE = 4.8828125e-4f;  // 2^11, 2^-11

// (va, vb) -> one granule {xh | xh' << 16, xl | xl' << 16} with the tag bit in bit 0 of the second word: two packed conversions
__device__ __forceinline__ void wq16_put(unsigned long long* p, const float va, const float vb, const unsigned tb) {
  const wq_f2 v = {va, vb};
  const wh16x2 h = __builtin_convertvector(v, wh16x2);
  const wq_f2 d = (v - __builtin_convertvector(h, wq_f2)) * WQ16_LO_SCALE;  // exact: the difference has <= 13 significant bits
  const wh16x2 l = __builtin_convertvector(d, wh16x2);
  const unsigned w0 = __builtin_bit_cast(unsigned, h), w1 = (__builtin_bit_cast(unsigned, l) & ~1u) | tb;
  __hip_atomic_store(p, ((unsigned long long)w1 << 32) | (unsigned long long)w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Range bookkeeping of a publishing lane: a running NaN-propagating maximum (v_maximum3_f32) of |a| + |b| over everything it published,
// looked at ONCE when its role's loop ends -- no compare-and-branch on the critical path of an item.  (A pair sum that trips although both
// terms are in range only sends the utterance to the exact chain.)
__device__ __forceinline__ float wq16_track2(const float rmax, const float a, const float b) {
  return __builtin_elementwise_maximum(rmax, __builtin_fabsf(a) + __builtin_fabsf(b));
}
__device__ __forceinline__ float wq16_track4(const float rmax, const float a, const float b, const float c, const float d) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(rmax, __builtin_fabsf(a) + __builtin_fabsf(b)), __builtin_fabsf(c) + __builtin_fabsf(d));
}
__device__ __forceinline__ void wq16_range_report(int* range_word, const float rmax) {
  if (!(rmax <= 65504.f)) __hip_atomic_store(range_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wq16_fresh(const unsigned long long v, const unsigned tb) { return (((unsigned)(v >> 32)) & 1u) == tb; }

}  // namespace mb
