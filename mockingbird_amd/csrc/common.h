// Shared host/device helpers for libmbhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/mbhip.h"

namespace mb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32 -> fp16 hi + fp16 residual scaled by 2^11 (the operand split of every error-compensated fp16 MFMA path: conv1d.hip,
// resblock_stage_f32.hip), two values at a time: v_cvt_pk_f16_f32, v_pk_add_f32, v_pk_mul_f32 -- 4 instructions per value instead
// of 6-7.  The value is clamped to fp16's range first (a finite hi leaves a residual of at most half an fp16 ulp, <= 16, x 2^11 <=
// 32768: the low half needs no clamp); results are bit for bit those of the scalar form (same roundings in the same order).
typedef _Float16 mb_h2 __attribute__((ext_vector_type(2)));
typedef float mb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(const float a, const float b, mb_h2& hi, mb_h2& lo) {
  const mb_f2 v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
  hi = __builtin_convertvector(v, mb_h2);
  lo = __builtin_convertvector((v - __builtin_convertvector(hi, mb_f2)) * 2048.f, mb_h2);
}

// Window loads of the support waves go through a buffer descriptor per (item, chunk): rows before the item (negative offsets wrap past
// the end) and rows beyond its valid length come back as zeros from the range check -- no predicate, no select.  (A select on the
// loaded value inside the REQUEST made the compiler wait for every load where it was issued: the "one interval ahead" pipeline of the
// support waves never had a load in flight across a barrier -- 7-10 k cycles per interval at 64 channels against 1.7 k of MFMAs.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tm_rsrc(const void* base, long long bytes) {  // wave-uniform inputs
  const unsigned long long q = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
  const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(bytes < 0 ? 0 : bytes > 0xffffffffll ? 0xffffffffll : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
__device__ __forceinline__ f32x4 tm_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define MB_HIP(expr)                                                     \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) return ::mb::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define MB_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ::mb::set_error(__VA_ARGS__);      \
      return MB_EINVAL;                  \
    }                                    \
  } while (0)

// Diagnostics, A/B knobs and test hooks share ONE environment variable so that they cannot be mistaken for supported switches:
//   MBHIP_DIAG="key=value,key,key=value"   (read at call time; a bare key reads as "1")
// diag_str returns false when the key is absent; diag_int returns `absent` then.  The keys are listed in DESIGN.md ("Run-time switches").
bool diag_str(const char* key, std::string* value);
int diag_int(const char* key, int absent = 0);
// value of a documented path switch (plain MBHIP_* variable) as an int; `absent` when unset
int env_int(const char* name, int absent);

// Side streams of the library: ONE process-wide pool of three non-blocking streams per device, created on first use and never
// destroyed.  HIP maps streams onto 4 hardware queues by default; streams beyond that SHARE a queue, and an event wait queued for one of
// them then holds back every stream behind it in that queue (a WaveRNN handle used to create 8 streams, every loop handle one, a GAN
// handle three: a GAN forward late in bench.py's process ran 9.4 instead of 8.4 ms).  Handles borrow from the pool: the resident loops
// run on stream 0, the parallel ResBlock chains of a GAN stage on 0..2; two handles on one pool stream are merely ordered.
int pool_stream(int i, hipStream_t* out);
constexpr int POOL_STREAMS = 3;

// the device word MBHIP_CONV_RANGE_CHECK=1 counts out-of-range staged values in (conv1d.hip; null when the check is off)
unsigned* conv_range_word();

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Simple bump allocator over a caller-provided workspace.
struct Arena {
  char* base;
  size_t cap, off;
  Arena(void* p, size_t c) : base((char*)p), cap(c), off(0) {}
  template <typename T>
  T* take(size_t n) {
    size_t o = align_up(off, 256);
    off = o + n * sizeof(T);
    return (T*)(base ? base + o : nullptr);
  }
  bool ok() const { return off <= cap; }
};

// Device buffer owned by a handle.
struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
  int upload(const float* h, size_t count);
  int alloc(size_t count);
  void release();
};

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Philox4x32-10 counter RNG (Salmon et al. 2011), used for the on-device
// sampling / dropout streams when no noise tensor is injected.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform strictly inside (0, 1): the centres of 2^23 equal cells (exact in fp32), so that -log(u) is a positive finite
// Exp(1) draw (torch's exponential_ never returns 0 either) and log(-log(u)) is finite
__device__ __forceinline__ float u32_to_unit(uint32_t x) {
  return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

}  // namespace mb
