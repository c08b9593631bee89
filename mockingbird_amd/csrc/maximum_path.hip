// monotonic_align.maximum_path on the device.
// Reference: monotonic_align/core.pyx:7-33 (maximum_path_each), :38-42 (maximum_path_c).
//
// One workgroup per batch item.  The forward DP walks rows y sequentially
// (row y depends on row y-1 only); lanes cover the banded x range, the previous
// row lives in LDS (double buffered) so each cell costs one global read and one
// global write of `value` (8 B/cell, the HBM-scan roofline of SURVEY.md section 8d).
// The backtrack is inherently serial: one lane follows the path through L2.
#include "common.h"

namespace mb {

static constexpr float MAX_NEG = -1e9f;

__global__ __launch_bounds__(256) void maximum_path_kernel(int* __restrict__ paths, float* __restrict__ values,
                                                           const int* __restrict__ t_ys,
                                                           const int* __restrict__ t_xs, int t_t, int t_s) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // [2][t_s]
  const int b = blockIdx.x;
  const int t_y = t_ys[b], t_x = t_xs[b];
  float* val = values + (size_t)b * t_t * t_s;
  int* path = paths + (size_t)b * t_t * t_s;
  for (int y = 0; y < t_y; ++y) {
    float* cur = rowbuf + (y & 1) * t_s;
    const float* prev = rowbuf + ((y & 1) ^ 1) * t_s;
    const int lo = max(0, t_x + y - t_y), hi = min(t_x, y + 1);
    for (int x = lo + (int)threadIdx.x; x < hi; x += blockDim.x) {
      const float v_cur = (x == y) ? MAX_NEG : prev[x];
      const float v_prev = (x == 0) ? (y == 0 ? 0.f : MAX_NEG) : prev[x - 1];
      const float v = val[(size_t)y * t_s + x] + fmaxf(v_prev, v_cur);
      val[(size_t)y * t_s + x] = v;
      cur[x] = v;
    }
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
      if (index >= 0 && index < t_s) path[(size_t)y * t_s + index] = 1;
      if (index != 0 && y > 0) {
        // value[y-1] as stored (band-updated or untouched input), read past L1
        const float a = __hip_atomic_load(val + (size_t)(y - 1) * t_s + index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float c = __hip_atomic_load(val + (size_t)(y - 1) * t_s + index - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (index == y || a < c) index -= 1;
      }
    }
  }
}

}  // namespace mb

using namespace mb;

extern "C" int mb_maximum_path(int32_t* d_paths, float* d_values, const int32_t* d_t_ys,
                               const int32_t* d_t_xs, int b, int t_t, int t_s, mb_stream_t stream) {
  MB_REQUIRE(b >= 0 && t_t >= 0 && t_s >= 0, "maximum_path: negative shape");
  if (b == 0 || t_t == 0 || t_s == 0) return MB_OK;
  MB_REQUIRE(d_paths && d_values && d_t_ys && d_t_xs, "maximum_path: null pointer");
  const size_t lds = (size_t)2 * t_s * sizeof(float);
  MB_REQUIRE(lds <= 64 * 1024, "maximum_path: t_s=%d too large for the LDS row buffer", t_s);
  hipLaunchKernelGGL(maximum_path_kernel, dim3(b), dim3(256), lds, (hipStream_t)stream, d_paths, d_values,
                     d_t_ys, d_t_xs, t_t, t_s);
  MB_HIP(hipGetLastError());
  return MB_OK;
}
