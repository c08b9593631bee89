// What shader clock does the chip sustain while every SIMD issues fp32 MFMAs back to back?
// Each wave runs ITERS x 48 independent v_mfma_f32_16x16x4_f32 (12 accumulators, the mix of the wide-batch
// recurrent GEMM, rnn_ts2_body.h) and stamps the shader clock (clock64, s_memtime) and the constant 100 MHz
// wall clock (wall_clock64, s_memrealtime) around the loop.  cycles / wall time = sustained clock;
// cycles / MFMA = issue cost (32 when nothing stalls).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_clock.hip -o mfma_clock ; run: ./mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* stamps, int iters, float a0, float b0) {
  f32x4 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  const float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = w1 - w0;
  }
}

static void run(int blocks, int iters, int launches) {
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&st, (size_t)blocks * 4 * 2 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.f, 2.f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < blocks * 4; ++w) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
  cyc /= blocks * 4; wall /= blocks * 4;
  const double mfmas = 48.0 * iters;
  printf("blocks %4d iters %5d launches %4d: %.1f cycles/MFMA, in-kernel %.2f us, clock %.2f GHz, launch-to-launch %.2f us, %.1f TFLOP/s over the launch train\n",
         blocks, iters, launches, cyc / mfmas, wall / 100.0, cyc / (wall * 10.0), ms * 1e3 / launches,
         (double)blocks * 4 * mfmas * 2048.0 * launches / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(st);
}

int main() {
  run(256, 16, 200);     // the wide-batch GEMM's shape: 768 MFMAs per wave, one wave per SIMD
  run(256, 16, 2000);
  run(256, 1600, 20);    // ~1 ms of sustained load
  run(64, 16, 200);      // a quarter of the chip
  run(512, 16, 200);     // two waves per SIMD
  return 0;
}
