// Is the instruction cache cold at every kernel launch?  A wave runs the SAME long straight-line
// code twice inside one launch (pass 0, pass 1) and stamps shader clocks around each pass.
// Build: hipcc -O3 --offload-arch=gfx950 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int N>
__device__ __forceinline__ float chain(float x, float y) {
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, y, 1.0f + i);   // distinct immediates: no rerolling
  return x;
}

template <int N>
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long* stamps, float y) {
  float x = threadIdx.x;
  unsigned long long t[3];
  t[0] = clock64();
  for (int pass = 0; pass < 2; ++pass) {
    x = chain<N>(x, y);
    asm volatile("" ::: "memory");
    t[pass + 1] = clock64();
  }
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 4 + 0] = t[1] - t[0];
    stamps[blockIdx.x * 4 + 1] = t[2] - t[1];
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

__global__ void other(float* p) { p[threadIdx.x] += 1.f; }

template <int N>
void run(const char* name, int blocks, bool interleave) {
  float* out; unsigned long long* st; float* q;
  hipMalloc(&out, blocks * 64 * 4); hipMalloc(&st, blocks * 4 * 8); hipMalloc(&q, 4096);
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  for (int i = 0; i < 20; ++i) {
    if (interleave) hipLaunchKernelGGL(other, dim3(64), dim3(256), 0, s, q);
    hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(64), 0, s, out, st, 1.0001f);
  }
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), st, blocks * 4 * 8, hipMemcpyDeviceToHost);
  std::vector<double> p0, p1;
  for (int b = 0; b < blocks; ++b) { p0.push_back((double)h[b * 4]); p1.push_back((double)h[b * 4 + 1]); }
  std::sort(p0.begin(), p0.end()); std::sort(p1.begin(), p1.end());
  printf("%-28s N=%5d blocks=%4d  pass0 median %8.0f cyc (%.2f us)  pass1 median %8.0f cyc (%.2f us)  max0 %.0f\n", name, N, blocks,
         p0[blocks / 2], p0[blocks / 2] / 2400.0, p1[blocks / 2], p1[blocks / 2] / 2400.0, p0.back());
}

int main() {
  run<256>("back-to-back same kernel", 1, false);
  run<256>("interleaved other kernel", 1, true);
  run<2048>("back-to-back same kernel", 1, false);
  run<2048>("interleaved other kernel", 1, true);
  run<2048>("interleaved, 256 blocks", 256, true);
  run<8192>("interleaved other kernel", 1, true);
  run<8192>("interleaved, 256 blocks", 256, true);
  return 0;
}
