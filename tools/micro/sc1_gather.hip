// Micro-benchmark behind DESIGN section 9: what an all-to-all gather costs when it has to bypass the (per-XCD, not mutually
// coherent) L2s.  NWG workgroups of 512 threads each read the SAME `bytes` region (an exchange vector every consumer needs),
// once with plain 16-byte loads (L2 hits after the first toucher of each XCD: what a launch boundary gives) and once with
// 8-byte relaxed agent-scope atomic loads (global_load_dwordx2 sc1: what a tagged-granule sweep issues).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/sc1_gather.hip -o build_variants/sc1_gather ; run: build_variants/sc1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void gather_plain(const uint4* __restrict__ src, size_t n16, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = threadIdx.x; i < n16; i += 512) { const uint4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 0x1234567ull) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(512) void gather_sc1(const unsigned long long* src, size_t n8, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = threadIdx.x; i < n8; i += 512 * 4) {  // four granules of a lane in flight per round, as the sweeps do
    unsigned long long v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const size_t j = i + (size_t)k * 512; v[k] = j < n8 ? __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k];
  }
  if (acc == 0x1234567ull) sink[blockIdx.x] = acc;
}
__global__ void fill(unsigned long long* p, size_t n, unsigned long long tag) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __hip_atomic_store(p + i, (tag << 32) | i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  const size_t maxb = 1 << 20;
  unsigned long long *buf, *sink;
  CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&sink, 4096 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%8s %5s %12s %12s   (us per gather, every workgroup reads the whole region; region rewritten before each gather)\n", "bytes", "wgs", "plain16", "sc1_8");
  for (int nwg : {32, 64, 256}) for (size_t bytes : {(size_t)16 << 10, (size_t)64 << 10, (size_t)256 << 10}) {
    float t[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {
      const int reps = 50;
      float tot = 0;
      for (int r = 0; r < reps + 5; ++r) {
        fill<<<64, 256>>>(buf, bytes / 8, (unsigned long long)r + 1);
        CK(hipEventRecord(e0));
        if (mode == 0) gather_plain<<<nwg, 512>>>((const uint4*)buf, bytes / 16, sink);
        else gather_sc1<<<nwg, 512>>>(buf, bytes / 8, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 5) tot += ms;
      }
      t[mode] = tot / reps * 1e3f;
    }
    printf("%8zu %5d %12.2f %12.2f\n", bytes, nwg, t[0], t[1]);
  }
  return 0;
}
