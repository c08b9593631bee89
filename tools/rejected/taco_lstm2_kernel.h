// MEASURED AND REJECTED in round 3, removed from libmbhip.so in round 4: both residual LSTM cells of a Tacotron
// decoder iteration in one launch with a fence + counter hand-off.  100.7 us per iteration against 48.7 with two launches
// (profiles/r03_taco_lstm_merged_ab.txt).  Kept as a record of the experiment; it was a member of csrc/taco_fast.h (uses its
// TfLstmK / fm_gemm / TF_* definitions) selected by MBHIP_TACO_LSTM_MERGED=1, diagnostics bits MBHIP_TACO_L2_DBG.  Not built.
// Both residual LSTM cells of an iteration in ONE launch: workgroups [0, n) are the first cell (exactly taco_lstm_kernel),
// [n, 2n) the second.  A second-cell workgroup requests its 16 x 1024 weight tile, bias, hidden-half quad and cell state, and
// only then waits for the first cell's output: its weight stream overlaps the first cell instead of starting behind a launch
// boundary.  Hand-off: every first-cell epilogue wave stores its outputs, executes an agent-scope release fence (the writes
// leave its XCD's L2) and bumps the counter of its blockIdx.y; a second-cell workgroup polls that counter with one lane
// (monotonic over the utterance: target = (iteration + 1) x producers), then every thread executes an agent-scope acquire fence
// before the activation loads.  No co-residency requirement: workgroups are dispatched in index order and a waiter only ever
// waits for lower indices, which are running or done.  Every wait has a wall-clock bail-out (TF_LOST; the host reports it).
// Same arithmetic, same summation order as the two launches: bit-identical (tests/test_tacotron_gpu.py).
// MEASURED AND LEFT OFF (MBHIP_TACO_LSTM_MERGED=1 selects it; tools/taco_lstm_ab.py, B = 32): 100.7 us per iteration against 48.7
// with two launches.  Without the acquire fence (buffer_inv sc1 in 2048 waves) 86.7, without the release fence (buffer_wbl2 sc1
// in 512 waves) 78.0, without both 63.5 (wrong results): an L2 write-back / invalidate per wave costs far more than the launch
// boundary it replaces, and even the fence-free skeleton loses 15 us to 256 pollers on one line and to the second cell's weight
// stream taking bandwidth from the first, which is the one on the critical path.  The granule form (wavernn_pipe.h) does not
// apply either: every second-cell workgroup needs all 32 x 1024 outputs, 256 KB of granules x 256 workgroups past the L2.
// What a launch boundary gives for free on this chip -- a coherent L2 -- is what an in-launch hand-off of a LARGE operand pays for.
struct TfLstm2K { TfLstmK l1, l2; int* sync; int it_off; int dbg; };  // dbg (MBHIP_TACO_L2_DBG, timing only): 1 = no acquire fence, 2 = no release fence
struct TfWaitCell {
  int* counter; int target; int* lost; int dbg;
  __device__ __forceinline__ void operator()() const {
    if (threadIdx.x == 0) {
      unsigned long long t0 = 0;
      int tries = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if ((++tries & 1023) == 0) {
          const unsigned long long now = (unsigned long long)wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > WP_TIMEOUT_TICKS) { atomicExch(lost, 1); break; }
          if (__hip_atomic_load(lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
      }
    }
    __syncthreads();
    if (!(dbg & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
};
template <int NT>
__global__ __launch_bounds__(512, 4) void taco_lstm2_kernel(TfLstm2K k) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int n_mt = gridDim.x >> 1;
  const bool second = (int)blockIdx.x >= n_mt;
  const TfLstmK& a = second ? k.l2 : k.l1;
  const int mt = second ? blockIdx.x - n_mt : blockIdx.x, nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE], it = a.flags[TF_ITER] + k.it_off;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int ntE = (nt0 + (wv < NT ? wv : 0) < a.nta) ? nt0 + (wv < NT ? wv : 0) : a.nta - 1;
  const float4 bq = a.b4[mt * 4 + du];
  const float4 hq = a.hpre[((size_t)mt * a.nta + ntE) * 64 + lane];
  float* cp = a.c + ((size_t)mt * a.nta + ntE) * 64 + lane;
  const float cprev = *cp;
  const size_t fo = ((size_t)(mt >> 2) * a.nta + ntE) * 256 + (mt & 3) * 64 + i * 4 + du;
  float sx[4], sh[4];
  float xr;
  const bool pick = mt == 100 && blockIdx.y == 0;
  tf_mark(a.trace, a.trace_slot, 0, pick);
  if (!second) {
    xr = a.x[fo];
    if (!fm_gemm<NT, 8, 8, 4, 1>(a.w, mt, a.x, a.x, a.nta, nt0, red, sx, sh, a.trace, a.trace_slot, pick)) return;
  } else {
    if (done) return;  // (uniform over the launch: nobody publishes, nobody waits)
    const int producers = n_mt * min(NT, a.nta - nt0);
    const TfWaitCell wait{k.sync + blockIdx.y, (it + 1) * producers, const_cast<int*>(a.flags) + TF_LOST, k.dbg};
    if (!fm_gemm<NT, 8, 8, 4, 1, TfWaitCell>(a.w, mt, a.x, a.x, a.nta, nt0, red, sx, sh, a.trace, a.trace_slot, pick, wait)) return;
    xr = a.x[fo];
  }
  if (nt0 + wv >= a.nta || done) return;
  // torch LSTMCell, gate order (i, f, g, o)
  const float gi = sigmoidf_((sx[0] + hq.x) + bq.x);
  const float gf = sigmoidf_((sx[1] + hq.y) + bq.y);
  const float gg = tanhf((sx[2] + hq.z) + bq.z);
  const float go = sigmoidf_((sx[3] + hq.w) + bq.w);
  const float cy = gf * cprev + gi * gg;
  const float hy = go * tanhf(cy);
  *cp = cy;
  a.h_out[fo] = hy;
  a.x_out[fo] = xr + hy;
  if (!second) {
    if (!(k.dbg & 2)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(k.sync + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  tf_mark_end(a.trace, a.trace_slot, 4, pick);
}

