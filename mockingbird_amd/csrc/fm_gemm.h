// Fragment-major (FM) skinny GEMM core shared by the Tacotron decoder loop (taco_fast.h) and the WaveRNN sample loop
// (wavernn_fast.h): layouts, the 8-wave K-split MFMA tile product with every fragment in flight, diagnostics marks.
//   FM activations: element (column n, feature k) of a K-vector set lives at
//       ((k/16 * NTA + n/16) * 64 + (k/4 % 4) * 16 + n % 16) * 4 + k % 4          (NTA = column tiles)
//     so the B fragment of a (k-block, column tile) is ONE contiguous 1 KB read -- the same shape as the packed A
//     fragments (rnn.h pack_rowtile) -- and the LINEAR / GRU / LSTM epilogues, whose lanes own (row quad | unit, column),
//     store contiguous 1 KB / 256 B pieces of the next launch's operand.
//   CM4 / CM1 ("cell-major") per-(unit, column) quads / scalars at (unit/4 * NTA + n/16) * 64 + lane: what one epilogue
//     lane writes is what the consuming epilogue lane reads (gate pre-activations, cell state).
#pragma once
#include "rnn_body.h"
#include "pair_granule.h"

namespace mb {

__host__ __device__ __forceinline__ size_t fm_floats(int K, int nta) { return (size_t)(K / 16) * nta * 256; }
__host__ __device__ __forceinline__ size_t cm_items(int units, int nta) { return (size_t)((units + 3) / 4) * nta * 64; }
// float index of element (column n, feature k) in an FM buffer
__host__ __device__ __forceinline__ size_t fm_index(int nta, int n, int k) {
  return ((size_t)(k >> 4) * nta + (n >> 4)) * 256 + ((k >> 2) & 3) * 64 + (n & 15) * 4 + (k & 3);
}

// diagnostics (MBHIP_DIAG=taco_trace=<file>): shader-clock stamps of one workgroup per kernel, 16 marks per kernel slot;
// mark 14 / 15 = 100 MHz wall clock at kernel start / end (aligns the kernels of an iteration with each other)
enum { TS_FC2 = 0, TS_GRU = 1, TS_LSA = 2, TS_RIN = 3, TS_LSTM1 = 4, TS_LSTM2 = 5, TS_MEL = 6, TS_MEL_FC1 = 7, TS_MEL_STOP = 8, TS_SLOTS = 9,
       TS_WG = 16 * TS_SLOTS, TS_WORDS = TS_WG + 2 * 128 };  // [TS_WG + 2 l], [.. + 1]: query-arrival and end wall clock of attention workgroup l of the fused launch
__device__ __forceinline__ void tf_mark(unsigned long long* tr, int slot, int k, bool pick) {
  if (tr && pick && threadIdx.x == 0) {
    tr[slot * 16 + k] = (unsigned long long)clock64();
    if (k == 0) tr[slot * 16 + 14] = (unsigned long long)wall_clock64();
  }
}
__device__ __forceinline__ void tf_mark_end(unsigned long long* tr, int slot, int k, bool pick) {
  if (tr && pick && (threadIdx.x & 63) == 0 && threadIdx.x < 128) {  // epilogue waves: the later one wins
    atomicMax(tr + slot * 16 + k, (unsigned long long)clock64());
    atomicMax(tr + slot * 16 + 15, (unsigned long long)wall_clock64());
  }
}

// row sum over the 16 lanes of a DPP row (every lane of the row receives it): 4 VALU ops, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// wave-wide sum / max on DPP row operations and four v_readlane (every lane receives the result): ~15 VALU ops where the
// __shfl_xor butterfly (common.h wave_sum) makes six trips through the LDS crossbar -- the attention's softmax is ONE wave on the chain
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
__device__ __forceinline__ float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float wave64_sum(float v) {
  v = row16_sum(v);
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave64_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// Skinny GEMM core: one 16-row weight tile (mt) x NT column tiles, K = 8 * PW k-blocks split over the 8 waves
// (wave w owns k-blocks w, w+8, ...), first PS steps from seg0, the rest from seg1 (both FM).  NPART = 2 keeps the
// seg1 sums apart (GRU hidden part).  All loads are issued before the first MFMA.  Returns true for the NT
// epilogue waves (wave w finishes column tile nt0 + w) with the reduced sums in sx / sh (k order: waves 0..7).
// WaitB: called between the weight loads and the activation loads (taco_lstm2_kernel: the second cell's workgroups have their
// weights in flight while they wait for the first cell's output); the default keeps the interleaved issue order.
struct FmNoWait { __device__ __forceinline__ void operator()() const {} };
template <int NT, int PW, int PS, int RL, int NPART, class WaitB = FmNoWait>
__device__ __forceinline__ bool fm_gemm(const float* __restrict__ w, const int mt, const float* __restrict__ seg0,
                                        const float* __restrict__ seg1, const int nta, const int nt0, float* red,
                                        float (&sx)[4], float (&sh)[4], unsigned long long* tr = nullptr, int slot = 0,
                                        bool pick = false, WaitB wait_b = WaitB()) {
  constexpr int BLK = 4 * RL * 16, NKB = 8 * PW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;  // dead 4th GRU row re-reads row 2 (never used)
  const float* wl = w + (size_t)mt * NKB * BLK + ((u * RL + tau) * 4 + kq) * 4;
  int ntc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ntc[nt] = (nt0 + nt < nta) ? nt0 + nt : nta - 1;  // odd tile count: duplicate, never stored
  float4 a[PW], b[PW][NT];
  constexpr bool SPLIT = !__is_same(WaitB, FmNoWait);
  if (SPLIT) {
#pragma unroll
    for (int p = 0; p < PW; ++p) a[p] = *reinterpret_cast<const float4*>(wl + (size_t)(wave + 8 * p) * BLK);
    __builtin_amdgcn_sched_barrier(0);
    wait_b();
  }
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int kb = wave + 8 * p;
    if (!SPLIT) a[p] = *reinterpret_cast<const float4*>(wl + (size_t)kb * BLK);
    const float4* sp = reinterpret_cast<const float4*>(p < PS ? seg0 : seg1);
    const int kl = p < PS ? kb : kb - 8 * PS;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[p][nt] = sp[((size_t)kl * nta + ntc[nt]) * 64 + lane];
  }
  // the machine scheduler would otherwise sink the loads next to their MFMAs to save registers (50 VGPRs, two or
  // three k-blocks in flight per wave); the whole point is to have every fragment of the wave in flight at once
  __builtin_amdgcn_sched_barrier(0);
  tf_mark(tr, slot, 1, pick);  // every load issued
  f32x4 accX[NT], accH[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { accX[nt] = {0.f, 0.f, 0.f, 0.f}; accH[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const bool hpart = NPART == 2 && p >= PS;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float av = c == 0 ? a[p].x : c == 1 ? a[p].y : c == 2 ? a[p].z : a[p].w;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = c == 0 ? b[p][nt].x : c == 1 ? b[p][nt].y : c == 2 ? b[p][nt].z : b[p][nt].w;
        if (hpart) accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accH[nt], 0, 0, 0);
        else accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accX[nt], 0, 0, 0);
      }
    }
  }
  float4* red4 = reinterpret_cast<float4*>(red);  // [8][NT][NPART][64]
  tf_mark(tr, slot, 2, pick);  // MFMAs of wave 0 issued (its last fragment has arrived)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    red4[((wave * NT + nt) * NPART + 0) * 64 + lane] = make_float4(accX[nt][0], accX[nt][1], accX[nt][2], accX[nt][3]);
    if (NPART == 2) red4[((wave * NT + nt) * NPART + 1) * 64 + lane] = make_float4(accH[nt][0], accH[nt][1], accH[nt][2], accH[nt][3]);
  }
  __syncthreads();
  tf_mark(tr, slot, 3, pick);  // all eight waves done
  if (wave >= NT) return false;
#pragma unroll
  for (int g = 0; g < 4; ++g) { sx[g] = 0.f; sh[g] = 0.f; }
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const float4 v = red4[((w8 * NT + wave) * NPART + 0) * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
    if (NPART == 2) {
      const float4 h = red4[((w8 * NT + wave) * NPART + 1) * 64 + lane];
      sh[0] += h.x; sh[1] += h.y; sh[2] += h.z; sh[3] += h.w;
    }
  }
  return true;
}

// The same tile product on the fp16 matrix pipe, fp32-grade (conv1d.hip's error-compensated scheme, as in wavernn_pipe16.h):
// w 2^s = wh + wl (host-split image), x = xh + 2^-11 xl' (split in registers from the fp32 FM fragments), three
// v_mfma_f32_16x16x32_f16 per 32 k in two accumulator chains -- 24 matrix instructions per wave where the fp32 form issues 64 four-pass
// ones (2048 of a launch's ~11 000 cycles per wave, twice that per SIMD).  A 32-k step of wave w pairs k-blocks w + 16 s and
// w + 16 s + 8: lane (i, kq) holds features 4 kq .. + 3 of both -- the matrix instruction does not care which k a lane's eight values
// are as long as A and B agree, so the B operand stays the FM float4 pair it is.  Image (pack_rowtile16, tacotron.hip):
// [tile][g = 8 s + w][hi | lo][64 lanes] x 8 halves, K padded to a multiple of 256 with zero columns (the B loads of a padded
// k-block re-read a valid one).  PS2 = steps taken from seg0 (k-blocks < kb0), the rest from seg1; NPART = 2 keeps the seg1 sums apart.
// A value beyond fp16's range (|x| > 65504) turns the sums into inf / NaN: the callers test their sums and raise flags[TF_LOST],
// the host reruns the call on the exact fp32 loop.
typedef float fm_f8 __attribute__((ext_vector_type(8)));
template <int NT, int PW2, int PS2, int NPART>
__device__ __forceinline__ bool fm_gemm16(const uint4* __restrict__ w16, const int mt, const float* __restrict__ seg0, const int kb0,
                                          const float* __restrict__ seg1, const int kb1, const int nta, const int nt0, float* red,
                                          const float unscale, float (&sx)[4], float (&sh)[4], unsigned long long* tr = nullptr, int slot = 0,
                                          bool pick = false) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint4* wl = w16 + (size_t)mt * (PW2 * 8) * 2 * 64 + lane;
  int ntc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ntc[nt] = (nt0 + nt < nta) ? nt0 + nt : nta - 1;
  uint4 ah[PW2], al[PW2];
  float4 b0[PW2][NT], b1[PW2][NT];
#pragma unroll
  for (int s = 0; s < PW2; ++s) {
    const int g = s * 8 + wave;
    ah[s] = wl[(size_t)(g * 2) * 64];
    al[s] = wl[(size_t)(g * 2 + 1) * 64];
    const int kbA = wave + 16 * s, kbB = kbA + 8;
    const float4* sp = reinterpret_cast<const float4*>(s < PS2 ? seg0 : seg1);
    const int nk = s < PS2 ? kb0 : kb1, base = s < PS2 ? 0 : PS2 * 16;
    const int ka = kbA - base, kb = (kbB - base < nk) ? kbB - base : ka;  // (padded k-block: its weights are zero)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      b0[s][nt] = sp[((size_t)ka * nta + ntc[nt]) * 64 + lane];
      b1[s][nt] = sp[((size_t)kb * nta + ntc[nt]) * 64 + lane];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  tf_mark(tr, slot, 1, pick);
  f32x4 accX[NT], aclX[NT], accH[NT], aclH[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    accX[nt] = {0.f, 0.f, 0.f, 0.f}; aclX[nt] = {0.f, 0.f, 0.f, 0.f};
    accH[nt] = {0.f, 0.f, 0.f, 0.f}; aclH[nt] = {0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int s = 0; s < PW2; ++s) {
    const bool hpart = NPART == 2 && s >= PS2;
    const wh16x8 wh = __builtin_bit_cast(wh16x8, ah[s]), wlo = __builtin_bit_cast(wh16x8, al[s]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const fm_f8 v = {b0[s][nt].x, b0[s][nt].y, b0[s][nt].z, b0[s][nt].w, b1[s][nt].x, b1[s][nt].y, b1[s][nt].z, b1[s][nt].w};
      const wh16x8 xh = __builtin_convertvector(v, wh16x8);
      const fm_f8 d = (v - __builtin_convertvector(xh, fm_f8)) * WQ16_LO_SCALE;
      const wh16x8 xl = __builtin_convertvector(d, wh16x8);
      if (hpart) {
        accH[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh, accH[nt], 0, 0, 0);
        aclH[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, aclH[nt], 0, 0, 0);
        accH[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, accH[nt], 0, 0, 0);
      } else {
        accX[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh, accX[nt], 0, 0, 0);
        aclX[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, aclX[nt], 0, 0, 0);
        accX[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, accX[nt], 0, 0, 0);
      }
    }
  }
  float4* red4 = reinterpret_cast<float4*>(red);  // [8][NT][NPART][64]
  tf_mark(tr, slot, 2, pick);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 x = (accX[nt] + aclX[nt] * WQ16_LO_UNSCALE) * unscale;
    red4[((wave * NT + nt) * NPART + 0) * 64 + lane] = make_float4(x[0], x[1], x[2], x[3]);
    if (NPART == 2) {
      const f32x4 h = (accH[nt] + aclH[nt] * WQ16_LO_UNSCALE) * unscale;
      red4[((wave * NT + nt) * NPART + 1) * 64 + lane] = make_float4(h[0], h[1], h[2], h[3]);
    }
  }
  __syncthreads();
  tf_mark(tr, slot, 3, pick);
  if (wave >= NT) return false;
#pragma unroll
  for (int g = 0; g < 4; ++g) { sx[g] = 0.f; sh[g] = 0.f; }
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const float4 v = red4[((w8 * NT + wave) * NPART + 0) * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
    if (NPART == 2) {
      const float4 h = red4[((w8 * NT + wave) * NPART + 1) * 64 + lane];
      sh[0] += h.x; sh[1] += h.y; sh[2] += h.z; sh[3] += h.w;
    }
  }
  return true;
}
// non-finite sums = an operand left fp16's range (see above): the launch flags the call for the exact loop
__device__ __forceinline__ void fm_range_check(const float (&sx)[4], int* lost) {
  const float m = fmaxf(fmaxf(__builtin_fabsf(sx[0]), __builtin_fabsf(sx[1])), fmaxf(__builtin_fabsf(sx[2]), __builtin_fabsf(sx[3])));
  if (!(m <= 3.0e38f) || !(sx[0] == sx[0]) || !(sx[1] == sx[1]) || !(sx[2] == sx[2]) || !(sx[3] == sx[3])) atomicCAS(lost, 0, 2);  // (2: range event; a timed-out wait's 1 stays)
}

template <int NT, int NPART> struct FmRed { static constexpr int floats = 8 * NT * NPART * 256; };

// flags block in the workspace (ints): [0] done, [1] n_frames, [2] arrival ticket, [3] utterances below the stop
// threshold, [4] iteration index of the first launch of the current graph replay, [6..7] 64-bit dropout seed
enum { TF_DONE = 0, TF_NFRAMES = 1, TF_ARRIVE = 2, TF_NOTREADY = 3, TF_ITER = 4, TF_LOST = 5, TF_SEED = 6 };

// ---- always-on PreNet dropout (pre_net.py:23,26) of a relu'd row quad; same masks / same Philox stream as the
//      general paths (rnn_body.h): Philox(iter, layer, n, row/4); masks [iteration][column][row] per layer ----
struct DropK {
  const float* mask;       // injected keep masks of THIS layer (iteration 0) or null
  long long it_stride;     // floats between the masks of consecutive iterations
  int ld;                  // row length of a mask
  int layer, it_add;       // prenet layer (Philox key); iteration offset (a job of launch `it` may prepare iteration it + 1)
  int it_limit;            // iterations covered by `mask`: the job that prepares iteration it_limit (never run) reads nothing
  unsigned thresh; float scale; int enabled;
};
// the factors alone (1 = keep unscaled): what relu_drop_quad multiplies by, computable before the sums exist
__device__ __forceinline__ void drop_quad_factors(const DropK& d, const int* flags, int it, int n, int row0, float (&f)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) f[r] = 1.f;
  if (!d.enabled) return;
  const int iter = it + d.it_add;
  if (d.mask) {
    if (iter >= d.it_limit) return;
    const float4 m = *reinterpret_cast<const float4*>(d.mask + (long long)iter * d.it_stride + (size_t)n * d.ld + row0);
    f[0] = m.x * d.scale; f[1] = m.y * d.scale; f[2] = m.z * d.scale; f[3] = m.w * d.scale;
  } else {
    const unsigned long long seed = *reinterpret_cast<const unsigned long long*>(flags + TF_SEED);
    uint32_t rr[4];
    philox4x32((uint32_t)iter, (uint32_t)d.layer, (uint32_t)n, (uint32_t)(row0 >> 2), (uint32_t)seed, (uint32_t)(seed >> 32), rr);
#pragma unroll
    for (int r = 0; r < 4; ++r) f[r] = (rr[r] >= d.thresh) ? d.scale : 0.f;
  }
}
__device__ __forceinline__ void relu_drop_quad(const DropK& d, const int* flags, int it, int n, int row0, float (&v)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  if (!d.enabled) return;
  const int iter = it + d.it_add;
  if (d.mask) {
    if (iter >= d.it_limit) return;
    const float4 m = *reinterpret_cast<const float4*>(d.mask + (long long)iter * d.it_stride + (size_t)n * d.ld + row0);
    v[0] *= m.x * d.scale; v[1] *= m.y * d.scale; v[2] *= m.z * d.scale; v[3] *= m.w * d.scale;
  } else {
    const unsigned long long seed = *reinterpret_cast<const unsigned long long*>(flags + TF_SEED);
    uint32_t rr[4];
    philox4x32((uint32_t)iter, (uint32_t)d.layer, (uint32_t)n, (uint32_t)(row0 >> 2), (uint32_t)seed, (uint32_t)(seed >> 32), rr);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= (rr[r] >= d.thresh) ? d.scale : 0.f;
  }
}


}  // namespace mb
