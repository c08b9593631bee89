// ppg2mel decoder loop at batch 1 (Decoder.inference, models/ppg2mel/rnn_decoder_mol.py:267-316; attend :187-198,
// decode :200-209, DecoderPrenet :10-22; utils/mol_attention.py:67-122) as ONE resident launch: VERDICT round 2, item 5.
//
// At one utterance the 6-launch step of ppg_fast.h is pure latency (6 x 4.1 us for 19 MB of weights and ~10 MFLOP).
// Here 217 workgroups stay resident for the whole utterance; every weight lives in the LDS (or, for prenet.1, the
// registers) of the workgroup that owns its rows from the first step on (19.1 MB = 75 KB per compute unit), and the
// layers hand their output vectors to each other as 8-byte {value, step tag} granules (granule.h).
//
//   role  workgroups  rows                         on the chain                                  in its shadow
//   ATT   64          8 units x 4 gates            p0 -> prenet.1 (redundantly, from registers)  W_hh . att_h, W_ctx . ctx for the
//                                                  -> W_x . p1 + parts -> LSTMCell -> att_h      next step (lane-private partial sums)
//   Q0    8           32 query rows                att_h -> q = relu(query_layer.0)
//   MOL   1           (memory rows in registers)   q -> mixture parameters -> alpha -> context
//   DEC   128         4 units x 4 gates            att_h part early, ctx part + LSTMCell -> h    W_hh . h for the next step (waves 4-7)
//   OUT   16          stop | 16 prenet.0' | mels   ctx part early, h part -> frames, p0(s + 1)   dropout factors of step s + 1
//
// Five hand-offs per step (p0 -> att_h -> q -> ctx -> h -> p0) instead of six launch boundaries.  A product is a
// lane-private fmaf chain over a 1/16 slice of K with the weights read as conflict-free ds_read_b128, the 16 slices of a
// row sit in 16 adjacent lanes and are summed with four DPP adds: a wave owns the four gate rows of one LSTM unit and
// finishes the cell without a barrier.  The summation order differs from ppg_fast.h's MFMA chains: the results agree to
// fp32 rounding (tests compare both with the oracle at the same tolerance), not bit for bit.
//
// Double-buffered by tag parity; the argument of wavernn_persist.h holds edge by edge because every workgroup publishes
// something the chain needs (the OUT workgroups each own 16 rows of the next prenet input next to their mel rows).
// Stop rule (:301-305): every OUT workgroup holds the stop row; on a stop, their first waves hold back p0(s + 1), so
// no workgroup can start step s + 1, and workgroup 0 of the role raises the abort word to 2 = "finished" -- the same
// word a lost hand-off raises to 1 (then the host runs the launch chain instead).
#pragma once
#include "ppg_fast.h"
#include "granule.h"

namespace mb {

constexpr int PR_ATT = 64, PR_DEC = 128, PR_Q0 = 8, PR_OUT = 16;
constexpr int PR_G_DEC = PR_ATT, PR_G_Q0 = PR_ATT + PR_DEC, PR_G_OUT = PR_G_Q0 + PR_Q0, PR_G_MOL = PR_G_OUT + PR_OUT;
constexpr int PR_WGS = PR_G_MOL + 1;  // 217
// exchange area, in granules, per parity (one utterance: feature k of a vector at vec + k)
enum { PRX_P0 = 0, PRX_AH = 256, PRX_Q = 768, PRX_CTX = 1024, PRX_DH = 1280, PRX_PER = 1792 };
inline size_t pr_exchange_bytes() { return (size_t)2 * PRX_PER * 8 + 256 + 8192; }  // + abort word + diagnostics marks
// per-workgroup LDS images (floats), packed on the host: float4 chunk c of thread t at (c * threads + t) * 4
constexpr int PR_IMG_ATT = 14 * 512 * 4;        // W_ih[:, prenet] 2 chunks | W_ih[:, context] 4 | W_hh 8      (K = 128 | 256 | 512)
constexpr int PR_IMG_DEC = (12 + 8) * 256 * 4;  // waves 0-3: W_ih 12 chunks (att_h 8 | context 4); waves 4-7: W_hh 8
constexpr int PR_IMG_Q0 = 8 * 512 * 4;
constexpr int PR_IMG_OUT = 12 * 512 * 4;        // h 8 chunks | context 4
constexpr int PR_IMG_W1 = 16 * 512 * 4;         // prenet.1 [128][256] in the registers of every ATT workgroup
constexpr int PR_LDS_X = 2048;                  // vectors + small state behind the weights
constexpr size_t PR_LDS_BYTES = (size_t)(PR_IMG_ATT + PR_LDS_X) * 4 + 64;
constexpr int PR_T_MAX = 4096;                  // memory rows (MOL workgroup: window in LDS, first 256 rows in registers)

struct PrK {
  const float* img_att; const float* img_w1; const float* img_dec; const float* img_q0; const float* img_out;
  const float4* att_b4; const float4* dec_b4;      // (b_ih + b_hh) as (i, f, g, o) per unit
  const float* q0_b; const float* out_b; const float* fc0_b;  // out_b: projection bias [RM] then the stop bias
  const float* w2; const float* b2;                // query_layer.2 [3M][256], [3M]
  const float* memory;                             // [T][256]
  float* mel_out; float* align_out; float* stop_out;
  const float* drop_mask;                          // injected keep masks (layer 0 [S][256], then layer 1 [S][128]) or null
  unsigned long long* ex; int* abort_word; int* flags;
  unsigned long long seed;
  int T, M, RM, S, min_steps;
  float thr, eps;
  int variant;                                     // A/B switches (MBHIP_DIAG=pr_variant=<bits>): 1 = DEC polls the context as a whole vector, 2 = ATT polls p0 as a whole vector
  unsigned long long* trace;                       // diagnostics (MBHIP_DIAG=pr_trace=<file>): wall-clock marks, steps 100..103
};

__device__ __forceinline__ float pr_dpp(const float v, const int ctrl_sel) {
  // 0: lane ^ 1, 1: lane ^ 2 (quad_perm), 2: row_half_mirror (7 - i inside 8), 3: row_mirror (15 - i inside 16)
  const int x = __float_as_int(v);
  int r;
  if (ctrl_sel == 0) r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);
  else if (ctrl_sel == 1) r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
  else if (ctrl_sel == 2) r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);
  else r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);
  return __int_as_float(r);
}
// sum over the 16 lanes of a DPP row; every lane ends with the same bits (each step adds two partners commutatively)
__device__ __forceinline__ float pr_sum16(float v) {
  v += pr_dpp(v, 0); v += pr_dpp(v, 1); v += pr_dpp(v, 2); v += pr_dpp(v, 3);
  return v;
}
__device__ __forceinline__ float pr_sum4(float v) { v += pr_dpp(v, 0); v += pr_dpp(v, 1); return v; }
__device__ __forceinline__ float pr_lane(const float v, const int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// NCH chunks of a lane's slice: weights w4[c * wstride] (this thread's float4 of chunk c), vector x4[c * 16] (its slice's float4).
// Every LDS read is issued before the first fmaf (left alone the compiler keeps two chunks in flight and the product waits
// for an LDS round trip per chunk), four partial sums keep the dependent chains short.
template <int NCH>
__device__ __forceinline__ float pr_dot(const float4* w4, const int wstride, const float4* x4, const float acc0) {
  float4 a[NCH], x[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { a[c] = w4[c * wstride]; x[c] = x4[c * 16]; }
  __builtin_amdgcn_sched_barrier(0);
  float s0 = acc0, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    s0 = fmaf(a[c].x, x[c].x, s0); s1 = fmaf(a[c].y, x[c].y, s1); s2 = fmaf(a[c].z, x[c].z, s2); s3 = fmaf(a[c].w, x[c].w, s3);
  }
  return (s0 + s1) + (s2 + s3);
}
// LSTM gate functions on the hardware exp2 / rcp (1 ulp each): sigmoid(x) = 1 / (1 + e^-x), tanh(x) = 1 - 2 / (1 + e^2x)
__device__ __forceinline__ float pr_sigmoid(const float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float pr_tanh(const float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
// softplus (F.softplus, beta = 1, threshold = 20) from e = exp(x): log1p(e) = log(u) . e / (u - 1) with u = 1 + e (exact for u == 1)
__device__ __forceinline__ float pr_softplus(const float x, const float e) {
  const float u = 1.f + e;
  const float l = u == 1.f ? e : __logf(u) * (e * __builtin_amdgcn_rcpf(u - 1.f));
  return x > 20.f ? x : l;
}

// One lane watches one granule of a vector (wavernn_persist.h wp_watch: what counts is how few requests sit in this
// unit's memory queue); ends in a barrier; false = the launch is over (finished or aborted: s_flag raised).
// `skip` (thread 0's view) raises the flag without waiting.
template <int SLEEP>
__device__ __forceinline__ bool pr_watch(const unsigned long long* p, const unsigned tag, int* abort_word, int* s_flag,
                                         const bool skip = false, unsigned long long* mk = nullptr) {
  if (threadIdx.x == 0) {
    if (skip) *s_flag = 1;
    else {
      unsigned long long t0 = 0;
      for (int tries = 0; (unsigned)(wp_get(p) >> 32) != tag; ++tries) {
        if ((tries & 7) == 7 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { *s_flag = 1; break; }
        if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) { *s_flag = 1; break; }
        __builtin_amdgcn_s_sleep(SLEEP);
      }
      if (mk) *mk = (unsigned long long)wall_clock64();  // diagnostics: the watched granule has arrived
    }
  }
  __syncthreads();
  return *s_flag == 0;
}
// the first n threads fetch their granule of the vector into xs[tid] (spinning on the tag if it is not there yet); no barrier
__device__ __forceinline__ void pr_sweep(const unsigned long long* vec, const int n, const unsigned tag, float* xs, int* abort_word,
                                         int* s_flag) {
  if ((int)threadIdx.x < n) {
    const unsigned long long* p = vec + threadIdx.x;
    unsigned long long v, t0 = 0;
    for (int tries = 0;; ++tries) {
      v = wp_get(p);
      if ((unsigned)(v >> 32) == tag) break;
      if ((tries & 7) == 7 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { *s_flag = 1; break; }
      if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) { *s_flag = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    xs[threadIdx.x] = __uint_as_float((unsigned)v);
  }
}
__device__ __forceinline__ bool pr_done(int* s_flag) {
  __syncthreads();
  return *s_flag == 0;
}
template <int SLEEP>
__device__ __forceinline__ bool pr_fetch(const unsigned long long* vec, const int n, const unsigned tag, float* xs, int* abort_word,
                                         int* s_flag, const bool skip = false, const int watched = -1,
                                         unsigned long long* mk = nullptr) {
  if (!pr_watch<SLEEP>(vec + (watched >= 0 ? watched : n - 1), tag, abort_word, s_flag, skip, mk)) return false;
  pr_sweep(vec, n, tag, xs, abort_word, s_flag);
  return pr_done(s_flag);
}

// Few consumers of an edge (Q0: 8 workgroups, MOL: 1, OUT: 16): the first wave polls the WHOLE vector (NG granules per
// lane, 64 NG in all) and hands it over through LDS -- one memory round trip instead of watch + sweep.  With 64+ consumer
// workgroups the polls of whole vectors would crowd the producers' stores out of the memory channel (pr_fetch there).
template <int NG>
__device__ __forceinline__ bool pr_poll(const unsigned long long* vec, const unsigned tag, float* xs, int* abort_word, int* s_flag,
                                        const bool skip = false, unsigned long long* mk = nullptr) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (skip) { if (lane == 0) *s_flag = 1; }
    else {
      unsigned long long v[NG], t0 = 0;
      for (int tries = 0;; ++tries) {
#pragma unroll
        for (int i = 0; i < NG; ++i) v[i] = wp_get(vec + i * 64 + lane);
        bool ok = true;
        {  // all tags in one xor / or chain (a chain of && compiled to nested exec-mask branches, wavernn_pipe16.h)
          unsigned stale_ = 0u;
#pragma unroll
          for (int i = 0; i < NG; ++i) stale_ |= (unsigned)(v[i] >> 32) ^ tag;
          ok = ok && stale_ == 0u;
        }
        if (__all(ok)) break;
        if ((tries & 7) == 7 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { if (lane == 0) *s_flag = 1; break; }
        if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) { if (lane == 0) *s_flag = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (mk && lane == 0) *mk = (unsigned long long)wall_clock64();
#pragma unroll
      for (int i = 0; i < NG; ++i) xs[i * 64 + lane] = __uint_as_float((unsigned)v[i]);
    }
  }
  __syncthreads();
  return *s_flag == 0;
}

// keep factor of prenet row `row` of `layer` at iteration `iter` (relu_drop_quad's draw: Philox word row & 3 of quad row >> 2)
__device__ __forceinline__ float pr_drop(const PrK& a, const int layer, const int iter, const int row) {
  if (iter >= a.S) return 0.f;  // prepared for a step that never runs
  if (a.drop_mask) return a.drop_mask[(layer ? (size_t)a.S * 256 : 0) + (size_t)iter * (layer ? 128 : 256) + row] * 2.f;
  uint32_t rr[4];
  philox4x32((uint32_t)iter, (uint32_t)layer, 0u, (uint32_t)(row >> 2), (uint32_t)a.seed, (uint32_t)(a.seed >> 32), rr);
  return rr[row & 3] >= 0x80000000u ? 2.f : 0.f;  // F.dropout(p = 0.5, training = True)   DecoderPrenet :18-21
}

__device__ __forceinline__ void pr_copy(float* dst, const float* __restrict__ src, const int floats) {
  for (int i = threadIdx.x * 4; i < floats; i += 512 * 4) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
}

__global__ __launch_bounds__(512) void ppg_resident_kernel(PrK a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lw = lds;
  float* xa = lds + PR_IMG_ATT;      // [512]
  float* xb = xa + 512;              // [256]
  float* xc = xb + 256;              // [512]
  float* s_p1 = xc + 512;            // [128]
  float* s_pre = s_p1 + 128;         // [256]
  int* s_flag = reinterpret_cast<int*>(s_pre + 256);
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (tests: the fallback path)
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sl = tid & 15;
  const int S = a.S;
  if (tid == 0) *s_flag = 0;
  auto EX = [&](int what, unsigned tag) { return a.ex + (size_t)(tag & 1) * PRX_PER + what; };
  const bool tr = a.trace && (g == 0 || g == PR_G_DEC || g == PR_G_Q0 || g == PR_G_OUT || g == PR_G_MOL);  // first workgroup of each role
#define PR_MK(role, k) ((tr && s >= 100 && s < 104) ? a.trace + ((role) * 4 + (s - 100)) * 16 + (k) : nullptr)
#define PR_MARK(role, k)                                                                             \
  do {                                                                                               \
    if (tr && tid == 0 && s >= 100 && s < 104) a.trace[((role) * 4 + (s - 100)) * 16 + (k)] = (unsigned long long)wall_clock64(); \
  } while (0)

  if (g < PR_ATT) {
    // ================================================================ ATT: prenet.1 + attention LSTMCell, units 8g .. 8g + 7
    pr_copy(lw, a.img_att + (size_t)g * PR_IMG_ATT, PR_IMG_ATT);
    float w1r[64];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 v = reinterpret_cast<const float4*>(a.img_w1)[c * 512 + tid];
      w1r[4 * c] = v.x; w1r[4 * c + 1] = v.y; w1r[4 * c + 2] = v.z; w1r[4 * c + 3] = v.w;
    }
    const float4* W4 = reinterpret_cast<const float4*>(lw) + tid;
    const int u = g * 8 + wave;
    const float4 bq = a.att_b4[u];
    const int r1 = tid >> 2, q1 = tid & 3;  // prenet.1: row r1, K quarter q1 (interleaved float4s)
    float cst = 0.f, acc_pre = 0.f;
    float dropf = pr_drop(a, 1, 0, r1);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
      float accx = 0.f;
      if (s > 0) {  // p0(0) = 0: the go frame through a bias-free prenet
        PR_MARK(0, 0);
        // (watched granule: row 0 of the next prenet input, one of those a stopping OUT workgroup holds back)
        if (a.variant & 2) { if (!pr_poll<4>(EX(PRX_P0, tag), tag, xa, a.abort_word, s_flag, false, PR_MK(0, 6))) return; }
        else if (!pr_fetch<1>(EX(PRX_P0, tag), 256, tag, xa, a.abort_word, s_flag, false, 0, PR_MK(0, 6))) return;
        PR_MARK(0, 1);
        const float4* x4 = reinterpret_cast<const float4*>(xa) + q1;
        float4 xv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) xv[c] = x4[c * 4];
        __builtin_amdgcn_sched_barrier(0);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          a0 = fmaf(w1r[4 * c], xv[c].x, a0); a1 = fmaf(w1r[4 * c + 1], xv[c].y, a1); a2 = fmaf(w1r[4 * c + 2], xv[c].z, a2); a3 = fmaf(w1r[4 * c + 3], xv[c].w, a3);
        }
        a1 = pr_sum4((a0 + a1) + (a2 + a3));
        if (q1 == 0) s_p1[r1] = fmaxf(a1, 0.f) * dropf;
        __syncthreads();
        accx = pr_dot<2>(W4, 512, reinterpret_cast<const float4*>(s_p1) + sl, 0.f);
        PR_MARK(0, 2);
      }
      const float v = pr_sum16(accx + acc_pre);
      const float gi = pr_sigmoid(pr_lane(v, 0) + bq.x), gf = pr_sigmoid(pr_lane(v, 16) + bq.y);
      const float gg = pr_tanh(pr_lane(v, 32) + bq.z), go = pr_sigmoid(pr_lane(v, 48) + bq.w);
      cst = gf * cst + gi * gg;
      const float h = go * pr_tanh(cst);
      // one store instruction per workgroup (8 granules = 64 contiguous bytes): granules stored one by one from eight waves
      // reached the consumers ~1 us later (8-byte pieces of a 32-byte ECC word are merged one after the other at the memory side);
      // padding every workgroup's granules to a whole 128-byte line on top of this measured no faster (8.81 vs 8.74 us per step)
      if (lane == 0) s_pre[wave] = h;
      __syncthreads();
      if (tid < 8) wp_put(EX(PRX_AH, tag) + g * 8 + tid, s_pre[tid], tag);
      PR_MARK(0, 3);
      if (a.trace && tid == 0 && s == 101) a.trace[512 + g] = (unsigned long long)wall_clock64();  // every workgroup's publish time
      if (s + 1 == S) break;
      // in the shadow of the rest of the step: the parts of the next step's gates that do not need its prenet output.
      // Nothing polls att_h here (Q0 and DEC are waiting for it on the chain): the context arrives two hand-offs later,
      // watched at a slow rate, and by then every granule of att_h is in place.
      dropf = pr_drop(a, 1, s + 1, r1);
      if (!pr_watch<16>(EX(PRX_CTX, tag) + 255, tag, a.abort_word, s_flag)) return;
      pr_sweep(EX(PRX_AH, tag), 512, tag, xc, a.abort_word, s_flag);
      pr_sweep(EX(PRX_CTX, tag), 256, tag, xb, a.abort_word, s_flag);
      if (!pr_done(s_flag)) return;
      acc_pre = pr_dot<8>(W4 + 6 * 512, 512, reinterpret_cast<const float4*>(xc) + sl, 0.f);
      acc_pre = pr_dot<4>(W4 + 2 * 512, 512, reinterpret_cast<const float4*>(xb) + sl, acc_pre);
      PR_MARK(0, 5);
    }
    return;
  }

  if (g < PR_G_Q0) {
    // ================================================================ DEC: decoder LSTMCell, units 4d .. 4d + 3
    const int d = g - PR_G_DEC;
    pr_copy(lw, a.img_dec + (size_t)d * PR_IMG_DEC, PR_IMG_DEC);
    const int half = wave >> 2, tl = tid & 255;  // waves 0-3: [att_h | context] part + the cell; waves 4-7: hidden part of the next step
    const float4* Wx = reinterpret_cast<const float4*>(lw) + tl;
    const float4* Wh = reinterpret_cast<const float4*>(lw) + 12 * 256 + tl;
    const int u = d * 4 + (wave & 3);
    const float4 bq = a.dec_b4[u];
    float cst = 0.f;
    if (tid < 256) s_pre[tid] = 0.f;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
      PR_MARK(1, 0);
      if (!pr_watch<1>(EX(PRX_AH, tag) + 511, tag, a.abort_word, s_flag)) return;
      pr_sweep(EX(PRX_AH, tag), 512, tag, xa, a.abort_word, s_flag);
      // h of the previous step: complete long ago (the OUT workgroups consumed it before p0 -> att_h of this step), and
      // fetched only now so that no DEC workgroup polls it while OUT is waiting for it
      if (s > 0) pr_sweep(EX(PRX_DH, tag - 1), 512, tag - 1, xc, a.abort_word, s_flag);
      if (!pr_done(s_flag)) return;
      PR_MARK(1, 1);
      float acc = 0.f;
      if (half == 0) acc = pr_dot<8>(Wx, 256, reinterpret_cast<const float4*>(xa) + sl, 0.f);
      else if (s > 0) s_pre[tl] = pr_dot<8>(Wh, 256, reinterpret_cast<const float4*>(xc) + sl, 0.f);
      PR_MARK(1, 2);
      if (a.variant & 1) { if (!pr_poll<4>(EX(PRX_CTX, tag), tag, xb, a.abort_word, s_flag, false, PR_MK(1, 6))) return; }
      else if (!pr_fetch<1>(EX(PRX_CTX, tag), 256, tag, xb, a.abort_word, s_flag, false, -1, PR_MK(1, 6))) return;
      PR_MARK(1, 3);
      if (half == 0) {
        acc = pr_dot<4>(Wx + 8 * 256, 256, reinterpret_cast<const float4*>(xb) + sl, acc);
        const float v = pr_sum16(acc + s_pre[tl]);
        const float gi = pr_sigmoid(pr_lane(v, 0) + bq.x), gf = pr_sigmoid(pr_lane(v, 16) + bq.y);
        const float gg = pr_tanh(pr_lane(v, 32) + bq.z), go = pr_sigmoid(pr_lane(v, 48) + bq.w);
        cst = gf * cst + gi * gg;
        const float h = go * pr_tanh(cst);
        if (lane == 0) s_p1[wave] = h;
      }
      __syncthreads();
      if (tid < 4) wp_put(EX(PRX_DH, tag) + d * 4 + tid, s_p1[tid], tag);  // one 32-byte store per workgroup
      PR_MARK(1, 4);
      if (a.trace && tid == 0 && s == 101) a.trace[512 + g] = (unsigned long long)wall_clock64();
    }
    return;
  }

  if (g < PR_G_OUT) {
    // ================================================================ Q0: q = relu(query_layer.0 . att_h + b)   mol_attention.py:75
    const int j = g - PR_G_Q0;
    pr_copy(lw, a.img_q0 + (size_t)j * PR_IMG_Q0, PR_IMG_Q0);
    const float4* W4 = reinterpret_cast<const float4*>(lw) + tid;
    const int row = j * 32 + (tid >> 4);
    const float b = a.q0_b[row];
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
      PR_MARK(2, 0);
      if (!pr_poll<8>(EX(PRX_AH, tag), tag, xa, a.abort_word, s_flag, false, PR_MK(2, 3))) return;
      PR_MARK(2, 1);
      const float v = pr_sum16(pr_dot<8>(W4, 512, reinterpret_cast<const float4*>(xa) + sl, 0.f));
      if (sl == 0) xb[tid >> 4] = fmaxf(v + b, 0.f);
      __syncthreads();
      if (tid < 32) wp_put(EX(PRX_Q, tag) + j * 32 + tid, xb[tid], tag);  // one 256-byte store per workgroup
      PR_MARK(2, 2);
      if (a.trace && tid == 0 && s == 101) a.trace[512 + g] = (unsigned long long)wall_clock64();
    }
    return;
  }

  if (g < PR_G_MOL) {
    // ================================================================ OUT: stop row | 16 rows of prenet.0' | RM / 16 mel rows
    const int j = g - PR_G_OUT;
    pr_copy(lw, a.img_out + (size_t)j * PR_IMG_OUT, PR_IMG_OUT);
    const float4* W4 = reinterpret_cast<const float4*>(lw) + tid;
    const int lr = tid >> 4, nmel = a.RM / 16;
    // kind of this lane's row: 0 stop, 1 prenet.0' row f, 2 mel row m, 3 dead
    const int kind = lr == 0 ? 0 : lr <= 16 ? 1 : lr < 17 + nmel ? 2 : 3;
    const int f = j * 16 + lr - 1, m = j * nmel + lr - 17;
    const float bias = kind == 0 ? a.out_b[a.RM] : kind == 1 ? a.fc0_b[f] : kind == 2 ? a.out_b[m] : 0.f;
    float dropf = kind == 1 ? pr_drop(a, 0, 1, f) : 0.f;
    bool stopped = false;  // wave 0's view (uniform there)
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
      PR_MARK(3, 0);
      if (!pr_fetch<1>(EX(PRX_CTX, tag), 256, tag, xb, a.abort_word, s_flag, stopped)) return;
      PR_MARK(3, 1);
      float acc = pr_dot<4>(W4 + 8 * 512, 512, reinterpret_cast<const float4*>(xb) + sl, 0.f);
      if (!pr_poll<8>(EX(PRX_DH, tag), tag, xa, a.abort_word, s_flag, false, PR_MK(3, 4))) return;
      PR_MARK(3, 2);
      acc = pr_dot<8>(W4, 512, reinterpret_cast<const float4*>(xa) + sl, acc);
      const float v = pr_sum16(acc) + bias;
      if (wave == 0) {  // stop_output = stop_layer([h, context]) (:288), stop rule (:301-305): sigmoid > threshold and enough steps
        const float lg = pr_lane(v, 0);
        stopped = (1.f / (1.f + expf(-lg)) > a.thr) && s + 1 >= a.min_steps;
        if (j == 0 && lane == 0) {
          a.stop_out[s] = lg;
          a.flags[TF_NFRAMES] = s + 1;
          if (stopped) {
            a.flags[TF_DONE] = 1;
            __hip_atomic_store(a.abort_word, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if (sl == 0) {
        if (kind == 1) {  // next step's prenet.0 through the projection's last frame (ppg_fast.h launch 6), relu, dropout
          s_p1[lr - 1] = fmaxf(v, 0.f) * dropf;
        } else if (kind == 2) {
          a.mel_out[(size_t)s * a.RM + m] = v;  // mel_output = linear_projection([h, context])   :281-287
        }
      }
      __syncthreads();
      if (tid < 16 && !stopped) wp_put(EX(PRX_P0, tag + 1) + j * 16 + tid, s_p1[tid], tag + 1);  // one 128-byte store per workgroup
      PR_MARK(3, 3);
      if (a.trace && tid == 0 && s == 101) a.trace[512 + g] = (unsigned long long)wall_clock64();
      if (kind == 1) dropf = pr_drop(a, 0, s + 2, f);
    }
    return;
  }

  // ================================================================== MOL: mixture-of-logistics attention   mol_attention.py:67-122
  // The first wave works the mixture parameters out while the others wait in a barrier (query_layer.2 rows as 16-lane
  // DPP sums; one exp / log1p sequence serves softmax and both softplus rows -- eight waves doing this redundantly, two
  // per SIMD, took 1.4 us).  Then every wave evaluates the logistic-CDF window at ITS 33 positions (wave w owns memory
  // rows 32 w .. 32 w + 31, kept in registers for the whole utterance), takes the differences through one lane shift and
  // accumulates its share of context = alpha . memory with the weights broadcast by v_readlane; the eight partial
  // contexts meet in LDS.
  {
    const int T = a.T, M = a.M;
    float* s_q = lds;                    // [256]
    float* s_mpw = s_q + 256;            // [16] raw mixture parameters
    float* s_mix = s_mpw + 16;           // [3][16]: w, 1 / sigma, mu of this step
    float* s_part = s_mpw + 128;         // [8][256]
    int* s_flag2 = reinterpret_cast<int*>(s_part + 2048);
    if (tid == 0) *s_flag2 = 0;
    const int rr = lane >> 4;
    const float* mem = a.memory + lane * 4;
    constexpr int NR = 32;
    float4 mv[NR];
#pragma unroll
    for (int jj = 0; jj < NR; ++jj) {
      const int t = wave * NR + jj;
      mv[jj] = t < T ? *reinterpret_cast<const float4*>(mem + (size_t)t * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // query_layer.2 in LDS: row o = 4 gq + rr belongs to lanes rr * 16 .. + 15, slice sl owns k = (c * 16 + sl) * 4 .. + 3;
    // float4 (gq * 4 + c) * 64 + lane is this lane's (conflict-free ds_read_b128)
    float4* s_w2 = reinterpret_cast<float4*>(s_part + 2048 + 4);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int o = gq * 4 + rr;
        if (wave == 0)
          s_w2[(gq * 4 + c) * 64 + lane] = o < 3 * M ? *reinterpret_cast<const float4*>(a.w2 + (size_t)o * 256 + (c * 16 + sl) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    // lanes 0..M-1: w_hat (softmax), 16..16+M-1: sigma_hat, 32..32+M-1: Delta_hat -- one exp / log1p sequence serves all three
    const int r3 = lane >> 4, m16 = lane & 15;
    const bool live = m16 < M && r3 < 3;
    const float b_l = live ? a.b2[r3 * M + m16] : 0.f;
    float mu_prev = 0.f;  // lanes 32..32+M-1 of the first wave carry the means
    float* mpw = s_mpw;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
      PR_MARK(4, 0);
      if (!pr_poll<4>(EX(PRX_Q, tag), tag, s_q, a.abort_word, s_flag2, false, PR_MK(4, 5))) return;
      PR_MARK(4, 1);
      // mixture_params = query_layer.2(q)  :75 -- waves 0..3 take four rows each
      if (wave < 4) {
        const float4* q4 = reinterpret_cast<const float4*>(s_q) + sl;
        float4 qv[4], wv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { qv[c] = q4[c * 16]; wv[c] = s_w2[(wave * 4 + c) * 64 + lane]; }
        __builtin_amdgcn_sched_barrier(0);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s0 = fmaf(wv[c].x, qv[c].x, s0); s1 = fmaf(wv[c].y, qv[c].y, s1); s2 = fmaf(wv[c].z, qv[c].z, s2); s3 = fmaf(wv[c].w, qv[c].w, s3);
        }
        const float acc = pr_sum16((s0 + s1) + (s2 + s3));
        if (sl == 0) mpw[wave * 4 + rr] = acc;
      }
      __syncthreads();
      // The first wave alone (the other seven wait in the barrier and leave it its SIMD):
      // w = softmax(w_hat) + eps; sigma = softplus(sigma_hat) + eps; mu = mu_prev + softplus(Delta_hat)   :92-96
      if (wave == 0) {
        const float x = live ? mpw[r3 * M + m16] + b_l : (r3 == 0 ? -INFINITY : 0.f);
        float mx = x;  // max over the DPP row (row 0: lanes >= M hold -inf)
        mx = fmaxf(mx, pr_dpp(mx, 0)); mx = fmaxf(mx, pr_dpp(mx, 1)); mx = fmaxf(mx, pr_dpp(mx, 2)); mx = fmaxf(mx, pr_dpp(mx, 3));
        const float e = expf(r3 == 0 ? x - mx : x);
        const float ew = (live && r3 == 0) ? e : 0.f;
        float se = 0.f;
#pragma unroll
        for (int mm = 0; mm < 5; ++mm) se += pr_lane(ew, mm);  // ascending m, as the sequential sum (lanes >= M hold 0)
        const float sp = pr_softplus(x, e);
        float val = ew * __builtin_amdgcn_rcpf(se) + a.eps;          // row 0: w
        if (r3 == 1) val = __builtin_amdgcn_rcpf(sp + a.eps);        // row 1: 1 / sigma
        if (r3 == 2) { val = mu_prev + sp; if (live) mu_prev = val; }  // row 2: mu
        if (lane < 48) s_mix[lane] = val;
      }
      __syncthreads();
      float wm[5], isg[5], mum[5];
#pragma unroll
      for (int mm = 0; mm < 5; ++mm) { wm[mm] = s_mix[mm]; isg[mm] = s_mix[16 + mm]; mum[mm] = s_mix[32 + mm]; }
      PR_MARK(4, 2);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};  // this wave's share of context = alpha . memory   :115 (v_pk_fma_f32)
      float* al = a.align_out + (size_t)s * T;
      for (int t0 = wave * NR; t0 < T; t0 += 8 * NR) {  // (one pass up to T_enc = 256)
        // alpha_full[j] = sum_m w_m / (1 + sigmoid((mu_m - (j + 0.5)) / sigma_m))   :101-107, at j = t0 + lane (33 of them matter).
        // With e = exp(-z): 1 / (1 + 1 / (1 + e)) = 1 - 1 / (2 + e); hardware exp2 / rcp (1 ulp) -- the window is compared at 1e-4.
        float af = 0.f;
        const float pos = (float)(t0 + lane) + 0.5f;
#pragma unroll
        for (int mm = 0; mm < 5; ++mm)
          if (mm < M) {
            const float e = __expf((pos - mum[mm]) * isg[mm]);
            af += wm[mm] * (1.f - __builtin_amdgcn_rcpf(2.f + e));
          }
        float v = __shfl_down(af, 1, 64) - af;  // alpha_t = diff; zeros -> eps   :108-109
        if (v == 0.f) v = a.eps;
        const int t = t0 + lane;
        if (lane >= NR || t >= T) v = 0.f;
        else al[t] = v;
        if (t0 == wave * NR) {
#pragma unroll
          for (int jj = 0; jj < NR; ++jj) {
            const float sc = pr_lane(v, jj);
            const f32x2 sc2 = {sc, sc}, m01 = {mv[jj].x, mv[jj].y}, m23 = {mv[jj].z, mv[jj].w};
            acc01 = __builtin_elementwise_fma(sc2, m01, acc01); acc23 = __builtin_elementwise_fma(sc2, m23, acc23);
          }
        } else {  // T_enc > 256: these rows come from L2, 4 in flight
#pragma unroll 1
          for (int j8 = 0; j8 < NR; j8 += 4) {
            float4 vv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int tr_ = t0 + j8 + jj;
              vv[jj] = tr_ < T ? *reinterpret_cast<const float4*>(mem + (size_t)tr_ * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float sc = __shfl(v, j8 + jj, 64);
              const f32x2 sc2 = {sc, sc}, m01 = {vv[jj].x, vv[jj].y}, m23 = {vv[jj].z, vv[jj].w};
              acc01 = __builtin_elementwise_fma(sc2, m01, acc01); acc23 = __builtin_elementwise_fma(sc2, m23, acc23);
            }
          }
        }
      }
      PR_MARK(4, 3);
      *reinterpret_cast<float4*>(s_part + wave * 256 + lane * 4) = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
      __syncthreads();
      if (tid < 256) {
        float r = s_part[tid];
#pragma unroll
        for (int w8 = 1; w8 < 8; ++w8) r += s_part[w8 * 256 + tid];
        wp_put(EX(PRX_CTX, tag) + tid, r, tag);
      }
      PR_MARK(4, 4);
    }
  }
#undef PR_MARK
#undef PR_MK
}

}  // namespace mb
