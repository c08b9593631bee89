// Device post-processing of WaveRNN.generate in float64:
//   xfade_and_unfold          models/vocoder/wavernn/models/fatchord_version.py:340-402
//   decode_mu_law             models/vocoder/wavernn/audio.py:102-107   (from_labels=False, :244-245)
//   de_emphasis               models/vocoder/wavernn/audio.py:92-93     (scipy lfilter([1],[1,-c]) :247-248)
//   truncate + linear fade    fatchord_version.py:250-253
// The reference runs these in numpy on the host after a D2H copy of the folds; here the folds never
// leave the GPU and only the finished waveform is downloaded.
//
// Numerics: every elementwise step is evaluated in the reference's operation order in float64 with
// contraction disabled (numpy does not fuse a*b+c); the IIR  y[n] = x[n] + c*y[n-1]  is a three-pass
// chunked scan (local recurrence per 256-sample chunk, serial carry over the chunk tails, carry * c^(i+1)
// added back), which reorders roundings: agreement with the sequential filter is ~1e-16 relative, the test
// gate is 1e-12 absolute.
#include "common.h"

// numpy evaluates a*b+c with two roundings: no fused multiply-add anywhere in this file.  Plain operators
// under this pragma; HIP's __dmul_rn / __dadd_rn are header inlines compiled with contraction allowed.
#pragma clang fp contract(off)

namespace mb {

struct PostK {
  const float* samples;  // [N][S]
  double* u;             // [L] unfolded (+ mu-law decoded) signal, then filtered in place
  double* carry;         // [nchunks]: chunk tails, then incoming carries
  double* wav;           // [out_len]
  int N, S, batched, overlap, L, out_len, fade_n;
  int mu, mu_law, deemph;
  double coef;
};

constexpr int POST_CHUNK = 256;

// np.linspace(start, stop, n)[m]: arange(n) * step + start, last element = stop exactly (two roundings)
__device__ __forceinline__ double linspace_at(double start, double stop, int n, int m) {
  if (n == 1) return start;
  if (m == n - 1) return stop;
  const double step = (stop - start) / (double)(n - 1);
  return (double)m * step + start;
}

// gain of position j inside a fold  (:363-385): [silence | sqrt(.5(1+t))] in, mirrored out, 1 between
__device__ __forceinline__ double fold_gain(int j, int S, int overlap) {
  if (overlap <= 0) return 1.0;
  const int silence = overlap / 2, fade_len = overlap - silence;
  if (j < overlap) {
    if (j < silence) return 0.0;
    const double t = linspace_at(-1.0, 1.0, fade_len, j - silence);
    return sqrt(0.5 * (1.0 + t));
  }
  const int q = j - (S - overlap);
  if (q >= 0) {
    if (q >= fade_len) return 0.0;
    const double t = linspace_at(-1.0, 1.0, fade_len, q);
    return sqrt(0.5 * (1.0 - t));
  }
  return 1.0;
}

__global__ __launch_bounds__(256) void wrn_unfold_decode_kernel(PostK a) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.L) return;
  double v;
  if (a.batched) {
    const int step = a.S - a.overlap;  // target + overlap
    int hi = p / step;
    if (hi > a.N - 1) hi = a.N - 1;
    v = 0.0;
    // folds are added in ascending order (:397-400); at most two cover a position.  numpy multiplies the
    // fade into EVERY sample of the head / tail (gain 1 elsewhere is no operation)
    for (int i = hi - 1; i <= hi; ++i) {
      if (i < 0) continue;
      const int j = p - i * step;
      if (j < 0 || j >= a.S) continue;
      double y = (double)a.samples[(size_t)i * a.S + j];
      if (a.overlap > 0 && (j < a.overlap || j >= a.S - a.overlap)) y = ((y) * (fold_gain(j, a.S, a.overlap)));
      v = ((v) + (y));
    }
  } else {
    v = (double)a.samples[p];
  }
  if (a.mu_law) {  // np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)
    const double mu = (double)(a.mu - 1);
    const double sg = v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0);
    v = (sg / mu) * (pow(1.0 + mu, fabs(v)) - 1.0);
  }
  a.u[p] = v;
}

// pass 1: local recurrence per chunk (zero initial state), tail -> carry[chunk]
__global__ __launch_bounds__(64) void wrn_deemph_local_kernel(PostK a, int nchunks) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchunks) return;
  const int p0 = c * POST_CHUNK, p1 = min(p0 + POST_CHUNK, a.L);
  double y = 0.0;
  for (int p = p0; p < p1; ++p) {
    y = a.u[p] + a.coef * y;
    a.u[p] = y;
  }
  a.carry[c] = y;
}

// pass 2 (one thread): carry[c] <- state entering chunk c
__global__ void wrn_deemph_carry_kernel(PostK a, int nchunks) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double cL = pow(a.coef, (double)POST_CHUNK);
  double in = 0.0;
  for (int c = 0; c < nchunks; ++c) {
    const double tail = a.carry[c];
    a.carry[c] = in;
    in = tail + cL * in;  // (chunks are full except the last, whose tail is unused)
  }
}

// pass 3: y[p] += carry_in * c^(i+1); truncate; linear fade-out over the last fade_n samples (:250-253)
__global__ __launch_bounds__(POST_CHUNK) void wrn_deemph_apply_fade_kernel(PostK a) {
  const int c = blockIdx.x, i = threadIdx.x, p = c * POST_CHUNK + i;
  __shared__ double pw[POST_CHUNK];
  if (a.deemph) {
    pw[i] = pow(a.coef, (double)(i + 1));
  }
  if (p >= a.out_len) return;
  double v = a.u[p];
  if (a.deemph) v = v + a.carry[c] * pw[i];  // pw[i] is this thread's own entry
  const int f0 = a.out_len - a.fade_n;
  if (p >= f0) v = ((v) * (linspace_at(1.0, 0.0, a.fade_n, p - f0)));
  a.wav[p] = v;
}

}  // namespace mb

using namespace mb;

static int post_lengths(int n_folds, int seq_len, int batched, int overlap, int* L) {
  MB_REQUIRE(n_folds >= 1 && seq_len >= 1 && overlap >= 0, "wavernn_finish: bad shape");
  if (batched) {
    MB_REQUIRE(seq_len > 2 * overlap, "wavernn_finish: seq_len %d <= 2*overlap", seq_len);
    *L = n_folds * (seq_len - overlap) + overlap;  // num_folds * (target + overlap) + overlap  (:359)
  } else {
    *L = seq_len;
  }
  return MB_OK;
}

extern "C" size_t mb_wavernn_finish_workspace_bytes(int n_folds, int seq_len, int batched, int overlap) {
  int L = 0;
  if (post_lengths(n_folds, seq_len, batched, overlap, &L)) return 0;
  return align_up((size_t)L * sizeof(double), 256) + align_up((size_t)cdiv(L, POST_CHUNK) * sizeof(double), 256) + 256;
}

extern "C" int mb_wavernn_finish(const float* d_samples, int n_folds, int seq_len, int batched, int overlap,
                                 int n_classes, int mu_law, int apply_preemphasis, double preemphasis,
                                 int wave_len, int fade_len, double* d_wav, int* out_len, void* d_workspace,
                                 size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(d_samples && d_wav && out_len, "wavernn_finish: null pointer");
  int L = 0;
  int rc = post_lengths(n_folds, seq_len, batched, overlap, &L);
  if (rc) return rc;
  MB_REQUIRE(batched || n_folds == 1, "wavernn_finish: unbatched output has one sequence");
  const int n_out = std::min(wave_len, L);  // output[:wave_len] (:251)
  // numpy raises when the fade window is longer than the waveform (mels < 26 frames, SURVEY finding 5)
  MB_REQUIRE(n_out >= fade_len && fade_len >= 1, "wavernn_finish: waveform of %d samples is shorter than the %d-sample fade-out", n_out, fade_len);
  const size_t need = mb_wavernn_finish_workspace_bytes(n_folds, seq_len, batched, overlap);
  if (!d_workspace || workspace_bytes < need) {
    set_error("wavernn_finish: workspace %zu B < required %zu B", workspace_bytes, need);
    return MB_ENOMEM;
  }
  Arena ar(d_workspace, workspace_bytes);
  PostK k;
  k.samples = d_samples; k.u = ar.take<double>(L); k.carry = ar.take<double>(cdiv(L, POST_CHUNK)); k.wav = d_wav;
  k.N = n_folds; k.S = seq_len; k.batched = batched; k.overlap = overlap; k.L = L; k.out_len = n_out; k.fade_n = fade_len;
  k.mu = n_classes; k.mu_law = mu_law; k.deemph = apply_preemphasis; k.coef = preemphasis;
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = cdiv(L, POST_CHUNK);
  hipLaunchKernelGGL(wrn_unfold_decode_kernel, dim3(cdiv(L, 256)), dim3(256), 0, s, k);
  if (apply_preemphasis) {
    hipLaunchKernelGGL(wrn_deemph_local_kernel, dim3(cdiv(nchunks, 64)), dim3(64), 0, s, k, nchunks);
    hipLaunchKernelGGL(wrn_deemph_carry_kernel, dim3(1), dim3(1), 0, s, k, nchunks);
  }
  hipLaunchKernelGGL(wrn_deemph_apply_fade_kernel, dim3(cdiv(n_out, POST_CHUNK)), dim3(POST_CHUNK), 0, s, k);
  MB_HIP(hipGetLastError());
  *out_len = n_out;
  return MB_OK;
}
