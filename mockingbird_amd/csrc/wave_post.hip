// Waveform wire format on device: peak normalisation and int16 PCM packing,
// the elementwise tail the reference runs in numpy / libsndfile on the host after the vocoder:
//   peak normalise   wav / np.abs(wav).max() * 0.97        gen_voice.py:41, control/toolbox/__init__.py:313
//   encode_16bits    clip(x * 2**15).astype(int16)         models/vocoder/wavernn/audio.py:38-39
//   save_wav         x * (32767 / max(0.01, max|x|))       models/synthesizer/audio.py:12-15
//   PCM_16           libsndfile f2s_clip_array/d2s_clip_array (sf.write(.., "PCM_16"), run.py:91)
// HBM-bound byte work: one read pass for max|x| (where the mode needs it), one read + one 2-byte write pass.
// Arithmetic is done in the array's own type (fp32 for the GAN vocoders, float64 after mb_wavernn_finish),
// one IEEE operation per numpy operation, no contraction.
#include "common.h"

#pragma clang fp contract(off)

namespace mb {

template <typename T> struct Bits;
template <> struct Bits<float> {
  using U = unsigned int;
  static __device__ __forceinline__ U of(float v) { return __builtin_bit_cast(unsigned int, v); }
  static __device__ __forceinline__ float from(U u) { return __builtin_bit_cast(float, u); }
};
template <> struct Bits<double> {
  using U = unsigned long long;
  static __device__ __forceinline__ U of(double v) { return __builtin_bit_cast(unsigned long long, v); }
  static __device__ __forceinline__ double from(U u) { return __builtin_bit_cast(double, u); }
};

// slot = max over |x| as a bit pattern: non-negative IEEE values order like unsigned integers, and a NaN
// (larger than +inf as a pattern) wins, which is np.max's NaN propagation
template <typename T>
__global__ __launch_bounds__(256) void wave_absmax_kernel(const T* __restrict__ x, long long n,
                                                          typename Bits<T>::U* __restrict__ slot) {
  using U = typename Bits<T>::U;
  U m = 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const U b = Bits<T>::of(x[i]) & (~(U)0 >> 1);  // clear the sign: |x|
    m = b > m ? b : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const U o = (U)__shfl_xor((unsigned long long)m, off);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(slot, m);
}

template <typename T>
__global__ __launch_bounds__(256) void wave_peak_normalize_kernel(T* __restrict__ x, long long n,
                                                                  const typename Bits<T>::U* __restrict__ slot,
                                                                  T target) {
  const T m = Bits<T>::from(*slot);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T q = x[i] / m;  // numpy: (wav / max) * 0.97, two roundings
    x[i] = q * target;
  }
}

__device__ __forceinline__ int round_half_even(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int round_half_even(double v) { return __double2int_rn(v); }

template <typename T, int MODE>
__global__ __launch_bounds__(256) void wave_pack_pcm16_kernel(const T* __restrict__ x, long long n,
                                                              const typename Bits<T>::U* __restrict__ slot,
                                                              short* __restrict__ y) {
  T scale = (T)32768;
  if (MODE == MB_PCM16_SAVE_WAV) {
    // 32767 / max(0.01, m): python's max keeps 0.01 unless m > 0.01; the quotient is rounded once in the
    // array's type (NumPy >= 2 scalar promotion; 32767 / 0.01 is 3276700 in either type)
    const T m = Bits<T>::from(*slot);
    scale = (m > (T)0.01) ? (T)32767 / m : (T)3276700;
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T v = x[i] * scale;
    int q;
    if (MODE == MB_PCM16_SNDFILE) {  // f2s_clip_array / d2s_clip_array
      q = v >= (T)32767 ? 32767 : (v <= (T)-32768 ? -32768 : round_half_even(v));
    } else if (MODE == MB_PCM16_ENCODE16) {  // np.clip then C truncation
      const T c = v < (T)-32768 ? (T)-32768 : (v > (T)32767 ? (T)32767 : v);
      q = (int)c;
    } else {
      q = (int)v;  // |v| <= 32767 (1 + ulp) by construction
    }
    y[i] = (short)q;
  }
}


// ---- the same tail for a BATCH of waveforms in two launches (gen_voice.py:30-41 per request: sentence breaks, peak normalisation, PCM) ----
// piece = {source offset (< 0: zeros), destination offset, length, item} in elements; the destination buffer is zeroed by the caller side of the
// ABI (mb_wave_finish_batch does it), so the gaps behind the sentences need no writes.  One IEEE operation per numpy operation, in
// the same order as the one-waveform kernels above: an item's result is bit for bit what insert_breaks -> peak_normalize -> pack give.
struct WavePiece { long long src, dst, len; long long item; };

__global__ __launch_bounds__(256) void wave_batch_absmax_kernel(const float* __restrict__ x, const WavePiece* __restrict__ pieces,
                                                                int blocks_per_piece, unsigned* __restrict__ slots) {
  const WavePiece p = pieces[blockIdx.x / blocks_per_piece];
  const int sub = blockIdx.x % blocks_per_piece;
  unsigned m = 0;
  if (p.src < 0) return;  // a gap
  for (long long i = (long long)sub * 256 + threadIdx.x; i < p.len; i += (long long)blocks_per_piece * 256) {
    const unsigned b = __builtin_bit_cast(unsigned, x[p.src + i]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(slots + p.item, m);
}

template <int MODE>  // -1: float out
__global__ __launch_bounds__(256) void wave_batch_write_kernel(const float* __restrict__ x, const WavePiece* __restrict__ pieces,
                                                               int blocks_per_piece, const unsigned* __restrict__ slots, int normalize,
                                                               float target, float* __restrict__ yf, short* __restrict__ ys) {
  const WavePiece p = pieces[blockIdx.x / blocks_per_piece];
  const int sub = blockIdx.x % blocks_per_piece;
  const float m = normalize ? __builtin_bit_cast(float, slots[p.item]) : 1.f;
  for (long long i = (long long)sub * 256 + threadIdx.x; i < p.len; i += (long long)blocks_per_piece * 256) {
    float v = p.src < 0 ? 0.f : x[p.src + i];  // (src < 0: a gap of zeros -- 0 / max * target, NaN for an all-zero item as in numpy)
    if (normalize) {
      const float q = v / m;  // numpy: (wav / max) * 0.97, two roundings
      v = q * target;
    }
    if (MODE < 0) { yf[p.dst + i] = v; continue; }
    const float w = v * 32768.f;
    int q;
    if (MODE == MB_PCM16_SNDFILE) q = w >= 32767.f ? 32767 : (w <= -32768.f ? -32768 : __float2int_rn(w));
    else { const float c = w < -32768.f ? -32768.f : (w > 32767.f ? 32767.f : w); q = (int)c; }
    ys[p.dst + i] = (short)q;
  }
}

static int wave_grid(long long n) { return (int)std::min<long long>((n + 255) / 256, 256 * 8); }

template <typename T>
static int absmax(const void* x, long long n, void* slot, hipStream_t s) {
  MB_HIP(hipMemsetAsync(slot, 0, 8, s));
  hipLaunchKernelGGL(wave_absmax_kernel<T>, dim3(wave_grid(n)), dim3(256), 0, s, (const T*)x, n,
                     (typename Bits<T>::U*)slot);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

template <typename T>
static int pack(const void* x, long long n, int mode, int16_t* y, void* slot, hipStream_t s) {
  using U = typename Bits<T>::U;
  const dim3 g(wave_grid(n)), b(256);
  if (mode == MB_PCM16_SAVE_WAV) {
    int rc = absmax<T>(x, n, slot, s);
    if (rc) return rc;
    hipLaunchKernelGGL((wave_pack_pcm16_kernel<T, MB_PCM16_SAVE_WAV>), g, b, 0, s, (const T*)x, n, (const U*)slot, (short*)y);
  } else if (mode == MB_PCM16_ENCODE16) {
    hipLaunchKernelGGL((wave_pack_pcm16_kernel<T, MB_PCM16_ENCODE16>), g, b, 0, s, (const T*)x, n, (const U*)slot, (short*)y);
  } else {
    hipLaunchKernelGGL((wave_pack_pcm16_kernel<T, MB_PCM16_SNDFILE>), g, b, 0, s, (const T*)x, n, (const U*)slot, (short*)y);
  }
  MB_HIP(hipGetLastError());
  return MB_OK;
}

}  // namespace mb

using namespace mb;

extern "C" size_t mb_wave_workspace_bytes(void) { return 256; }

extern "C" int mb_wave_peak_normalize(void* d_wav, int dtype, long long n, double target, void* d_workspace,
                                      size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(dtype == MB_F32 || dtype == MB_F64, "wave_peak_normalize: dtype %d (MB_F32 or MB_F64)", dtype);
  MB_REQUIRE(n >= 0, "wave_peak_normalize: n=%lld", n);
  if (n == 0) return MB_OK;  // np.abs(empty).max() raises in the reference; callers never pass an empty waveform
  MB_REQUIRE(d_wav, "wave_peak_normalize: null pointer");
  if (!d_workspace || workspace_bytes < mb_wave_workspace_bytes()) {
    set_error("wave_peak_normalize: workspace %zu B < required %zu B", workspace_bytes, mb_wave_workspace_bytes());
    return MB_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == MB_F32) {
    if ((rc = absmax<float>(d_wav, n, d_workspace, s))) return rc;
    hipLaunchKernelGGL(wave_peak_normalize_kernel<float>, dim3(wave_grid(n)), dim3(256), 0, s, (float*)d_wav, n,
                       (const unsigned int*)d_workspace, (float)target);
  } else {
    if ((rc = absmax<double>(d_wav, n, d_workspace, s))) return rc;
    hipLaunchKernelGGL(wave_peak_normalize_kernel<double>, dim3(wave_grid(n)), dim3(256), 0, s, (double*)d_wav, n,
                       (const unsigned long long*)d_workspace, target);
  }
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_wave_pack_pcm16(const void* d_wav, int dtype, long long n, int mode, int16_t* d_pcm,
                                  void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(dtype == MB_F32 || dtype == MB_F64, "wave_pack_pcm16: dtype %d (MB_F32 or MB_F64)", dtype);
  MB_REQUIRE(mode == MB_PCM16_SNDFILE || mode == MB_PCM16_ENCODE16 || mode == MB_PCM16_SAVE_WAV,
             "wave_pack_pcm16: mode %d", mode);
  MB_REQUIRE(n >= 0, "wave_pack_pcm16: n=%lld", n);
  if (n == 0) return MB_OK;
  MB_REQUIRE(d_wav && d_pcm, "wave_pack_pcm16: null pointer");
  if (mode == MB_PCM16_SAVE_WAV && (!d_workspace || workspace_bytes < mb_wave_workspace_bytes())) {
    set_error("wave_pack_pcm16: workspace %zu B < required %zu B", workspace_bytes, mb_wave_workspace_bytes());
    return MB_ENOMEM;
  }
  return dtype == MB_F32 ? pack<float>(d_wav, n, mode, d_pcm, d_workspace, (hipStream_t)stream)
                         : pack<double>(d_wav, n, mode, d_pcm, d_workspace, (hipStream_t)stream);
}

extern "C" size_t mb_wave_finish_batch_workspace_bytes(int n_items) { return n_items > 0 ? (size_t)n_items * sizeof(unsigned) : 0; }

extern "C" int mb_wave_finish_batch(const float* d_wav, const long long* d_pieces, int n_pieces, int n_items, long long out_elems,
                                    double normalize_target, int pcm_mode, void* d_out, void* d_workspace, size_t workspace_bytes,
                                    mb_stream_t stream) {
  MB_REQUIRE(n_pieces >= 0 && n_items >= 0 && out_elems >= 0, "wave_finish_batch: bad counts");
  MB_REQUIRE(pcm_mode == -1 || pcm_mode == MB_PCM16_SNDFILE || pcm_mode == MB_PCM16_ENCODE16,
             "wave_finish_batch: pcm_mode %d (-1 float out, MB_PCM16_SNDFILE, MB_PCM16_ENCODE16; save_wav takes the one-waveform calls)", pcm_mode);
  if (out_elems == 0) return MB_OK;
  MB_REQUIRE(d_out, "wave_finish_batch: null pointer");
  hipStream_t s = (hipStream_t)stream;
  MB_HIP(hipMemsetAsync(d_out, 0, (size_t)out_elems * (pcm_mode < 0 ? sizeof(float) : sizeof(short)), s));  // the gaps
  if (n_pieces == 0) return MB_OK;
  MB_REQUIRE(d_wav && d_pieces, "wave_finish_batch: null pointer");
  const bool norm = normalize_target == normalize_target && normalize_target > 0.0;  // (NaN / <= 0: no normalisation)
  if (norm) {
    if (!d_workspace || workspace_bytes < mb_wave_finish_batch_workspace_bytes(n_items)) {
      set_error("wave_finish_batch: workspace %zu B < required %zu B", workspace_bytes, mb_wave_finish_batch_workspace_bytes(n_items));
      return MB_ENOMEM;
    }
    MB_HIP(hipMemsetAsync(d_workspace, 0, mb_wave_finish_batch_workspace_bytes(n_items), s));
  }
  const WavePiece* pieces = reinterpret_cast<const WavePiece*>(d_pieces);
  const int bpp = std::max(1, std::min(64, 2048 / std::max(1, n_pieces)));  // blocks per piece: ~2048 workgroups in all
  const dim3 g((unsigned)n_pieces * bpp), b(256);
  unsigned* slots = reinterpret_cast<unsigned*>(d_workspace);
  if (norm) hipLaunchKernelGGL(wave_batch_absmax_kernel, g, b, 0, s, d_wav, pieces, bpp, slots);
  if (pcm_mode < 0) hipLaunchKernelGGL((wave_batch_write_kernel<-1>), g, b, 0, s, d_wav, pieces, bpp, slots, norm ? 1 : 0, (float)normalize_target, (float*)d_out, (short*)nullptr);
  else if (pcm_mode == MB_PCM16_SNDFILE) hipLaunchKernelGGL((wave_batch_write_kernel<MB_PCM16_SNDFILE>), g, b, 0, s, d_wav, pieces, bpp, slots, norm ? 1 : 0, (float)normalize_target, (float*)nullptr, (short*)d_out);
  else hipLaunchKernelGGL((wave_batch_write_kernel<MB_PCM16_ENCODE16>), g, b, 0, s, d_wav, pieces, bpp, slots, norm ? 1 : 0, (float)normalize_target, (float*)nullptr, (short*)d_out);
  MB_HIP(hipGetLastError());
  return MB_OK;
}
