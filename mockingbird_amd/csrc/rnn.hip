// Row-tile MFMA recurrent GEMM kernels (see rnn.h for the design).
#include "rnn.h"

namespace mb {

template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void rnn_rowtile_kernel(RnnK a) {
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr int BLK = 4 * RL * 16;  // floats per (tile, k-block)
  constexpr int NPART = (EPI == EPI_GRU) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[NW * NPART * 256];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt = blockIdx.x, ntile = blockIdx.y;
  const int i = lane & 15, kq = lane >> 4;  // A: row i, k-quad kq;  B: column i, k-quad kq
  const int u = i >> 2, tau = i & 3;
  const bool live = tau < RL;
  const float* wl = a.w + (size_t)mt * a.nkb_total * BLK + ((u * RL + tau) * 4 + kq) * 4;
  int ncol = ntile * 16 + i;
  if (ncol >= a.N) ncol = a.N - 1;  // duplicate a live column; its result is never stored

  if (a.step_counter && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.step_counter += 1;

  f32x4 accX = {0.f, 0.f, 0.f, 0.f}, accH = {0.f, 0.f, 0.f, 0.f};
  int kb0 = 0;
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const RnnSeg sg = a.seg[sgi];
    const float* xb = sg.p + (size_t)ncol * sg.ld + kq * 4;
    // this wave's blocks inside the segment: local index j with (kb0 + j) % NW == wave
    int j = (wave - kb0 % NW + NW) % NW;
    for (; j < sg.nkb; j += NW) {
      const float4 bv = *reinterpret_cast<const float4*>(xb + j * 16);
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) av = *reinterpret_cast<const float4*>(wl + (size_t)(kb0 + j) * BLK);
      if (NPART == 2 && sg.part == 1) {
        accH = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, accH, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, accH, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, accH, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, accH, 0, 0, 0);
      } else {
        accX = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, accX, 0, 0, 0);
        accX = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, accX, 0, 0, 0);
        accX = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, accX, 0, 0, 0);
        accX = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, accX, 0, 0, 0);
      }
    }
    kb0 += sg.nkb;
  }
  // cross-wave reduction through LDS: D fragment lane = (unit = lane>>4, col = lane&15), reg = gate
  float4* red4 = reinterpret_cast<float4*>(red);
  red4[(wave * NPART + 0) * 64 + lane] = make_float4(accX[0], accX[1], accX[2], accX[3]);
  if (NPART == 2) red4[(wave * NPART + 1) * 64 + lane] = make_float4(accH[0], accH[1], accH[2], accH[3]);
  __syncthreads();
  if (wave != 0) return;
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float4 v = red4[(w * NPART + 0) * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
    if (NPART == 2) {
      const float4 h = red4[(w * NPART + 1) * 64 + lane];
      sh[0] += h.x; sh[1] += h.y; sh[2] += h.z; sh[3] += h.w;
    }
  }
  const int n = ntile * 16 + (lane & 15);
  const int du = lane >> 4;  // unit within tile
  if (n >= a.N) return;
  const int prow = a.pre_idx ? a.pre_idx[n] : 0;
  const float* pre = a.pre_table ? a.pre_table + (size_t)prow * a.pre_stride : nullptr;

  if (EPI == EPI_LINEAR) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 16 + du * 4 + r;
      if (row < a.units) {
        float v = sx[r];
        if (a.biasX) v += a.biasX[row];
        if (pre) v += pre[row];
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = sigmoidf_(v);
        else if (a.act == 3) v = tanhf(v);
        const size_t o = (size_t)n * a.ldy + row;
        if (a.mask) v = v * a.mask[o] * a.mask_scale;
        a.y[o] = v;
      }
    }
    return;
  }
  const int j = mt * 4 + du;  // hidden unit
  if (j >= a.units) return;
  const int H = a.units;
  const size_t so = (size_t)n * H + j;
  if (EPI == EPI_GRU) {
    // torch GRUCell (gate order r,z,n): r = s(i_r+h_r), z = s(i_z+h_z), n = tanh(i_n + r*h_n),
    // h' = n + z*(h - n).   models/vocoder/wavernn/models/fatchord_version.py:196-200,265-271;
    // models/synthesizer/models/tacotron.py:60,98
    float ir = sx[0], iz = sx[1], in_ = sx[2], hr = sh[0], hz = sh[1], hn = sh[2];
    if (a.biasX) { ir += a.biasX[j]; iz += a.biasX[H + j]; in_ += a.biasX[2 * H + j]; }
    if (pre) { ir += pre[j]; iz += pre[H + j]; in_ += pre[2 * H + j]; }
    if (a.biasH) { hr += a.biasH[j]; hz += a.biasH[H + j]; hn += a.biasH[2 * H + j]; }
    const float rg = sigmoidf_(ir + hr);
    const float zg = sigmoidf_(iz + hz);
    const float ng = tanhf(in_ + rg * hn);
    const float hp = a.h_prev[so];
    const float hy = ng + zg * (hp - ng);
    a.h_out[so] = hy;
    if (a.x_out) a.x_out[so] = (a.x_res ? a.x_res[so] : 0.f) + hy;
  } else {
    // torch LSTMCell (gate order i,f,g,o).  tacotron.py:62-63,112-125
    float gi = sx[0], gf = sx[1], gg = sx[2], go = sx[3];
    if (a.biasX) { gi += a.biasX[j]; gf += a.biasX[H + j]; gg += a.biasX[2 * H + j]; go += a.biasX[3 * H + j]; }
    if (a.biasH) { gi += a.biasH[j]; gf += a.biasH[H + j]; gg += a.biasH[2 * H + j]; go += a.biasH[3 * H + j]; }
    gi = sigmoidf_(gi); gf = sigmoidf_(gf); gg = tanhf(gg); go = sigmoidf_(go);
    const float cy = gf * a.c_prev[so] + gi * gg;
    const float hy = go * tanhf(cy);
    a.c_out[so] = cy;
    a.h_out[so] = hy;
    if (a.x_out) a.x_out[so] = (a.x_res ? a.x_res[so] : 0.f) + hy;
  }
}

void pack_rowtile(const float* rows, int n_live_rows, int K, int RL, std::vector<float>* out) {
  const int per_tile = 4 * RL;
  const int n_mt = (n_live_rows + per_tile - 1) / per_tile;
  const int nkb = K / 16;
  out->assign((size_t)n_mt * nkb * per_tile * 16, 0.f);
  for (int mt = 0; mt < n_mt; ++mt)
    for (int kb = 0; kb < nkb; ++kb)
      for (int r = 0; r < per_tile; ++r) {
        const int row = mt * per_tile + r;
        if (row >= n_live_rows) continue;
        float* dst = out->data() + (((size_t)mt * nkb + kb) * per_tile + r) * 16;
        const float* src = rows + (size_t)row * K + kb * 16;
        for (int q = 0; q < 16; ++q) dst[q] = src[q];
      }
}

void cell_rows(const float* w_ih, int kx, int ldx, const float* w_hh, int kh, int H, int G,
               std::vector<float>* rows) {
  const int K = kx + kh;
  rows->assign((size_t)H * G * K, 0.f);
  for (int j = 0; j < H; ++j)
    for (int g = 0; g < G; ++g) {
      float* dst = rows->data() + ((size_t)j * G + g) * K;
      memcpy(dst, w_ih + (size_t)(g * H + j) * ldx, sizeof(float) * kx);
      memcpy(dst + kx, w_hh + (size_t)(g * H + j) * kh, sizeof(float) * kh);
    }
}

int rnn_launch(int epi, const RnnK& k, hipStream_t s) {
  MB_REQUIRE(k.N >= 1 && k.units >= 1 && k.nseg >= 1 && k.nseg <= 4, "rnn_launch: bad shape");
  constexpr int NW = 8;
  const int n_mt = (epi == EPI_LINEAR) ? cdiv(k.units, 16) : cdiv(k.units, 4);
  dim3 grid(n_mt, cdiv(k.N, 16));
  if (epi == EPI_LINEAR) hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_LINEAR, NW>), grid, dim3(NW * 64), 0, s, k);
  else if (epi == EPI_GRU) hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_GRU, NW>), grid, dim3(NW * 64), 0, s, k);
  else hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_LSTM, NW>), grid, dim3(NW * 64), 0, s, k);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

}  // namespace mb
