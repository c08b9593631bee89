// The product loop shared by the time-major split kernels (resblock_pair_split.hip, conv_split_tm.hip): error-compensated fp16 MFMA
// products of a tap over LDS-resident hi / scaled-lo B planes and a register ring of {hi, lo} weight fragments.
// The including kernel defines, with these names: constexpr KB, MT, NTW, TD (= 2); h16x8 ring[TD][KB][MT][2]; const h16x8* wp[MT];
// int ftn; auto sp_wrap (a callable: the stream wrapped -- empty, or the resident conv kernel's switch to the next channel group);
// const int NFT, ntaps; f32x16 acc[MT][NTW], acl[MT][NTW]; constexpr bool DUAL; const h16 k2m11 = 2^-11; and SPAIR_DBG (0 in the
// product).  DUAL (instances with one M tile per wave, where the registers allow a second accumulator set): the product of the scaled
// low activation half runs on the plain high weight fragment into acl, and the epilogue adds 2^-11 acl -- the wave no longer makes the
// third weight image (4 v_pk_mul_f16 per k-step beside 6-12 MFMAs).
#pragma once
#include "common.h"

namespace mb {
typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

}  // namespace mb

#ifndef SPAIR_DBG
#define SPAIR_DBG 0
#endif

// one tap: KB k-steps from ring slot S; refills the slot with flat tap ftn AFTER the MFMAs that read it.  The B fragments (LDS, hi and
// lo planes LO_ halves apart) run one k-step ahead of the MFMAs (bhc_ / blc_ = current); NEXT = first row of the next tap (or any
// valid row after the chunk's last tap: that read is discarded).  The three products of a k-step run product-major over the wave's
// MT x NTW accumulators: consecutive MFMAs never wait on each other's result.
#define SP_TAP(S, BPTR, NEXT, RS, LO_)                                                             \
  do {                                                                                             \
    const h16* bp_ = (BPTR);                                                                       \
    const h16* np_ = (NEXT);                                                                       \
    const size_t nf_ = (size_t)ftn * KB;                                                           \
    _Pragma("unroll") for (int u = 0; u < KB; ++u) {                                               \
      h16x8 bhn_[NTW], bln_[NTW], ws_[MT];                                                         \
      const h16* rp_ = u + 1 < KB ? bp_ + (u + 1) * 16 : np_;                                      \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                            \
        if (SPAIR_DBG & 2) { bhn_[n] = bhc_[n]; bln_[n] = blc_[n]; continue; }                     \
        bhn_[n] = *reinterpret_cast<const h16x8*>(rp_ + n * 32 * (RS));                            \
        bln_[n] = *reinterpret_cast<const h16x8*>(rp_ + (LO_) + n * 32 * (RS));                    \
      }                                                                                            \
      if (!DUAL) { _Pragma("unroll") for (int i = 0; i < MT; ++i) ws_[i] = ring[S][u][i][0] * k2m11; } \
      if (SPAIR_DBG & 32) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                             \
          _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                        \
            acc[i][n][0] += (float)ring[S][u][i][1][0] * (float)bhc_[n][0] + (float)ws_[i][0] * (float)blc_[n][0]; \
            acc[i][n][1] += (float)ring[S][u][i][0][0];                                            \
          }                                                                                        \
      } else {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                            \
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i][1], bhc_[n], acc[i][n], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                          \
          if (DUAL) acl[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i][0], blc_[n], acl[i][n], 0, 0, 0); \
          else acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ws_[i], blc_[n], acc[i][n], 0, 0, 0); \
        }                                                                                          \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                            \
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i][0], bhc_[n], acc[i][n], 0, 0, 0); \
      }                                                                                            \
      if (!(SPAIR_DBG & 1)) {                                                                      \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                             \
          _Pragma("unroll") for (int p = 0; p < 2; ++p) ring[S][u][i][p] = wp[i][((nf_ + u) * 2 + p) * 64]; \
      }                                                                                            \
      /* issue order: the next k-step's B fragments go out ONE behind each of the first 2 NTW MFMAs (left to itself the compiler */ \
      /* issued them behind the last MFMA and waited for the first of them at once: an LDS round trip exposed per k-step -- 39 */ \
      /* cycles per MFMA at 256 channels, 90 at 64, against 32), then the rest of the MFMAs, then the ring refills */ \
      if (!(SPAIR_DBG & 64)) {                                                                     \
        _Pragma("unroll") for (int m = 0; m < 3 * MT * NTW; ++m) {                                 \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
          if (m < 2 * NTW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        }                                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * MT, 0);                                    \
      }                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n) { bhc_[n] = bhn_[n]; blc_[n] = bln_[n]; }    \
    }                                                                                              \
    if (ftn + 1 == NFT) { ftn = 0; sp_wrap(); } else ++ftn;  /* (sp_wrap: the includer's hook at the end of the circular stream) */ \
  } while (0)

// a chunk = ntaps taps (ntaps odd); S0 = ring slot of its first tap
#define SP_TAPJ(S, J)                                                                              \
  SP_TAP(S, cb_ + (size_t)(J) * ts_, cb_ + (size_t)((J) + 1 < ntaps ? (J) + 1 : 0) * ts_, rs_, lo_)
#define SP_CHUNK(S0, BASE, RS, TAPSTEP, LOFF)                                                      \
  do {                                                                                             \
    const h16* cb_ = (BASE);                                                                       \
    const int rs_ = (RS);                                                                          \
    const size_t ts_ = (size_t)(TAPSTEP);                                                          \
    const int lo_ = (LOFF);                                                                        \
    h16x8 bhc_[NTW], blc_[NTW];                                                                    \
    _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                              \
      bhc_[n] = *reinterpret_cast<const h16x8*>(cb_ + n * 32 * rs_);                               \
      blc_[n] = *reinterpret_cast<const h16x8*>(cb_ + lo_ + n * 32 * rs_);                         \
    }                                                                                              \
    int j_ = 0;                                                                                    \
    if (S0 == 1) { SP_TAPJ(TD - 1, 0); j_ = 1; }                                                   \
    for (; j_ + 1 < ntaps; j_ += 2) {                                                              \
      SP_TAPJ(0, j_);                                                                              \
      SP_TAPJ(TD - 1, j_ + 1);                                                                     \
    }                                                                                              \
    if (S0 == 0) SP_TAPJ(0, ntaps - 1);                                                            \
  } while (0)
