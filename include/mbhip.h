/*
 * mbhip.h -- C ABI of libmbhip.so: the MI355X (gfx950) native hot path of
 * babysor/MockingBird (mel synthesis + waveform vocoding + monotonic_align).
 *
 * The reference has no FFI layer of its own: its boundary is a
 * set of Python callables.  Every entry point below is what a binding for
 * that boundary would call; the reference interface each one replaces is
 * cited as file:line relative to the MockingBird repository.
 *
 * Conventions
 *  - All `d_*` pointers are DEVICE pointers (HBM, fp32 unless stated); all
 *    `h_*` pointers are HOST pointers.  No torch types cross this boundary.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    work is enqueued asynchronously on it; nothing here synchronises the
 *    device except the *_create functions (weight upload) and where stated.
 *  - Functions return 0 on success, a negative MB_E* code on failure;
 *    mb_last_error() returns a thread-local description.
 *  - Handles own their device weights (hipMalloc); workspaces are caller
 *    owned so the caller's allocator (e.g. torch's caching allocator) decides
 *    placement.  Query the size first.
 *  - A handle is single-threaded: the loop handles (mb_wavernn, mb_taco, mb_ppg2mel) keep mutable
 *    stream / event / hipGraph state, so concurrent calls need one handle each.  When a call that
 *    already queued work on the handle's own streams fails, it drains those streams before it
 *    returns the error, so the caller may release the buffers it passed.
 */
#ifndef MBHIP_H
#define MBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MB_OK 0
#define MB_EINVAL (-1)   /* bad argument / shape */
#define MB_EHIP (-2)     /* HIP runtime error (see mb_last_error) */
#define MB_ENOMEM (-3)   /* workspace too small / allocation failed */
#define MB_ESTATE (-4)   /* handle not usable */

typedef void* mb_stream_t;

const char* mb_last_error(void);
/* Version of this header's contract; bumped whenever the MEANING of an existing argument or the layout of a packed image
 * changes (new symbols alone do not bump it).  Callers compare mb_abi_version() with the MB_ABI_VERSION they were built
 * against (mockingbird_amd/_lib.py refuses a library that answers something else).
 *   1: rounds 1-2.
 *   2: mb_taco_config.dropout = 0 means the reference's 0.5 and a negative value disables dropout (was: 0 disables);
 *      mb_conv1d_pack images carry the fp16 hi / lo fragments behind the fp32 ones; the uniform draws of the on-device RNGs
 *      are centres of 2^23 cells (the same seed gives other dropout masks / samples than version 1).
 *   3: mb_wavernn_loop_path takes one `resident` argument (MBHIP_WAVERNN_RESIDENT) in place of the three env_* ones; the library reads
 *      19 environment switches instead of 52 (DESIGN.md "Run-time switches": renamed / merged / moved under MBHIP_DIAG). */
#define MB_ABI_VERSION 4
int mb_abi_version(void);
/* Host-logic hook (tests/test_host_logic.py): the library's view of MBHIP_DIAG="key=value,key,..." -- the one variable behind every
 * diagnostic, A/B knob and test hook (DESIGN.md "Run-time switches").  Copies the value of `key` (a bare key reads as "1") into
 * out[0..n) and returns its length, or -1 when the key is absent.  No reference counterpart. */
int mb_diag_lookup(const char* key, char* out, int n);

/* ------------------------------------------------------------------------
 * 1. Conv1d / ConvTranspose1d primitive (fp32 MFMA implicit GEMM).
 *    Building block of every conv stack on the path; exported so the parity
 *    tests can exercise it per layer.
 *    Replaces: torch.nn.Conv1d / ConvTranspose1d as used in
 *      models/vocoder/hifigan/models.py:11-48,120-123,134-150
 *      models/vocoder/fregan/generator.py:11-52,98-120,137-166
 *      models/synthesizer/models/sublayer/common/batch_norm_conv.py:4-14
 *      models/vocoder/wavernn/models/fatchord_version.py:9-44
 * ---------------------------------------------------------------------- */

/* Number of floats of the packed weight image: the fp32 A fragments of the exact fp32-input MFMA kernel, a 64-float header
 * and the fp16 hi / lo A fragments of the error-compensated fp16 MFMA kernel (conv1d.hip; the default from 16 input channels
 * up, MBHIP_CONV_SPLIT=0 selects the fp32-input kernel). */
size_t mb_conv1d_packed_floats(int c_out, int c_in, int ksize, int up);

/* Pack torch-layout weights on the host.
 *  transposed == 0: h_w is Conv1d weight   [c_out][c_in][ksize], up must be 1
 *  transposed == 1: h_w is ConvTranspose1d [c_in][c_out][ksize], stride = up,
 *                   padding = pad (ksize must be a multiple of up). */
int mb_conv1d_pack(const float* h_w, int c_out, int c_in, int ksize, int up,
                   int transposed, int pad, float* h_packed);

typedef struct mb_conv1d_args {
  const float* d_x;       /* [B][c_in][t_in] (batch stride x_bstride floats)     */
  const float* d_wpacked; /* image from mb_conv1d_pack                            */
  const float* d_bias;    /* [c_out] or NULL                                      */
  const float* d_res;     /* residual, same layout as y, or NULL                  */
  const float* d_post_scale; /* [c_out] or NULL: y = act(.)*scale + shift (BN)   */
  const float* d_post_shift;
  float* d_y;             /* [B][c_out][t_out], or [B][t_out][c_out] if transpose_out */
  long long x_bstride, y_bstride, res_bstride;
  int batch, c_in, c_out, t_in, t_out;
  int ksize, dilation, pad; /* conv: padding; transposed conv: torch `padding`   */
  int up;                 /* 1 = Conv1d, >1 = ConvTranspose1d with stride up      */
  int in_act;             /* 0 none, 1 leaky_relu(in_slope) applied to x on load  */
  float in_slope;
  float in_scale;         /* x is multiplied by this before in_act (0 or 1.0 = none) */
  int out_act;            /* 0 none, 1 relu, 2 tanh, 3 sigmoid, 5 leaky_relu(out_slope), 4 highway:
                             y = g*relu(conv+b) + (1-g)*res with g = d_gate
                             (common/highway_network.py:12-17)                  */
  float out_scale;        /* result *= out_scale after the residual add (0 or 1.0 = none) */
  int accumulate;         /* y += result instead of y = result                    */
  int in_repeat;          /* >1: x is read through nearest-neighbour upsampling,
                             source index = t / in_repeat (t_in counts upsampled
                             positions); fregan/generator.py:104-110            */
  int transpose_out;      /* store y[b][t][c_out] (time-major)                    */
  const float* d_gate;    /* out_act 4 only: sigmoid gate, same layout as y       */
  /* ragged batches: item b is valid for its first d_valid[b] * valid_mul stored rows of x (int32 device array, NULL = all of
   * t_in); positions beyond read as zero padding, output tiles wholly beyond an item's valid output are skipped */
  const int* d_valid; int valid_mul;
  int down;               /* >1: strided Conv1d (torch `stride`), t_out = (t_in + 2*pad - dilation*(ksize-1) - 1)/down + 1;
                             needs up == 1, in_repeat <= 1, d_valid == NULL (models/ppg2mel/__init__.py:55-70) */
  float out_slope;        /* out_act 5: leaky_relu(out_slope) */
} mb_conv1d_args;

int mb_conv1d(const mb_conv1d_args* a, mb_stream_t stream);

/* Range diagnostics of the error-compensated path (no reference counterpart: the reference's fp32 Conv1d of
 * models/vocoder/hifigan/models.py:11-48 has no such limit).  With MBHIP_CONV_RANGE_CHECK=1 in the environment every launch
 * of that path counts the input values it stages with |x| > 65504 (the fp16 hi half saturates there) or NaN / Inf into one
 * word per device.  Returns the count of the current device (0 if the check never ran there), < 0 on a HIP error;
 * reset != 0 clears it.  Synchronises the device. */
long long mb_conv1d_range_events(int reset);

/* ------------------------------------------------------------------------
 * 1b. fp16 Conv1d / ConvTranspose1d primitive (fp16 storage, fp32 accumulate on
 *     v_mfma_f32_32x32x16_f16) -- the throughput path of the GAN vocoders
 *     (BASELINE configs[4]: "Fre-GAN vocoder fp16 with MFMA Conv1d").
 *     Activations are TIME-MAJOR [B][T][C] fp16 (channel contiguous).
 *     Replaces the same torch.nn.Conv1d / ConvTranspose1d call sites as
 *     section 1, run under .half():
 *      models/vocoder/hifigan/models.py:11-48,120-123,134-150
 *      models/vocoder/fregan/generator.py:11-52,98-120,137-166
 * ---------------------------------------------------------------------- */
size_t mb_conv1d_f16_packed_halves(int c_out, int c_in, int ksize, int up);
/* h_w: fp32 torch-layout weights (as mb_conv1d_pack) -> fp16 A-fragment image. */
int mb_conv1d_f16_pack(const float* h_w, int c_out, int c_in, int ksize, int up,
                       int transposed, int pad, uint16_t* h_packed);

typedef struct mb_conv1d_f16_args {
  const void* d_x;        /* fp16 [B][t_in/in_repeat][c_in], c_in % 8 == 0        */
  const void* d_wpacked;  /* image from mb_conv1d_f16_pack                        */
  const float* d_bias;    /* fp32 [c_out] or NULL                                 */
  const void* d_res;      /* fp16 residual, same layout as y, or NULL             */
  void* d_y;              /* fp16 [B][t_out][c_out] (fp32 if y_f32)               */
  long long x_bstride, y_bstride, res_bstride; /* in elements                     */
  int batch, c_in, c_out, t_in, t_out;
  int ksize, dilation, pad, up;  /* as mb_conv1d_args                             */
  int in_act;             /* 0 none, 1 leaky_relu(in_slope), 0 < slope < 1        */
  float in_slope;
  int out_act;            /* 0 none, 1 relu, 2 tanh                               */
  float out_scale;        /* result *= out_scale after the residual add (0 = 1.0) */
  int accumulate;         /* y += result                                          */
  int in_repeat;          /* nearest-neighbour upsampled read, as mb_conv1d_args  */
  int y_f32;              /* store y as fp32 (final conv_post -> waveform)        */
  const int* d_valid; int valid_mul;  /* ragged batches, as mb_conv1d_args */
} mb_conv1d_f16_args;

int mb_conv1d_f16(const mb_conv1d_f16_args* a, mb_stream_t stream);

/* Fused ResBlock unit (fp16 path):  y = x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2  with conv1
 * (k taps, dilation d, "same" padding) and conv2 (k taps, dilation 1), i.e. ONE iteration of
 *   ResBlock1.forward  models/vocoder/hifigan/models.py:39-46
 *                      models/vocoder/fregan/generator.py:43-50
 * in a single launch, the intermediate activation kept in LDS (resblock_f16.hip).  Optionally the
 * result is scaled and accumulated into y (the mean over the parallel ResBlocks, models.py:141-145):
 *   y = (accumulate ? y : 0) + out_scale * (x + conv2(...)).
 * Supported: channels in {16,32,64,128,256}, k odd >= 3 (mb_resblock_pair_f16_supported). */
int mb_resblock_pair_f16_supported(int channels, int ksize, int dilation);
size_t mb_resblock_pair_f16_packed_halves(int channels, int ksize);
/* h_w1, h_w2: fp32 torch Conv1d weights [C][C][k], weight norm folded -> one fp16 A-fragment stream */
int mb_resblock_pair_f16_pack(const float* h_w1, const float* h_w2, int channels, int ksize,
                              uint16_t* h_packed);
typedef struct mb_resblock_pair_f16_args {
  const void* d_x;        /* fp16 [B][t][channels]                                */
  void* d_y;              /* fp16 [B][t][channels], must not alias d_x            */
  const void* d_wpacked;  /* image from mb_resblock_pair_f16_pack                 */
  const float* d_b1;      /* fp32 [channels] bias of conv1                        */
  const float* d_b2;      /* fp32 [channels] bias of conv2                        */
  int batch, channels, t;
  int ksize, dilation;    /* conv1 dilation; conv2 has dilation 1                 */
  float slope;            /* leaky_relu slope in (0,1), applied before both convs */
  float out_scale;        /* 0 = 1.0                                              */
  int accumulate;
  const int* d_valid; int valid_mul;  /* ragged batches: item b has d_valid[b] * valid_mul positions (NULL = t) */
} mb_resblock_pair_f16_args;
int mb_resblock_pair_f16(const mb_resblock_pair_f16_args* a, mb_stream_t stream);

/* One upsampling stage's whole ResBlock group (fp16 path) in ONE launch:
 *   y = out_scale * sum_j ResBlock_j(x),  ResBlock_j = num_dilations units x <- x + conv2(lrelu(conv1_d(lrelu(x)) + b1)) + b2
 * i.e. the loop over self.resblocks of Generator.forward
 *   models/vocoder/hifigan/models.py:139-145, models/vocoder/fregan/generator.py:150-157
 * with ResBlock1.forward (hifigan/models.py:39-46, fregan/generator.py:43-50), the activations of a tile LDS-resident across
 * all units of all chains (resblock_stage_f16.hip): x is read once and y written once per stage.
 * Supported: channels in {16, 32, 64, 128}, 1..4 kernels (k odd >= 3), 1..4 dilations, the tile must fit LDS
 * (mb_resblock_stage_f16_supported).  At 16 / 32 channels the whole group is one launch; at 64 / 128 channels LDS holds short
 * windows (256 / 128 rows), so it pays for single ResBlocks with a small reach only (mb_resblock_stage_f16_efficiency = useful
 * rows / window rows): the others run mb_resblock_pair_f16 per unit, every launch accumulating into y. */
int mb_resblock_stage_f16_supported(int channels, int num_kernels, const int* ksizes, int num_dilations,
                                    const int* dilations /* [num_kernels][num_dilations] */);
float mb_resblock_stage_f16_efficiency(int channels, int num_kernels, const int* ksizes, int num_dilations,
                                       const int* dilations);
size_t mb_resblock_stage_f16_packed_halves(int channels, int num_kernels, const int* ksizes, int num_dilations);
/* h_w1[j * num_dilations + u], h_w2[...]: fp32 torch Conv1d weights [C][C][k_j] of convs1[u] / convs2[u] of ResBlock j */
int mb_resblock_stage_f16_pack(const float* const* h_w1, const float* const* h_w2, int channels, int num_kernels,
                               const int* ksizes, int num_dilations, uint16_t* h_packed);
typedef struct mb_resblock_stage_f16_args {
  const void* d_x;        /* fp16 [B][t][channels]                                          */
  void* d_y;              /* fp16 [B][t][channels], must not alias d_x                      */
  const void* d_wpacked;  /* image from mb_resblock_stage_f16_pack                          */
  const float* d_bias;    /* fp32 [num_kernels][num_dilations][2][channels]: b1, b2 per unit */
  int batch, channels, t;
  int num_kernels, num_dilations;
  int ksize[4];
  int dilation[4][4];     /* [kernel][unit]: dilation of convs1[unit]; convs2 have dilation 1 */
  float slope;            /* leaky_relu slope in (0,1)                                      */
  float out_scale;        /* 0 = 1 / num_kernels                                            */
  int accumulate;         /* y += result (a stage run as one launch per ResBlock)           */
  const int* d_valid; int valid_mul;  /* ragged batches, as mb_resblock_pair_f16_args */
} mb_resblock_stage_f16_args;
int mb_resblock_stage_f16(const mb_resblock_stage_f16_args* a, mb_stream_t stream);

/* The same ResBlock group at the reference's precision (round 4, resblock_stage_f32.hip): fp32 [B][channels][t] tensors in and out (the
 * reference's layout), fp32-grade results from error-compensated fp16 MFMA products on LDS-resident hi / lo operands (section 1's scheme),
 * residual chain and the kernels' mean in fp32 registers.  Replaces, on the fp32 path, the 2 x num_dilations x num_kernels Conv1d calls of
 * one stage of Generator.forward (models/vocoder/hifigan/models.py:139-145 with ResBlock1.forward :39-46; fregan/generator.py:150-157,
 * :43-50).  One launch covers the kernels / dilations the arguments name: gan.hip's default plan calls it once per ResBlock (k = 3, 7)
 * or per unit (k = 11) at 32 channels (256-row windows; the whole group in one launch only under MBHIP_DIAG=gan_s32_group), once per
 * ResBlock or unit at 64 channels (192-row windows), accumulating into y.  Same argument struct as mb_resblock_stage_f16 with d_x / d_y = fp32 [B][channels][t] and d_bias =
 * [num_kernels][num_dilations][2][channels] biases followed by the [num_kernels][num_dilations][2] unscale factors mb_resblock_stage_f32_pack
 * returns in h_unscale. */
int mb_resblock_stage_f32_supported(int channels, int num_kernels, const int* ksizes, int num_dilations, const int* dilations);
float mb_resblock_stage_f32_efficiency(int channels, int num_kernels, const int* ksizes, int num_dilations, const int* dilations);
size_t mb_resblock_stage_f32_packed_halves(int channels, int num_kernels, const int* ksizes, int num_dilations);
int mb_resblock_stage_f32_pack(const float* const* h_w1, const float* const* h_w2, int channels, int num_kernels, const int* ksizes,
                               int num_dilations, uint16_t* h_packed, float* h_unscale);
int mb_resblock_stage_f32(const mb_resblock_stage_f16_args* a, mb_stream_t stream);

/* Round 6 -- the fused ResBlock unit at the REFERENCE's precision on TIME-major fp32 tensors (resblock_pair_split.hip):
 *   y = (accumulate ? y : 0) + out_scale * (x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2)
 * = one (convs1[i], convs2[i]) iteration of ResBlock1.forward (models/vocoder/hifigan/models.py:39-46,
 * models/vocoder/fregan/generator.py:43-50; vits.py:251 builds the same block) in ONE launch: d_x / d_y are fp32 [B][t][channels]
 * (every staged load 16 bytes, no transposition), the intermediate h stays in LDS, the products are the error-compensated fp16 MFMA
 * triple of section 1 (w 2^s = wh + wl, a = ah + 2^-11 al: fp32-grade sums), the residual chain stays fp32.  This is what the fp32
 * generators' ResBlocks run on since round 6 (gan.hip: channel-major tensors are turned once per stage, mb_f32_cm_to_tm / _tm_to_cm).
 * Supported: channels in {16,32,64,128,256}, k odd >= 3, the tile must fit LDS (mb_resblock_pair_split_supported). */
int mb_resblock_pair_split_supported(int channels, int ksize, int dilation);
size_t mb_resblock_pair_split_packed_halves(int channels, int ksize);
/* h_w1, h_w2: fp32 torch Conv1d weights [C][C][k], weight norm folded -> one stream of {hi, lo} fp16 A fragments per 32-row output
 * tile in consumption order; h_unscale[2] = 2^-s of conv1 / conv2 (pass them back as unscale1 / unscale2) */
int mb_resblock_pair_split_pack(const float* h_w1, const float* h_w2, int channels, int ksize, uint16_t* h_packed, float* h_unscale);
typedef struct mb_resblock_pair_split_args {
  const float* d_x;       /* fp32 [B][t][channels]                                */
  float* d_y;             /* fp32 [B][t][channels], must not alias d_x            */
  const void* d_wpacked;  /* image from mb_resblock_pair_split_pack               */
  const float* d_b1;      /* fp32 [channels] bias of conv1                        */
  const float* d_b2;      /* fp32 [channels] bias of conv2                        */
  int batch, channels, t;
  int ksize, dilation;    /* conv1 dilation; conv2 has dilation 1                 */
  float slope;            /* leaky_relu slope in (0,1), applied before both convs */
  float out_scale;        /* 0 = 1.0                                              */
  float unscale1, unscale2; /* h_unscale[0], h_unscale[1] of the pack             */
  int accumulate;
  const int* d_valid; int valid_mul;  /* ragged batches: item b has d_valid[b] * valid_mul positions (NULL = t) */
} mb_resblock_pair_split_args;
int mb_resblock_pair_split(const mb_resblock_pair_split_args* a, mb_stream_t stream);
/* Round 6 -- every other conv of the fp32 generators on the same time-major tensors (conv_split_tm.hip): conv_pre, the ConvTranspose1d
 * upsamplers as three-tap polyphase convs whose [t][u C'] result IS the upsampled [u t][C'] tensor, Fre-GAN's cond_up / res_output,
 * conv_post (models/vocoder/hifigan/models.py:103-127,134-150; models/vocoder/fregan/generator.py:95-118,137-166):
 *   y[b][t][m] = act(out_scale * (sum_{ci,j} W[m][ci][j] lrelu(x[b][t - pad + j dilation][ci]) + bias[m] + res[b][t][m])) (+ y)
 * x fp32 [B][t][c_in], y / res fp32 [B][t][c_out]; error-compensated fp16 MFMA products, fp32-grade sums.  Supported: k odd,
 * (k - 1) dilation <= 80, c_in a multiple of 4 (mb_conv_split_tm_supported). */
int mb_conv_split_tm_supported(int c_out, int c_in, int ksize, int dilation);
size_t mb_conv_split_tm_packed_halves(int c_out, int c_in, int ksize);
/* h_w: fp32 [c_out][c_in][ksize] (torch Conv1d layout) -> {hi, lo} fp16 A-fragment streams; *h_unscale = 2^-s (pass back as unscale) */
int mb_conv_split_tm_pack(const float* h_w, int c_out, int c_in, int ksize, uint16_t* h_packed, float* h_unscale);
typedef struct mb_conv_split_tm_args {
  const float* d_x;       /* fp32 [B][t][c_in]                                      */
  float* d_y;             /* fp32 [B][t][c_out], must not alias d_x                 */
  const void* d_wpacked;  /* image from mb_conv_split_tm_pack                       */
  const float* d_bias;    /* fp32 [c_out] or NULL                                   */
  const float* d_res;     /* fp32 [B][t][c_out] added before the activation, or NULL (needs c_out % 4 == 0) */
  int batch, t, c_in, c_out;
  int ksize, dilation, pad;
  float in_slope;         /* leaky_relu slope in (0,1] applied to x; 1 = none       */
  float unscale;          /* *h_unscale of the pack                                 */
  float out_scale;        /* 0 = 1.0                                                */
  int out_act;            /* 0 none, 1 relu, 2 tanh, 3 sigmoid, 4 highway: y = g relu(v) + (1 - g) res with g = d_gate (common/highway_network.py:12-17) */
  int accumulate;         /* y += result (needs c_out % 4 == 0)                     */
  const int* d_valid; int valid_mul;  /* ragged batches: item b has d_valid[b] * valid_mul rows (NULL = t) */
  const float* d_gate;    /* fp32 [B][t][c_out], out_act 4 only                     */
  const float* d_post_scale; const float* d_post_shift;  /* fp32 [c_out] affine behind the activation (BatchNorm after ReLU,
                                                            common/batch_norm_conv.py:11-14), or NULL */
  long long x_bstride;    /* floats between batch items of x (0 = t * c_in)         */
  int x_row_stride;       /* floats between rows of x (0 = c_in): e.g. a [t][B][c] scan output read as B items */
  int x_split;            /* d_x is a SPLIT tensor -- fp16 [B][t][hi | lo][c_in], hi = fp16(x), lo = fp16((x - hi) 2^11): what d_ysplit of a
                             producer, mb_maxpool2_tm or mb_highway_tm wrote -- and is staged by copy (c_in % 8 == 0, > 32, in_slope 1) */
  void* d_ysplit;         /* also write the result as such a split tensor [B][t][hi | lo][c_out], or NULL */
} mb_conv_split_tm_args;
int mb_conv_split_tm(const mb_conv_split_tm_args* a, mb_stream_t stream);
/* MaxPool1d(2, stride 1, padding 1)[:t] over time (sublayer/cbhg.py:20,61-62) on fp32 [B][t][channels]: y[t] = max(x[t-1], x[t]), as
 * fp32 (d_y) and / or as a split tensor (d_ysplit); one of them may be NULL */
int mb_maxpool2_tm(const float* d_x, float* d_y, void* d_ysplit, int batch, int t, int channels, mb_stream_t stream);
/* A Conv1d to ONE output channel on a time-major fp32 tensor (the generators' conv_post: models/vocoder/hifigan/models.py:146-148,
   fregan/generator.py:161-163, vits.py:296-297): d_y[b][t] = act(bias + sum_{j, c} d_w[j][c] * lrelu(d_x[b][t - pad + j * dilation][c])),
   'same' padding, exact fp32 FMAs; d_w is [ksize][c_in] fp32 on the device (tap-major), out_act 0 none / 2 tanh, in_slope in (0, 1]
   (1 = no activation); d_valid / valid_mul as in mb_conv_split_tm (rows beyond an item's length read as zeros and are not written). */
int mb_conv_c1_tm(const float* d_x, const float* d_w, float bias, float* d_y, int batch, int t, int c_in, int ksize, int dilation,
                  int pad, float in_slope, int out_act, const int32_t* d_valid, int valid_mul, mb_stream_t stream);
/* Highway combine (common/highway_network.py:12-17): d_hg fp32 [rows][2 channels] = (W1 x + b1 | W2 x + b2) of one conv launch, d_x fp32
 * [rows][channels] -> d_y = g relu(h) + (1 - g) x with g = sigmoid(W2 x + b2), also as a split tensor (d_ysplit, may be NULL) */
int mb_highway_tm(const float* d_hg, const float* d_x, float* d_y, void* d_ysplit, long long rows, int channels, mb_stream_t stream);
/* fp32 [B][channels][t] (the reference's layout) <-> fp32 [B][t][channels] (the layout above), out of place */
int mb_f32_cm_to_tm(const float* d_x, float* d_y, int batch, int channels, int t, mb_stream_t stream);
int mb_f32_tm_to_cm(const float* d_x, float* d_y, int batch, int channels, int t, mb_stream_t stream);

/* Layout/precision converters between the reference's [B][C][T] fp32 tensors and the
 * time-major fp16 activations above (mel upload; tests). */
int mb_f32_to_f16_tm(const float* d_x, void* d_y, int batch, int channels, int t, mb_stream_t stream);
int mb_f16_tm_to_f32(const void* d_x, float* d_y, int batch, int channels, int t, mb_stream_t stream);

/* ------------------------------------------------------------------------
 * 2. GAN vocoders: HiFi-GAN and Fre-GAN generator forward.
 *    Replaces: Generator.forward  models/vocoder/hifigan/models.py:134-150
 *              FreGAN.forward     models/vocoder/fregan/generator.py:137-166
 *    behind infer_waveform        models/vocoder/{hifigan,fregan}/inference.py:60-74
 * ---------------------------------------------------------------------- */
#define MB_GAN_HIFIGAN 0
#define MB_GAN_FREGAN 1
#define MB_GAN_MAX_UPS 8
#define MB_GAN_MAX_KERNELS 4
#define MB_GAN_MAX_DIL 4

typedef struct mb_gan_config {
  int kind;                 /* MB_GAN_HIFIGAN | MB_GAN_FREGAN                    */
  int num_mels;             /* 80                                                */
  int upsample_initial_channel;
  int num_upsamples;
  int upsample_rates[MB_GAN_MAX_UPS];
  int upsample_kernel_sizes[MB_GAN_MAX_UPS];
  int num_kernels;          /* resblocks per stage                               */
  int resblock_kernel_sizes[MB_GAN_MAX_KERNELS];
  int num_dilations;        /* convs1/convs2 pairs per resblock (3 or 4)         */
  int resblock_dilations[MB_GAN_MAX_KERNELS][MB_GAN_MAX_DIL];
  int top_k;                /* Fre-GAN only (generator.py:80), default 4         */
  int interp_ups;           /* HiFi-GAN h.sampling_rate == 24000 (models.py:107-118): each
                               ups[i] is nearest-interpolate x u (InterpolationBlock,
                               models.py:74-91) + Conv1d(k, padding (k-1)//2) instead of a
                               ConvTranspose1d; weight i+1 is then a Conv1d weight
                               [C_out][C_in][k] (state name ups.i.1).  An even k makes the
                               stage one sample short (T*u - 1), as the reference's does:
                               size d_wav with mb_gan_out_samples.                       */
  int resblock_type;        /* 0 / 1 = ResBlock1 (three (convs1[d], convs2[d]) units per block), 2 = ResBlock2 (h.resblock == '2',
                               models.py:51-72,100; vits.py:251): two units x <- x + conv_d(lrelu(x)) with the block's first two
                               dilations; its weights are resblocks[i].convs[0..1] in place of convs1 / convs2 (ABI 4)             */
} mb_gan_config;

typedef struct mb_gan mb_gan;

/* Number of weight tensors mb_gan_create expects and, for tensor i, its
 * element count.  Tensors are the reference module's parameters with
 * weight_g/weight_v folded to `weight` (what remove_weight_norm() leaves,
 * hifigan/models.py:152-162), each conv as (weight, bias), in this order:
 *   conv_pre, ups[0..], [fregan: cond_up[0..], res_output[0..].1],
 *   resblocks[0..]: convs1[0..] then convs2[0..] (resblock_type 2: convs[0], convs[1]), conv_post.
 * See mockingbird_amd/weights.py:gan_weight_list. */
int mb_gan_num_weights(const mb_gan_config* cfg);
size_t mb_gan_weight_numel(const mb_gan_config* cfg, int index);

int mb_gan_create(const mb_gan_config* cfg, const float* const* h_weights,
                  int n_weights, mb_gan** out);
/* Same, choosing the arithmetic of the conv stacks: MB_F32 = fp32 MFMA (the 1e-4 RMS parity
 * path, what mb_gan_create builds), MB_F16 = fp16 storage / fp32 accumulate (section 1b; parity
 * gate: relative RMS <= 5e-3).  mb_gan_forward's signature is the same for
 * both: fp32 mel in, fp32 waveform out. */
#define MB_F32 0
#define MB_F16 1
int mb_gan_create_ex(const mb_gan_config* cfg, const float* const* h_weights,
                     int n_weights, int dtype, mb_gan** out);
int mb_gan_dtype(const mb_gan* g);
void mb_gan_destroy(mb_gan* g);
int mb_gan_hop(const mb_gan* g);                 /* product of upsample rates   */
/* Samples per utterance mb_gan_forward writes for `frames` mel frames: frames*hop, except
 * for interp_ups configs with even upsample kernels (see mb_gan_config.interp_ups). */
long long mb_gan_out_samples(const mb_gan* g, int frames);
size_t mb_gan_workspace_bytes(const mb_gan* g, int batch, int frames);
/* d_mel [batch][num_mels][frames] -> d_wav [batch][mb_gan_out_samples(g, frames)] */
int mb_gan_forward(const mb_gan* g, const float* d_mel, int batch, int frames,
                   float* d_wav, void* d_workspace, size_t workspace_bytes,
                   mb_stream_t stream);
/* As mb_gan_forward, plus an optional per-utterance channel bias added to conv_pre's output:
 * d_chan_bias fp32 [batch][upsample_initial_channel] or NULL.  With kind = MB_GAN_HIFIGAN and num_mels =
 * the latent width this is the VITS decoder, Generator.forward models/synthesizer/models/vits.py:273-291
 * (x = conv_pre(x) + cond(g); cond(g) is constant over time), conv_post's missing bias passed as zeros. */
/* Ragged batch: item b has d_frames[b] <= frames mel frames (int32 device array); d_mel is zero- (or anything-) padded to
 * [batch][num_mels][frames], d_wav is [batch][mb_gan_out_samples(g, frames)] and item b's first mb_gan_out_samples(g,
 * d_frames[b]) samples equal its own single-utterance forward: every conv reads positions beyond an item's length as
 * its zero padding (the generators are not causal: merely zero-padding the mel changes the tail).  One launch sequence
 * for the whole batch.  Not for interp_ups configs (their stage lengths are not multiples of the frame count). */
int mb_gan_forward_ragged(const mb_gan* g, const float* d_mel, int batch, int frames, const int32_t* d_frames, float* d_wav,
                          const float* d_chan_bias, void* d_workspace, size_t workspace_bytes, mb_stream_t stream);
int mb_gan_forward_ex(const mb_gan* g, const float* d_mel, int batch, int frames, float* d_wav,
                      const float* d_chan_bias, void* d_workspace, size_t workspace_bytes, mb_stream_t stream);

/* ------------------------------------------------------------------------
 * 3. WaveRNN (fatchord) vocoder.
 *    Replaces: WaveRNN.generate  models/vocoder/wavernn/models/fatchord_version.py:153-257
 *    (UpsampleNetwork :60-85, fold_with_overlap :288-338, sample loop :190-234)
 *    behind infer_waveform       models/vocoder/wavernn/inference.py:45-64
 * ---------------------------------------------------------------------- */
typedef struct mb_wavernn_config {
  int rnn_dims, fc_dims, bits, pad;
  int n_upsample; int upsample_factors[4];
  int feat_dims, compute_dims, res_out_dims, res_blocks;
  int mode;  /* 0 = RAW (softmax over 2^bits classes, fatchord_version.py:222-228); 1 = MOL (30 fc3 outputs = 10 x (logit, mean,
              * log scale) of a discretised mixture of logistics, :213-220 + models/vocoder/distribution.py:87-123) */
} mb_wavernn_config;

typedef struct mb_wavernn mb_wavernn;

/* Weight tensors in state_dict() order of the reference WaveRNN module
 * (fatchord_version.py:88-122), BatchNorm as (weight,bias,running_mean,
 * running_var); see mockingbird_amd/weights.py:wavernn_weight_list. */
int mb_wavernn_num_weights(const mb_wavernn_config* cfg);
size_t mb_wavernn_weight_numel(const mb_wavernn_config* cfg, int index);
int mb_wavernn_create(const mb_wavernn_config* cfg, const float* const* h_weights,
                      int n_weights, mb_wavernn** out);
void mb_wavernn_destroy(mb_wavernn* w);

/* Geometry of one generate() call: `frames` mel frames (unpadded) produce
 * total_len = frames*prod(upsample_factors) conditioning positions, cut into
 * n_folds windows of seq_len steps (batched) or 1 x total_len (unbatched). */
typedef struct mb_wavernn_plan {
  int frames, total_len, n_folds, seq_len, fold_stride; /* fold_stride = target+overlap */
  size_t workspace_bytes;
} mb_wavernn_plan;
int mb_wavernn_plan_generate(const mb_wavernn* w, int frames, int batched, int target,
                             int overlap, mb_wavernn_plan* plan);

/* d_mel: [feat_dims][frames] (already divided by max_abs_value if the caller
 *        normalises, inference.py:60-61).
 * d_noise: NULL -> on-device counter RNG (seed); else Exp(1) draws
 *        [seq_len][n_folds][n_classes] consumed exactly like
 *        torch.multinomial(p,1) == argmax(p / noise).
 *        MOL mode: uniform(1e-5, 1 - 1e-5) draws [seq_len][n_folds][11]: per step and fold the 10 mixture-indicator
 *        draws, then the logistic draw -- the order sample_from_discretized_mix_logistic consumes them
 *        (models/vocoder/distribution.py:105,118).
 * d_samples: [n_folds][seq_len] float32 in [-1,1] (RAW: 2*k/(C-1)-1; MOL: the logistic sample), the tensor the
 *        reference stacks at fatchord_version.py:236.
 * d_logits_out: optional [seq_len][n_folds][n_classes] dump of fc3 outputs (tests).
 * d_forced: optional teacher forcing: [n_folds][seq_len] samples fed back
 *        instead of the drawn ones (tests).
 * h_progress: optional host-visible (pinned/host-coherent) int32 the kernels
 *        update with the number of completed steps (progress_callback support).
 * HOST-BLOCKING on the default path: for 1..96 fold columns the loop is ONE resident launch (wavernn_persist.h /
 *        wavernn_pipe16.h / wavernn_pipe.h) whose 192 / 224 workgroups must be co-resident; the call waits for the launch and
 *        reads its abort word (a lost hand-off -> the launch chain computes the utterance instead, and the device is remembered
 *        as unsuitable) and, for wf_pipe16_kernel, its range word (an activation beyond the operand pairs' |x| <= 65504 -> the
 *        fp32 launch chain computes the utterance), so it returns with the samples complete on `stream`.  The chain's stream
 *        equals the exact resident kernels' bit for bit but NOT wf_pipe16_kernel's (same noise, fp32 instead of fp32-grade sums:
 *        the two may pick differently at near-ties): mb_wavernn_last_path tells which form produced a call's samples.  Wider
 *        calls (the launch chain, hipGraph replays) are stream-asynchronous as before; MBHIP_WAVERNN_RESIDENT=0 selects the
 *        chain for every width. */
int mb_wavernn_generate(const mb_wavernn* w, const mb_wavernn_plan* plan,
                        const float* d_mel, const float* d_noise, uint64_t seed,
                        float* d_samples, float* d_logits_out, const float* d_forced,
                        int* h_progress, void* d_workspace, size_t workspace_bytes,
                        mb_stream_t stream);
/* time (ms) the sample-loop kernels of the LAST generate call took, measured
 * with hipEvents on `stream` (valid after the stream is synchronised), and the
 * number of kernel launches in it. */
/* Several utterances in ONE sample loop (additive; the reference vocodes one utterance per call): the folds
 * of all utterances are the columns of the same 5 launches per step, amortising the launch latency that
 * bounds a single utterance.  Production chain only (Philox sampling); batched fold geometry
 * (fold_with_overlap, fatchord_version.py:288-338) per utterance; utterance u draws exactly the noise it
 * would draw alone with seed h_seeds[u]; its samples equal those of mb_wavernn_generate(seed = h_seeds[u]) on the same form of the loop
 * and, across forms (a wide batch multiplies operand pairs on the fp16 pipe, rnn_ts3_body.h; a single utterance runs wf_pipe16_kernel), up to
 * picks at provable near-ties.  Above 64 fold columns the call is HOST-BLOCKING: the range word of the operand-pair GEMMs is read behind the
 * loop (hipStreamSynchronize) and a raised word reruns the loop on the fp32 instances (mb_wavernn_last_path: MB_WRN_FALLBACK_RANGE).
 * h_fold_offsets [n_utt+1] (out of the plan call): utterance u owns rows [off[u], off[u+1]) of
 * d_samples [n_folds][seq_len].  h_d_mels: HOST array of n_utt DEVICE pointers, mel u = fp32 [feat][frames[u]]. */
typedef struct mb_wavernn_batch_plan {
  int n_utt, n_folds, seq_len, fold_stride;
  size_t workspace_bytes;
} mb_wavernn_batch_plan;
int mb_wavernn_plan_generate_batch(const mb_wavernn* w, int n_utt, const int* h_frames, int target, int overlap,
                                   mb_wavernn_batch_plan* plan, int* h_fold_offsets);
int mb_wavernn_generate_batch(const mb_wavernn* w, const mb_wavernn_batch_plan* plan, const int* h_frames,
                              const float* const* h_d_mels, const uint64_t* h_seeds, float* d_samples,
                              void* d_workspace, size_t workspace_bytes, mb_stream_t stream);

/* Device post-processing of the generated folds, float64 (wavernn_post.hip).
 * Replaces the numpy tail of WaveRNN.generate  models/vocoder/wavernn/models/fatchord_version.py:236-257:
 *   xfade_and_unfold :340-402 (if batched), decode_mu_law audio.py:102-107 (if mu_law),
 *   de_emphasis audio.py:92-93 (if apply_preemphasis), output[:wave_len], linear fade over fade_len samples.
 * d_samples: fp32 [n_folds][seq_len] as written by mb_wavernn_generate; d_wav: float64, capacity >=
 * min(wave_len, unfolded length); *out_len receives that length.  Fails (MB_EINVAL) when the waveform is
 * shorter than the fade window, where the reference raises (mels < 26 frames). */
size_t mb_wavernn_finish_workspace_bytes(int n_folds, int seq_len, int batched, int overlap);
int mb_wavernn_finish(const float* d_samples, int n_folds, int seq_len, int batched, int overlap,
                      int n_classes, int mu_law, int apply_preemphasis, double preemphasis,
                      int wave_len, int fade_len, double* d_wav, int* out_len, void* d_workspace,
                      size_t workspace_bytes, mb_stream_t stream);
/* Host-logic hook (no GPU needed; tests/test_host_logic.py): which form of the sample loop mb_wavernn_generate runs -- the one table
 * the library itself consults (csrc/wavernn.hip wavernn_pick_path).  No reference counterpart: WaveRNN.generate
 * (models/vocoder/wavernn/models/fatchord_version.py:153-235) is a Python loop.
 *   columns: fold columns (1 = batched=False); mode: 0 RAW / 1 MOL; production: 1 = production-dims model sampling on the device
 *   (no injected noise, forced samples, logits dump, trace); have_q16: the K = 512 split weight images exist; resident_cus: compute
 *   units a resident launch may count on; dev_failed: a resident launch lost a hand-off on this device before;
 *   resident: MBHIP_WAVERNN_RESIDENT as an int: -1 unset / "auto", 0 = no resident launch, 1 = resident wherever legal (also on a
 *   device that failed before), 2 = "exact": like 1 with the exact fp32 kernel instead of the operand-pair one. */
#define MB_WRN_PATH_CHAIN 0     /* the 5-launch chain (csrc/wavernn_fast.h), hipGraph replays                          */
#define MB_WRN_PATH_PERSIST1 1  /* one column, resident = "exact" (RAW): wf_persist1_kernel (csrc/wavernn_persist.h)     */
#define MB_WRN_PATH_PIPE 2      /* 2..32 columns, exact fp32 MFMA: wf_pipe_kernel (csrc/wavernn_pipe.h); resident = "exact" */
#define MB_WRN_PATH_PIPE16 3    /* 1..96 columns (MOL: 1..64), 22-bit operand pairs: wf_pipe16_kernel (csrc/wavernn_pipe16.h) */
int mb_wavernn_loop_path(int columns, int mode, int production, int have_q16, int resident_cus, int dev_failed, int resident);
/* Which form of the loop produced the samples of the LAST mb_wavernn_generate call on this handle (*path: MB_WRN_PATH_*), and
 * whether a resident launch was discarded on the way (*fallback): a caller that needs a reproducible stream per (mel, seed) checks
 * that the path is the one it expects (the chain's stream differs from wf_pipe16_kernel's at near-ties).  No reference counterpart. */
#define MB_WRN_FALLBACK_NONE 0   /* the planned form ran                                                                  */
#define MB_WRN_FALLBACK_ABORT 1  /* a resident launch lost a hand-off (workgroups not co-resident): the chain ran instead    */
#define MB_WRN_FALLBACK_RANGE 2  /* wf_pipe16_kernel met |x| > 65504 or a NaN in an exchange vector: the fp32 chain ran instead */
int mb_wavernn_last_path(const mb_wavernn* w, int* path, int* fallback);
/* Test hook: the Exp(1) noise the on-device sampler of the production paths draws for `seed`:
 * d_out [steps][folds][n_classes] = E for steps step0 .. step0+steps-1, i.e. exactly the tensor which, passed as d_noise
 * to the oracle's sample loop (argmax(softmax(l) / E), torch.multinomial's rule), reproduces what
 * mb_wavernn_generate(d_noise = NULL, seed) samples with its fused Gumbel-argmax (argmax(l - log E)).
 * For mb_wavernn_generate_batch, `folds` is utterance u's own fold count and seed = h_seeds[u]. */
int mb_wavernn_debug_noise(uint64_t seed, int step0, int steps, int folds, int n_classes, float* d_out, mb_stream_t stream);
/* MOL mode: d_out [steps][folds][nr_mix + 1] = the uniform(1e-5, 1 - 1e-5) draws of the on-device mixture sampler, in the layout
 * mb_wavernn_generate takes as d_noise for a MOL model. */
int mb_wavernn_debug_noise_mol(uint64_t seed, int step0, int steps, int folds, int nr_mix, float* d_out, mb_stream_t stream);
int mb_wavernn_last_loop_ms(const mb_wavernn* w, float* ms, int* launches);
/* Measurement hook (bench.py roofline leg): runs the real generate loop twice on its
 * stream, bracketed by hipEvents, with and without kernel `which` (0 rnn1 GRU, 1 rnn2 GRU,
 * 2 fc1, 3 fc2, 4 fc3); the loop's kernels run back to back, so the per-step difference is that
 * kernel's launch-to-launch duration.  Also reports that launch's algorithmic bytes
 * (weights once + vectors in/out + table rows, fp32).  `iters` is ignored. */
int mb_wavernn_bench_kernel(mb_wavernn* w, const mb_wavernn_plan* plan, const float* d_mel,
                            float* d_samples, void* d_workspace, size_t workspace_bytes,
                            int which, int iters, float* avg_us, double* algorithmic_bytes,
                            mb_stream_t stream);

/* ------------------------------------------------------------------------
 * 4. Tacotron decoder loop + CBHG postnet.
 *    Replaces: Decoder.forward loop  models/synthesizer/models/tacotron.py:71-138,264-275
 *              CBHG + post_proj      models/synthesizer/models/sublayer/cbhg.py:40-84,
 *                                    tacotron.py:281-283
 *    behind Synthesizer.synthesize_spectrograms models/synthesizer/inference.py:75-142
 * ---------------------------------------------------------------------- */
typedef struct mb_taco_config {
  int n_mels, project_dims, decoder_dims, lstm_dims, max_r, r;
  int postnet_dims, postnet_K, num_highways;
  int lsa_kernel, lsa_filters;
  /* optional text encoder (tacotron.py:11-44,255): present when has_encoder != 0 */
  int has_encoder, num_chars, embed_dims, encoder_dims, encoder_K, speaker_dims, style_dims;
  /* optional global style tokens (global_style_token.py:9-145, gst_hyperparameters.py): present when
   * has_gst != 0 (needs has_encoder); style_dims = E */
  int has_gst, gst_tokens, gst_heads, gst_n_convs, gst_width;
  int gst_filters[8];
  /* PreNet dropout probability (hparams.tts_dropout, pre_net.py:23,26: applied at inference too); kept values are
   * scaled by 1/(1-p).  0 (a zero-initialised struct) = the reference's 0.5; a negative value disables dropout. */
  float dropout;
} mb_taco_config;
typedef struct mb_taco mb_taco;
int mb_taco_num_weights(const mb_taco_config* cfg);
size_t mb_taco_weight_numel(const mb_taco_config* cfg, int index);
int mb_taco_create(const mb_taco_config* cfg, const float* const* h_weights,
                   int n_weights, mb_taco** out);
void mb_taco_destroy(mb_taco* t);
size_t mb_taco_workspace_bytes(const mb_taco* t, int batch, int t_text, int max_steps);
/* Runs the autoregressive decoder until the reference's batch-wide stop rule
 * fires (tacotron.py:275) or max_steps frames are produced, then the postnet.
 *  d_memory      [B][T][project_dims]  encoder_seq after speaker/GST concat
 *  d_memory_proj [B][T][decoder_dims]  encoder_proj(encoder_seq)
 *  d_chars       [B][T] int32 token ids (0 = padding, lsa.py:34)
 *  d_dropout     NULL -> on-device RNG(seed); else Bernoulli(0.5) keep masks
 *                [n_iter][2][B][2*decoder_dims] (1.0 keep / 0.0 drop) drawn in
 *                the reference's program order (pre_net.py:23,26)
 *  d_mel         [B][n_mels][max_steps] decoder mel_outputs (cat of frames)
 *  d_linear      [B][n_mels][max_steps] postnet output (what generate returns)
 *  d_attn        [B][max_steps/r][T]
 *  h_n_frames    host int: frames actually produced (synchronises the stream) */
int mb_taco_decode(const mb_taco* t, const float* d_memory, const float* d_memory_proj,
                   const int32_t* d_chars, int batch, int t_text, int max_steps,
                   float min_stop_token, const float* d_dropout, uint64_t seed,
                   float* d_mel, float* d_linear, float* d_attn, int* h_n_frames,
                   void* d_workspace, size_t workspace_bytes, mb_stream_t stream);

/* Duration of the decoder loop of the last mb_taco_decode call (HIP events on the loop's own stream: iterations
 * only, without the state initialisation in front of it and the postnet behind it) and the number of decoder
 * iterations it ran.  Production-dims handles only (the hipGraph-replayed loop); MB_ESTATE otherwise. */
int mb_taco_last_loop_ms(const mb_taco* t, float* ms, int* iterations);
/* Duration of the CBHG postnet + post_proj of the last mb_taco_decode call (tacotron.py:281-283; HIP events on the call's stream) --
 * what bench.py's tacotron.postnet_roofline prices (16.1 MFLOP per frame, MFMA-bound). */
int mb_taco_last_postnet_ms(const mb_taco* t, float* ms);

/* Launches per decoder iteration of the last mb_taco_decode call on the hipGraph-replayed loop:
 *   7 = one launch per stage, every product on the fp32 matrix pipe (also what a call falls back to when a hand-off of the fused launch
 *       times out or an operand leaves fp16's range, and what MBHIP_DIAG=taco_front=0 selects);
 *   5 = prenet fc2, attention GRU and attention as ROLES of one launch with tagged-granule hand-offs (taco_front_kernel: batch <= 32,
 *       text <= 192 symbols, 48 + 4 batch compute units) -- with fp32 products the 7-launch loop's bits;
 *   4 = additionally without the rnn_input launch (text <= 128 symbols): rnn_input, the next attention-GRU pre-activation and the stop
 *       logit's context half are linear in the context, so they are formed by the attention workgroups from a memory projected once per
 *       call (the workspace holds the projection); always with the fp16-pipe products below;
 *  -1 = no decode on that loop yet. */
int mb_taco_last_loop_form(const mb_taco* t);

/* Host logic only (no device access): THE selection of that loop's form -- (batch, text length, whether the fast attention kernel serves
 * the call (production dims, <= 192 symbols), compute units of the device, whether the handle has its fp16 images / projection weights,
 * the handle's lost-hand-off memo, MBHIP_DIAG taco_front / taco_f16 / taco_fold as -1 (unset) | 0 | 1) -> 4 | 5 | 7 launches per iteration,
 * *f16_out (may be null) = the fp16-pipe products.  mb_taco_decode calls the same function (csrc/tacotron.hip taco_pick_form; the table is
 * its header comment and tests/test_host_logic.py::test_taco_loop_form_table). */
int mb_taco_loop_form(int batch, int t_text, int lsa_fast, int n_cus, int images, int front_failed, int front_sw, int f16_sw, int fold_sw, int* f16_out);

/* 1 when the last mb_taco_decode call on that loop multiplied its K >= 1024 tiles (LSTM input halves, rnn_input, mel_proj / prenet fc1' /
 * stop rows, hidden halves) on the fp16 matrix pipe with error-compensated split operands (w 2^s = wh + wl, x = xh + 2^-11 xl: 22-bit
 * operands, fp32 accumulate -- the 4-launch form, and the 5-launch form with more than 16 utterances; a value beyond fp16's range makes
 * the call run again on the exact fp32 loop), 0 when every product ran on the fp32 pipe, -1 = no decode on that loop yet.  MBHIP_DIAG=taco_f16=0|1 overrides. */
int mb_taco_last_loop_f16(const mb_taco* t);

/* Text encoder + global style token + attention-memory assembly (the once-per-chunk front
 * half of Tacotron.forward, tacotron.py:234-255):
 *  d_chars   [B][T] int32, d_speaker [B][speaker_dims],
 *  style_idx 0..gst_tokens-1: that token's value vector, broadcast (tacotron.py:243-248);
 *            anything else: ReferenceEncoder(zeros) (+) speaker embedding -> multi-head attention
 *            over all tokens, per utterance (tacotron.py:249-251; gen_voice.py passes -1).
 *            The ReferenceEncoder output on the all-zero input is a constant of the checkpoint
 *            and is folded at mb_taco_create time.
 *  d_dropout NULL -> on-device RNG(seed); else keep masks [2][B][T][encoder_dims]
 *            (encoder PreNet, pre_net.py:23,26)
 *  -> d_memory [B][T][project_dims], d_memory_proj [B][T][decoder_dims].
 * Encoder weights follow the decoder/postnet list of mb_taco_create:
 *   encoder.embedding, encoder.pre_net.fc1/fc2 (w,b), encoder.cbhg (bank K x (conv,BN4),
 *   proj1, proj2, highways, rnn fwd/rev), encoder_proj.weight, then (has_gst)
 *   gst.encoder.convs[i] (weight,bias) + bns[i] (weight,bias,running_mean,running_var) per conv,
 *   gst.encoder.gru (weight_ih,weight_hh,bias_ih,bias_hh), gst.stl.embed,
 *   gst.stl.attention.W_query/W_key/W_value. */
size_t mb_taco_encode_workspace_bytes(const mb_taco* t, int batch, int t_text);
int mb_taco_encode(const mb_taco* t, const int32_t* d_chars, const float* d_speaker, int style_idx,
                   int batch, int t_text, const float* d_dropout, uint64_t seed,
                   float* d_memory, float* d_memory_proj, void* d_workspace, size_t workspace_bytes,
                   mb_stream_t stream);

/* ------------------------------------------------------------------------
 * 5. monotonic_align.maximum_path
 *    Replaces: maximum_path_c  monotonic_align/core.pyx:38-42 (and :7-33)
 *    d_values float32 [b][t_t][t_s] is mutated in place exactly like the
 *    Cython code mutates `value`; d_paths int32 [b][t_t][t_s] must be zeroed
 *    by the caller (monotonic_align/__init__.py:13 allocates zeros).
 * ---------------------------------------------------------------------- */
int mb_maximum_path(int32_t* d_paths, float* d_values, const int32_t* d_t_ys,
                    const int32_t* d_t_xs, int b, int t_t, int t_s, mb_stream_t stream);

/* ------------------------------------------------------------------------
 * 6. ppg2mel voice-conversion decoder: the autoregressive loop of
 *    Decoder.inference / inference_batched   models/ppg2mel/rnn_decoder_mol.py:267-374
 *    (DecoderPrenet :10-22, attention LSTMCell + MOLAttention models/ppg2mel/utils/mol_attention.py:67-122,
 *    decoder LSTMCell stack :200-209, linear_projection + stop_layer :281-288, stop rule :301-305/:349-354).
 *    Weights (fp32, torch layout), in this order: prenet.layers[i].weight (n_prenet, bias-free);
 *    attention_rnn weight_ih, weight_hh, bias_ih, bias_hh; attention_layer.query_layer.0 weight, bias;
 *    query_layer.2 weight, bias; per decoder layer weight_ih, weight_hh, bias_ih, bias_hh;
 *    linear_projection weight, bias; stop_layer weight, bias.
 * ---------------------------------------------------------------------- */
typedef struct mb_ppg2mel_config {
  int enc_dim, num_mels, frames_per_step, attention_rnn_dim, decoder_rnn_dim;
  int n_prenet, prenet_dims[4];
  int num_mixtures, encoder_down_factor, num_decoder_rnn_layer, concat_context_to_last;
} mb_ppg2mel_config;
typedef struct mb_ppg2mel mb_ppg2mel;
int mb_ppg2mel_num_weights(const mb_ppg2mel_config* cfg);
size_t mb_ppg2mel_weight_numel(const mb_ppg2mel_config* cfg, int index);
int mb_ppg2mel_create(const mb_ppg2mel_config* cfg, const float* const* h_weights, int n_weights, mb_ppg2mel** out);
void mb_ppg2mel_destroy(mb_ppg2mel* p);
size_t mb_ppg2mel_workspace_bytes(const mb_ppg2mel* p, int batch);
/* d_memory: fp32 [batch][t_enc][enc_dim].  Runs until every utterance's sigmoid(stop) > stop_threshold with at
 * least min_steps steps done, or max_steps (the reference: max = t_enc*down/r, min = max - 5).
 * d_dropout: optional prenet keep masks (0/1), layer l at offset sum_{k<l} max_steps*batch*prenet_dims[k],
 * [max_steps][batch][prenet_dims[l]] inside; NULL -> Philox(seed).  Outputs are per step, untruncated:
 * d_mel [batch][max_steps][frames_per_step*num_mels], d_align [batch][max_steps][t_enc],
 * d_stop [batch][max_steps] (logits); *h_n_steps = steps produced. */
/* mb_ppg2mel_decode is HOST-BLOCKING (it returns *h_n_steps): batch 1 runs the resident launch of ppg_resident.h, a batch of 2..32 the
 * resident launch of ppg_batch.h (round 5: Decoder.inference_batched, rnn_decoder_mol.py:317-374, on MFMA tiles; fp32-grade results, not
 * bit-identical to the chain's), and reads its abort / range words behind it (a lost hand-off or an activation beyond the operand
 * pairs' |x| <= 65504 -> the 6-launch chain redoes the batch); wider batches poll the stop flag.
 * The same holds for mb_taco_encode / mb_taco_decode's CBHG GRU scans (gru_scan.h): one stream synchronisation per CBHG. */
/* Duration of the decoder loop of the last mb_ppg2mel_decode call (HIP events on the loop's stream) and the steps it
 * produced; production-dims handles only (the graph-replayed step of ppg_fast.h), MB_ESTATE otherwise. */
int mb_ppg2mel_last_loop_ms(const mb_ppg2mel* p, float* ms, int* steps);
/* Kernel launches the decoder loop of the last mb_ppg2mel_decode call took: 1 = a resident loop (one utterance:
 * csrc/ppg_resident.h, the whole Decoder.inference loop, rnn_decoder_mol.py:267-316, as one launch; 2..32 utterances: csrc/ppg_batch.h), 6 per step on the
 * graph-replayed chain (csrc/ppg_fast.h).  Production-dims handles only, MB_ESTATE otherwise. */
int mb_ppg2mel_last_loop_launches(const mb_ppg2mel* p, int* launches);
int mb_ppg2mel_decode(const mb_ppg2mel* p, const float* d_memory, int batch, int t_enc, int max_steps,
                      int min_steps, float stop_threshold, const float* d_dropout, uint64_t seed,
                      float* d_mel, float* d_align, float* d_stop, int* h_n_steps, void* d_workspace,
                      size_t workspace_bytes, mb_stream_t stream);

/* 6b. The one-shot networks either side of that loop in MelDecoderMOLv2.inference
 *     (models/ppg2mel/__init__.py:166-192): encode = bnf_prenet + pitch_convs (:50-98; strided Conv1d, LeakyReLU(0.1),
 *     InstanceNorm1d), their sum, concat with F.normalize(spembs), reduce_proj -> decoder memory;
 *     postnet = mel + Postnet(mel) (utils/cnn_postnet.py:8-52, eval-mode BatchNorm folded on the host).
 *    h_weights (fp32 host, state_dict order): bnf_prenet.0.weight, .3.weight, .3.bias, .6.weight, .6.bias;
 *    pitch_convs.0.weight, .3.weight, .3.bias, .6.weight, .6.bias; reduce_proj.weight, .bias; per postnet layer
 *    conv.weight, conv.bias, BatchNorm weight, bias, running_mean, running_var. */
typedef struct mb_ppg2mel_net_config {
  int bnf_dim, spk_dim, enc_dim;   /* bottle_neck_feature_dim, spk_embed_dim, encoder_dim */
  int down0, down1;                /* encoder_downsample_rates */
  int num_mels, postnet_layers, postnet_dim, postnet_ksize;
} mb_ppg2mel_net_config;
typedef struct mb_ppg2mel_net mb_ppg2mel_net;
int mb_ppg2mel_net_num_weights(const mb_ppg2mel_net_config* cfg);
size_t mb_ppg2mel_net_weight_numel(const mb_ppg2mel_net_config* cfg, int index);
int mb_ppg2mel_net_create(const mb_ppg2mel_net_config* cfg, const float* const* h_weights, int n_weights,
                          mb_ppg2mel_net** out);
void mb_ppg2mel_net_destroy(mb_ppg2mel_net* p);
/* decoder memory length for t input frames (two strided convolutions), 0 if t is too short */
int mb_ppg2mel_net_t_enc(const mb_ppg2mel_net_config* cfg, int t);
size_t mb_ppg2mel_net_workspace_bytes(const mb_ppg2mel_net* p, int batch, int t);  /* t: frames of either call */
/* d_bnf fp32 [batch][t][bnf_dim], d_logf0_uv [batch][t][2], d_spk [batch][spk_dim] (not normalised)
 * -> d_memory [batch][t_enc][enc_dim], the layout mb_ppg2mel_decode takes. */
int mb_ppg2mel_net_encode(const mb_ppg2mel_net* p, const float* d_bnf, const float* d_logf0_uv, const float* d_spk,
                          int batch, int t, float* d_memory, void* d_workspace, size_t workspace_bytes,
                          mb_stream_t stream);
/* d_mel fp32 [batch][t][num_mels] (the decoder's frames) -> d_out = d_mel + postnet(d_mel), same layout. */
int mb_ppg2mel_net_postnet(const mb_ppg2mel_net* p, const float* d_mel, int batch, int t, float* d_out,
                           void* d_workspace, size_t workspace_bytes, mb_stream_t stream);

/* ----------------------------------------------------------------------
 * 7. Waveform wire format on device: the elementwise tail between the
 *    vocoder and the file / the rank-0 gather.  d_wav is fp32 (HiFi-GAN / Fre-GAN output) or float64
 *    (mb_wavernn_finish output); arithmetic is done in THAT type, as numpy does on the array it is given.
 * ---------------------------------------------------------------------- */
#define MB_F64 2
/* wav = wav / max|wav| * target, in place            gen_voice.py:41, control/toolbox/__init__.py:313
 * (target 0.97).  An all-zero waveform gives NaNs, as the reference's 0/0 does. */
int mb_wave_peak_normalize(void* d_wav, int dtype, long long n, double target, void* d_workspace,
                           size_t workspace_bytes, mb_stream_t stream);
/* float -> int16 PCM.  mode:
 *   MB_PCM16_SNDFILE   x*32768, clipped to [-32768, 32767], round-half-even: what libsndfile's
 *                      f2s_clip_array / d2s_clip_array (src/pcm.c) do for sf.write(.., "PCM_16") with
 *                      python-soundfile's SFC_SET_CLIPPING on (run.py:91, gen_voice.py:47).  libsndfile is
 *                      not vendored by the reference nor installed here: restated, parity unpinned.
 *   MB_PCM16_ENCODE16  np.clip(x * 2**15, -2**15, 2**15 - 1).astype(np.int16)  (truncation)
 *                      models/vocoder/wavernn/audio.py:38-39
 *   MB_PCM16_SAVE_WAV  x * (32767 / max(0.01, max|x|)), .astype(np.int16)      (truncation)
 *                      models/synthesizer/audio.py:12-15 */
#define MB_PCM16_SNDFILE 0
#define MB_PCM16_ENCODE16 1
#define MB_PCM16_SAVE_WAV 2
size_t mb_wave_workspace_bytes(void);
int mb_wave_pack_pcm16(const void* d_wav, int dtype, long long n, int mode, int16_t* d_pcm,
                       void* d_workspace, size_t workspace_bytes, mb_stream_t stream);

/* The same tail for a BATCH of fp32 waveforms in two launches (pipeline.gen_wavs: gen_voice.py:30-41 per request -- cut at the sentence
   boundaries, silence behind every sentence, peak normalisation over the request, PCM).  d_pieces: n_pieces x {source offset in d_wav (< 0: a piece of zeros),
   destination offset in d_out, length, item} as four int64 each (elements); d_out (out_elems floats, or int16 with a pcm_mode) is zeroed
   here, so the gaps need no pieces.  normalize_target <= 0 or NaN: no normalisation; pcm_mode -1: float out, else MB_PCM16_SNDFILE /
   MB_PCM16_ENCODE16 (save_wav needs the peak of the normalised signal: one-waveform calls).  Every item's samples are bit for bit those of
   mb_wave_peak_normalize + mb_wave_pack_pcm16 on its own waveform. */
size_t mb_wave_finish_batch_workspace_bytes(int n_items);
int mb_wave_finish_batch(const float* d_wav, const long long* d_pieces, int n_pieces, int n_items, long long out_elems,
                         double normalize_target, int pcm_mode, void* d_out, void* d_workspace, size_t workspace_bytes,
                         mb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MBHIP_H */
