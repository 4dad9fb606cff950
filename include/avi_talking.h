/*
 * avi_talking.h -- C ABI of libavi_talking_hip.so (MI355X / gfx950).
 *
 * The reference (sunyasheng/AVI-Talking) is pure Python over torch; it has no FFI for this
 * path.  Its boundary is a set of Python call signatures.  This header is
 * the C-ABI a binding for those call sites targets: plain device pointers, sizes and a
 * hipStream_t, no torch types.  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless a parameter is documented as host;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *     nothing allocates, frees or synchronises, so every call is hipGraph-capturable;
 *   - activations are fp32, row-major, "channels last" ([batch][time][channel]);
 *   - GEMM weights are stored [N_pad][K] in bf16 as a hi part and (optionally) a lo part
 *     (w = hi + lo to ~2^-17 relative), produced by avi_pack_weight_split();
 *   - return value: 0 on success, a negative AVI_E* code on a bad argument, or the positive
 *     hipError_t of a failed launch.
 */
#ifndef AVI_TALKING_H
#define AVI_TALKING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVI_OK 0
#define AVI_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define AVI_ENOSPC (-2)   /* workspace too small */

/* activation codes for AviGemm.act */
#define AVI_ACT_NONE 0
#define AVI_ACT_GELU 1    /* exact erf GELU (torch.nn.functional.gelu default) */
#define AVI_ACT_LRELU02 2 /* LeakyReLU(0.2) (L2lMotionPrior.py:376,389) */
#define AVI_ACT_RELU 3    /* nn.TransformerDecoderLayer default (models/faceformer.py:148) */
#define AVI_ACT_SILU 4
#define AVI_ACT_QUICK_GELU 5 /* x*sigmoid(1.702x): CLIP text model hidden_act (models/diffusion_prior.py:29-55) */

/* precision codes: how fp32 operands are fed to the bf16 matrix cores */
#define AVI_PREC_BF16 1   /* x,w rounded to bf16; 1 MFMA per product; ~4e-3 relative */
#define AVI_PREC_BF16X3 3 /* x=xh+xl, w=wh+wl; xh*wh + xh*wl + xl*wh; ~1e-5 relative (parity mode) */
#define AVI_PREC_F16X2 2  /* OPT-IN: x=xh+xl (fp16 planes), w rounded to ONE fp16 plane (Whi; Wlo ignored); xh*w + xl*w:
                             two MFMAs per product, weight rounding 2^-12 (the reference itself runs fp16 autocast).
                             Plane-operand GEMMs only (Ahi/Alo given); Chi/Clo outputs are fp16 hi/lo planes. */
/* split-plane activation formats (x = hi + lo, 16-bit patterns) of the plane-producing entry points */
#define AVI_PLANES_BF16 0
#define AVI_PLANES_F16 1

const char* avi_version(void);

/* ------------------------------------------------------------------------------------------
 * Status words: failures a launch can only find out about on the device.  `words` points to AVI_STATUS_WORDS 32-bit
 * words; a later launch of this process that meets failure k STORES 1 into words[k] (a plain system-scope store, no
 * read-modify-write: the words may live in pinned, device-mapped host memory and be read by the host WITHOUT a
 * synchronisation; a platform without PCIe atomics is fine).  NULL (the initial state) switches the reporting off.  The
 * library never clears a word; the owner does.  Nothing like it exists in the reference, whose guards are host-side NaN
 * sweeps (inferno/utils/batch.py:22-34, train_diffusion_prior.py:135-137).
 *   words[AVI_STATUS_F16_OVERFLOW]  a value written into an fp16 hi plane (AVI_PLANES_F16 producers: conv layer 0, the
 *                                   plane-operand GEMM epilogues, LayerNorm, attention) was non-finite or beyond the fp16
 *                                   range (|x| >= 65520): the planes hold inf, results are invalid.  bf16 planes have
 *                                   fp32's range and are not guarded.
 *   words[AVI_STATUS_F16_TINY]      a wave's whole share of an fp16 plane tile was below 2^-12 in magnitude (and not all
 *                                   zero): the lo plane has gone subnormal and x = hi + lo no longer carries ~22 bits - the
 *                                   2-term fp16 GEMM that consumes the planes is then less accurate than its weight
 *                                   rounding (2^-12).
 *   words[AVI_STATUS_PAIR_TIMEOUT]  a paired-sampler workgroup gave up on its partner (bounded spin,
 *                                   avi_prior_sample_paired); that launch's output is NaN.
 *   words[AVI_STATUS_EXCHANGE_TIMEOUT]  a workgroup of the persistent FaceFormer decode never saw a granule it waited for
 *                                   (bounded spin, avi_faceformer_decode_persistent: the launch did not get all 256 CUs);
 *                                   that launch's output is NaN from the frame it happened in.
 */
#define AVI_STATUS_F16_OVERFLOW 0
#define AVI_STATUS_F16_TINY 1
#define AVI_STATUS_PAIR_TIMEOUT 2
#define AVI_STATUS_EXCHANGE_TIMEOUT 3
#define AVI_STATUS_WORDS 4
int avi_set_status_words(void* words);
void* avi_status_words(void);
/* Diagnostics, so that the failure paths above can be tested.
 * avi_debug_fault_inject: make later launches fail in a chosen way (0 = none):
 *   AVI_FAULT_PAIR_PARTNER_ABSENT   the second workgroup of every sample pair of avi_prior_sample_paired leaves at once.
 *   AVI_FAULT_EXCHANGE_ABSENT       the last workgroup of avi_faceformer_decode_persistent leaves at once.
 * avi_debug_raise_status: one launch on `stream` that stores 1 into words[k] the way a failing kernel would. */
#define AVI_FAULT_PAIR_PARTNER_ABSENT 1
#define AVI_FAULT_EXCHANGE_ABSENT 2
int avi_debug_fault_inject(int faults);
int avi_debug_raise_status(int k, void* stream);
/* avi_debug_where: out[b] = XCC_ID << 16 | HW_ID[15:0] of workgroup b of a launch of `blocks` x `threads` with `lds_bytes` of
 * dynamic LDS, each holding its slot for `spin` x 64 sleep cycles: where a stream's workgroups land (CU masks, the dealing
 * of workgroups over the XCDs).  Speed diagnostics only. */
int avi_debug_where(unsigned* out, int blocks, int threads, int lds_bytes, int spin, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched strided GEMM with fused epilogue -- the dense contraction under every Linear / Conv1d
 * on the path:
 *     C[z][m][n] = affine( act( sum_k A[z][m*lda + k] * W[z][n][k] + bias[z][n] ) ) + R[z][m][n]
 * A rows may overlap (lda < K): a channels-last Conv1d(C_in, C_out, k, stride s) is exactly this
 * GEMM with lda = s*C_in, K = k*C_in (HF Wav2Vec2 conv layers 1-6, models/lib/wav2vec.py:97;
 * FLINT convs, L2lMotionPrior.py:370-392,419-424).  Linear layers use lda = K.
 * Batch index z = zo*z_inner + zi; every operand has an outer and an inner batch stride
 * (in elements), which covers grouped convolution (pos-conv, groups = z_inner).
 * ---------------------------------------------------------------------------------------- */
typedef struct AviGemm {
    const float* A;       long long lda, sAo, sAi;
    const uint16_t* Whi;  /* bf16 [N_pad][K]; N_pad = 64 if N <= 64, else N rounded up to 128 */
    const uint16_t* Wlo;  /* bf16 lo part; required when prec == AVI_PREC_BF16X3 */
    long long sWo, sWi;
    float* C;             long long ldc, sCo, sCi;
    const float* bias;    long long sBo, sBi;     /* [N] or NULL */
    const float* R;       long long ldr, sRo, sRi; /* residual, or NULL */
    const float* scale;   const float* shift;      /* post-activation per-n affine (BatchNorm eval), or NULL */
    int M, N, K;          /* K % 64 == 0 */
    int batch, z_inner;   /* batch >= 1, z_inner >= 1, batch % z_inner == 0 */
    int act, prec;
    /* Optional "split-plane" activations: a value x is stored as two bf16 planes, x = hi + lo to 2^-17 relative
     * (same 4 bytes per element as fp32).  When Ahi/Alo are set, A is ignored and the activation tile goes
     * global -> LDS by LDS-DMA with no conversion in the GEMM loop (gemm_dma.hip; lda and the A batch strides then
     * count bf16 elements of ONE plane, lda % 8 == 0).  When Chi/Clo are set the epilogue also emits the result as
     * planes with row stride ldc (C may then be NULL), so chains of GEMMs never pass through fp32. */
    const uint16_t *Ahi, *Alo;
    uint16_t *Chi, *Clo;
    /* Row stride of the weight planes in elements (0 = K).  With ldw > K a batch of launches can walk K slices of
     * one weight matrix (split-K for skinny problems: sAo = sWo = K_slice, partial sums per batch entry, folded by
     * avi_splitk_epilogue). */
    int ldw;
    /* Compute units this launch can count on (0 = all 256).  A caller that knows another kernel holds some CUs for the
     * whole duration (the sampling pipeline: 32 sampler workgroups) passes the remainder, and the tile shape is chosen
     * to fill whole rounds of THAT many workgroups. */
    int cus;
    /* Optional IEEE-half copy of the result (fp32-operand kernel only: A set, Ahi NULL), same ldc / batch strides as C,
     * counted in elements; C may then be NULL.  The EMOTE head's last layer writes its coefficients this way for long-form
     * batches (BASELINE.json configs[4]: "fp16 coeffs"): half the bytes, |rounding| <= 2^-11 |value|. */
    uint16_t* C16;
    /* Optional STREAM-K workspace for the 128-row plane-operand kernel (batch 1, cus > 0): device memory of sk_ws_floats
     * floats, zero-filled ONCE by the caller (the kernel leaves its tile counters at zero), used by one launch at a time
     * (stream order).  With it, a problem whose tiles do not fill the last round of `cus` workgroups is cut along K into
     * `cus` equal shares; tiles shared by several workgroups are finished by the contributor that arrives last (csrc/
     * gemm_pp192.hip).  Size: tiles(128 x 256) * 4 * 128 * 256 + tiles + 64 floats (74 MB at M = 8000, N = 768).  NULL = off. */
    float* sk_ws;
    long long sk_ws_floats;
} AviGemm;
int avi_gemm(const AviGemm* g, void* stream);

/* fp32 [N][K] -> bf16 hi/lo [N_pad][K] (rows N..N_pad-1 zero).  lo may be NULL. */
int avi_pack_weight_split(const float* W, int N, int K, int N_pad, uint16_t* hi, uint16_t* lo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio front end.  Replaces HF Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm as called from
 * inferno/models/temporal/AudioEncoders.py:170-178 (joint over the batch) and
 * dataset/data_loader.py:289-290 (per clip), and conv layer 0 + GroupNorm + GELU of the HF
 * feature encoder (models/lib/wav2vec.py:97).
 * ---------------------------------------------------------------------------------------- */
/* pcm: int16 [B][N] (is_int16 != 0) or fp32 [B][N]; out fp32 [B][N]; stats: scratch >= 2*B doubles */
int avi_audio_normalize(const void* pcm, int is_int16, int B, int N, int joint, float eps,
                        float* out, double* stats, void* stream);
/* x [B][N] -> y [B][T0][512] = GELU(GroupNorm_512(Conv1d(1,512,10,stride 5)(x))), T0 = (N-10)/5+1.
 * w0 [512][10], gamma/beta [512]; moments: scratch >= 65*B*ceil(T0/512) doubles, T0 = (N-10)/5+1 (per-chunk partial moments, no zeroing needed); scale_shift: scratch >= 1024*B floats. */
int avi_conv0_gn_gelu(const float* x, int B, int N, const float* w0, const float* gamma, const float* beta,
                      float eps, float* y, double* moments, float* scale_shift, void* stream);
/* same, result as hi/lo planes [B][T0][512] (the A operand format of the LDS-DMA GEMM; plane_fmt = AVI_PLANES_*) */
int avi_conv0_gn_gelu_planes(const float* x, int B, int N, const float* w0, const float* gamma, const float* beta,
                             float eps, uint16_t* y_hi, uint16_t* y_lo, double* moments, float* scale_shift,
                             int plane_fmt, void* stream);

/* 50->25 Hz resample (F.interpolate linear, align_corners=True; models/lib/wav2vec.py:67-73) fused
 * with LayerNorm(C) of the feature projection (HF Wav2Vec2FeatureProjection).
 * in [B][Tin][C] -> out [B][Tout][C]; gamma/beta may be NULL (interpolation only). */
int avi_interp_layernorm(const float* in, int B, int Tin, int C, int Tout, const float* gamma,
                         const float* beta, float eps, float* out, void* stream);
/* same with the input given as bf16 hi/lo planes (x = hi + lo) */
int avi_interp_layernorm_planes(const uint16_t* in_hi, const uint16_t* in_lo, int B, int Tin, int C, int Tout,
                                const float* gamma, const float* beta, float eps, float* out, void* stream);

/* row LayerNorm: out[r] = LN(in[r]) * gamma + beta, rows x C.  in == out allowed. */
int avi_layernorm(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                  float* out, void* stream);
/* Split-K epilogue for skinny GEMMs (M <= 32: the aligner MLP, models/diffusion_prior.py:58-117, whose Linear ->
 * LayerNorm -> GELU (+ residual) blocks run at batch-size rows): y[r][:] = sum_z parts[z*part_stride + r*C ..] + bias,
 * then, if do_ln, LayerNorm(gamma, beta, eps); then act; then + residual.  Partials are added in z order.
 * C % 4 == 0, C <= 4096. */
int avi_splitk_epilogue(const float* parts, int nparts, long long part_stride, int rows, int C, const float* bias,
                        const float* gamma, const float* beta, float eps, int do_ln, int act, const float* residual,
                        float* out, void* stream);

/* LayerNorm whose result is written as fp32 (out, may be NULL) and as hi/lo planes (the next GEMM's operand;
 * plane_fmt = AVI_PLANES_*) */
int avi_layernorm_planes(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                         float* out, uint16_t* out_hi, uint16_t* out_lo, int plane_fmt, void* stream);
/* out[r] = act(LN(in[r])) + residual[r]  (BrainNetwork blocks: Linear -> LayerNorm -> GELU -> +residual,
 * models/diffusion_prior.py:64-76,106-110).  residual may be NULL; in == out allowed. */
int avi_layernorm_act(const float* in, int rows, int C, const float* gamma, const float* beta, float eps,
                      int act, const float* residual, float* out, void* stream);

/* pos-conv input packing: h [B][T][G*Cg] -> xg [B][G][T+2*pad][Cg], zero padded (pad = 64, Cg = 48). */
int avi_group_pad_pack(const float* h, int B, int T, int G, int Cg, int pad, float* xg, void* stream);
/* wav2vec2's positional conv embedding in one launch (HF Wav2Vec2PositionalConvEmbedding behind models/lib/wav2vec.py:142-148):
 * out[b][t][:] = x[b][t][:] + gelu(conv1d(x, k = taps, groups, padding = taps/2)[t] + bias), last extra frame dropped.
 * x, out [B][T][C] fp32 (C = groups * 48, taps = 128: wav2vec2-base); w_hi / w_lo: bf16 hi / lo planes
 * [groups][rows_per_group >= 48][taps * 48], W[g][n][tap * 48 + ch] = conv.weight[g * 48 + n][ch][tap] (weight norm folded).
 * NOT in place: `out` must not overlap `x` (a workgroup reads a +/-64-frame halo of x that its neighbours store as out);
 * overlapping ranges return AVI_EINVAL. */
int avi_posconv_gelu_residual(const float* x, int B, int T, int C, int groups, int taps, const uint16_t* w_hi,
                              const uint16_t* w_lo, int rows_per_group, const float* bias, float* out, void* stream);

/* time-axis padding / repetition: in [B][T][C] -> out [B][padL + T*rep + padR][C];
 * mode 0 zeros, 1 replicate edge rows (Conv1d padding_mode='replicate', L2lMotionPrior.py:387). */
int avi_pad_repeat(const float* in, int B, int T, int C, int rep, int padL, int padR, int mode,
                   float* out, void* stream);

/* out[b][t][c] = table[ids[b][t]][c] + pos[t][c]: CLIPTextEmbeddings (token + position embedding) under
 * FrozenCLIPEmbedder.forward (models/diffusion_prior.py:48-53).  ids int64 [B][T], every id in [0, vocab) (an id out
 * of range is clamped into the table and counted in *bad_ids, int32 device, caller-zeroed, may be NULL). */
int avi_embed_tokens(const long long* ids, const float* table, const float* pos, int B, int T, int C, int vocab,
                     float* out, int* bad_ids, void* stream);
/* out[b][c] = mean_t in[b][t][c]: the caller's pooling of the text feature, `CLIP(text).mean(dim=1)`
 * (train_diffusion_prior.py:438-439,710-711), which yields the (B,768) voxel the aligner consumes. */
int avi_mean_tokens(const float* in, int B, int T, int C, float* out, void* stream);

/* out[b][t][c] = in[b][t][c] + add[b][c]  (EMOTE style_op "add", FaceFormerDecoder.py:667-668) */
int avi_add_rowbcast(const float* in, const float* add, int B, int T, int C, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head attention over packed projections (fp32 math, online softmax).
 *   q [B][Tq][ldq], k/v [B][Tk][ldk] with head h at column h*D; out [B][Tq][ldo].
 *   bias_mode: 0 none (HF wav2vec2 eager attention; EMOTE bert_decoder, temporal_bias none)
 *              1 ALiBi "future": -slope_h*|i-j|   (TransformerMasking.py:80-98, FLINT decoder)
 *              2 FaceFormer causal: j<=i: -slope_h*floor((i-j)/period), j>i: -inf
 *                (models/faceformer.py:51-72)
 *   slopes [H] (device) for modes 1,2.  scale multiplies q.k (1/sqrt(D)).
 * ---------------------------------------------------------------------------------------- */
int avi_attention(const float* q, const float* k, const float* v, float* out, int B, int H, int Tq, int Tk,
                  int D, int ldq, int ldk, int ldo, float scale, int bias_mode, const float* slopes,
                  int period, void* stream);

/* Head-dim-64 unbiased attention on the matrix cores (bf16 3-term split, fp32 accumulate): the wav2vec2 encoder
 * layers.  qkv packed [B][T][ld] (q | k | v, each H*64 wide, ld >= 3*H*64), out [B][T][ldo].
 * scratch >= 6 * B*H*Tp*64 bf16 values (Tp = T rounded up to 64), 16-byte aligned. */
int avi_attention_d64(const float* qkv, int B, int H, int T, int ld, float scale, float* out, int ldo,
                      uint16_t* scratch, void* stream);
/* Same attention, result as split bf16 planes out_hi/out_lo [B][T][ldo] (x = hi + lo; `out` fp32 optional, may be NULL):
 * the activation format of the LDS-DMA GEMMs, so encoder.layers.*.attention.out_proj reads it without conversion. */
int avi_attention_d64_planes(const float* qkv, int B, int H, int T, int ld, float scale, float* out,
                             uint16_t* out_hi, uint16_t* out_lo, int ldo, int plane_fmt, void* stream);
/* Same, under the bias modes of avi_attention (slopes [H] device for modes 1, 2).  Mode 2 with zero slopes and
 * period 1 is the causal mask of the CLIP text transformer behind FrozenCLIPEmbedder (models/diffusion_prior.py:52-53). */
int avi_attention_d64_planes_biased(const float* qkv, int B, int H, int T, int ld, float scale, int bias_mode,
                                    const float* slopes, int period, float* out, uint16_t* out_hi, uint16_t* out_lo,
                                    int ldo, int plane_fmt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Diffusion prior.  Replaces VersatileDiffusionPriorNetwork.forward (models/diffusion_prior.py:223-313),
 * FlaggedCausalTransformer.forward (:154-166) with the dalle2 Attention/FeedForward/LayerNorm blocks,
 * and InstructDiffusionPrior.p_sample / p_sample_loop_ddpm (:329-367) + dalle2 q_posterior.
 * All pointers are fp32 device arrays.  Linear weights are stored TRANSPOSED, [K][N] (in x out).
 * ---------------------------------------------------------------------------------------- */
#define AVI_PRIOR_MAX_DEPTH 8
typedef struct AviPriorLayer {
    const float* norm_g;   /* [128]       layers.{l}.0.norm.g */
    const float* wqkv;     /* [128][640]  cat(to_q.weight, to_kv.weight)^T : q 512 | k 64 | v 64 */
    const float* null_kv;  /* [2][64]     layers.{l}.0.null_kv */
    const float* wout;     /* [512][128]  to_out.0.weight^T */
    const float* out_g;    /* [128]       to_out.1.g */
    const float* ff_g;     /* [128]       layers.{l}.1.0.g */
    const float* w1;       /* [128][1024] layers.{l}.1.1.weight^T (value 512 | gate 512) */
    const float* w2;       /* [512][128]  layers.{l}.1.5.weight^T */
} AviPriorLayer;
typedef struct AviPriorWeights {
    int depth, timesteps;
    const float* time_table;                 /* [timesteps][128] SinusoidalPosEmb(t) */
    const float *t_w0, *t_b0;                /* [128][256], [256]   to_time_embeds MLP */
    const float *t_w1, *t_b1;                /* [256][256], [256] */
    const float *t_w2, *t_b2;                /* [256][128], [128] */
    const float* learned_query;              /* [128] (learned_query_mode "pos_emb") */
    const float* null_brain;                 /* [128] null_brain_embeds */
    const float* null_image;                 /* [128] null_image_embed */
    const float* rel_bias;                   /* [8][3][4] RelPosBias(n=3, n+1=4) gathered from the (32,8) table */
    const float *rot_cos, *rot_sin;          /* [3][32] rotary cos/sin for positions 0..2 */
    AviPriorLayer layer[AVI_PRIOR_MAX_DEPTH];
    const float* final_g;                    /* [128] causal_transformer.norm.g (stable LayerNorm) */
    const float* wproj;                      /* [128][128] project_out.weight^T */
    const float *coef1, *coef2, *logvar;     /* [timesteps] posterior_mean_coef1/2, posterior_log_variance_clipped */
} AviPriorWeights;

/* bf16 hi/lo planes of the streamed denoiser matrices for the batched matrix-core sampler, FRAGMENT-MAJOR:
 * plane[((n/16)*(K/32) + k/32)*512 + ((n%16) + 16*((k%32)/8))*8 + k%8] = W[n][k]  (W in torch layout [N][K]),
 * i.e. [N/16][K/32][64 lanes][8]: one MFMA fragment load of a wave is one contiguous 1-KiB read. */
typedef struct AviPriorLayerPlanes {
    const uint16_t *qkv_hi, *qkv_lo;   /* [640][128]  cat(to_q.weight, to_kv.weight) */
    const uint16_t *out_hi, *out_lo;   /* [128][512]  to_out.0.weight */
    const uint16_t *w1_hi, *w1_lo;     /* [1024][128] layers.{l}.1.1.weight */
    const uint16_t *w2_hi, *w2_lo;     /* [128][512]  layers.{l}.1.5.weight */
    /* w1_lo == w2_lo == NULL: w1_hi / w2_hi hold ONE plane of IEEE fp16 values (same fragment order); the kernel then
     * splits the activation into fp16 hi + lo and spends two MFMAs per product (half the streamed bytes). */
} AviPriorLayerPlanes;
typedef struct AviPriorPlanes {
    AviPriorLayerPlanes layer[AVI_PRIOR_MAX_DEPTH];
    const uint16_t *proj_hi, *proj_lo; /* [128][128]  project_out.weight */
} AviPriorPlanes;

/* One denoiser evaluation per sample (training forward / unit of the sampler):
 * x_t [B][128], t [B] int32, text_embed [B][128], keep masks [B] bytes or NULL (= keep) -> pred [B][128]. */
int avi_prior_forward(const AviPriorWeights* w, const float* x_t, const int* t, const float* text_embed,
                      const unsigned char* brain_keep, const unsigned char* image_keep, int B, float* pred,
                      void* stream);
/* Whole DDPM loop in ONE launch (one workgroup per sample): noise [timesteps+1][B][128], noise[0] = x_T,
 * noise[1+k] = z of the k-th step; out [B][128] = x_0 * inv_scale (inv_scale = 1/sqrt(128)).
 * temb_scratch >= timesteps*128 floats (time embeddings of all timesteps, filled by a first tiny launch). */
int avi_prior_sample(const AviPriorWeights* w, const float* text_embed, const float* noise, int B,
                     float inv_scale, float* out, float* temb_scratch, void* stream);
/* Same loop with `samples_per_group` (1..5) samples per workgroup on the matrix cores (bf16 3-term split): the
 * weights are streamed once per step per GROUP instead of per sample.  Small vectors (gains, null kv, biases,
 * schedule) come from `w`, the streamed matrices from `p`. */
int avi_prior_sample_batched(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                             const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                             float* temb_scratch, void* stream);
/* The time embeddings of all timesteps, [timesteps][128] (to_time_embeds of models/diffusion_prior.py:188-191,284): a
 * constant of the weights, so a caller may build it once ... */
int avi_prior_time_table(const AviPriorWeights* w, float* temb, void* stream);
/* ... and sample with it: ONE launch on `stream` (no table kernel in front of the sampler). */
int avi_prior_sample_batched_tab(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed,
                                 const float* noise, int B, int samples_per_group, float inv_scale, float* out,
                                 const float* temb_table, void* stream);

/* PAIRED sampler (csrc/prior_pair.hip): two samples share two workgroups on two CUs and each streams HALF of every layer's
 * matrices (q heads / feed-forward halves; the partial sums of to_out and ff2 cross between the partners as tagged 8-byte
 * granules), because the loop above is bound by what ONE CU can take in from L2.  Same contract as
 * avi_prior_sample_batched_tab; plane formats: feed-forward matrices one fp16 plane, attention matrices bf16 hi / lo (the
 * default).  workspace: avi_prior_pair_workspace_bytes(B) bytes, zero-filled ONCE by the caller, then owned by the library
 * (launch epoch, exchange slots); one launch at a time (stream order).  A partner that never answers within the bounded
 * spin makes the launch FAIL LOUDLY: the half that gave up sends NaN from then on, so every output of the pair is NaN, and
 * AVI_STATUS_PAIR_TIMEOUT is raised in the status word (avi_set_status_words) and in workspace word 1 (u64, sticky until the
 * owner clears it).  An exchange is tagged with 16 bits of the launch epoch and a 16-bit exchange number:
 * 2 * depth * timesteps must stay below 65536 (AVI_EINVAL otherwise). */
long long avi_prior_pair_workspace_bytes(int B);
int avi_prior_sample_paired(const AviPriorWeights* w, const AviPriorPlanes* p, const float* text_embed, const float* noise,
                            int B, float inv_scale, float* out, const float* temb_table, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * FaceFormer-style autoregressive decoder.  Replaces the loop of Faceformer.predict
 * (models/faceformer.py:710-729; teacher-free branch of forward_switch_frame :392-409) over
 * init_biased_mask (:51-72), enc_dec_mask (:75-83), PeriodicPositionalEncoding (:87-102) and the
 * nn.TransformerDecoderLayer(d, 4 heads, ff 2d, ReLU, post-LN) built at :148-149.
 * All pointers fp32 device arrays; Linear weights TRANSPOSED to [K][N].
 * ---------------------------------------------------------------------------------------- */
typedef struct AviFaceformerWeights {
    int D, V, period;                 /* feature_dim (D/4 a power of two >= 4), vertice_dim (<= 64), PPE/ALiBi period */
    const float *wqkv, *bqkv;         /* [D][3D], [3D]  self_attn.in_proj */
    const float *wo, *bo;             /* [D][D], [D]    self_attn.out_proj */
    const float *n1g, *n1b, *n2g, *n2b, *n3g, *n3b;   /* norm1..3 */
    const float *w1, *b1;             /* [D][2D], [2D]  linear1 */
    const float *w2, *b2;             /* [2D][D], [D]   linear2 */
    const float *wr, *br;             /* [D][V], [V]    vertice_map_r */
    const float *wm, *bm;             /* [V][D], [D]    vertice_map */
    const float* pe;                  /* [period][D]    one period of the PPE table */
    const float* slopes;              /* [4]            ALiBi head slopes */
    const float* obj_embedding;       /* [D]            start token */
    const float *coeff_mean, *coeff_std;  /* [V] or both NULL: out = out*std + mean (misc/coeff_{mean,std}.npy) */
} AviFaceformerWeights;
/* cross [B][T][D] = multihead_attn.out_proj(v_proj(memory)) (the diagonal memory mask makes cross-attention at
 * step i read memory row i only); kv_scratch >= B*T*2*D floats; out [B][T][V]. One launch for all T steps. */
int avi_faceformer_decode(const AviFaceformerWeights* w, const float* cross, int B, int T, float* kv_scratch,
                          float* out, void* stream);
/* Long-form decode (BASELINE.json configs[4], T up to tens of thousands).  The reference cannot decode more than 600
 * frames: its ALiBi mask and PPE table stop there (models/faceformer.py:88,147) and `predict` fails on the slice.
 * CHUNKED-CAUSAL semantics defined here: frame i attends to frames floor(i/chunk)*chunk .. i only (the KV cache restarts
 * at every multiple of `chunk`), the ALiBi distance is i-j as before, the PPE phase is i mod period (chunk must be a
 * multiple of period, so the phase runs on across the boundary), and the input embedding of the first frame of a
 * chunk is vertice_map(previous output) as for every other frame - motion stays continuous, only the attention window
 * is cut.  chunk >= T (or chunk = 0) is exactly the reference's predict(); oracle: oracle/faceformer.py
 * predict_cached(chunk=...). */
int avi_faceformer_decode_chunked(const AviFaceformerWeights* w, const float* cross, int B, int T, int chunk,
                                  float* kv_scratch, float* out, void* stream);
/* Same decode, the (un-normalised) coefficients stored as IEEE half [B][T][V] (configs[4] "fp16 coeffs"); the fed-back
 * frame stays fp32, so the decode itself is unchanged and out16 == half(out) element by element. */
int avi_faceformer_decode_chunked_f16(const AviFaceformerWeights* w, const float* cross, int B, int T, int chunk,
                                      float* kv_scratch, uint16_t* out16, void* stream);

/* Input rows of the TEACHER-FORCED decoder pass (models/faceformer.py:382-384): out[b][t] = vertice_map(coeff[b][t-1]) +
 * pe[t mod period] with coeff[b][-1] = 0 (`torch.cat([zeros_like(coeff[:, -1:]), coeff[:, :-1]], 1)`), coeff [B][T][V]
 * normalised coefficients, out [B][T][D]; fp32 on the vector pipe (K = 53).  The rest of the pass (:385-391) is
 * avi_gemm / avi_attention (bias mode 2) / avi_layernorm_act, see avi-talking_amd/host/faceformer.py. */
int avi_faceformer_tf_embed(const AviFaceformerWeights* w, const float* coeff, int B, int T, float* out, void* stream);

/* The same decode for WIDE decoders (D >= 256; config/vocaset/demo.yaml uses feature_dim 1024) as a chain of small
 * launches per frame, every one spread over the whole chip, instead of one workgroup per utterance streaming all
 * 8 D^2 weights through one CU per frame: self-attention as split-key partials over the KV cache, the four Linear
 * layers as 16-column slices on the matrix cores (bf16 3-term split of fragment-major weight planes), LayerNorms in
 * between.  The caller captures the whole chain (6-7 launches x T) in one hipGraph and replays it.
 * Derived constants, built once per model by the host (avi-talking_amd/host/faceformer.py): */
typedef struct AviFaceformerPlanes {
    const uint16_t *wo_hi, *wo_lo;   /* [self_attn.out_proj | vertice_map.weight (V columns padded to 64)]: [D][D + 64] as bf16
                                        hi/lo planes in fragment-major order [N/16][K/32][64 lanes][8] */
    const uint16_t *w1_hi, *w1_lo;   /* linear1 [2D][D] */
    const uint16_t *w2_hi, *w2_lo;   /* linear2 [D][2D] */
    const uint16_t *wr_hi, *wr_lo;   /* vertice_map_r [64 (V padded with zero rows)][D] */
    const float* wf_t;               /* [64][3D]: (in_proj . vertice_map)^T, rows >= V zero: qkv of frame i from frame i-1 */
    const float* bf;                 /* [period][3D]: in_proj (vertice_map.bias + pe[p]) + in_proj.bias */
    const float* qkv0;               /* [3D]: in_proj (obj_embedding + pe[0]) + in_proj.bias (frame 0) */
    const float* x0;                 /* [D]:  obj_embedding + pe[0] */
} AviFaceformerPlanes;
/* *floats = size of `work` (in floats) the chain needs for B utterances of width D */
int avi_faceformer_steps_work_floats(int D, int B, long long* floats);
/* The same decode for wide decoders and ONE utterance (B = 1) as ONE persistent launch of 256 workgroups, one per CU
 * (csrc/faceformer_persist.hip): every workgroup keeps its rows of all five matrices in LDS for the whole decode (fp32:
 * 137 KB at D = 1024) and the frame's activation vectors travel between the CUs as data-tagged 8-byte granules, six edges per
 * frame.  Replaces the per-frame launch chain below where its launches are latency, not work (40 us per frame at B = 1).
 * D in {256, 512, 1024}; 6 T + 6 < 65 535; chunk <= 1024; the device must have >= 256 CUs, all free (AVI_EINVAL otherwise;
 * a launch that does not get them ends with NaN output and AVI_STATUS_EXCHANGE_TIMEOUT, never hangs).
 *   avi_faceformer_persist_sizes  *image_floats = size of the per-model LDS image, *xch_bytes = the exchange workspace
 *                                 (zero-filled ONCE by the caller, then left to the library: it carries the launch epoch;
 *                                 one launch at a time may use it)
 *   avi_faceformer_persist_pack   builds the image from the fp32 weights (once per model)
 *   avi_faceformer_decode_persistent  out [B][T][V] fp32, or out16 IEEE half (the other NULL); kv_scratch >= B*T*2*D floats */
int avi_faceformer_persist_sizes(int D, long long* image_floats, long long* xch_bytes);
int avi_faceformer_persist_pack(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, float* image, void* stream);
int avi_faceformer_decode_persistent(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, const float* image,
                                     const float* cross, int B, int T, int chunk, float* kv_scratch, void* xch, float* out,
                                     uint16_t* out16, void* stream);
/* Enqueues the whole chain on `stream`.  D a multiple of 64 with D/4 in {16,...,256}; B <= 32 per call. */
int avi_faceformer_decode_steps(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, const float* cross,
                                int B, int T, int chunk, float* kv_scratch, float* work, float* out, void* stream);
int avi_faceformer_decode_steps_f16(const AviFaceformerWeights* w, const AviFaceformerPlanes* p, const float* cross,
                                    int B, int T, int chunk, float* kv_scratch, float* work, uint16_t* out16, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (aligner + prior; train_diffusion_prior.py:434-499).  Backward GEMMs reuse avi_gemm:
 *   dX = dY . W      -> avi_gemm(A = dY, W = packed W^T)
 *   dW = dY^T . X    -> avi_gemm(A = dY^T (avi_transpose), W = packed X^T), bias grad = avi_colsum(dY).
 * ---------------------------------------------------------------------------------------- */
/* LayerNorm with everything the training forward needs: y = act(LN(x / amax if stable)) * mask + residual.
 * mask = dropout keep-mask pre-scaled by 1/(1-p) or NULL; beta NULL = gain-only (dalle2 LayerNorm). */
int avi_layernorm_ex(const float* in, int rows, int C, const float* gamma, const float* beta, float eps, int act,
                     const float* mask, const float* residual, int stable, float* out, void* stream);
/* Backward of the above w.r.t. x, gamma, beta (residual passes through at the caller).
 * stats: scratch >= 3*rows floats.  accumulate != 0 adds into dgamma/dbeta.  dbeta may be NULL. */
int avi_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mask,
                      int rows, int C, float eps, int act, int stable, const float* dx_add, float* dx, float* dgamma,
                      float* dbeta, int accumulate, float* stats, void* stream);   /* dx = dLN/dx + dx_add (or NULL) */
/* The training forward of the 3-token denoiser in ONE launch (the pass p_losses differentiates,
 * models/diffusion_prior.py:119-313,369-400): token rows in, every intermediate the backward pass needs out.  Arrays are
 * [3B][C] per layer, stacked over the layers ([depth][3B][C]); `qkv` holds the projections BEFORE the rotary rotation.
 * `w` supplies the small vectors (LayerNorm gains, null_kv, the gathered rel_bias, rotary tables, final_g), `p` the
 * fragment-major bf16 hi/lo planes of the CURRENT weights (every *_lo non-NULL), re-packed each step by
 * avi_pack_fragment_planes. */
typedef struct AviPriorTrainDump {
    const float* tok0;                 /* [3B][128] token rows (avi_prior_tokens_fwd) */
    float *tok_in, *n1, *qkv, *ao, *o1, *tokm, *n2, *hff, *sw;   /* C = 128, 128, 640, 512, 128, 128, 128, 1024, 512 */
    float *tok_out, *fin, *po;         /* [3B][128]: input of the final LayerNorm, its output, project_out */
} AviPriorTrainDump;
int avi_prior_train_forward(const AviPriorWeights* w, const AviPriorPlanes* p, const AviPriorTrainDump* d, int B,
                            int samples_per_group, void* stream);
/* The dX chain of the backward pass of the same six layers in ONE launch (+ one small launch that reduces the LayerNorm
 * gain gradients): from `dtok_top`, the gradient at the output of the last layer, to `dtok0`, the gradient of the token
 * rows.  `pT` holds the fragment-major planes of the TRANSPOSED matrices (avi_pack_fragment_planes, transpose = 1).  The
 * output gradient of every matrix is stored for the weight-gradient GEMMs (dy_*: [depth][3B][C], C = 128, 1024, 128, 640 for
 * linear2, linear1, to_out, to_q|to_kv); dgamma_part: scratch >= ceil(B / samples_per_group) * depth * 3 * 128 floats.
 * attn_part: scratch of depth * B * 224 floats - every (layer, sample) STORES its 96 relative-bias and 128 null-kv gradient
 * contributions there and the launch's last kernel sums them in sample order into dnull_kv[l] ([2][64]) and drel ([8][3][4]),
 * plain stores: the step is run-to-run deterministic.  attn_part == NULL: dnull_kv[l] is zeroed and then accumulated with
 * float atomics, the relative-bias gradient is ADDED (atomics) to drel (cleared by the caller): last-bit differences run
 * to run. */
typedef struct AviPriorTrainBwd {
    const float* dtok_top;
    const float *tok_in, *qkv, *o1, *tokm, *hff;      /* forward intermediates (AviPriorTrainDump) */
    float *dy_w2, *dy_w1, *dy_out, *dy_qkv;
    float* dtok0;
    float* dgamma_part;
    float* dnull_kv[AVI_PRIOR_MAX_DEPTH];
    float* drel;
    float* attn_part;
} AviPriorTrainBwd;
typedef struct AviPriorGainGrads {
    float* g[AVI_PRIOR_MAX_DEPTH][3];                 /* gradients of norm.g, to_out.1.g, ff 0.g of every layer ([128], written) */
} AviPriorGainGrads;
int avi_prior_train_backward(const AviPriorWeights* w, const AviPriorPlanes* pT, const AviPriorTrainBwd* d,
                             const AviPriorGainGrads* gains, int B, int samples_per_group, void* stream);
/* Table-driven re-layout of row-major [N][K] 16-bit planes into the fragment-major order of AviPriorPlanes (transpose != 0:
 * the source is the [K][N] plane of the transposed matrix).  jobs_dev: device array; first_block = running sum of
 * ceil(N*K/8/256) over the jobs; N % 16 == 0, K % 32 == 0. */
typedef struct AviPlaneJob {
    const uint16_t *src_hi, *src_lo;
    uint16_t *dst_hi, *dst_lo;
    int N, K, first_block, transpose;
} AviPlaneJob;
int avi_pack_fragment_planes(const AviPlaneJob* jobs_dev, int njobs, int total_blocks, void* stream);

/* Data movement of the step, so that a captured training step contains library kernels only
 * (tests/test_gpu_library_only.py): p[0..n) = 0 (p 16-byte aligned) */
int avi_zero(float* p, long long n, void* stream);
/* dst[r][0..C) = src[row_index ? row_index[r] : r][0..C), row strides in elements: the time-embedding lookup
 * (models/diffusion_prior.py:284), pred = tokens[:, -1] (:311) and its gradient scatter */
int avi_copy_rows(const float* src, long long src_stride, const int* row_index, float* dst, long long dst_stride,
                  int rows, int C, void* stream);
/* T5 relative-position bias of the n-token denoiser (dalle2 RelPosBias(n, n+1), models/diffusion_prior.py:159):
 * forward (emb [32][heads] -> bias [heads][n][n+1], dbias = demb = NULL) or backward (demb [32][heads] = scatter of dbias,
 * every entry written, emb = bias = NULL); distances below 16 are their own bucket, so n <= 16 */
int avi_prior_rel_bias(const float* emb, float* bias, const float* dbias, float* demb, int heads, int n, void* stream);
/* [R][C] -> [C][R] */
int avi_transpose(const float* in, int R, int C, float* out, void* stream);
/* hi/lo [C_pad][R] = bf16 hi/lo split of in^T (in is [R][Cc] fp32; rows Cc..C_pad-1 are zero): the transposed
 * "weight" operand of avi_gemm for dW = dY^T X and dX = dY W in one pass. */
int avi_transpose_pack_split(const float* in, int R, int Cc, int C_pad, uint16_t* hi, uint16_t* lo, void* stream);
/* Several transposes in ONE launch (the training step is a chain of ~400 short launches: every launch saved is ~5 us).
 * A job writes out (fp32 [C][R]) and/or hi/lo (split planes [C_pad][R], rows >= C zero); NULL outputs are skipped. */
typedef struct AviTransposeJob {
    const float* in;   /* [R][C] fp32 */
    float* out;        /* [C][R] fp32 or NULL */
    uint16_t* hi;      /* [C_pad][R] or NULL */
    uint16_t* lo;
    int R, C, C_pad;   /* C_pad >= C; ignored (taken as C) when hi is NULL */
    int first_block;   /* device tables only: index of the job's first block (prefix sum over the jobs) */
    float* colsum;     /* non-NULL (with out = hi = NULL): the job is colsum[c] = sum_r in[r][c] instead (a bias
                        * gradient riding in the same launch), ceil(C/16) blocks */
} AviTransposeJob;
/* jobs: HOST array of 1..4 jobs, passed to the kernel by value (dY^T and X^T of one backward GEMM pair) */
int avi_transpose_jobs(const AviTransposeJob* jobs, int njobs, void* stream);
/* jobs_dev: DEVICE table with first_block filled in, total_blocks = sum of the jobs' blocks (the per-step refresh of
 * every transposed weight plane of the trainer: the table is built once).  Blocks of a job: ceil(C/16) for a column sum;
 * (C_pad/64) * (R/64) for a plane-only job (hi/lo set, out NULL) with R % 64 == 0 and C_pad % 64 == 0 (64 x 64 tiles, 16-byte
 * stores; hi/lo 16-byte aligned); otherwise ceil(Cp/32) * ceil(R/32) with Cp = C_pad for plane jobs, C else. */
int avi_transpose_table(const AviTransposeJob* jobs_dev, int njobs, int total_blocks, void* stream);
int avi_colsum(const float* in, int R, int C, float* out, int accumulate, void* stream);     /* out[c] = sum_r */
int avi_act_fwd(const float* x, long long n, int act, float* y, void* stream);
int avi_act_bwd(const float* x_pre, const float* dy, long long n, int act, float* dx, void* stream);
/* dalle2 SwiGLU: h [R][2F] = value | gate, y [R][F] = value * silu(gate) */
int avi_swiglu_fwd(const float* h, int R, int F, float* y, void* stream);
int avi_swiglu_bwd(const float* h, const float* dy, int R, int F, float* dh, void* stream);
/* q_sample + token assembly of p_losses (models/diffusion_prior.py:372,255-303):
 * x0 = target*scale; x_t = sqrt_ac[t] x0 + sqrt_1mac[t] noise; tokens [B][3][128] = [text|null, time, x_t|null + query]. */
int avi_prior_tokens_fwd(const float* target, const float* noise, const int* t, const float* sqrt_ac,
                         const float* sqrt_1mac, float scale, const float* text_embed, const float* time_emb,
                         const unsigned char* brain_keep, const unsigned char* image_keep, const float* null_brain,
                         const float* null_image, const float* learned_query, int B, float* x0, float* tokens,
                         void* stream);
/* dtokens -> dtext [B][128], dtime [B][128]; dnull_brain/dnull_image/dlearned_query [128] are ADDED to. */
int avi_prior_tokens_bwd(const float* dtokens, const unsigned char* brain_keep, const unsigned char* image_keep, int B,
                         float* dtext, float* dtime, float* dnull_brain, float* dnull_image, float* dlearned_query,
                         void* stream);
/* dalle2 Attention core on 3 tokens (+ null kv): qkv [B][3][640] (q 512 | k 64 | v 64) -> out [B][3][512].
 * Backward recomputes the forward; dnull_kv [2][64] and drel_bias [8][3][4] are ADDED to (atomics). */
int avi_prior_attn_fwd(const float* qkv, const float* null_kv, const float* rel_bias, const float* rot_cos,
                       const float* rot_sin, int B, float* out, void* stream);
int avi_prior_attn_bwd(const float* qkv, const float* null_kv, const float* rel_bias, const float* rot_cos,
                       const float* rot_sin, const float* dout, int B, float* dqkv, float* dnull_kv, float* drel_bias,
                       void* stream);
/* loss[0] = mean((pred-x0)^2) (models/diffusion_prior.py:391-399, predict_x_start); dpred = weight*dloss/dpred. */
int avi_mse_loss(const float* pred, const float* x0, int n, float weight, float* loss, float* dpred, void* stream);
/* soft_clip_loss (train_diffusion_prior.py:125-133) on L2-normalised rows (:454-455): proj/target [B][D], D <= 256.
 * dproj = grad_scale * dloss/dproj.  scratch >= 2*B*D + 2*B + 3*B*B floats. */
int avi_soft_clip_loss(const float* proj, const float* target, int B, int D, float temp, float grad_scale, float* loss,
                       float* dproj, float* scratch, void* stream);
/* Fused AdamW over one flat parameter region (torch.optim.AdamW semantics, train_diffusion_prior.py:997-1004),
 * n % 4 == 0, step >= 1; g is multiplied by grad_scale first (1/world for DP-averaged sums).
 * hi/lo (both or neither): also emit the bf16 hi/lo planes of the updated values for avi_gemm.
 * dyn (device pointer or NULL, 4 floats): {lr, 1-beta1^step, 1/sqrt(1-beta2^step), beta1 (0 = keep the argument)}
 * override the scalar arguments, so a captured hipGraph can be replayed with a new learning rate AND a new beta1 every
 * step: the reference's OneCycleLR has cycle_momentum on (train_diffusion_prior.py:351-357), which moves AdamW's beta1
 * between 0.95 and 0.85; the first bias correction is formed from the step's own beta1, as torch does. */
int avi_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, float grad_scale, const float* dyn, uint16_t* hi, uint16_t* lo,
              void* stream);

/* ------------------------------------------------------------------------------------------
 * Random draws INSIDE the captured passes.  Replaces the torch-generator calls of the reference's steps:
 * torch.randn(..., generator) for x_T and per DDPM step (models/diffusion_prior.py:337,349-351); timesteps, q_sample
 * noise, prob_mask_like cond-drop masks and nn.Dropout masks of one training step (models/diffusion_prior.py:445,453,
 * 255-259,62-75 via train_diffusion_prior.py:449).
 * Generator: Philox-4x32-10; state = device uint64[2] {seed, offset}.  Element i of a fill is word (i & 3) of the block
 * with counter (i >> 2 [48 bits] | subsequence << 48, offset) under key seed: fills with different `subsequence`
 * (< 65536) or different offsets never overlap.  avi_rng_advance adds `delta` to the offset (the last node of a captured
 * pass: every replay then draws fresh numbers; a replayed sequence is reproduced by resetting the state).
 * kinds: RAW uint32 words | NORMAL fp32 (Box-Muller on word pairs) | KEEP_SCALED fp32 (u >= param ? 1/(1-param) : 0, a
 * dropout keep mask with drop probability param) | BERNOULLI_U8 (u < param) | RANDINT_I32 (floor(word * param / 2^32),
 * param = exclusive upper bound <= 2^24) | UNIFORM fp32 in [0,1) (24 bits).  `out` aligned to 4 elements. */
#define AVI_RNG_RAW 0
#define AVI_RNG_NORMAL 1
#define AVI_RNG_KEEP_SCALED 2
#define AVI_RNG_BERNOULLI_U8 3
#define AVI_RNG_RANDINT_I32 4
#define AVI_RNG_UNIFORM 5
int avi_rng_fill(const unsigned long long* state, unsigned subsequence, int kind, float param, long long n, void* out,
                 void* stream);
int avi_rng_advance(unsigned long long* state, unsigned long long delta, void* stream);

/* ------------------------------------------------------------------------------------------
 * FLAME vertices.  Replaces `FLAME.forward(shape_params, expression_params, pose_params,
 * eye_pose_params)[0]` (third_party/inferno/inferno/models/DecaFLAME.py:222-244) = `lbs` of
 * third_party/inferno/inferno/utils/lbs.py:142-235 on the 5-joint head model (parents [-1,0,1,1,1]); landmarks are
 * not produced.  The basis arrays are the FLAME buffers re-laid out once per model:
 *   shape_basis[k][v*3+c] = shapedirs[v][c][k] (k < n_shape), frame_basis = [expression directions | posedirs rows]
 *   ((n_exp + 36) x V*3), j_* = J_regressor folded into template / shape / expression bases. */
typedef struct AviFlameBasis {
    const float* v_template;   /* [V][3] */
    const float* shape_basis;  /* [n_shape][V*3] */
    const float* frame_basis;  /* [n_exp + 36][V*3] */
    const float* j_template;   /* [5][3]          J_regressor . v_template */
    const float* j_shape;      /* [5*3][n_shape]  J_regressor . shapedirs[..., :n_shape] */
    const float* j_exp;        /* [5*3][n_exp] */
    const float* lbs_weights;  /* [V][5] */
    int V, n_shape, n_exp;
    /* optional (NULL = fp32 vector-pipe kernel): split bf16 planes of frame_basis written by avi_flame_pack_basis,
     * [3][Vp][KP] each (Vp = V rounded up to 16; KP = 96 if n_exp + 36 <= 96, else 160): the blend runs on the
     * matrix cores in 3-term bf16 */
    const uint16_t* basis_hi;
    const uint16_t* basis_lo;
} AviFlameBasis;
/* shape [B][n_shape] (one per clip), exp [B*T][n_exp], pose [B*T][15] = axis-angle of (global, neck, jaw, eye_l, eye_r)
 * -> verts [B*T][V][3].  Scratch (16-byte aligned; tiles = B * ceil(T/16): the matrix-core kernel takes its per-frame
 * operands in 16-frame fragment tiles per clip): v_shaped B*V*3 floats, coef max(ceil(B*T/8)*8*160, tiles*2560) floats,
 * xf max(B*T*60, tiles*3072) + B*16 floats (per-frame transforms, then the per-clip rest joints).  n_exp <= 124. */
int avi_flame_vertices(const AviFlameBasis* fb, const float* shape, const float* exp, const float* pose, int B, int T,
                       float* v_shaped, float* coef, float* xf, float* verts, void* stream);
/* Fill the optional basis planes of `fb` (see AviFlameBasis): hi / lo hold 3*Vp*KP uint16 each.  Once per model. */
int avi_flame_pack_basis(const AviFlameBasis* fb, uint16_t* hi, uint16_t* lo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVI_TALKING_H */
