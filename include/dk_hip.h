/*
 * dk_hip.h -- C ABI of libdk_hip.so: the MI355X (gfx950) denoising engine behind
 * DiffusionKit's DiffusionPipeline / FluxPipeline.
 *
 * The reference (argmaxinc/DiffusionKit, python/src/diffusionkit/mlx/) has no FFI layer: its
 * hot path is Python calling MLX ops.  This header *creates* the boundary a maintainer would
 * bind (ctypes stub in INTEGRATION.md).  Each entry point cites the reference call site it
 * replaces; paths are relative to python/src/diffusionkit/mlx/.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses
 *     (torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 *     (torch.cuda.current_stream().cuda_stream); 0 = the null stream.
 *   - the caller owns every buffer, including the workspaces sized by *_workspace_bytes().
 *   - all activations / weights are bfloat16 (raw uint16 storage) unless a name says f32.
 *   - every function returns 0 on success, <0 on error; dk_last_error() describes the last
 *     failure of the calling thread.  Nothing throws across the ABI.
 *   - no hidden host synchronisation and no internal threads: work is enqueued on `stream`.
 *   - tensors are token-major / NHWC, exactly the layouts of the reference's MLX arrays.
 */
#ifndef DK_HIP_H
#define DK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DK_ABI_VERSION 5

int dk_abi_version(void);
const char* dk_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Operator level (one MLX op call site each).  These are what the parity tests drive.
 * ---------------------------------------------------------------------------------------- */

/* epilogues of dk_gemm_bf16 / dk_conv3x3_bf16 */
enum {
  DK_EPI_BIAS = 0,      /* C = A W^T + b                      nn.Linear, mmdit.py:56,821-832          */
  DK_EPI_BIAS_GELU = 1, /* C = gelu_erf(A W^T + b)            FFN fc1 + nn.GELU(), mmdit.py:421,835   */
  DK_EPI_GATE_RES = 2,  /* C = res + gate[b,:] * (A W^T + b)  post_sdpa gating, mmdit.py:533-548      */
  DK_EPI_RES = 3,       /* C = res + (A W^T + b)              VAE residual adds, vae.py:99,54         */
  DK_EPI_BIAS_SILU = 4  /* C = silu(A W^T + b)                embedder MLPs, mmdit.py:357-361,372-376 */
};

typedef struct dk_gemm_desc {
  const void* A;    /* [M, K] bf16, row stride lda (elements)                       */
  const void* W;    /* [N, K] bf16 contiguous (nn.Linear weight layout [out, in])   */
  void* C;          /* [M, N] bf16, row stride ldc                                  */
  const void* bias; /* [N] bf16 or NULL                                             */
  const void* gate; /* DK_EPI_GATE_RES: [n_batch, gate_stride] bf16                 */
  const void* res;  /* DK_EPI_GATE_RES / DK_EPI_RES: residual, row stride ldr       */
  int32_t M, N, K;
  int32_t lda, ldc, ldr;
  /* logical row m -> physical row (m / seg_len) * seg_stride + m % seg_len.  seg_len = M,
   * seg_stride = 0 addresses a plain matrix; (S_img, S) addresses the image rows of a joint
   * [B, S, h] buffer whose base pointer was advanced to the first image row. */
  int32_t a_seg_len, a_seg_stride;
  int32_t c_seg_len, c_seg_stride;
  int32_t r_seg_len, r_seg_stride;
  int32_t gate_seg_len; /* batch index of row m is m / gate_seg_len */
  int32_t gate_stride;
  float alpha;          /* scales the accumulator before the bias (1.0 for Linear) */
  int32_t epilogue;
  int32_t ldw;          /* row stride of W in elements; 0 = K (contiguous nn.Linear weight) */
  /* Optional scratch for the remainder-wave K split of the 256x256 kernel (large M, N % 256 == 0):
   * dk_gemm_workspace_bytes() bytes, 256-byte aligned, whose LAST 4096 bytes are zero before the first use (the
   * kernels leave them zero).  NULL = no K split.  One GEMM at a time may use a given workspace. */
  void* workspace;
  size_t workspace_bytes;
} dk_gemm_desc;

size_t dk_gemm_workspace_bytes(void);

/* nn.Linear call sites of the hot path: mmdit.py:56,358-360,373-375,432,771,777,821-832;
 * vae.py:36-39,84 */
int dk_gemm_bf16(const dk_gemm_desc* d, void* stream);

/* What dk_gemm_bf16(d) -- or, with d2 != NULL, the grouped launch the engines issue for the image and text streams of a double block -- WOULD
 * launch on the current device (ABI 5, round 6).  Host only: no kernel runs, pointers in the descriptors are not dereferenced (only their alignment
 * is looked at; `workspace` non-NULL tells the rules that the K split is available), and without a GPU the rules assume 256 compute units.  The
 * decision code is the launch code itself, so tests/test_dispatch_plan.py sweeps shapes over the dispatch rules on the CPU. */
typedef struct dk_gemm_plan_t {
  int32_t kernel;      /* 128: 128 x 128-tile kernel; 3: 256 x 256 tiles, 8 waves (gemm256v3.hip); 4: one wave per SIMD (gemm256v4.hip) */
  int32_t tile_rows;   /* 128 / 224 / 256 */
  int32_t tiles;       /* output tiles of the launch */
  int32_t workgroups;  /* grid size: a tile that is cut along K counts once per piece */
  int32_t split_tiles; /* tiles cut along K */
  int32_t k_pieces;    /* pieces per cut tile (1: none) */
  int32_t ks;          /* K steps of 64 elements a workgroup runs (the finisher piece of a cut tile) */
  int32_t n_cu;        /* compute units the rules assumed */
  int32_t launches;    /* kernel launches the call expands to */
} dk_gemm_plan_t;
int dk_gemm_plan(const dk_gemm_desc* d, const dk_gemm_desc* d2, dk_gemm_plan_t* plan);

typedef struct dk_conv_desc {
  const void* x;    /* NHWC bf16 [B, H(/2), W(/2), C]; C multiple of 64              */
  const void* w;    /* [O, 3, 3, C] bf16 (MLX nn.Conv2d weight layout)               */
  void* y;          /* NHWC bf16 [B, H, W, ldy>=O]                                   */
  const void* bias; /* [O] or NULL                                                   */
  const void* res;  /* DK_EPI_RES: NHWC [B, H, W, ldr]                               */
  const void* zeros;/* >= 128 bytes of zeros on the device (padding taps)            */
  int32_t B, H, W, C, O; /* H, W are OUTPUT sizes                                     */
  int32_t ldy, ldr;
  int32_t upsample; /* 1: conv over the nearest-x2 upsampling of x (vae.py:20-25,146);
                     * 2: stride-2 conv over x = [B, 2H, 2W, C] padded by one zero row / column at the
                     *    bottom / right (the encoder's downsample, vae.py:141-143)          */
  int32_t epilogue;
} dk_conv_desc;

/* nn.Conv2d 3x3 / stride 1 / pad 1 call sites: vae.py:73,79,134,349,384 */
int dk_conv3x3_bf16(const dk_conv_desc* d, void* stream);

/* The norm -> silu -> conv sequence of ResnetBlock2D (vae.py:72-73,78-79,91-100) and of the decoder's output head
 * (conv_norm_out -> silu -> conv_out, vae.py:381,384,397-399) as ONE operator for the high-resolution stage: a 3x3 / stride 1 /
 * pad 1 convolution whose workgroups stage an 18 x 18 halo of the RAW input in LDS per 16 x 16 output tile, applying the
 * GroupNorm (as the per-channel fp32 pair from dk_groupnorm_table_bf16) and the SiLU on the way in -- the normalised tensor is
 * never written.  Optionally: + residual (the block's "+ x", vae.py:100); the 1x1 conv_shortcut of a channel-changing block
 * (vae.py:86-89,98-99) as extra reduction columns over x2 (w = [conv weight | shortcut weight], ldw = 9 C + C2 or more);
 * (sum, sum of squares) partials of the output per channel group for the GroupNorm that reads it next (dk_groupnorm_table_bf16
 * with x == NULL); and, for O <= 4 (conv_out), the clip / uint8 tail of decode_latents_to_image (__init__.py:581-584,525-526).
 * H, W multiples of 16; C (and C2) multiples of 64; O a multiple of 128, or <= 4 with the image tail. */
typedef struct dk_conv_gn_desc {
  const void* x;       /* NHWC bf16 [B, H(/2), W(/2), C]                                          */
  const void* w;       /* bf16 [O, ldw]: column (ky * 3 + kx) * C + c, then 9 C + c2              */
  const void* bias;    /* [O]                                                                      */
  void* y;             /* NHWC bf16 [B, H, W, ldy] (NULL with the image tail)                      */
  const void* res;     /* NHWC bf16 [B, H, W, ldr] or NULL                                         */
  const float* gn_scale_shift; /* [B, 2, C] fp32 from dk_groupnorm_table_bf16, or NULL: x as it is */
  int32_t gn_silu;
  const void* x2;      /* NHWC bf16 [B, H, W, C2] or NULL                                          */
  const void* bias2;   /* [O] or NULL                                                              */
  float* stats_partial;/* [B, (H/16)*(W/16), stats_groups, 2] fp32 or NULL                         */
  float* image_f32;    /* image tail: [B, H, W, 3] in [0, 1]                                       */
  uint8_t* image_u8;   /* ... [B, H, W, 3]                                                         */
  void* raw_bf16;      /* ... conv output, bf16 [B, H, W, 4]                                       */
  int32_t B, H, W, C, O, C2, ldw, ldy, ldr, upsample, stats_groups;
} dk_conv_gn_desc;
int dk_conv3x3_gn_bf16(const dk_conv_gn_desc* d, void* stream);
/* nn.GroupNorm statistics (vae.py:34,72,78,381) as the table dk_conv3x3_gn_bf16 applies: scale = rstd * gamma, shift =
 * beta - mean * scale, fp32 [B, 2, C].  x != NULL: statistics of x (NHWC bf16 [B, HW, C]); x == NULL: from `n_partial` partials
 * per batch row that a dk_conv3x3_gn_bf16 launch left in scratch (layout [B, n_partial, G, 2]).  scratch:
 * dk_groupnorm_scratch_floats(B, G) floats, or B * n_partial * G * 2 + B * G * 2 if that is more. */
int dk_groupnorm_table_bf16(const void* x, int32_t B, int64_t HW, int32_t C, int32_t G, const void* gamma, const void* beta,
                            float eps, float* scratch, int32_t n_partial, float* scale_shift, void* stream);

/* mx.fast.scaled_dot_product_attention call sites mmdit.py:562,643,687,736.
 * q/k/v: row (b*S + s) at ptr + (b*S + s)*ld + head*D; out likewise with ldo.  D in {64,128}. */
int dk_attention_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                      int32_t S, int32_t D, int32_t ld, int32_t ldo, float scale, void* stream);

/* Workspace of the attention launches THIS host thread enqueues (256-byte aligned, dk_attention_workspace_bytes() bytes; NULL: none).
 * The one-wave-per-SIMD D = 128 kernel (attention5.hip) splits the query blocks of a launch's last, partial round of the CUs along the
 * keys and merges the partial results through it.  Optional: without a workspace nothing is split (same results up to the bf16
 * rounding of the partials; FLUX 1024^2, one image: ~10 % longer attention launches).  ONE split launch at a time may use a given
 * workspace: launches that share it must be ordered on one stream.  The pointer is per host thread, not per device -- a thread that
 * moves to another device installs that device's buffer again.  dk_mmdit_* calls do not use this buffer: every engine carves its own
 * region from its workspace and installs it for the duration of the call (two engines on two streams never share partial results). */
size_t dk_attention_workspace_bytes(void);
int dk_attention_set_workspace(void* workspace, size_t bytes);

/* Single-head attention over head_dim 512: the VAE mid block's Attention (vae.py:28-57: softmax((q / sqrt 512) k^T) v over all
 * H * W tokens), flash-style -- the [T, T] score matrix of the reference (537 MB at T = 16384) is never written.  q / k / v / out:
 * bf16 [B, T, ld] with ld == 512 EXACTLY (dense rows: the transposed copy of v is taken from a dense [T, 512] matrix; any other row
 * pitch is rejected) and ldo >= 512, a multiple of 8; vt_scratch: B * 512 * dk_attention_d512_tp(T) bf16 (the kernel reads a transposed,
 * zero-padded copy of v that this call writes there first). */
int32_t dk_attention_d512_tp(int32_t T);
int dk_attention_d512_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t T, int32_t ld, int32_t ldo,
                           float scale, void* vt_scratch, void* stream);

/* The same with an additive score bias (text encoders): CLIP's causal mask
 * (clip.py:83-89, one [S, ldb] table for every head: bias_head_stride 0) and T5's relative-position
 * bias (t5.py:61-88, [H, S, ldb]); scores = scale * q.k + bias.  ldb: multiple of 64, >= S. */
int dk_attention_bias_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                           int32_t S, int32_t D, int32_t ld, int32_t ldo, float scale, const void* bias,
                           int64_t bias_head_stride, int32_t ldb, void* stream);

/* Text-conditioning helpers (clip.py:28-120, t5.py:60-243):
 * dk_embedding_bf16: out[i,:] = table[ids[i],:] (+ pos[i % pos_rows,:]); out_bf16 and / or out_f32.
 * dk_layernorm_bf16: nn.LayerNorm with weight / bias.   dk_t5_rmsnorm_bf16: the T5 RMSNorm over the fp32 stream.
 * dk_text_elementwise: op 0 y = quick_gelu(a); 1 y = a * b; 2 r_f32 += a.
 * dk_t5_bias_bf16: out[h,q,k] = emb[rel_bucket[k - q + S - 1], h] (rel_bucket: int32 [2S-1], host-computed). */
int dk_embedding_bf16(const void* table, const int32_t* ids, const void* pos, int32_t pos_rows, void* out_bf16,
                      float* out_f32, int32_t n, int32_t dim, int32_t vocab, void* stream);
int dk_layernorm_bf16(const void* x, void* out, int32_t M, int32_t h, const void* weight, const void* bias,
                      float eps, void* stream);
int dk_t5_rmsnorm_bf16(const float* x, void* out, int32_t M, int32_t h, const void* weight, float eps, void* stream);
int dk_text_elementwise(const void* a, const void* b, void* y, float* r, int64_t n, int32_t op, void* stream);
int dk_t5_bias_bf16(const void* emb, const int32_t* rel_bucket, int32_t H, int32_t S, int32_t ld, void* out,
                    void* stream);

/* ---- fp8 path (BASELINE.json configs[3]: fp8 weights, CDNA4 block-scaled fp8 MFMA) -----------------------------------
 * Weights: OCP e4m3 [N, K] + one f32 scale per output channel.  Activations: MX-fp8 = e4m3 [rows, K] + one E8M0 scale byte
 * per row and 32 consecutive columns, the scale bytes in a side array of dk_mx_scale_bytes(rows, K) bytes whose layout is
 * private to the library (written by dk_quantize_mx8 / dk_ln_modulate_mx8 / an MX-fp8 GEMM output, read by dk_gemm_fp8;
 * tests/_fp8.py restates it).  Strides of fp8 buffers are in bytes.  No reference counterpart: the reference quantises
 * weights to 4 bits with MLX (model_io.py:728-734) and keeps fp16 / bf16 activations. */
typedef struct dk_gemm_fp8_desc {
  const void* A;        /* e4m3 [rows, K], row stride lda; logical row m -> (m / a_seg_len) * a_seg_stride + m % a_seg_len */
  const void* A_scales; /* scale side array of the BUFFER A points into                                                  */
  const void* W;        /* e4m3 [N, K], row stride ldw                                                                   */
  const float* w_scale; /* [N]                                                                                           */
  void* C;              /* bf16 [., ldc], or with c_mx8: e4m3 [., ldc] + scales into C_scales                            */
  const void* bias;     /* bf16 [N] or NULL                                                                              */
  const void* gate;     /* as dk_gemm_desc                                                                               */
  const void* res;
  int32_t M, N, K;      /* N % 256 == 0, K % 128 == 0                                                                    */
  int32_t lda, ldw, ldc, ldr;
  int32_t a_seg_len, a_seg_stride; /* multiples of 128 rows                                                              */
  int32_t a_row0;       /* physical row of its buffer that A points at (multiple of 128)                                 */
  int32_t a_rows;       /* rows of the buffer A points into (sizes its scale array)                                      */
  int32_t c_seg_len, c_seg_stride;
  int32_t r_seg_len, r_seg_stride;
  int32_t gate_seg_len, gate_stride;
  int32_t epilogue;     /* DK_EPI_*                                                                                      */
  int32_t c_mx8;        /* 1: MX-fp8 output (M % 256 == 0)                                                               */
  void* C_scales;
  int32_t c_rows, c_row0, c_col0; /* rows of the output buffer, physical row / column (multiple of 32) that C points at  */
  /* ABI 5: optional K-split scratch, the same buffer and rules as dk_gemm_desc.workspace (dk_gemm_workspace_bytes() bytes, 256-byte aligned, last
   * 4096 bytes zero before the first use): a launch of at most half a round of 256 x 256 tiles with a long reduction is cut along K.  NULL: never. */
  void* workspace;
  size_t workspace_bytes;
} dk_gemm_fp8_desc;
int dk_gemm_fp8(const dk_gemm_fp8_desc* d, void* stream);
size_t dk_mx_scale_bytes(int64_t rows, int32_t k);
/* bf16 [M, h] (row stride ldx) -> MX-fp8 rows out_row0 .. of a [out_rows, ldo] e4m3 buffer at column out_col0 (multiple of 32) */
int dk_quantize_mx8(const void* x, int32_t ldx, int32_t M, int32_t h, void* out, int32_t ldo, void* out_scales,
                    int64_t out_rows, int32_t out_row0, int32_t out_col0, void* stream);
/* dk_ln_modulate_bf16 whose output row leaves as MX-fp8 (same arithmetic, same bf16 rounding, then quantised) */
int dk_ln_modulate_mx8(const void* x, int32_t ldx, int32_t M, int32_t h, const void* shift, const void* scale,
                       int32_t mod_stride, int32_t mod_seg_len, float eps, void* out, int32_t ldo, void* out_scales,
                       int64_t out_rows, int32_t out_row0, void* stream);
/* row pitch in bytes of an fp8 matrix with k columns that is read with a long reduction (k + 128 from 8192 on, see dk_weight_pitch) */
int32_t dk_weight_pitch_fp8(int32_t k);

/* affine_transform + LayerNorm, mmdit.py:958-972, 838-849.
 * out[m,:] = LN(x[m,:]) * (1 + scale[b,:]) + shift[b,:], b = m / mod_seg_len. */
int dk_ln_modulate_bf16(const void* x, int32_t ldx, void* out, int32_t ldo, int32_t M, int32_t h,
                        const void* shift, const void* scale, int32_t mod_stride, int32_t mod_seg_len,
                        int32_t x_seg_len, int32_t x_seg_stride, float eps, void* stream);

/* QKNorm (mmdit.py:754-764) + RoPE.apply (mmdit.py:934-942), in place on the q / k column groups
 * of a token-major QKV buffer.  q_weight/k_weight NULL = no norm; rope_table NULL = no rotation.
 * rope_table: f32 [S_pos, D/2, 2] (cos, sin); row m uses position pos_off + m % row_seg_len. */
int dk_qk_norm_rope_bf16(void* qkv, int32_t ld, int32_t q_off, int32_t k_off, int32_t rows, int32_t H,
                         int32_t D, const void* q_weight, const void* k_weight, float eps,
                         const float* rope_table, int32_t row_seg_len, int32_t row_seg_stride,
                         int32_t pos_off, void* stream);

/* RoPE.rope / _get_positions, mmdit.py:865-911: table f32 [S_txt + gh*gw, sum(axes)/2, 2] */
int dk_rope_table_f32(float* table, int32_t S_txt, int32_t gh, int32_t gw, const int32_t* axes_dim,
                      int32_t n_axes, float theta, void* stream);

/* TimestepAdapter.timestep_embedding, mmdit.py:379-389 (quirk Q2). embed_dtype: 0 bf16, 1 fp16, 2 fp32 */
int dk_timestep_embedding_bf16(const float* t_dev, int32_t n, int32_t dim, float max_period,
                               int32_t embed_dtype, void* out, void* stream);

/* LatentImageAdapter patchify, mmdit.py:292-300 (reshape_order=1) / :285-290 (0):
 * latent f32 [n_img,Hl,Wl,C] -> tokens bf16 [n_img*dup, S_i, p*p*C] */
int dk_latent_to_tokens(const float* x, void* tokens, int32_t n_img, int32_t dup, int32_t Hl, int32_t Wl,
                        int32_t C, int32_t p, int32_t reshape_order, void* stream);

/* One Euler step incl. unpatchify, x0 prediction, CFG and re-patchify:
 * CFGDenoiser.__call__ (__init__.py:691-719) after the MMDiT call, to_d (:756),
 * sample_euler body (:778-781), unpack/unpatchify (mmdit.py:304-321, 975-988).
 * x: f32 [n_img,Hl,Wl,C] updated in place; model_out: bf16 [n_img*(1+cfg_on), S_i, ld_out];
 * tokens: next step's patchified input. */
int dk_euler_cfg_step(float* x, const void* model_out, int32_t ld_out, void* tokens, int32_t n_img,
                      int32_t cfg_on, int32_t Hl, int32_t Wl, int32_t C, int32_t p, int32_t reshape_order,
                      float sigma, float sigma_next, float cfg_weight, void* stream);

/* LatentFormat.process_in/out, __init__.py:729-733: y = x * a + b (f32) */
int dk_affine_f32(const float* x, float* y, int64_t n, float a, float b, void* stream);

/* nn.GroupNorm(pytorch_compatible=True) [+ nn.silu], vae.py:34,72,78,381,91,96,398.
 * x, y: NHWC bf16 [B, HW, C]; scratch_f32 needs dk_groupnorm_scratch_floats(B, G) floats. */
size_t dk_groupnorm_scratch_floats(int32_t B, int32_t G);
int dk_groupnorm_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* gamma,
                      const void* beta, float eps, int32_t fuse_silu, float* scratch_f32, void* stream);

/* mx.softmax(scores, axis=-1), vae.py:51: in place over rows of a bf16 matrix */
int dk_softmax_rows_bf16(void* x, int32_t rows, int32_t cols, int32_t ld, void* stream);
int dk_transpose_bf16(const void* x, void* y, int32_t R, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Model level: MMDiT (mmdit.py:22-266) and VAEDecoder (vae.py:336-401)
 * ---------------------------------------------------------------------------------------- */
typedef struct dk_mmdit_config {
  int32_t num_heads, depth_multimodal, depth_unified, hidden_size, mlp_ratio;
  int32_t vae_latent_dim, patch_size, patchify_via_reshape;
  int32_t use_qk_norm, use_rope, rope_axes_dim[4], n_rope_axes, rope_theta;
  int32_t use_pos_embed, max_latent_resolution;
  int32_t pooled_text_embed_dim, token_level_text_embed_dim, frequency_embed_dim, max_period;
  int32_t embed_dtype; /* dtype the timestep embedding is evaluated in: 0 bf16, 1 fp16, 2 fp32 */
  float layer_norm_eps;
  /* 1: the FLUX.1-dev guidance embedding (MLPEmbedder "guidance_in", mmdit.py:31-36,945-955, config.py:97-111) is added to
   * the modulation vector; needs the guidance_in.mlp.layers.{0,2}.{weight,bias} tensors and dk_mmdit_set_guidance.
   * 0 (the reference's behaviour: model_io.py:109 runs FLUX.1-dev on the schnell preset, quirk Q7): absent. */
  int32_t guidance_embed;
  /* 1: the Linear layers of the transformer blocks (q/k/v, o_proj, fc1, fc2, linear1, linear2) run on the fp8 MFMA with
   * e4m3 weights ("<name>.weight_fp8" uint8 [N, dk_weight_pitch_fp8(K)] + "<name>.wscale" f32 [N] instead of
   * "<name>.weight") and MX-fp8 activations quantised on the fly (BASELINE.json configs[3]; the reference's counterpart is
   * its 4-bit nn.QuantizedLinear checkpoints, model_io.py:728-734,772-775).  Needs head_dim 128 and text / image token
   * counts that are multiples of 128.  0: bf16 weights. */
  int32_t fp8_linears;
  /* fp8 precision policy (round 5; only read with fp8_linears): the Linears of the first fp8_bf16_double_blocks double-stream blocks
   * stay bf16 ("<name>.weight", bf16 activations, the bf16 GEMM): errors made in the first blocks travel through all 57 -- measured
   * dB per step against what the blocks cost in DESIGN.md section 4.  0: every block on the fp8 path. */
  int32_t fp8_bf16_double_blocks;
} dk_mmdit_config;

typedef struct dk_mmdit dk_mmdit;

int dk_mmdit_create(const dk_mmdit_config* cfg, dk_mmdit** out);
void dk_mmdit_destroy(dk_mmdit* m);

/* Weight binding by name (model.load_weights / model.update, model_io.py:743,783).  Names follow
 * the reference module tree with the fused tensors of diffusionkit_amd/weights.py:
 *   <block>.attn.qkv.{weight,bias}, <single block>.linear2.{weight,bias}, adaLN.{weight,bias}.
 * The pointer must stay valid for the lifetime of the handle. */
int dk_mmdit_bind(dk_mmdit* m, const char* name, const void* dev_ptr);
/* number of hidden_size-wide rows of the packed adaLN output, and the row offset of one module
 * (kind 0 image stream of double block i, 1 text stream, 2 single block i, 3 final layer) */
int dk_mmdit_mod_rows(const dk_mmdit* m);
int dk_mmdit_mod_offset(const dk_mmdit* m, int32_t kind, int32_t index);

size_t dk_mmdit_workspace_bytes(const dk_mmdit* m, int32_t batch, int32_t latent_h, int32_t latent_w,
                                int32_t text_len, int32_t n_timesteps);
/* fixes the problem shape, carves the workspace, builds RoPE table / cropped pos-emb */
int dk_mmdit_prepare(dk_mmdit* m, int32_t batch, int32_t latent_h, int32_t latent_w, int32_t text_len,
                     int32_t n_timesteps, void* workspace, size_t workspace_bytes, void* stream);

/* MMDiT.cache_modulation_params(pooled_text_embeddings, timesteps), mmdit.py:77-180.
 * pooled: bf16 [batch, pooled_dim] on the device; timesteps: n host floats (already rounded to
 * the activation dtype by the caller, quirk Q1). */
int dk_mmdit_cache_modulation_params(dk_mmdit* m, const void* pooled, const float* timesteps_host,
                                     int32_t n, void* stream);
/* Guidance strength fed to the guidance embedding (configs with guidance_embed = 1; FLUX.1-dev's distilled guidance, 3.5 by
 * default): the next dk_mmdit_cache_modulation_params adds guidance_in(timestep_embedding(1000 * guidance)) to every
 * modulation vector -- the published FLUX.1-dev conditioning, which the reference's module tree declares (mmdit.py:31-36)
 * but whose call site (:219-220) it never reaches (quirk Q7). */
int dk_mmdit_set_guidance(dk_mmdit* m, float guidance);

/* Step-invariant hoist of `self.context_embedder(token_level_text_embeddings)` (mmdit.py:195, recomputed by the reference
 * in every MMDiT.__call__): embeds `text` (bf16 [batch, S_t, text_dim]) once into the engine's workspace; later
 * dk_mmdit_forward calls with text == NULL copy that result into the joint stream instead of recomputing it (same values).
 * Invalidated by dk_mmdit_prepare. */
int dk_mmdit_cache_context(dk_mmdit* m, const void* text, void* stream);

/* MMDiT.__call__ (mmdit.py:188-266) between patchify and unpatchify, for cached timestep
 * `step_index` (quirk Q11: index instead of float key).
 * tokens_in: bf16 [batch, S_i, p*p*C]; text: bf16 [batch, S_t, text_dim], or NULL after dk_mmdit_cache_context;
 * tokens_out: bf16 [batch, S_i, p*p*C] (FinalLayer output). */
int dk_mmdit_forward(dk_mmdit* m, const void* tokens_in, const void* text, int32_t step_index,
                     void* tokens_out, void* stream);
/* MultiModalTransformerBlock.__call__ / UnifiedTransformerBlock.__call__ (mmdit.py:568-675, 693-751) on a caller-supplied
 * residual stream: copies x_in (bf16 [batch, S, h], the engine's joint layout: text rows first, then image rows, per batch row)
 * into the engine's stream, runs blocks [first_block, first_block + n_blocks) of the global order (double blocks
 * 0 .. depth_multimodal - 1, then the single blocks) with the modulation parameters cached for `step_index`, and copies the
 * stream to x_out.  The operator-level boundary of one reference block: the teacher-forced parity tests drive single blocks
 * of the full-size model through it (tests/test_gpu_fullsize.py). */
int dk_mmdit_run_blocks(dk_mmdit* m, const void* x_in, void* x_out, int32_t step_index, int32_t first_block,
                        int32_t n_blocks, void* stream);
/* read-only view of an internal buffer for parity taps: 0 = joint residual stream [B,S,h],
 * 1 = modulation table [n*B, rows*h] */
const void* dk_mmdit_debug_buffer(const dk_mmdit* m, int32_t which);

typedef struct dk_vae_config {
  int32_t in_channels, out_channels, block_out_channels[4], n_blocks, layers_per_block, resnet_groups;
  float group_norm_eps;
} dk_vae_config;
typedef struct dk_vae dk_vae;
int dk_vae_create(const dk_vae_config* cfg, dk_vae** out);
void dk_vae_destroy(dk_vae* v);
/* names = reference module tree (vae.py / model_io.py:411-486), conv weights flattened to
 * [O, 9*I]; conv_in.weight zero-padded to I = 64 */
int dk_vae_bind(dk_vae* v, const char* name, const void* dev_ptr);
size_t dk_vae_workspace_bytes(const dk_vae* v, int32_t batch, int32_t latent_h, int32_t latent_w);
/* VAEDecoder.__call__ (vae.py:386-401) + decode_latents_to_image (__init__.py:581-584) +
 * uint8 conversion (__init__.py:525-526).  latent: f32 [B,h,w,16];
 * image_f32: [B,8h,8w,3] in [0,1] or NULL; image_u8: [B,8h,8w,3] or NULL;
 * raw_bf16: decoder output before the clip, [B,8h,8w,4] (3 used) or NULL. */
int dk_vae_decode(dk_vae* v, const float* latent, int32_t batch, int32_t latent_h, int32_t latent_w,
                  float* image_f32, uint8_t* image_u8, void* raw_bf16, void* workspace,
                  size_t workspace_bytes, void* stream);

/* VAEEncoder.__call__ (vae.py:456-467) on a dk_vae created with the encoder's configuration and
 * bound to its module names (conv_in, down_blocks.{i}.resnets.{r}, down_blocks.{i}.downsample,
 * mid_blocks.{0,1,2}, conv_norm_out, conv_out).  image: f32 [B,H,W,in_channels] in [-1,1]
 * (read_image, __init__.py:536-551); moments (mean | logvar, __init__.py:588-589):
 * bf16 [B,H/8,W/8,ldm] (ldm multiple of 4, >= out_channels) and / or f32 [B,H/8,W/8,out_channels]. */
size_t dk_vae_encoder_workspace_bytes(const dk_vae* v, int32_t batch, int32_t image_h, int32_t image_w);
int dk_vae_encode(dk_vae* v, const float* image, int32_t batch, int32_t image_h, int32_t image_w,
                  void* moments_bf16, int32_t ldm, float* moments_f32, void* workspace,
                  size_t workspace_bytes, void* stream);
/* encode_image_to_latents tail (__init__.py:589-594):
 * latent = mean + exp(0.5 * clip(logvar, -30, 20)) * noise, f32 [n_pixels, latent_channels] */
int dk_latent_sample_f32(const void* moments_bf16, int32_t ldm, const float* noise, float* latent,
                         int64_t n_pixels, int32_t latent_channels, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart; the reference times phases with time.time(),
 * mlx/__init__.py:315-530).  When enabled, every bf16 GEMM (class 0), conv (1), attention (2) and fp8 GEMM (3) launch
 * is bracketed by HIP events on its launch stream; dk_profile_read sums elapsed time, algorithmic
 * FLOPs and launch count of one class since the last dk_profile_enable call.
 * ---------------------------------------------------------------------------------------- */
int dk_profile_enable(int32_t on);
int dk_profile_read(int32_t kernel_class, double* total_ms, double* total_flops, int64_t* launches);

/* Tuning knobs for A/B measurements (no reference counterpart); -1 = automatic (the shipped default) for every key.
 * "gemm": 128 = 128x128 tiles only, 9 = the 8-wave 256x256 kernel on every shape it accepts, 10 = the one-wave-per-SIMD 256x256 kernel
 * (asm body) on every shape IT accepts; "gemm_v4": 0 = the automatic choice never takes the latter; "gemm_skew": start skew of that kernel's
 * multi-round launches in 0.25 us steps (-1: none); "gemm_mf": 8 / 7 = 256- / 224-row
 * tiles; "gemm_split": 0 / 1 = remainder-wave K split never / whenever possible; "gemm_split_min": K-tile steps a workgroup must save before a
 * Linear of at most half a round of tiles is cut along K as a whole (default 32); "gemm_pair_nk": K-tile steps from which an image + text pair with
 * a small extra round is grouped and cut (default 24); "gemm_fuse_k" / "gemm_fuse_q": 0 / 1 = the keys' /
 * queries' QKNorm + RoPE in the q/k/v projection's tail off / on; "attn": kernel of dk_attention_bf16 (4 lean kernel,
 * 9 phase-alternating kernel: head_dim 128 only, falls back to 4 otherwise; 10 one-wave-per-SIMD kernel with the generated asm
 * tile loop: head_dim 128, S a multiple of 256 and >= 768, falls back to 9 otherwise -- the automatic choice from S = 2048 on, and from
 * S = 1024 for launches of at least three quarters of a round of 256-query blocks);
 * "attn_split": key ranges of the one-wave-per-SIMD kernel's last-round query blocks (-1 automatic, 0 never, 2..4);
 * "attn_fuse_q": 0 = stand-alone query QKNorm + RoPE pass; "conv_halo": 0 = VAE convolutions through the GEMM form;
 * "conv_v4": the fused VAE convolutions with >= 256 output channels on the one-wave-per-SIMD kernel (1 where one image fills the
 * CUs, 0 never, 2 wherever eligible);
 * "pitch_min_k": rows of at least this many elements are stored padded (dk_weight_pitch).
 * Returns 0, or -1 for an unknown key. */
int dk_tune_set(const char* key, int32_t value);

/* Row pitch, in elements, the engine expects for a weight matrix whose rows hold k elements and that it reads with a long
 * reduction: k itself below 8192, k + 64 from there on (the 24-30 KB row stride of the [h, 4h] "mlp.fc2.weight" of the
 * double-stream blocks and the [h, 5h] "linear2.weight" of the single-stream blocks of FLUX camps on a few memory channels;
 * the pad columns are never read).  The host packer (diffusionkit_amd/weights.py: pack_mmdit) lays those two tensors out
 * with this pitch before dk_mmdit_bind; every other tensor is dense.  No reference counterpart (mlx arrays are dense). */
int32_t dk_weight_pitch(int32_t k);

#ifdef __cplusplus
}
#endif
#endif /* DK_HIP_H */
