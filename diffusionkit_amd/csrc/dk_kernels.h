// Internal launcher declarations shared by the kernel translation units and the C-ABI layer.
#pragma once
#include "../../include/dk_hip.h"
#include "dk_common.h"

// ---- GEMM / implicit-GEMM conv ---------------------------------------------------------
// C[m, n] = epi(alpha * sum_k A[m, k] * W[n, k] + bias[n])   (bf16 in, fp32 accumulate, bf16 out)
// Logical row m maps to physical row (m / seg_len) * seg_stride + (m % seg_len); this is how a
// stream (text rows / image rows of a joint [B, S, h] buffer) is addressed without copies.
// epilogue codes: DK_EPI_* in include/dk_hip.h

struct GemmParams {
  const bf16_t* A;
  const bf16_t* W;
  bf16_t* C;
  const bf16_t* bias;  // [N] or null
  const bf16_t* gate;  // [n_batch, gate_stride] or null
  const bf16_t* res;   // same row mapping as C, leading dim ldr
  int M, N, K;
  int lda, ldc, ldr;
  int ldw;  // row stride of W in elements (>= K; a padded stride avoids L2-channel camping on power-of-two K)
  int a_seg_len, a_seg_stride;
  int c_seg_len, c_seg_stride;
  int r_seg_len, r_seg_stride;
  int gate_seg_len, gate_stride;
  float alpha;
  int epi;
  // implicit-GEMM 3x3 conv (pad 1, stride 1) over NHWC input; M = cB*cH*cW output pixels,
  // K = 9*cC. If ups != 0 the conv input is the nearest-neighbour x2 upsampling of the stored
  // [cB, cH/2, cW/2, cC] tensor (vae.py:20-25 folded into the gather); ups == 2: stride-2 conv over the stored
  // [cB, 2*cH, 2*cW, cC] tensor padded by one zero row / column at the bottom / right (vae.py:141-143).
  int conv;
  int cB, cH, cW, cC, ups;
  const bf16_t* zeros;  // >= 128 B of zeros (padding taps)
  void* workspace;      // optional K-split workspace (dk_gemm_split_workspace_bytes()), else null
  size_t workspace_bytes;
  // optional column split (256^2 kernel only): output columns >= n_split go to C2 (leading dim ldc2, same row
  // map as C, column index rebased to 0) with epilogue epi2 -- the fused linear1 of the single-stream blocks
  // (q/k/v projection | fc1 + GELU over one read of the modulated activations, mmdit.py:693-751)
  int n_split;
  bf16_t* C2;
  int ldc2;
  int epi2;
  // optional QKNorm + RoPE of the KEY columns (mmdit.py:754-764, 934-942) inside the tail of a q / k / v projection on the 256^2
  // kernel (bias-only epilogue): columns [kn_col0, kn_col1) -- multiples of 256 -- are heads of kn_D (128 or 64) columns; a row's
  // head is normalised with weight kn_w [kn_D] and rotated by the cos / sin table kn_rope [S_pos, kn_D / 2, 2] (null: no rotation)
  // at position kn_pos_off + (row % kn_seg_len).  Same arithmetic and rounding points as dk_qk_norm_rope_kernel.  Null kn_w: off.
  // (a launch that does not reach the 256^2 kernel runs dk_launch_qk_norm_rope on its output instead: gemm.hip)
  const bf16_t* kn_w;
  const float* kn_rope;
  int kn_col0, kn_col1, kn_D, kn_pos_off, kn_seg_len;
  float kn_eps;
  // round 4: the QUERY columns [qn_col0, qn_col1) the same way with their own weight qn_w (same head size, table, positions, eps) -- the
  // attention kernel then loads finished queries.  Needs kn_w; null qn_w: queries untouched
  const bf16_t* qn_w;
  int qn_col0, qn_col1;
};
int dk_launch_gemm(const GemmParams& p, hipStream_t stream);
// Plan mode (dk_gemm_plan / dk_gemm_pair_plan, host only): while g_dk_gemm_plan points at a record, the launchers fill it with what they WOULD launch and
// return without touching the device -- the decision code is the launch code itself, so a CPU test can sweep shapes over the dispatch rules
struct DkGemmPlan {
  int kernel;      // 128: dk_gemm_bf16_kernel (128 x 128 tiles), 3: dk_gemm256v3_kernel (8 waves), 4: dk_gemm256v4_kernel (one wave per SIMD)
  int tile_rows;   // 128 / 224 / 256
  int tiles;       // output tiles of the launch (both problems of a grouped one)
  int workgroups;  // grid size (tiles that are cut along K count once per piece)
  int split_tiles; // tiles cut along K (0: none)
  int k_pieces;    // pieces per cut tile
  int ks;          // K-tile steps (64 elements each) of the finisher piece; whole tiles: K / 64
  int n_cu;        // compute units the rules assumed
  int launches;    // kernel launches the call expands to (a column split or a fused-norm fallback on the small kernel: 2)
};
extern thread_local DkGemmPlan* g_dk_gemm_plan;
extern int g_dk_gemm_mode;
extern int g_dk_v3_split;  // gemm256v3.hip: remainder-wave K split (-1 auto, 0 off, 1 whenever possible)
extern int g_dk_pair_split_nk;  // gemm.hip: see dk_launch_gemm_pair
extern int g_dk_v3_split_min;  // ... saved K-tile steps below which an all-remainder Linear stays whole (-1: default)
extern int g_dk_v3_mf;     // gemm256v3.hip: wave-tile height in 16-row fragments (-1 auto, 8 = 256-row tiles, 7 = 224-row tiles)
// workspace of the remainder-wave K split (fp32 slabs + flags; its last 4 KiB -- the flag region -- must be zero before the first
// launch; the kernels leave it zero)
size_t dk_gemm_split_workspace_bytes();
bool dk_gemm256v3_eligible(const GemmParams& p);  // N % 128 == 0, K % 64 == 0, any M, any row-segment maps
int dk_launch_gemm256v3(const GemmParams& p, const GemmParams* p2, hipStream_t stream);
bool dk_gemm256v3_splits_whole_launch(const GemmParams& p, const GemmParams* p2);  // <= half a round of tiles, every tile cut along K (needs p.workspace)
// gemm256v4.hip: one wave per SIMD, 256 accumulators in AGPRs, hand-scheduled asm body (N % 256 == 0, no conv / K split / half tiles)
extern int g_dk_v4_auto;  // gemm.hip
extern int g_dk_v4_skew;  // gemm256v4.hip
bool dk_gemm256v4_eligible(const GemmParams& p);
bool dk_gemm256v4_uniform_tiles(const GemmParams& p, int bm);
int dk_gemm256v4_pick_mf(const GemmParams& p, const GemmParams* p2, int n_cu);
int dk_launch_gemm256v4(const GemmParams& p, const GemmParams* p2, hipStream_t stream);
int dk_launch_gemm256v3_raw(const GemmParams& p, const GemmParams& pb, int tiles_a, int tiles_b, hipStream_t stream);  // gemm256v3.hip (16x16x32 MFMA K loop)
// two problems with the same N, K, epilogue in one launch (image + text stream of a double block); falls
// back to two launches when the pair is not eligible for the grouped kernel
int dk_launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t stream);

// ---- fp8 GEMM (gemm256f8.hip): e4m3 weights with per-output-channel fp32 scales, MX-fp8 activations -------------------
// C[m, n] = epi(wscale[n] * sum_k A[m, k] * 2^(SA[m, k / 32] - 127) * W[n, k] + bias[n]); same row-segment maps, epilogues,
// grouped launch and column split as GemmParams.  Strides of A / W and of an MX-fp8 output are in BYTES (= elements).
struct GemmF8Params {
  const unsigned char* A;   // [rows, K] e4m3, row pitch lda
  const unsigned char* SA;  // E8M0 scales of the A BUFFER (dk_mx_scale_index over its physical rows), sa_nblk 128-row blocks per K-tile
  const unsigned char* W;   // [N, K] e4m3, row pitch ldw
  const float* wscale;      // [N]
  void* C;                  // bf16 [., ldc] or, with c_mx8, e4m3 [., ldc] + scales into SC
  const bf16_t* bias;
  const bf16_t* gate;
  const bf16_t* res;
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int a_seg_len, a_seg_stride, a_row0;  // a_row0: physical row of the A buffer that pointer A addresses (scale indexing)
  int sa_nblk;
  int c_seg_len, c_seg_stride;
  int r_seg_len, r_seg_stride;
  int gate_seg_len, gate_stride;
  int epi;
  int c_mx8;
  // optional column split: output columns >= n_split go to C2 (column index rebased to 0) with epilogue epi2
  int n_split;
  void* C2;
  int ldc2, epi2, c2_mx8;
  // MX-fp8 outputs: scale side array of the OUTPUT buffer, its 128-row block count, the physical row that pointer C / C2
  // addresses, and the 32-column block index of output column 0 inside that buffer's rows
  unsigned char* SC;
  int sc_nblk, c_row0, sc_kb0;
  // optional QKNorm + RoPE of the key columns in the tail (see GemmParams; bf16 first output, bias-only epilogue)
  const bf16_t* kn_w;
  const float* kn_rope;
  int kn_col0, kn_col1, kn_D, kn_pos_off, kn_seg_len;
  float kn_eps;
  const bf16_t* qn_w;  // ... and of the query columns [qn_col0, qn_col1) (see GemmParams)
  int qn_col0, qn_col1;
  // optional K-split scratch (dk_gemm_split_workspace_bytes(), the bf16 kernels' buffer: fp32 slabs + flags, flag region zero between launches):
  // round 6 -- a launch of at most half a round of tiles with a long reduction is cut along K (FLUX below 1024 x 1024)
  void* workspace;
  size_t workspace_bytes;
};
bool dk_gemm256f8_eligible(const GemmF8Params& p);
int dk_launch_gemm256f8(const GemmF8Params& p, const GemmF8Params* p2, hipStream_t stream);
// bf16 rows -> MX-fp8 rows + scales (fp8_ops.hip).  x: [M, h] bf16 through the row map (seg_len, seg_stride); out row m at
// physical row out_row0 + (m / o_seg_len) * o_seg_stride + m % o_seg_len of the fp8 buffer (pitch ldo bytes, column
// offset col0, a multiple of 32).  dk_launch_ln_modulate_mx8 = dk_launch_ln_modulate with that output.
struct Mx8Out {
  unsigned char* out;  // buffer base (physical row 0, column 0)
  unsigned char* scales;
  int ldo, n_blk128, row0, seg_len, seg_stride, col0;
};
int dk_launch_quantize_mx8(const bf16_t* x, int ldx, int x_seg_len, int x_seg_stride, int M, int h, const Mx8Out& o, hipStream_t stream);
int dk_launch_ln_modulate_mx8(const bf16_t* x, int ldx, int M, int h, const bf16_t* shift, const bf16_t* scale, int mod_stride, int seg_len,
                              int x_seg_len, int x_seg_stride, float eps, const Mx8Out& o, hipStream_t stream);
int dk_launch_ln_modulate2_mx8(const bf16_t* x0, int M0, const bf16_t* shift0, const bf16_t* scale0, int seg0, const Mx8Out& o0,
                               const bf16_t* x1, int M1, const bf16_t* shift1, const bf16_t* scale1, int seg1, const Mx8Out& o1, int ldx, int h,
                               int mod_stride, int x_seg_stride, float eps, hipStream_t stream);

// optional HIP-event timing of the dominant kernels (profile.hip); cls: 0 GEMM, 1 conv, 2 attention, 3 fp8 GEMM
void dk_prof_begin(int cls, double work, hipStream_t st);
void dk_prof_end(hipStream_t st);

// ---- attention -------------------------------------------------------------------------
struct AttnParams {
  const bf16_t* Q;  // row s of batch b: Q + (b*S + s)*ld + head*D
  const bf16_t* K;
  const bf16_t* V;
  bf16_t* O;  // O + (b*S + s)*ldo + head*D
  int B, H, S, D;
  int ld, ldo;
  float scale;
  // optional additive score bias (text encoders: CLIP's causal mask clip.py:83-89, T5's relative-position bias
  // t5.py:61-88): scores = scale * q.k + bias[head * bias_head_stride + q * ldb + k]; ldb a multiple of 64 >= S
  const bf16_t* bias = nullptr;
  long bias_head_stride = 0;
  int ldb = 0;
  // optional QKNorm (mmdit.py:754-764) + RoPE (:934-942) of the QUERY rows inside the kernel's Q load (the keys keep their
  // own pass, dk_launch_qk_norm_rope*(..., k_only)): rows s < qn_split use weight qn_a, the others qn_b; q_rope = the
  // [S, D/2, 2] cos/sin table indexed by s, or null
  const bf16_t* qn_a = nullptr;
  const bf16_t* qn_b = nullptr;
  int qn_split = 0;
  float qn_eps = 1e-6f;
  const float* q_rope = nullptr;
  // lab only: trace buffer of attention4.hip's DK4_TRACE builds (scripts/attn_trace.py)
  void* bal_ws = nullptr;
  // optional MX-fp8 copy of the output (fp8_linears: the o-projection's activation operand): row (b*S + s) of O8 at o8_ld bytes
  // per row, head h at byte column h*D; E8M0 scales of 32-column blocks in O8_scales (dk_mx_scale_index over o8_nblk 128-row
  // blocks).  dk_attn4_fwd_kernel writes it INSTEAD of O from its accumulators (values rounded to bf16 first, as the separate
  // quantiser pass over O sees them); for the other kernels dk_launch_attention runs that pass behind the launch.
  unsigned char* O8 = nullptr;
  unsigned char* O8_scales = nullptr;
  int o8_ld = 0, o8_nblk = 0;
  // attention5.hip, filled in by its launcher: workgroups 0 .. a5_whole - 1 take whole query blocks; the others one of a5_split key ranges of
  // a block of the last, partial round of the CUs and leave (O / l in bf16, offset, l) in a5_ws for dk_attn5_merge_kernel
  int a5_whole = 0, a5_split = 1;
  void* a5_ws = nullptr;
};
void dk_set_attention_workspace(void* ws, size_t bytes = 0);  // attention.hip: thread-local workspace, picked up by dk_launch_attention
void* dk_get_attention_workspace();
size_t dk_get_attention_workspace_bytes();
extern int g_dk_attn_mode;
int dk_launch_attention(const AttnParams& p, hipStream_t stream);
int dk_launch_attention2(const AttnParams& p, int waves, hipStream_t stream);  // attention2.hip (VALU-lean variant)
int dk_launch_attention4(const AttnParams& p, hipStream_t stream);             // attention4.hip (the waves of a SIMD in opposite phases; D = 128, no score bias)
bool dk_attention5_eligible(const AttnParams& p);                              // attention5.hip (one wave per SIMD, asm tile loop; D = 128, S % 256 == 0, no score bias)
int dk_launch_attention5(const AttnParams& p, hipStream_t stream);
extern int g_dk_attn5_split;

// ---- single-head D = 512 attention of the VAE's mid block (attention512.hip) -------------------------------
struct Attn512Params {
  const bf16_t* Q;   // [B, T, ld]
  const bf16_t* K;   // [B, T, ld]
  const bf16_t* Vt;  // TRANSPOSED values: [B, 512, Tp], rows zero-padded beyond T (dk_launch_transpose with ldy = Tp)
  bf16_t* O;         // [B, T, ldo]
  int T, Tp, B, ld, ldo;
  float scale;
};
int dk_launch_attention512(const Attn512Params& p, hipStream_t stream);

// ---- text-conditioning kernels (text_ops.hip) ---------------------------------------------------
int dk_launch_embedding(const bf16_t* table, const int* ids, const bf16_t* pos, int pos_rows, bf16_t* out, float* out_f32, int n, int dim,
                        int vocab, hipStream_t stream);
int dk_launch_layernorm(const bf16_t* x, bf16_t* out, int M, int h, const bf16_t* w, const bf16_t* b, float eps, hipStream_t stream);
int dk_launch_t5_rmsnorm(const float* x, bf16_t* out, int M, int h, const bf16_t* w, float eps, hipStream_t stream);
int dk_launch_text_elementwise(const bf16_t* a, const bf16_t* b, bf16_t* y, float* r, long n, int op, hipStream_t stream);
int dk_launch_t5_bias(const bf16_t* emb, const int* rel_bucket, int H, int S, int ld, bf16_t* out, hipStream_t stream);

// ---- elementwise / normalisation ---------------------------------------------------------
// out[m, :] = bf16( LN(x[m, :]) * bf16(1 + scale[b, :]) + shift[b, :] ), b = m / seg_len
int dk_launch_ln_modulate(const bf16_t* x, int ldx, bf16_t* out, int ldo, int M, int h,
                          const bf16_t* shift, const bf16_t* scale, int mod_stride, int seg_len,
                          int x_seg_len, int x_seg_stride, float eps, hipStream_t stream);
// in-place per-head RMSNorm (learned weight) + RoPE on the q and k column groups of a QKV buffer
int dk_launch_qk_norm_rope(bf16_t* qkv, int ld, int q_off, int k_off, int rows, int H, int D,
                           const bf16_t* qw, const bf16_t* kw, float eps, const float* rope,
                           int row_seg_len, int row_seg_stride, int pos_off, int S_pos,
                           hipStream_t stream, int k_only = 0);
// two row sets (image / text stream of a double block) per launch
int dk_launch_ln_modulate2(const bf16_t* x0, bf16_t* out0, int M0, const bf16_t* shift0, const bf16_t* scale0, int seg0, const bf16_t* x1,
                           bf16_t* out1, int M1, const bf16_t* shift1, const bf16_t* scale1, int seg1, int ldx, int ldo, int h,
                           int mod_stride, int x_seg_stride, float eps, hipStream_t stream);
int dk_launch_qk_norm_rope2(bf16_t* qkv0, int rows0, const bf16_t* qw0, const bf16_t* kw0, int seg0, int pos0, bf16_t* qkv1, int rows1,
                            const bf16_t* qw1, const bf16_t* kw1, int seg1, int pos1, int ld, int q_off, int k_off, int H, int D,
                            float eps, const float* rope, int row_seg_stride, hipStream_t stream, int k_only = 0);
int dk_launch_silu(const bf16_t* x, bf16_t* y, long n, hipStream_t stream);
int dk_launch_add(const bf16_t* a, const bf16_t* b, int b_rows, bf16_t* y, int rows, int cols, hipStream_t stream);
int dk_launch_timestep_embedding(const float* t, int n, int rep, int dim, float max_period, int embed_dtype,
                                 bf16_t* out, hipStream_t stream);
int dk_launch_rope_table(float* table, int S_txt, int gh, int gw, const int* axes, int n_axes, float theta,
                         hipStream_t stream);
int dk_launch_f32_to_bf16(const float* x, bf16_t* y, long n, hipStream_t stream);
int dk_launch_affine_f32(const float* x, float* y, long n, float a, float b, hipStream_t stream);
// latent [n_img, Hl, Wl, C] fp32 -> tokens [B, S_i, p*p*C] bf16 (B = n_img * dup)
int dk_launch_latent_to_tokens(const float* x, bf16_t* tok, int n_img, int dup, int Hl, int Wl, int C, int p,
                               int reshape_order, hipStream_t stream);
// fused x0-prediction + CFG + Euler update (+ re-patchify for the next step)
int dk_launch_euler_step(float* x, const bf16_t* model_out, int ld_out, bf16_t* tok, int n_img, int cfg_on,
                         int Hl, int Wl, int C, int p, int reshape_order, float sigma, float sigma_next,
                         float cfg_weight, hipStream_t stream);

// ---- 3x3 conv with LDS halo staging and the GroupNorm-apply + SiLU prologue (conv_halo.hip) ----------------------------------
struct ConvHaloParams {
  const bf16_t* x;        // stored input NHWC [B, H >> ups, W >> ups, C] (RAW values when gn_ss is given)
  const bf16_t* w;        // [O, ldw] K-major: column tap * C + c (tap = ky * 3 + kx), then (x2) 9 * C + c2
  const bf16_t* bias;     // [O]
  const bf16_t* bias2;    // bias of the shortcut extension, or null
  const bf16_t* res;      // residual NHWC [B, H, W, ldr], or null
  bf16_t* y;              // NHWC [B, H, W, ldy]
  const float* gn_ss;     // [B][2][C] fp32 (scale | shift) of the GroupNorm applied to x on load, or null (x is used as it is)
  int gn_silu;            // SiLU behind that GroupNorm
  const bf16_t* x2;       // 1x1 shortcut input NHWC [B, H, W, C2] (raw), or null
  int C2;
  float* stats_out;       // [B][tiles per image][G_out][2] (sum, sum of squares) of the stored output, or null
  int G_out;
  float* img;             // image tail (O <= 4): clip(y / 2 + 0.5) f32 [npix, 3]
  unsigned char* u8;      // ... (x 255) truncated to uint8 [npix, 3]
  bf16_t* raw;            // ... y itself, bf16 [npix, 4]
  int out_channels;
  int B, H, W, C, O, ups, ldw, ldy, ldr;
};
bool dk_conv_halo_eligible(const ConvHaloParams& p, bool img);
int dk_launch_conv_halo(const ConvHaloParams& p, hipStream_t stream);
// conv256v4.hip: the same contract in the one-wave-per-SIMD frame (16 x 16 pixels x 256 channels per workgroup, asm body); no shortcut
// extension, no image tail.  dk_launch_conv_halo routes to it when dk_conv256v4_wanted (dk_tune_set("conv_v4", 0 | 1 | 2))
extern int g_dk_conv_v4;
bool dk_conv256v4_eligible(const ConvHaloParams& p);
bool dk_conv256v4_wanted(const ConvHaloParams& p);
int dk_launch_conv256v4(const ConvHaloParams& p, hipStream_t stream);

// ---- VAE ops -------------------------------------------------------------------------------
// the second pass of the GroupNorm statistics alone: per (batch, group) the partials [B][nchunk][G][2] -> mean / rstd, and
// (gamma given) the per-channel table scale_shift [B][2][C]: scale = rstd * gamma, shift = beta - mean * scale
int dk_launch_groupnorm_finalize(const float* partial, int nchunk, int B, int G, double count, float eps, float* mean_rstd,
                                 const bf16_t* gamma, const bf16_t* beta, int C, float* scale_shift, hipStream_t stream);
int dk_launch_groupnorm_partials(const bf16_t* x, int B, long HW, int C, int G, float* partial, int nchunk, hipStream_t stream);
int dk_launch_groupnorm_stats(const bf16_t* x, int B, long HW, int C, int G, float* partial, int nchunk,
                              float* mean_rstd, float eps, hipStream_t stream);
int dk_launch_groupnorm_apply(const bf16_t* x, bf16_t* y, int B, long HW, int C, int G, const float* mean_rstd,
                              const bf16_t* gamma, const bf16_t* beta, int do_silu, hipStream_t stream);
int dk_launch_softmax_rows(bf16_t* x, int rows, int cols, int ld, hipStream_t stream);
int dk_launch_transpose(const bf16_t* x, bf16_t* y, int R, int Cc, hipStream_t stream, int ldy = 0);  // ldy > R: zero-padded rows
int dk_launch_pad_channels(const float* x, bf16_t* y, long npix, int C, int Cpad, hipStream_t stream);
int dk_launch_image_post(const bf16_t* x, int ldx, float* img, unsigned char* u8, long npix, hipStream_t stream);
int dk_launch_latent_sample(const bf16_t* mom, int ldm, const float* noise, float* out, long npix, int L, hipStream_t stream);
int dk_launch_bf16_rows_to_f32(const bf16_t* x, int ldx, float* y, long npix, int C, hipStream_t stream);
