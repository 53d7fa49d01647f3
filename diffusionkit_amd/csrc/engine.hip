// C-ABI layer (include/dk_hip.h) + the host-side MMDiT / VAE-decoder engines that sequence the
// gfx950 kernels.  Host code only: no kernels are defined here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dk_hip.h"
#include "dk_kernels.h"

static thread_local std::string g_last_error;
void dk_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" int dk_abi_version(void) { return DK_ABI_VERSION; }
extern "C" const char* dk_last_error(void) { return g_last_error.c_str(); }

int g_dk_attn_mode = -1;
int g_dk_fuse_k = 1;  // dk_tune_set("gemm_fuse_k", v): QKNorm + RoPE of the keys inside the q / k / v projection's tail (1, default) or as a separate pass (0)
int g_dk_fuse_qg = -1;  // dk_tune_set("gemm_fuse_q", v): ... and of the QUERIES there too (1) instead of in the attention kernel's Q load (0); -1 (default): on
                         // the fp8 path only -- measured in the model (profiles/r04_attention_qfuse_in_model.log): +1.0 % per step with fp8 weights, flat in bf16
int g_dk_fuse_q = 1;  // dk_tune_set("attn_fuse_q", v): QKNorm + RoPE of the queries inside the attention kernel's Q load (1, default) or as a separate pass (0)
// Rows of K >= g_dk_pitch_min_k elements (the [h, 5h] linear2 and [h, 4h] fc2 weights of FLUX and the activations they
// multiply) are stored with 64 elements of padding: a 24-30 KB row stride makes the K-tile DMA of 256 rows camp on a few
// memory channels (linear2 of FLUX: 369 -> 345 us with the padded pitch, profiles/archive/r01_gemm_lab_pitch.log).
int g_dk_pitch_min_k = 8192;
// dk_tune_set("conv_halo", v): the VAE's norm -> silu -> conv stages on the halo-staged kernel with the GroupNorm applied on load
// (conv_halo.hip): -1 (default) / 1 wherever the shape allows (measured: decode 15.1 -> 12.5 ms, against 13.0 ms when only the
// stages with fewer than 256 output channels use it -- the 256 / 512-channel convs are ~7 % slower than on the 256 x 256 implicit-GEMM
// kernel, but lose their GroupNorm-apply passes), 2 only below 256 output channels, 0 never
int g_dk_conv_halo = -1;
extern "C" int32_t dk_weight_pitch(int32_t k) { return k >= g_dk_pitch_min_k ? k + 64 : k; }
extern "C" int dk_tune_set(const char* key, int32_t value) {
  DK_REQUIRE(key != nullptr, "null key");
  if (strcmp(key, "gemm") == 0) { g_dk_gemm_mode = value; return 0; }
  if (strcmp(key, "gemm_v4") == 0) { g_dk_v4_auto = value; return 0; }
  if (strcmp(key, "gemm_skew") == 0) { g_dk_v4_skew = value; return 0; }
  if (strcmp(key, "attn") == 0) { g_dk_attn_mode = value; return 0; }
  if (strcmp(key, "attn_fuse_q") == 0) { g_dk_fuse_q = value; return 0; }
  if (strcmp(key, "attn_split") == 0) { g_dk_attn5_split = value; return 0; }
  if (strcmp(key, "gemm_fuse_k") == 0) { g_dk_fuse_k = value; return 0; }
  if (strcmp(key, "gemm_fuse_q") == 0) { g_dk_fuse_qg = value; return 0; }
  if (strcmp(key, "gemm_split") == 0) { g_dk_v3_split = value; return 0; }
  if (strcmp(key, "gemm_split_min") == 0) { g_dk_v3_split_min = value; return 0; }
  if (strcmp(key, "gemm_pair_nk") == 0) { g_dk_pair_split_nk = value; return 0; }
  if (strcmp(key, "gemm_mf") == 0) { g_dk_v3_mf = value; return 0; }
  if (strcmp(key, "pitch_min_k") == 0) { g_dk_pitch_min_k = value; return 0; }
  if (strcmp(key, "conv_halo") == 0) { g_dk_conv_halo = value; return 0; }
  if (strcmp(key, "conv_v4") == 0) { g_dk_conv_v4 = value; return 0; }
  dk_set_error(std::string("unknown tuning key: ") + key);
  return -1;
}

static inline hipStream_t S_(void* s) { return (hipStream_t)s; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// operator-level wrappers
// ---------------------------------------------------------------------------------------------
static GemmParams gemm_params_from_desc(const dk_gemm_desc* d) {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16_t*)d->A; p.W = (const bf16_t*)d->W; p.C = (bf16_t*)d->C;
  p.bias = (const bf16_t*)d->bias; p.gate = (const bf16_t*)d->gate; p.res = (const bf16_t*)d->res;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldc = d->ldc; p.ldr = d->ldr;
  p.a_seg_len = d->a_seg_len > 0 ? d->a_seg_len : d->M; p.a_seg_stride = d->a_seg_stride;
  p.c_seg_len = d->c_seg_len > 0 ? d->c_seg_len : d->M; p.c_seg_stride = d->c_seg_stride;
  p.r_seg_len = d->r_seg_len > 0 ? d->r_seg_len : d->M; p.r_seg_stride = d->r_seg_stride;
  p.gate_seg_len = d->gate_seg_len > 0 ? d->gate_seg_len : d->M; p.gate_stride = d->gate_stride;
  p.alpha = d->alpha; p.epi = d->epilogue; p.ldw = d->ldw;
  p.workspace = d->workspace; p.workspace_bytes = d->workspace_bytes;
  return p;
}

extern "C" size_t dk_gemm_workspace_bytes(void) { return dk_gemm_split_workspace_bytes(); }

extern "C" int dk_gemm_bf16(const dk_gemm_desc* d, void* stream) {
  DK_REQUIRE(d != nullptr, "null descriptor");
  return dk_launch_gemm(gemm_params_from_desc(d), S_(stream));
}

extern "C" int dk_gemm_plan(const dk_gemm_desc* d, const dk_gemm_desc* d2, dk_gemm_plan_t* plan) {
  DK_REQUIRE(d != nullptr && plan != nullptr, "null descriptor / plan");
  static_assert(sizeof(dk_gemm_plan_t) == sizeof(DkGemmPlan), "the ABI record mirrors the launchers' record");
  DkGemmPlan rec;
  memset(&rec, 0, sizeof(rec));
  struct Scope {  // (the launchers see the record only for the duration of this call, whatever path returns)
    explicit Scope(DkGemmPlan* r) { g_dk_gemm_plan = r; }
    ~Scope() { g_dk_gemm_plan = nullptr; }
  } scope(&rec);
  const int rc = d2 != nullptr ? dk_launch_gemm_pair(gemm_params_from_desc(d), gemm_params_from_desc(d2), nullptr) : dk_launch_gemm(gemm_params_from_desc(d), nullptr);
  memcpy(plan, &rec, sizeof(rec));
  return rc;
}

// workspace: optional K-split scratch (dk_gemm_split_workspace_bytes) for stages whose tiles fill only half the CUs
static int conv3x3_launch(const dk_conv_desc* d, void* workspace, hipStream_t stream) {
  DK_REQUIRE(d != nullptr, "null descriptor");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const bf16_t*)d->x; p.W = (const bf16_t*)d->w; p.C = (bf16_t*)d->y;
  p.bias = (const bf16_t*)d->bias; p.res = (const bf16_t*)d->res;
  p.M = d->B * d->H * d->W; p.N = d->O; p.K = 9 * d->C;
  p.lda = d->C; p.ldc = d->ldy; p.ldr = d->ldr;
  p.a_seg_len = p.c_seg_len = p.r_seg_len = p.gate_seg_len = p.M;
  p.alpha = 1.0f; p.epi = d->epilogue;
  p.conv = 1; p.cB = d->B; p.cH = d->H; p.cW = d->W; p.cC = d->C; p.ups = d->upsample;
  p.zeros = (const bf16_t*)d->zeros;
  if (workspace) { p.workspace = workspace; p.workspace_bytes = dk_gemm_split_workspace_bytes(); }
  DK_REQUIRE(d->upsample >= 0 && d->upsample <= 2, "upsample: 0 plain, 1 nearest-x2 input view, 2 stride-2 (downsample)");
  if (d->upsample == 1) DK_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0, "upsampled conv needs even output size");
  return dk_launch_gemm(p, stream);
}
extern "C" int dk_conv3x3_bf16(const dk_conv_desc* d, void* stream) { return conv3x3_launch(d, nullptr, S_(stream)); }

// Workspace of the attention launches of this host thread.  attention5.hip splits the query blocks of a launch's last, partial round of the
// CUs along the keys (FLUX, one image: 408 blocks on 256 CUs -- 152 blocks in three key ranges each fill the second round to two thirds
// of a block's time); the partial results (bf16 O / l, offset, l per row) go through this buffer.  Without one (or with one too small for a
// launch) the blocks are not split: same results up to the rounding of the partials, a longer last round.
extern "C" size_t dk_attention_workspace_bytes(void) { return (size_t)1020 * (65536 + 2048); }  // <= 255 blocks x 4 key ranges
extern "C" int dk_attention_set_workspace(void* workspace, size_t bytes) {
  DK_REQUIRE(workspace == nullptr || ((uintptr_t)workspace & 255) == 0, "attention workspace: 256-byte aligned (or NULL)");
  dk_set_attention_workspace(workspace, bytes);
  return 0;
}

extern "C" int dk_attention_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t S,
                                 int32_t D, int32_t ld, int32_t ldo, float scale, void* stream) {
  AttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)out;
  p.B = B; p.H = H; p.S = S; p.D = D; p.ld = ld; p.ldo = ldo; p.scale = scale;
  return dk_launch_attention(p, S_(stream));
}

extern "C" int dk_attention_bias_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t S, int32_t D,
                                      int32_t ld, int32_t ldo, float scale, const void* bias, int64_t bias_head_stride, int32_t ldb,
                                      void* stream) {
  DK_REQUIRE(bias != nullptr, "bias missing (use dk_attention_bf16 without one)");
  AttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)out;
  p.B = B; p.H = H; p.S = S; p.D = D; p.ld = ld; p.ldo = ldo; p.scale = scale;
  p.bias = (const bf16_t*)bias; p.bias_head_stride = (long)bias_head_stride; p.ldb = ldb;
  return dk_launch_attention(p, S_(stream));
}
extern "C" int32_t dk_attention_d512_tp(int32_t T) { return (int32_t)align_up((size_t)(T > 0 ? T : 0), 64); }
// transpose of every image's V into [512, Tp] rows (zero-padded), then the flash kernel
static int attention_d512(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int B, int T, int ld, int ldo, float scale,
                          bf16_t* vt, hipStream_t st) {
  DK_REQUIRE(ld == 512, "attention_d512: q / k / v rows of exactly 512 columns (the transpose reads dense [T, 512] matrices)");
  const int Tp = dk_attention_d512_tp(T);
  for (int b = 0; b < B; ++b) {
    const int rc = dk_launch_transpose(v + (size_t)b * T * ld, vt + (size_t)b * 512 * Tp, T, 512, st, Tp);
    if (rc) return rc;
  }
  Attn512Params a;
  a.Q = q; a.K = k; a.Vt = vt; a.O = out; a.T = T; a.Tp = Tp; a.B = B; a.ld = ld; a.ldo = ldo; a.scale = scale;
  return dk_launch_attention512(a, st);
}
extern "C" int dk_attention_d512_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t T, int32_t ld, int32_t ldo,
                                      float scale, void* vt_scratch, void* stream) {
  DK_REQUIRE(q && k && v && out && vt_scratch && B > 0 && T > 0, "null / empty argument");
  return attention_d512((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, B, T, ld, ldo, scale, (bf16_t*)vt_scratch,
                        S_(stream));
}
extern "C" int dk_embedding_bf16(const void* table, const int32_t* ids, const void* pos, int32_t pos_rows, void* out_bf16, float* out_f32,
                                 int32_t n, int32_t dim, int32_t vocab, void* stream) {
  DK_REQUIRE(table && ids && (out_bf16 || out_f32), "null argument");
  return dk_launch_embedding((const bf16_t*)table, ids, (const bf16_t*)pos, pos_rows, (bf16_t*)out_bf16, out_f32, n, dim, vocab, S_(stream));
}
extern "C" int dk_layernorm_bf16(const void* x, void* out, int32_t M, int32_t h, const void* weight, const void* bias, float eps,
                                 void* stream) {
  DK_REQUIRE(x && out && weight && M > 0 && h > 0, "bad argument");
  return dk_launch_layernorm((const bf16_t*)x, (bf16_t*)out, M, h, (const bf16_t*)weight, (const bf16_t*)bias, eps, S_(stream));
}
extern "C" int dk_t5_rmsnorm_bf16(const float* x, void* out, int32_t M, int32_t h, const void* weight, float eps, void* stream) {
  DK_REQUIRE(x && out && weight && M > 0 && h > 0, "bad argument");
  return dk_launch_t5_rmsnorm(x, (bf16_t*)out, M, h, (const bf16_t*)weight, eps, S_(stream));
}
extern "C" int dk_text_elementwise(const void* a, const void* b, void* y, float* r, int64_t n, int32_t op, void* stream) {
  DK_REQUIRE(a && n > 0 && op >= 0 && op <= 2 && (op == 2 ? r != nullptr : y != nullptr) && (op != 1 || b != nullptr), "bad argument");
  return dk_launch_text_elementwise((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, r, (long)n, op, S_(stream));
}
extern "C" int dk_t5_bias_bf16(const void* emb, const int32_t* rel_bucket, int32_t H, int32_t S, int32_t ld, void* out, void* stream) {
  DK_REQUIRE(emb && rel_bucket && out && H > 0 && S > 0 && ld >= S && ld % 64 == 0, "bad argument");
  return dk_launch_t5_bias((const bf16_t*)emb, rel_bucket, H, S, ld, (bf16_t*)out, S_(stream));
}

extern "C" int dk_ln_modulate_bf16(const void* x, int32_t ldx, void* out, int32_t ldo, int32_t M, int32_t h, const void* shift,
                                   const void* scale, int32_t mod_stride, int32_t mod_seg_len, int32_t x_seg_len,
                                   int32_t x_seg_stride, float eps, void* stream) {
  return dk_launch_ln_modulate((const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, h, (const bf16_t*)shift, (const bf16_t*)scale,
                               mod_stride, mod_seg_len > 0 ? mod_seg_len : M, x_seg_len > 0 ? x_seg_len : M, x_seg_stride,
                               eps, S_(stream));
}

extern "C" int32_t dk_weight_pitch_fp8(int32_t k) { return k >= g_dk_pitch_min_k ? k + 128 : k; }
static int mx_nblk(long rows) { return (int)((rows + 127) / 128 + 1); }
extern "C" size_t dk_mx_scale_bytes(int64_t rows, int32_t k) { return (size_t)((k + 127) / 128) * (size_t)mx_nblk((long)rows) * 512; }

extern "C" int dk_gemm_fp8(const dk_gemm_fp8_desc* d, void* stream) {
  DK_REQUIRE(d != nullptr, "null descriptor");
  GemmF8Params p;
  memset(&p, 0, sizeof(p));
  p.A = (const unsigned char*)d->A; p.SA = (const unsigned char*)d->A_scales; p.W = (const unsigned char*)d->W; p.wscale = d->w_scale;
  p.C = d->C; p.bias = (const bf16_t*)d->bias; p.gate = (const bf16_t*)d->gate; p.res = (const bf16_t*)d->res;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
  p.a_seg_len = d->a_seg_len > 0 ? d->a_seg_len : (d->M + 127) / 128 * 128; p.a_seg_stride = d->a_seg_stride; p.a_row0 = d->a_row0;
  p.sa_nblk = mx_nblk(d->a_rows);
  p.c_seg_len = d->c_seg_len > 0 ? d->c_seg_len : d->M; p.c_seg_stride = d->c_seg_stride;
  p.r_seg_len = d->r_seg_len > 0 ? d->r_seg_len : d->M; p.r_seg_stride = d->r_seg_stride;
  p.gate_seg_len = d->gate_seg_len > 0 ? d->gate_seg_len : d->M; p.gate_stride = d->gate_stride;
  p.epi = d->epilogue; p.c_mx8 = d->c_mx8;
  if (d->c_mx8) {
    DK_REQUIRE(d->M % 256 == 0 && d->c_col0 % 32 == 0 && d->C_scales != nullptr, "MX-fp8 output: M a multiple of 256, column offset a multiple of 32");
    p.SC = (unsigned char*)d->C_scales; p.sc_nblk = mx_nblk(d->c_rows); p.c_row0 = d->c_row0; p.sc_kb0 = d->c_col0 / 32;
  }
  p.workspace = d->workspace; p.workspace_bytes = d->workspace_bytes;
  return dk_launch_gemm256f8(p, nullptr, S_(stream));
}
static Mx8Out mx8_out(void* out, void* scales, int ldo, long rows, int row0, int seg_len, int seg_stride, int col0) {
  Mx8Out o;
  o.out = (unsigned char*)out; o.scales = (unsigned char*)scales; o.ldo = ldo; o.n_blk128 = mx_nblk(rows); o.row0 = row0;
  o.seg_len = seg_len; o.seg_stride = seg_stride; o.col0 = col0;
  return o;
}
extern "C" int dk_quantize_mx8(const void* x, int32_t ldx, int32_t M, int32_t h, void* out, int32_t ldo, void* out_scales, int64_t out_rows,
                               int32_t out_row0, int32_t out_col0, void* stream) {
  DK_REQUIRE(x && out && out_scales && M > 0, "bad argument");
  DK_REQUIRE(out_row0 >= 0 && (int64_t)out_row0 + M <= out_rows, "rows [out_row0, out_row0 + M) must lie inside the [out_rows, ldo] output");
  DK_REQUIRE(out_col0 >= 0 && ldo >= out_col0 + h, "columns [out_col0, out_col0 + h) must lie inside a row of ldo bytes");
  return dk_launch_quantize_mx8((const bf16_t*)x, ldx, M, 0, M, h, mx8_out(out, out_scales, ldo, (long)out_rows, out_row0, M, 0, out_col0), S_(stream));
}
extern "C" int dk_ln_modulate_mx8(const void* x, int32_t ldx, int32_t M, int32_t h, const void* shift, const void* scale, int32_t mod_stride,
                                  int32_t mod_seg_len, float eps, void* out, int32_t ldo, void* out_scales, int64_t out_rows,
                                  int32_t out_row0, void* stream) {
  DK_REQUIRE(x && out && out_scales && M > 0, "bad argument");
  DK_REQUIRE(out_row0 >= 0 && (int64_t)out_row0 + M <= out_rows, "rows [out_row0, out_row0 + M) must lie inside the [out_rows, ldo] output");
  DK_REQUIRE(ldo >= h, "a row of the output holds h bytes");
  return dk_launch_ln_modulate_mx8((const bf16_t*)x, ldx, M, h, (const bf16_t*)shift, (const bf16_t*)scale, mod_stride,
                                   mod_seg_len > 0 ? mod_seg_len : M, M, 0, eps, mx8_out(out, out_scales, ldo, (long)out_rows, out_row0, M, 0, 0),
                                   S_(stream));
}

extern "C" int dk_qk_norm_rope_bf16(void* qkv, int32_t ld, int32_t q_off, int32_t k_off, int32_t rows, int32_t H, int32_t D,
                                    const void* q_weight, const void* k_weight, float eps, const float* rope_table,
                                    int32_t row_seg_len, int32_t row_seg_stride, int32_t pos_off, void* stream) {
  return dk_launch_qk_norm_rope((bf16_t*)qkv, ld, q_off, k_off, rows, H, D, (const bf16_t*)q_weight, (const bf16_t*)k_weight,
                                eps, rope_table, row_seg_len > 0 ? row_seg_len : rows, row_seg_stride, pos_off, 0, S_(stream));
}

extern "C" int dk_rope_table_f32(float* table, int32_t S_txt, int32_t gh, int32_t gw, const int32_t* axes_dim, int32_t n_axes,
                                 float theta, void* stream) {
  return dk_launch_rope_table(table, S_txt, gh, gw, axes_dim, n_axes, theta, S_(stream));
}

extern "C" int dk_timestep_embedding_bf16(const float* t_dev, int32_t n, int32_t dim, float max_period, int32_t embed_dtype,
                                          void* out, void* stream) {
  return dk_launch_timestep_embedding(t_dev, n, 1, dim, max_period, embed_dtype, (bf16_t*)out, S_(stream));
}

extern "C" int dk_latent_to_tokens(const float* x, void* tokens, int32_t n_img, int32_t dup, int32_t Hl, int32_t Wl, int32_t C,
                                   int32_t p, int32_t reshape_order, void* stream) {
  DK_REQUIRE(Hl % p == 0 && Wl % p == 0, "latent size must be divisible by the patch size");
  return dk_launch_latent_to_tokens(x, (bf16_t*)tokens, n_img, dup, Hl, Wl, C, p, reshape_order, S_(stream));
}

extern "C" int dk_euler_cfg_step(float* x, const void* model_out, int32_t ld_out, void* tokens, int32_t n_img, int32_t cfg_on,
                                 int32_t Hl, int32_t Wl, int32_t C, int32_t p, int32_t reshape_order, float sigma,
                                 float sigma_next, float cfg_weight, void* stream) {
  DK_REQUIRE(sigma != 0.0f, "sigma must be non-zero");
  return dk_launch_euler_step(x, (const bf16_t*)model_out, ld_out, (bf16_t*)tokens, n_img, cfg_on, Hl, Wl, C, p, reshape_order,
                              sigma, sigma_next, cfg_weight, S_(stream));
}

extern "C" int dk_affine_f32(const float* x, float* y, int64_t n, float a, float b, void* stream) {
  return dk_launch_affine_f32(x, y, (long)n, a, b, S_(stream));
}

static int gn_nchunk(long HW, int C) {
  const long ppi = 256 / (C / 8);
  long n = HW / (ppi * 8);
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}
extern "C" size_t dk_groupnorm_scratch_floats(int32_t B, int32_t G) { return (size_t)B * 1024 * 2 * G + (size_t)B * G * 2; }
extern "C" int dk_groupnorm_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* gamma,
                                 const void* beta, float eps, int32_t fuse_silu, float* scratch, void* stream) {
  const int nchunk = gn_nchunk((long)HW, C);
  float* mean_rstd = scratch + (size_t)B * 1024 * 2 * G;
  int rc = dk_launch_groupnorm_stats((const bf16_t*)x, B, (long)HW, C, G, scratch, nchunk, mean_rstd, eps, S_(stream));
  if (rc) return rc;
  return dk_launch_groupnorm_apply((const bf16_t*)x, (bf16_t*)y, B, (long)HW, C, G, mean_rstd, (const bf16_t*)gamma,
                                   (const bf16_t*)beta, fuse_silu, S_(stream));
}

// scratch layout shared by dk_groupnorm_bf16 / dk_groupnorm_table_bf16: [partials: B * n * 2G][mean_rstd: B * G * 2], n = the
// larger of 1024 and the caller's n_partial
extern "C" int dk_groupnorm_table_bf16(const void* x, int32_t B, int64_t HW, int32_t C, int32_t G, const void* gamma, const void* beta,
                                       float eps, float* scratch, int32_t n_partial, float* scale_shift, void* stream) {
  DK_REQUIRE(gamma && beta && scratch && scale_shift && B > 0 && G > 0 && C % G == 0, "groupnorm table arguments");
  DK_REQUIRE(x != nullptr || n_partial > 0, "either x or the number of partials a conv launch left in scratch");
  const int nchunk = x ? gn_nchunk((long)HW, C) : n_partial;
  float* mean_rstd = scratch + (size_t)B * (nchunk > 1024 ? nchunk : 1024) * 2 * G;
  if (x) {  // the partial sums only: the one finalisation below builds mean / rstd AND the table
    const int rc = dk_launch_groupnorm_partials((const bf16_t*)x, B, (long)HW, C, G, scratch, nchunk, S_(stream));
    if (rc) return rc;
  }
  return dk_launch_groupnorm_finalize(scratch, nchunk, B, G, (double)HW * (double)(C / G), eps, mean_rstd, (const bf16_t*)gamma,
                                      (const bf16_t*)beta, C, scale_shift, S_(stream));
}

static ConvHaloParams conv_halo_params(const dk_conv_gn_desc* d) {
  ConvHaloParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w; p.bias = (const bf16_t*)d->bias; p.bias2 = (const bf16_t*)d->bias2;
  p.res = (const bf16_t*)d->res; p.y = (bf16_t*)d->y; p.gn_ss = d->gn_scale_shift; p.gn_silu = d->gn_silu;
  p.x2 = (const bf16_t*)d->x2; p.C2 = d->C2; p.stats_out = d->stats_partial; p.G_out = d->stats_groups;
  p.img = d->image_f32; p.u8 = d->image_u8; p.raw = (bf16_t*)d->raw_bf16; p.out_channels = d->O <= 4 ? d->O : 0;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C = d->C; p.O = d->O; p.ups = d->upsample; p.ldw = d->ldw; p.ldy = d->ldy; p.ldr = d->ldr;
  return p;
}
extern "C" int dk_conv3x3_gn_bf16(const dk_conv_gn_desc* d, void* stream) {
  DK_REQUIRE(d && d->x && d->w && d->bias, "null argument");
  const bool img = d->image_f32 || d->image_u8 || d->raw_bf16;
  DK_REQUIRE(img || d->y, "no output");
  return dk_launch_conv_halo(conv_halo_params(d), S_(stream));
}

extern "C" int dk_softmax_rows_bf16(void* x, int32_t rows, int32_t cols, int32_t ld, void* stream) {
  return dk_launch_softmax_rows((bf16_t*)x, rows, cols, ld, S_(stream));
}
extern "C" int dk_transpose_bf16(const void* x, void* y, int32_t R, int32_t C, void* stream) {
  return dk_launch_transpose((const bf16_t*)x, (bf16_t*)y, R, C, S_(stream));
}

// ---------------------------------------------------------------------------------------------
// helpers shared by the engines
// ---------------------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off, cap;
  bool dry;
  Carver(void* b, size_t c) : base((char*)b), off(0), cap(c), dry(b == nullptr) {}
  void* take(size_t bytes) {
    off = align_up(off, 256);
    void* p = dry ? nullptr : (void*)(base + off);
    off += bytes;
    return p;
  }
};

// every attention kernel can normalise / rotate the queries in its Q load
static int fuse_q() { return g_dk_fuse_q ? 1 : 0; }
// the keys' QKNorm + RoPE rides in the q / k / v projection's tail (only together with the fused query side: the stand-alone pass
// then has nothing left to do)
static bool fuse_k(const bf16_t* kn) { return g_dk_fuse_k != 0 && fuse_q() && kn != nullptr; }
// ... and the queries' with it (round 4): the attention kernel then loads finished queries
static bool fuse_qg(const bf16_t* qn, bool f8) { return (g_dk_fuse_qg < 0 ? f8 : g_dk_fuse_qg != 0) && qn != nullptr; }
template <typename P>
static void set_key_norm(P& p, const bf16_t* kn, const bf16_t* qn, int h, int D, const float* rope, int pos_off, int seg_len) {
  p.kn_w = kn; p.kn_rope = rope; p.kn_col0 = h; p.kn_col1 = 2 * h; p.kn_D = D; p.kn_pos_off = pos_off; p.kn_seg_len = seg_len; p.kn_eps = 1e-6f;
  if (fuse_qg(qn, std::is_same<P, GemmF8Params>::value)) { p.qn_w = qn; p.qn_col0 = 0; p.qn_col1 = h; }
}

// Split workspace (fp32 slabs + flags) handed to the GEMMs an engine call builds: every dk_mmdit_* entry point sets it to ITS
// engine's region (carved from that engine's workspace) before it enqueues anything and all launches of the call are
// enqueued before it returns, so two engines -- on one host thread in turn, or on two threads / streams at once
// (thread_local) -- never share a flag region.  A single engine must not be driven from two streams concurrently (its
// activations live in one workspace anyway).
static thread_local void* g_linear_ws = nullptr;
struct LinearWsScope {  // an engine call's GEMMs split through that engine's region; the previous setting comes back afterwards
  void* prev;
  explicit LinearWsScope(void* ws) : prev(g_linear_ws) { g_linear_ws = ws; }
  ~LinearWsScope() { g_linear_ws = prev; }
};

static GemmParams linear_params(const bf16_t* A, int lda, int a_seg_len, int a_seg_stride, const bf16_t* W, const bf16_t* bias,
                                bf16_t* C, int ldc, int c_seg_len, int c_seg_stride, int M, int N, int K, int epi,
                                const bf16_t* gate, int gate_seg_len, int gate_stride, const bf16_t* res, int ldr, int r_seg_len,
                                int r_seg_stride, int ldw = 0) {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.gate = gate; p.res = res;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldw = ldw;
  p.a_seg_len = a_seg_len; p.a_seg_stride = a_seg_stride;
  p.c_seg_len = c_seg_len; p.c_seg_stride = c_seg_stride;
  p.r_seg_len = r_seg_len > 0 ? r_seg_len : M; p.r_seg_stride = r_seg_stride;
  p.gate_seg_len = gate_seg_len > 0 ? gate_seg_len : M; p.gate_stride = gate_stride;
  p.alpha = 1.0f; p.epi = epi;
  if (g_linear_ws) { p.workspace = g_linear_ws; p.workspace_bytes = dk_gemm_split_workspace_bytes(); }
  return p;
}

static int linear_call(const bf16_t* A, int lda, int a_seg_len, int a_seg_stride, const bf16_t* W, const bf16_t* bias,
                       bf16_t* C, int ldc, int c_seg_len, int c_seg_stride, int M, int N, int K, int epi,
                       const bf16_t* gate, int gate_seg_len, int gate_stride, const bf16_t* res, int ldr, int r_seg_len,
                       int r_seg_stride, hipStream_t st, int ldw = 0) {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.gate = gate; p.res = res;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldw = ldw;
  p.a_seg_len = a_seg_len; p.a_seg_stride = a_seg_stride;
  p.c_seg_len = c_seg_len; p.c_seg_stride = c_seg_stride;
  p.r_seg_len = r_seg_len > 0 ? r_seg_len : M; p.r_seg_stride = r_seg_stride;
  p.gate_seg_len = gate_seg_len > 0 ? gate_seg_len : M; p.gate_stride = gate_stride;
  p.alpha = 1.0f; p.epi = epi;
  return dk_launch_gemm(p, st);
}
// plain [M,K] x [N,K]^T -> [M,N]
static int linear_plain(const bf16_t* A, const bf16_t* W, const bf16_t* bias, bf16_t* C, int M, int N, int K, int epi,
                        hipStream_t st) {
  return linear_call(A, K, M, 0, W, bias, C, N, M, 0, M, N, K, epi, nullptr, 0, 0, nullptr, 0, 0, 0, st);
}

#define DK_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

// ---------------------------------------------------------------------------------------------
// MMDiT engine
// ---------------------------------------------------------------------------------------------
struct StreamW {  // one TransformerBlock's weights (mmdit.py:395-438)
  const bf16_t *qkv_w = nullptr, *qkv_b = nullptr, *qn = nullptr, *kn = nullptr;
  const bf16_t *o_w = nullptr, *o_b = nullptr, *fc1_w = nullptr, *fc1_b = nullptr, *fc2_w = nullptr, *fc2_b = nullptr;
  const bf16_t *l2_w = nullptr, *l2_b = nullptr;  // single blocks: [h, 5h] = [o_proj | fc2]
  // fp8_linears: e4m3 weights + per-output-channel scales instead of the bf16 matrices above (biases stay bf16)
  const unsigned char *qkv_w8 = nullptr, *o_w8 = nullptr, *fc1_w8 = nullptr, *fc2_w8 = nullptr, *l2_w8 = nullptr;
  const float *qkv_ws = nullptr, *o_ws = nullptr, *fc1_ws = nullptr, *fc2_ws = nullptr, *l2_ws = nullptr;
};

struct dk_mmdit {
  dk_mmdit_config cfg;
  std::unordered_map<std::string, const void*> named;
  bool resolved = false;
  // resolved weights
  const bf16_t *xemb_w, *xemb_b, *pos_w, *ctx_w, *ctx_b, *y0_w, *y0_b, *y2_w, *y2_b, *t0_w, *t0_b, *t2_w, *t2_b;
  const bf16_t *adaln_w, *adaln_b, *final_w, *final_b;
  std::vector<StreamW> dimg, dtxt, single;
  // shape
  int B = 0, Hl = 0, Wl = 0, S_t = 0, S_i = 0, S = 0, n_t = 0;
  bool prepared = false, mod_ready = false;
  // workspace views
  bf16_t *X, *XN, *QKV, *ATT, *CAT, *HID, *MOD, *POS;
  bf16_t* CTXE = nullptr;  // context_embedder(text) [B, S_t, h], step-invariant (dk_mmdit_cache_context)
  bool ctx_ready = false;
  int ldh = 0, ldcat = 0;  // row pitch of HID / CAT (dk_weight_pitch of r*h / (1+r)*h at carve time; fc2 / linear2 weights use the same)
  void* GWS = nullptr;  // GEMM split workspace (fp32 slabs + flags), dk_gemm_split_workspace_bytes()
  void* AWS = nullptr;  // attention5.hip's key-split partial results of THIS engine's launches (dk_attention_workspace_bytes(); D = 128 only)
  size_t AWS_bytes = 0;
  bf16_t *temb, *t1, *tvec, *y1, *yvec, *vec;
  float *rope, *tdev;
  // guidance embedding (cfg.guidance_embed): MLPEmbedder weights, the value set by dk_mmdit_set_guidance, scratch rows
  const bf16_t *g0_w = nullptr, *g0_b = nullptr, *g2_w = nullptr, *g2_b = nullptr;
  float guidance = 3.5f;
  bf16_t *gemb = nullptr, *g1 = nullptr, *gvec = nullptr;
  // fp8_linears: MX-fp8 activation buffers + their scale side arrays (layout dk_mx_scale_index)
  unsigned char *XN8 = nullptr, *ATT8 = nullptr, *HC8 = nullptr;    // [BS, h], [BS, h], max([BS, ldh8], [BS, ldcat8])
  unsigned char *SXN = nullptr, *SATT = nullptr, *SHID = nullptr, *SCAT = nullptr;
  int ldh8 = 0, ldcat8 = 0, nblk = 0;
  bool fp8() const { return cfg.fp8_linears != 0; }
  // precision policy: the first n_bf16() double-stream blocks keep bf16 Linears under fp8_linears (global block index = double-block index)
  int n_bf16() const { return fp8() ? (cfg.fp8_bf16_double_blocks < cfg.depth_multimodal ? cfg.fp8_bf16_double_blocks : cfg.depth_multimodal) : 0; }

  int h() const { return cfg.hidden_size; }
  int D() const { return cfg.hidden_size / cfg.num_heads; }
  int F() const { return cfg.patch_size * cfg.patch_size * cfg.vae_latent_dim; }
  bool txt_skipped(int i) const { return i == cfg.depth_multimodal - 1 && cfg.depth_unified < 1; }
  int mod_rows() const {
    int n = 0;
    for (int i = 0; i < cfg.depth_multimodal; ++i) n += 6 + (txt_skipped(i) ? 2 : 6);
    return n + 3 * cfg.depth_unified + 2;
  }
  int mod_offset(int kind, int index) const {
    int n = 0;
    for (int i = 0; i < cfg.depth_multimodal; ++i) {
      if (kind == 0 && i == index) return n;
      n += 6;
      if (kind == 1 && i == index) return n;
      n += txt_skipped(i) ? 2 : 6;
    }
    for (int i = 0; i < cfg.depth_unified; ++i) {
      if (kind == 2 && i == index) return n;
      n += 3;
    }
    if (kind == 3) return n;
    return -1;
  }
};

extern "C" int dk_mmdit_create(const dk_mmdit_config* cfg, dk_mmdit** out) {
  DK_REQUIRE(cfg && out, "null argument");
  DK_REQUIRE(cfg->hidden_size % cfg->num_heads == 0, "hidden_size must be divisible by num_heads");
  const int D = cfg->hidden_size / cfg->num_heads;
  DK_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
  DK_REQUIRE(cfg->hidden_size % 64 == 0, "hidden_size must be a multiple of 64");
  DK_REQUIRE((cfg->patch_size * cfg->patch_size * cfg->vae_latent_dim) % 64 == 0, "patch feature size must be a multiple of 64");
  DK_REQUIRE(cfg->token_level_text_embed_dim % 64 == 0 && cfg->pooled_text_embed_dim % 64 == 0 &&
                 cfg->frequency_embed_dim % 64 == 0,
             "embedding dims must be multiples of 64");
  if (cfg->use_rope) {
    int half = 0;
    for (int i = 0; i < cfg->n_rope_axes; ++i) half += cfg->rope_axes_dim[i] / 2;
    DK_REQUIRE(half * 2 == D, "rope axes must sum to head_dim");
  }
  if (cfg->fp8_linears) DK_REQUIRE(D == 128 && cfg->hidden_size % 256 == 0, "fp8_linears: head_dim 128 and hidden_size a multiple of 256");
  dk_mmdit* m = new dk_mmdit();
  m->cfg = *cfg;
  *out = m;
  return 0;
}
extern "C" int dk_mmdit_set_guidance(dk_mmdit* m, float guidance) {
  DK_REQUIRE(m != nullptr, "null handle");
  DK_REQUIRE(m->cfg.guidance_embed, "this configuration has no guidance embedding (guidance_embed = 0)");
  m->guidance = guidance;
  m->mod_ready = false;
  return 0;
}
extern "C" void dk_mmdit_destroy(dk_mmdit* m) { delete m; }

extern "C" int dk_mmdit_bind(dk_mmdit* m, const char* name, const void* dev_ptr) {
  DK_REQUIRE(m && name && dev_ptr, "null argument");
  m->named[name] = dev_ptr;
  m->resolved = false;
  return 0;
}
extern "C" int dk_mmdit_mod_rows(const dk_mmdit* m) { return m->mod_rows(); }
extern "C" int dk_mmdit_mod_offset(const dk_mmdit* m, int32_t kind, int32_t index) { return m->mod_offset(kind, index); }

static int need(const std::unordered_map<std::string, const void*>& named, const std::string& name, const bf16_t** out,
                bool optional = false) {
  auto it = named.find(name);
  if (it == named.end()) {
    *out = nullptr;
    if (optional) return 0;
    dk_set_error("weight not bound: " + name);
    return -3;
  }
  *out = (const bf16_t*)it->second;
  return 0;
}

// the matrix of one Linear: bf16 "<name>.weight", or (fp8_linears) e4m3 "<name>.weight_fp8" + f32 "<name>.wscale"
static int need_matrix(dk_mmdit* m, bool f8, const std::string& name, const bf16_t** w, const unsigned char** w8, const float** ws) {
  if (!f8) return need(m->named, name + ".weight", w);
  const bf16_t *a = nullptr, *b = nullptr;
  DK_TRY(need(m->named, name + ".weight_fp8", &a));
  DK_TRY(need(m->named, name + ".wscale", &b));
  *w8 = (const unsigned char*)a;
  *ws = (const float*)b;
  return 0;
}
static int resolve_stream(dk_mmdit* m, const std::string& p, StreamW& w, bool single, bool skip_post, bool f8) {
  const auto& n = m->named;
  if (single) {  // fused [q|k|v|fc1] matrix; fc1 views point into it
    DK_TRY(need_matrix(m, f8, p + ".linear1", &w.qkv_w, &w.qkv_w8, &w.qkv_ws));
    DK_TRY(need(n, p + ".linear1.bias", &w.qkv_b));
    if (!f8) w.fc1_w = w.qkv_w + (size_t)3 * m->h() * m->h();
    w.fc1_b = w.qkv_b + 3 * m->h();
  } else {
    DK_TRY(need_matrix(m, f8, p + ".attn.qkv", &w.qkv_w, &w.qkv_w8, &w.qkv_ws));
    DK_TRY(need(n, p + ".attn.qkv.bias", &w.qkv_b));
  }
  if (m->cfg.use_qk_norm) {
    DK_TRY(need(n, p + ".qk_norm.q_norm.weight", &w.qn));
    DK_TRY(need(n, p + ".qk_norm.k_norm.weight", &w.kn));
  }
  if (skip_post) return 0;
  if (!single) {
    DK_TRY(need_matrix(m, f8, p + ".mlp.fc1", &w.fc1_w, &w.fc1_w8, &w.fc1_ws));
    DK_TRY(need(n, p + ".mlp.fc1.bias", &w.fc1_b));
  }
  if (single) {
    DK_TRY(need_matrix(m, f8, p + ".linear2", &w.l2_w, &w.l2_w8, &w.l2_ws));
    DK_TRY(need(n, p + ".linear2.bias", &w.l2_b));
  } else {
    DK_TRY(need_matrix(m, f8, p + ".attn.o_proj", &w.o_w, &w.o_w8, &w.o_ws));
    DK_TRY(need(n, p + ".attn.o_proj.bias", &w.o_b));
    DK_TRY(need_matrix(m, f8, p + ".mlp.fc2", &w.fc2_w, &w.fc2_w8, &w.fc2_ws));
    DK_TRY(need(n, p + ".mlp.fc2.bias", &w.fc2_b));
  }
  return 0;
}

static int mmdit_resolve(dk_mmdit* m) {
  if (m->resolved) return 0;
  const auto& n = m->named;
  DK_TRY(need(n, "x_embedder.proj.weight", &m->xemb_w));
  DK_TRY(need(n, "x_embedder.proj.bias", &m->xemb_b));
  DK_TRY(need(n, "x_pos_embedder.pos_embed.weight", &m->pos_w, !m->cfg.use_pos_embed));
  DK_TRY(need(n, "context_embedder.weight", &m->ctx_w));
  DK_TRY(need(n, "context_embedder.bias", &m->ctx_b));
  DK_TRY(need(n, "y_embedder.mlp.layers.0.weight", &m->y0_w));
  DK_TRY(need(n, "y_embedder.mlp.layers.0.bias", &m->y0_b));
  DK_TRY(need(n, "y_embedder.mlp.layers.2.weight", &m->y2_w));
  DK_TRY(need(n, "y_embedder.mlp.layers.2.bias", &m->y2_b));
  DK_TRY(need(n, "t_embedder.mlp.layers.0.weight", &m->t0_w));
  DK_TRY(need(n, "t_embedder.mlp.layers.0.bias", &m->t0_b));
  DK_TRY(need(n, "t_embedder.mlp.layers.2.weight", &m->t2_w));
  DK_TRY(need(n, "t_embedder.mlp.layers.2.bias", &m->t2_b));
  DK_TRY(need(n, "adaLN.weight", &m->adaln_w));
  DK_TRY(need(n, "adaLN.bias", &m->adaln_b));
  DK_TRY(need(n, "final_layer.linear.weight", &m->final_w));
  DK_TRY(need(n, "final_layer.linear.bias", &m->final_b));
  if (m->cfg.guidance_embed) {
    DK_TRY(need(n, "guidance_in.mlp.layers.0.weight", &m->g0_w));
    DK_TRY(need(n, "guidance_in.mlp.layers.0.bias", &m->g0_b));
    DK_TRY(need(n, "guidance_in.mlp.layers.2.weight", &m->g2_w));
    DK_TRY(need(n, "guidance_in.mlp.layers.2.bias", &m->g2_b));
  }
  m->dimg.assign(m->cfg.depth_multimodal, StreamW());
  m->dtxt.assign(m->cfg.depth_multimodal, StreamW());
  m->single.assign(m->cfg.depth_unified, StreamW());
  for (int i = 0; i < m->cfg.depth_multimodal; ++i) {
    const std::string b = "multimodal_transformer_blocks." + std::to_string(i);
    const bool f8 = m->fp8() && i >= m->n_bf16();
    DK_TRY(resolve_stream(m, b + ".image_transformer_block", m->dimg[i], false, false, f8));
    DK_TRY(resolve_stream(m, b + ".text_transformer_block", m->dtxt[i], false, m->txt_skipped(i), f8));
    // (the fused QKNorm decisions of a double block are taken per stream from these pointers and must agree: ADVICE r4)
    DK_REQUIRE((m->dimg[i].qn == nullptr) == (m->dtxt[i].qn == nullptr) && (m->dimg[i].kn == nullptr) == (m->dtxt[i].kn == nullptr),
               "QKNorm weights must be bound for both streams of a double block or for neither");
  }
  for (int i = 0; i < m->cfg.depth_unified; ++i)
    DK_TRY(resolve_stream(m, "unified_transformer_blocks." + std::to_string(i) + ".transformer_block", m->single[i], true, false, m->fp8()));
  m->resolved = true;
  return 0;
}

static size_t mmdit_carve(dk_mmdit* m, Carver& c, int B, int Hl, int Wl, int S_t, int n_t) {
  const int h = m->h(), p = m->cfg.patch_size;
  const int S_i = (Hl / p) * (Wl / p), S = S_t + S_i;
  const size_t BS = (size_t)B * S;
  m->X = (bf16_t*)c.take(BS * h * 2);
  m->XN = (bf16_t*)c.take(BS * h * 2);
  m->QKV = (bf16_t*)c.take(BS * 3 * h * 2);
  m->ATT = (bf16_t*)c.take(BS * h * 2);
  m->ldh = dk_weight_pitch(m->cfg.mlp_ratio * h);
  m->ldcat = dk_weight_pitch((1 + m->cfg.mlp_ratio) * h);
  const size_t hid = BS * m->ldh * 2;  // MLP hidden of both streams of a double block (image rows first)
  const size_t cat = m->cfg.depth_unified > 0 ? BS * (size_t)m->ldcat * 2 : 0;
  // single blocks: [attn | gelu(fc1)]; double blocks: MLP hidden (fp8_linears: only the bf16 double blocks of the precision policy need it)
  m->CAT = (bf16_t*)c.take(m->fp8() ? (m->n_bf16() > 0 ? hid : 0) : (cat > hid ? cat : hid));
  m->HID = m->CAT;
  if (m->fp8()) {  // the same buffers in MX-fp8 (the bf16 X, QKV, ATT above stay: residual stream, attention operands / output)
    const int r = m->cfg.mlp_ratio;
    m->ldh8 = dk_weight_pitch_fp8(r * h);
    m->ldcat8 = dk_weight_pitch_fp8((1 + r) * h);
    m->nblk = mx_nblk((long)BS);
    m->XN8 = (unsigned char*)c.take(BS * h);
    m->ATT8 = (unsigned char*)c.take(BS * h);
    const size_t hid8 = BS * (size_t)m->ldh8, cat8 = m->cfg.depth_unified > 0 ? BS * (size_t)m->ldcat8 : 0;
    m->HC8 = (unsigned char*)c.take(cat8 > hid8 ? cat8 : hid8);
    m->SXN = (unsigned char*)c.take(dk_mx_scale_bytes((long)BS, h));
    m->SATT = (unsigned char*)c.take(dk_mx_scale_bytes((long)BS, h));
    m->SHID = (unsigned char*)c.take(dk_mx_scale_bytes((long)BS, r * h));
    m->SCAT = (unsigned char*)c.take(m->cfg.depth_unified > 0 ? dk_mx_scale_bytes((long)BS, (1 + r) * h) : 0);
  }
  m->MOD = (bf16_t*)c.take((size_t)n_t * B * m->mod_rows() * h * 2);
  m->POS = (bf16_t*)c.take(m->cfg.use_pos_embed ? (size_t)S_i * h * 2 : 0);
  m->CTXE = (bf16_t*)c.take((size_t)B * S_t * h * 2);
  m->temb = (bf16_t*)c.take((size_t)n_t * m->cfg.frequency_embed_dim * 2);
  m->t1 = (bf16_t*)c.take((size_t)n_t * h * 2);
  m->tvec = (bf16_t*)c.take((size_t)n_t * h * 2);
  m->y1 = (bf16_t*)c.take((size_t)B * h * 2);
  m->yvec = (bf16_t*)c.take((size_t)B * h * 2);
  m->vec = (bf16_t*)c.take((size_t)n_t * B * h * 2);
  m->gemb = (bf16_t*)c.take(m->cfg.guidance_embed ? (size_t)m->cfg.frequency_embed_dim * 2 : 0);
  m->g1 = (bf16_t*)c.take(m->cfg.guidance_embed ? (size_t)h * 2 : 0);
  m->gvec = (bf16_t*)c.take(m->cfg.guidance_embed ? (size_t)h * 2 : 0);
  m->rope = (float*)c.take(m->cfg.use_rope ? (size_t)S * m->D() * 4 : 0);
  m->tdev = (float*)c.take((size_t)n_t * 4);
  m->GWS = c.take(dk_gemm_split_workspace_bytes());
  // this engine's own region for the key-split jobs of attention5.hip (two engines, or an engine and ops.attention, driven from one
  // host thread on different streams must not share the partial results of their split launches)
  m->AWS_bytes = m->D() == 128 ? dk_attention_workspace_bytes() : 0;
  m->AWS = c.take(m->AWS_bytes);
  return c.off;
}

extern "C" size_t dk_mmdit_workspace_bytes(const dk_mmdit* m, int32_t batch, int32_t latent_h, int32_t latent_w, int32_t text_len,
                                           int32_t n_timesteps) {
  dk_mmdit tmp = *m;
  Carver c(nullptr, 0);
  return mmdit_carve(&tmp, c, batch, latent_h, latent_w, text_len, n_timesteps) + 256;
}

extern "C" int dk_mmdit_prepare(dk_mmdit* m, int32_t batch, int32_t latent_h, int32_t latent_w, int32_t text_len,
                                int32_t n_timesteps, void* workspace, size_t workspace_bytes, void* stream) {
  DK_REQUIRE(m && workspace, "null argument");
  DK_REQUIRE(batch > 0 && text_len > 0 && n_timesteps > 0, "empty problem");
  const int p = m->cfg.patch_size;
  DK_REQUIRE(latent_h % p == 0 && latent_w % p == 0, "latent size must be divisible by the patch size");
  DK_TRY(mmdit_resolve(m));
  if (m->fp8()) {
    const int S_i_ = (latent_h / p) * (latent_w / p);
    DK_REQUIRE(text_len % 128 == 0 && S_i_ % 128 == 0,
               "fp8_linears: text_len and the number of image tokens must be multiples of 128 (MX scale blocks of 128 rows)");
  }
  DK_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  Carver c(workspace, workspace_bytes);
  const size_t need_bytes = mmdit_carve(m, c, batch, latent_h, latent_w, text_len, n_timesteps);
  DK_REQUIRE(need_bytes <= workspace_bytes, "workspace too small");
  m->B = batch; m->Hl = latent_h; m->Wl = latent_w; m->S_t = text_len;
  m->S_i = (latent_h / p) * (latent_w / p); m->S = m->S_t + m->S_i; m->n_t = n_timesteps;
  hipStream_t st = S_(stream);
  if (m->cfg.use_rope)
    DK_TRY(dk_launch_rope_table(m->rope, m->S_t, latent_h / p, latent_w / p, m->cfg.rope_axes_dim, m->cfg.n_rope_axes,
                                (float)m->cfg.rope_theta, st));
  if (m->cfg.use_pos_embed) {
    // centre crop of the learned table (mmdit.py:334-349): rows y0..y0+gh, cols x0..x0+gw
    const int mh = m->cfg.max_latent_resolution, gh = latent_h / p, gw = latent_w / p;
    DK_REQUIRE(gh <= mh && gw <= mh, "latent larger than the positional table");
    const int y0 = (mh - gh) / 2, x0 = (mh - gw) / 2;
    DK_CHECK_HIP(hipMemcpy2DAsync(m->POS, (size_t)gw * m->h() * 2, m->pos_w + ((size_t)y0 * mh + x0) * m->h(),
                                  (size_t)mh * m->h() * 2, (size_t)gw * m->h() * 2, gh, hipMemcpyDeviceToDevice, st));
  }
  // the flag region of the GEMM split workspace must be zero before the first launch (the kernels leave it zero)
  DK_CHECK_HIP(hipMemsetAsync((char*)m->GWS + dk_gemm_split_workspace_bytes() - 4096, 0, 4096, st));
  m->prepared = true;
  m->mod_ready = false;
  m->ctx_ready = false;
  return 0;
}

static bool c_guidance(const dk_mmdit* m) { return m->cfg.guidance_embed != 0; }
extern "C" int dk_mmdit_cache_modulation_params(dk_mmdit* m, const void* pooled, const float* timesteps_host, int32_t n,
                                                void* stream) {
  DK_REQUIRE(m && m->prepared, "dk_mmdit_prepare must be called first");
  DK_REQUIRE(n > 0 && n <= m->n_t, "more timesteps than the workspace was prepared for");
  hipStream_t st = S_(stream);
  LinearWsScope ws_scope(m->GWS);
  const int h = m->h(), B = m->B, P = m->cfg.pooled_text_embed_dim, Fq = m->cfg.frequency_embed_dim;
  DK_CHECK_HIP(hipMemcpyAsync(m->tdev, timesteps_host, (size_t)n * 4, hipMemcpyHostToDevice, st));
  DK_TRY(dk_launch_timestep_embedding(m->tdev, n, 1, Fq, (float)m->cfg.max_period, m->cfg.embed_dtype, m->temb, st));
  // t_embedder / y_embedder: Linear -> SiLU -> Linear (mmdit.py:352-392)
  DK_TRY(linear_plain(m->temb, m->t0_w, m->t0_b, m->t1, n, h, Fq, DK_EPI_BIAS_SILU, st));
  DK_TRY(linear_plain(m->t1, m->t2_w, m->t2_b, m->tvec, n, h, h, DK_EPI_BIAS, st));
  DK_TRY(linear_plain((const bf16_t*)pooled, m->y0_w, m->y0_b, m->y1, B, h, P, DK_EPI_BIAS_SILU, st));
  DK_TRY(linear_plain(m->y1, m->y2_w, m->y2_b, m->yvec, B, h, h, DK_EPI_BIAS, st));
  if (c_guidance(m)) {
    // FLUX.1-dev guidance embedding (MLPEmbedder, mmdit.py:31-36,945-955): g = guidance_in(timestep_embedding(1000 * guidance)),
    // added to the pooled-text embedding of every batch row, hence to every modulation vector
    const float gt = 1000.0f * m->guidance;
    DK_CHECK_HIP(hipMemcpyAsync(m->tdev, &gt, 4, hipMemcpyHostToDevice, st));  // (tdev[0] was consumed by the launch above, same stream)
    DK_TRY(dk_launch_timestep_embedding(m->tdev, 1, 1, Fq, (float)m->cfg.max_period, m->cfg.embed_dtype, m->gemb, st));
    DK_TRY(linear_plain(m->gemb, m->g0_w, m->g0_b, m->g1, 1, h, Fq, DK_EPI_BIAS_SILU, st));
    DK_TRY(linear_plain(m->g1, m->g2_w, m->g2_b, m->gvec, 1, h, h, DK_EPI_BIAS, st));
    DK_TRY(dk_launch_add(m->yvec, m->gvec, 1, m->yvec, B, h, st));  // y[b] += g
  }
  // vec[step*B + b] = silu(y[b] + t[step]); adaLN_modulation = SiLU -> Linear (mmdit.py:94-96,430-435)
  DK_TRY(dk_launch_add(m->yvec, m->tvec, n, m->vec, n * B, h, st));
  DK_TRY(dk_launch_silu(m->vec, m->vec, (long)n * B * h, st));
  DK_TRY(linear_plain(m->vec, m->adaln_w, m->adaln_b, m->MOD, n * B, m->mod_rows() * h, h, DK_EPI_BIAS, st));
  m->mod_ready = true;
  return 0;
}

// one TransformerBlock.pre_sdpa (mmdit.py:440-519) on a row range of the joint stream
static int pre_sdpa(dk_mmdit* m, const StreamW& w, int row_off, int S_s, const bf16_t* mod, int mod_stride, hipStream_t st) {
  const int h = m->h(), B = m->B, S = m->S, M = B * S_s;
  DK_TRY(dk_launch_ln_modulate(m->X + (size_t)row_off * h, h, m->XN, h, M, h, mod, mod + h, mod_stride, S_s, S_s, S,
                               m->cfg.layer_norm_eps, st));
  DK_TRY(linear_call(m->XN, h, M, 0, w.qkv_w, w.qkv_b, m->QKV + (size_t)row_off * 3 * h, 3 * h, S_s, S, M, 3 * h, h,
                     DK_EPI_BIAS, nullptr, 0, 0, nullptr, 0, 0, 0, st));
  DK_TRY(dk_launch_qk_norm_rope(m->QKV + (size_t)row_off * 3 * h, 3 * h, 0, h, M, m->cfg.num_heads, m->D(), w.qn, w.kn, 1e-6f,
                                m->cfg.use_rope ? m->rope : nullptr, S_s, S, row_off, m->S, st));
  return 0;
}

// sequential TransformerBlock.post_sdpa (mmdit.py:537-548) on a row range
static int post_sdpa_seq(dk_mmdit* m, const StreamW& w, int row_off, int S_s, const bf16_t* mod, int mod_stride, hipStream_t st) {
  const int h = m->h(), B = m->B, S = m->S, M = B * S_s, r = m->cfg.mlp_ratio;
  bf16_t* Xs = m->X + (size_t)row_off * h;
  // residual += gate_attn * o_proj(attn)
  DK_TRY(linear_call(m->ATT + (size_t)row_off * h, h, S_s, S, w.o_w, w.o_b, Xs, h, S_s, S, M, h, h, DK_EPI_GATE_RES, mod + 2 * h,
                     S_s, mod_stride, Xs, h, S_s, S, st));
  // residual += gate_mlp * fc2(gelu(fc1(LN-mod(residual))))
  DK_TRY(dk_launch_ln_modulate(Xs, h, m->XN, h, M, h, mod + 3 * h, mod + 4 * h, mod_stride, S_s, S_s, S, m->cfg.layer_norm_eps, st));
  DK_TRY(linear_call(m->XN, h, M, 0, w.fc1_w, w.fc1_b, m->HID, m->ldh, M, 0, M, r * h, h, DK_EPI_BIAS_GELU, nullptr, 0, 0, nullptr,
                     0, 0, 0, st));
  DK_TRY(linear_call(m->HID, m->ldh, M, 0, w.fc2_w, w.fc2_b, Xs, h, S_s, S, M, h, r * h, DK_EPI_GATE_RES, mod + 5 * h, S_s,
                     mod_stride, Xs, h, S_s, S, st, m->ldh));
  return 0;
}

extern "C" int dk_mmdit_cache_context(dk_mmdit* m, const void* text, void* stream) {
  DK_REQUIRE(m && m->prepared && text, "prepare must precede cache_context");
  LinearWsScope ws_scope(m->GWS);
  const dk_mmdit_config& c = m->cfg;
  const int M = m->B * m->S_t;
  DK_TRY(linear_call((const bf16_t*)text, c.token_level_text_embed_dim, M, 0, m->ctx_w, m->ctx_b, m->CTXE, m->h(), M, 0, M, m->h(),
                     c.token_level_text_embed_dim, DK_EPI_BIAS, nullptr, 0, 0, nullptr, 0, 0, 0, S_(stream)));
  m->ctx_ready = true;
  return 0;
}

// FinalLayer (mmdit.py:767-796) on the image rows
static int mmdit_final_layer(dk_mmdit* m, const bf16_t* mod_step, bf16_t* tokens_out, hipStream_t st) {
  const int h = m->h(), B = m->B, S = m->S, S_t = m->S_t, S_i = m->S_i, F = m->F();
  const int mod_stride = m->mod_rows() * h;
  const bf16_t* mod_fin = mod_step + (size_t)m->mod_offset(3, 0) * h;
  DK_TRY(dk_launch_ln_modulate(m->X + (size_t)S_t * h, h, m->XN, h, B * S_i, h, mod_fin, mod_fin + h, mod_stride, S_i, S_i, S,
                               m->cfg.layer_norm_eps, st));
  return linear_plain(m->XN, m->final_w, m->final_b, tokens_out, B * S_i, F, h, DK_EPI_BIAS, st);
}

// ---- fp8_linears: the transformer blocks on the fp8 GEMM (gemm256f8.hip) -------------------------------------------------
// A Linear over MX-fp8 activations.  abuf / sa: the fp8 activation buffer and its scale side array (pitch lda bytes); the GEMM
// reads logical rows through (a_row0, a_seg_len, a_seg_stride) -- all multiples of 128 rows (checked in dk_mmdit_prepare).
static GemmF8Params f8_params(const dk_mmdit* m, const unsigned char* abuf, const unsigned char* sa, int lda, int a_row0, int a_seg_len,
                              int a_seg_stride, const unsigned char* W, int ldw, const float* ws, const bf16_t* bias, int M, int N, int K,
                              int epi) {
  GemmF8Params p;
  memset(&p, 0, sizeof(p));
  p.A = abuf + (size_t)a_row0 * lda; p.SA = sa; p.W = W; p.wscale = ws; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
  p.a_seg_len = a_seg_len; p.a_seg_stride = a_seg_stride; p.a_row0 = a_row0; p.sa_nblk = m->nblk;
  p.epi = epi;
  if (g_linear_ws) { p.workspace = g_linear_ws; p.workspace_bytes = dk_gemm_split_workspace_bytes(); }  // (the K split of small launches, round 6)
  return p;
}
static void f8_out_bf16(GemmF8Params& p, bf16_t* C, int ldc, int c_seg_len, int c_seg_stride) {
  p.C = C; p.ldc = ldc; p.c_seg_len = c_seg_len; p.c_seg_stride = c_seg_stride; p.c_mx8 = 0;
  p.r_seg_len = p.M; p.gate_seg_len = p.M;
}
static void f8_gate_res(GemmF8Params& p, const bf16_t* gate, int gate_seg_len, int gate_stride, const bf16_t* res, int ldr, int r_seg_len,
                        int r_seg_stride) {
  p.gate = gate; p.gate_seg_len = gate_seg_len; p.gate_stride = gate_stride; p.res = res; p.ldr = ldr; p.r_seg_len = r_seg_len;
  p.r_seg_stride = r_seg_stride;
}
static int f8_pair(const GemmF8Params& a, const GemmF8Params* b, hipStream_t st) { return dk_launch_gemm256f8(a, b, st); }

// blocks [first, first + count) of the global order (double blocks, then single blocks)
static int mmdit_blocks_fp8(dk_mmdit* m, const bf16_t* mod_step, int first, int count, hipStream_t st) {
  const dk_mmdit_config& c = m->cfg;
  const int h = m->h(), B = m->B, S = m->S, S_t = m->S_t, S_i = m->S_i, r = c.mlp_ratio;
  const int mod_stride = m->mod_rows() * h;
  const float scale = 1.0f / sqrtf((float)m->D());
  bf16_t* X_img = m->X + (size_t)S_t * h;
  bf16_t* X_txt = m->X;
  const int Mi = B * S_i, Mt = B * S_t;
  const long BS = (long)B * S;
  const int ldh8 = m->ldh8, ldcat8 = m->ldcat8;
  // XN8 / HID8 keep the bf16 path's row order: image rows [0, Mi), text rows [Mi, Mi + Mt); ATT8 / CAT8 are indexed like the
  // joint stream (b * S + s)
  const Mx8Out xn_img = mx8_out(m->XN8, m->SXN, h, BS, 0, Mi, 0, 0), xn_txt = mx8_out(m->XN8, m->SXN, h, BS, Mi, Mt, 0, 0);
  for (int i = 0; i < c.depth_multimodal; ++i) {
    if (i < first || i >= first + count) continue;
    const bf16_t* mod_img = mod_step + (size_t)m->mod_offset(0, i) * h;
    const bf16_t* mod_txt = mod_step + (size_t)m->mod_offset(1, i) * h;
    const StreamW& wi = m->dimg[i];
    const StreamW& wt = m->dtxt[i];
    const bool txt_post = !m->txt_skipped(i);
    // pre_sdpa (mmdit.py:440-519): LN-modulate -> MX-fp8, q/k/v projection, QK-norm (+ RoPE)
    DK_TRY(dk_launch_ln_modulate2_mx8(X_img, Mi, mod_img, mod_img + h, S_i, xn_img, X_txt, Mt, mod_txt, mod_txt + h, S_t, xn_txt, h, h, mod_stride,
                                      S, c.layer_norm_eps, st));
    {
      GemmF8Params qi = f8_params(m, m->XN8, m->SXN, h, 0, Mi, 0, wi.qkv_w8, h, wi.qkv_ws, wi.qkv_b, Mi, 3 * h, h, DK_EPI_BIAS);
      GemmF8Params qt = f8_params(m, m->XN8, m->SXN, h, Mi, Mt, 0, wt.qkv_w8, h, wt.qkv_ws, wt.qkv_b, Mt, 3 * h, h, DK_EPI_BIAS);
      f8_out_bf16(qi, m->QKV + (size_t)S_t * 3 * h, 3 * h, S_i, S);
      f8_out_bf16(qt, m->QKV, 3 * h, S_t, S);
      if (fuse_k(wi.kn) && fuse_k(wt.kn)) {
        set_key_norm(qi, wi.kn, wi.qn, h, m->D(), c.use_rope ? m->rope : nullptr, S_t, S_i);
        set_key_norm(qt, wt.kn, wt.qn, h, m->D(), c.use_rope ? m->rope : nullptr, 0, S_t);
      }
      DK_TRY(f8_pair(qi, &qt, st));
    }
    if (!(fuse_k(wi.kn) && fuse_k(wt.kn)))
      DK_TRY(dk_launch_qk_norm_rope2(m->QKV + (size_t)S_t * 3 * h, Mi, wi.qn, wi.kn, S_i, S_t, m->QKV, Mt, wt.qn, wt.kn, S_t, 0, 3 * h, 0, h,
                                     c.num_heads, m->D(), 1e-6f, c.use_rope ? m->rope : nullptr, S, st, fuse_q()));
    AttnParams ap;
    ap.Q = m->QKV; ap.K = m->QKV + h; ap.V = m->QKV + 2 * h; ap.O = m->ATT;
    ap.B = B; ap.H = c.num_heads; ap.S = S; ap.D = m->D(); ap.ld = 3 * h; ap.ldo = h; ap.scale = scale;
    // (queries finished by the projection's tail: plain Q load)
    if (fuse_q() && !(fuse_k(wi.kn) && fuse_k(wt.kn) && fuse_qg(wi.qn, true) && fuse_qg(wt.qn, true))) { ap.qn_a = wt.qn; ap.qn_b = wi.qn; ap.qn_split = S_t; ap.q_rope = c.use_rope ? m->rope : nullptr; }
    // the o-projection's MX-fp8 operand comes out of the attention kernel (or out of a quantiser pass behind it: attention.hip)
    ap.O8 = m->ATT8; ap.O8_scales = m->SATT; ap.o8_ld = h; ap.o8_nblk = m->nblk;
    DK_TRY(dk_launch_attention(ap, st));
    // post_sdpa (mmdit.py:537-548): residual += gate_attn * o_proj(attn)
    {
      GemmF8Params oi = f8_params(m, m->ATT8, m->SATT, h, S_t, S_i, S, wi.o_w8, h, wi.o_ws, wi.o_b, Mi, h, h, DK_EPI_GATE_RES);
      f8_out_bf16(oi, X_img, h, S_i, S);
      f8_gate_res(oi, mod_img + 2 * h, S_i, mod_stride, X_img, h, S_i, S);
      if (txt_post) {
        GemmF8Params ot = f8_params(m, m->ATT8, m->SATT, h, 0, S_t, S, wt.o_w8, h, wt.o_ws, wt.o_b, Mt, h, h, DK_EPI_GATE_RES);
        f8_out_bf16(ot, X_txt, h, S_t, S);
        f8_gate_res(ot, mod_txt + 2 * h, S_t, mod_stride, X_txt, h, S_t, S);
        DK_TRY(f8_pair(oi, &ot, st));
      } else {
        DK_TRY(f8_pair(oi, nullptr, st));
      }
    }
    // residual += gate_mlp * fc2(gelu(fc1(LN-mod(residual)))); the MLP hidden leaves fc1 as MX-fp8
    if (txt_post)
      DK_TRY(dk_launch_ln_modulate2_mx8(X_img, Mi, mod_img + 3 * h, mod_img + 4 * h, S_i, xn_img, X_txt, Mt, mod_txt + 3 * h, mod_txt + 4 * h, S_t,
                                        xn_txt, h, h, mod_stride, S, c.layer_norm_eps, st));
    else
      DK_TRY(dk_launch_ln_modulate_mx8(X_img, h, Mi, h, mod_img + 3 * h, mod_img + 4 * h, mod_stride, S_i, S_i, S, c.layer_norm_eps, xn_img, st));
    {
      GemmF8Params f1i = f8_params(m, m->XN8, m->SXN, h, 0, Mi, 0, wi.fc1_w8, h, wi.fc1_ws, wi.fc1_b, Mi, r * h, h, DK_EPI_BIAS_GELU);
      f1i.C = m->HC8; f1i.ldc = ldh8; f1i.c_seg_len = Mi; f1i.c_mx8 = 1; f1i.SC = m->SHID; f1i.sc_nblk = m->nblk; f1i.c_row0 = 0; f1i.sc_kb0 = 0;
      f1i.r_seg_len = Mi; f1i.gate_seg_len = Mi;
      GemmF8Params f2i = f8_params(m, m->HC8, m->SHID, ldh8, 0, Mi, 0, wi.fc2_w8, ldh8, wi.fc2_ws, wi.fc2_b, Mi, h, r * h, DK_EPI_GATE_RES);
      f8_out_bf16(f2i, X_img, h, S_i, S);
      f8_gate_res(f2i, mod_img + 5 * h, S_i, mod_stride, X_img, h, S_i, S);
      if (txt_post) {
        GemmF8Params f1t = f8_params(m, m->XN8, m->SXN, h, Mi, Mt, 0, wt.fc1_w8, h, wt.fc1_ws, wt.fc1_b, Mt, r * h, h, DK_EPI_BIAS_GELU);
        f1t.C = m->HC8 + (size_t)Mi * ldh8; f1t.ldc = ldh8; f1t.c_seg_len = Mt; f1t.c_mx8 = 1; f1t.SC = m->SHID; f1t.sc_nblk = m->nblk; f1t.c_row0 = Mi;
        f1t.sc_kb0 = 0; f1t.r_seg_len = Mt; f1t.gate_seg_len = Mt;
        GemmF8Params f2t = f8_params(m, m->HC8, m->SHID, ldh8, Mi, Mt, 0, wt.fc2_w8, ldh8, wt.fc2_ws, wt.fc2_b, Mt, h, r * h, DK_EPI_GATE_RES);
        f8_out_bf16(f2t, X_txt, h, S_t, S);
        f8_gate_res(f2t, mod_txt + 5 * h, S_t, mod_stride, X_txt, h, S_t, S);
        DK_TRY(f8_pair(f1i, &f1t, st));
        DK_TRY(f8_pair(f2i, &f2t, st));
      } else {
        DK_TRY(f8_pair(f1i, nullptr, st));
        DK_TRY(f8_pair(f2i, nullptr, st));
      }
    }
  }
  // UnifiedTransformerBlock x depth_unified (mmdit.py:693-751)
  const int M = B * S;
  const Mx8Out xn_all = mx8_out(m->XN8, m->SXN, h, BS, 0, M, 0, 0);
  for (int i = 0; i < c.depth_unified; ++i) {
    if (c.depth_multimodal + i < first || c.depth_multimodal + i >= first + count) continue;
    const StreamW& w = m->single[i];
    const bf16_t* mod = mod_step + (size_t)m->mod_offset(2, i) * h;
    DK_TRY(dk_launch_ln_modulate_mx8(m->X, h, M, h, mod, mod + h, mod_stride, S, M, 0, c.layer_norm_eps, xn_all, st));
    {  // linear1: [q|k|v] -> QKV (bf16), gelu(fc1) -> CAT8[:, h:] (MX-fp8), one pass over the modulated activations
      GemmF8Params l1 = f8_params(m, m->XN8, m->SXN, h, 0, M, 0, w.qkv_w8, h, w.qkv_ws, w.qkv_b, M, (3 + r) * h, h, DK_EPI_BIAS);
      f8_out_bf16(l1, m->QKV, 3 * h, M, 0);
      l1.n_split = 3 * h; l1.C2 = m->HC8 + h; l1.ldc2 = ldcat8; l1.epi2 = DK_EPI_BIAS_GELU; l1.c2_mx8 = 1;
      l1.SC = m->SCAT; l1.sc_nblk = m->nblk; l1.c_row0 = 0; l1.sc_kb0 = h / 32;
      if (fuse_k(w.kn)) set_key_norm(l1, w.kn, w.qn, h, m->D(), c.use_rope ? m->rope : nullptr, 0, S);
      DK_TRY(f8_pair(l1, nullptr, st));
    }
    if (!fuse_k(w.kn))
      DK_TRY(dk_launch_qk_norm_rope(m->QKV, 3 * h, 0, h, M, c.num_heads, m->D(), w.qn, w.kn, 1e-6f, c.use_rope ? m->rope : nullptr, S, S, 0, S,
                                    st, fuse_q()));
    AttnParams ap;
    ap.Q = m->QKV; ap.K = m->QKV + h; ap.V = m->QKV + 2 * h; ap.O = m->ATT;
    ap.B = B; ap.H = c.num_heads; ap.S = S; ap.D = m->D(); ap.ld = 3 * h; ap.ldo = h; ap.scale = scale;
    if (fuse_q() && !(fuse_k(w.kn) && fuse_qg(w.qn, true))) { ap.qn_a = ap.qn_b = w.qn; ap.qn_split = 0; ap.q_rope = c.use_rope ? m->rope : nullptr; }
    ap.O8 = m->HC8; ap.O8_scales = m->SCAT; ap.o8_ld = ldcat8; ap.o8_nblk = m->nblk;  // [attn | gelu] operand of linear2: columns [0, h)
    DK_TRY(dk_launch_attention(ap, st));
    {  // x += gate * ([attn | gelu] @ [o_proj | fc2]^T + bias)   (one bias: quirk Q8)
      GemmF8Params l2 = f8_params(m, m->HC8, m->SCAT, ldcat8, 0, M, 0, w.l2_w8, ldcat8, w.l2_ws, w.l2_b, M, h, (1 + r) * h, DK_EPI_GATE_RES);
      f8_out_bf16(l2, m->X, h, M, 0);
      f8_gate_res(l2, mod + 2 * h, S, mod_stride, m->X, h, M, 0);
      DK_TRY(f8_pair(l2, nullptr, st));
    }
  }
  return 0;
}

static int mmdit_blocks_fp8(dk_mmdit* m, const bf16_t* mod_step, int first, int count, hipStream_t st);
static int mmdit_blocks_bf16(dk_mmdit* m, const bf16_t* mod_step, int first, int count, hipStream_t st);
// blocks [first, first + count): the bf16 double blocks of the precision policy on the bf16 path, everything else on the model's own
static int mmdit_blocks(dk_mmdit* m, const bf16_t* mod_step, int first, int count, hipStream_t st) {
  if (!m->fp8()) return mmdit_blocks_bf16(m, mod_step, first, count, st);
  const int nb = m->n_bf16(), end = first + count;
  if (first < nb) DK_TRY(mmdit_blocks_bf16(m, mod_step, first, (end < nb ? end : nb) - first, st));
  if (end > nb) DK_TRY(mmdit_blocks_fp8(m, mod_step, first > nb ? first : nb, end - (first > nb ? first : nb), st));
  return 0;
}

struct AttnWsScope {  // an engine call's attention launches split through that engine's region; the host thread's own setting comes back
  void* prev;
  size_t prev_bytes;
  AttnWsScope(void* ws, size_t bytes) : prev(dk_get_attention_workspace()), prev_bytes(dk_get_attention_workspace_bytes()) {
    dk_set_attention_workspace(ws, bytes);  // (an engine without a region -- D = 64 -- runs unsplit: never the thread's buffer on another engine's stream)
  }
  ~AttnWsScope() { dk_set_attention_workspace(prev, prev_bytes); }
};
struct MmditCallScope {  // an engine call's GEMM and attention splits go through THAT engine's regions; the caller's settings come back
  LinearWsScope lin;
  AttnWsScope att;
  explicit MmditCallScope(dk_mmdit* m) : lin(m->GWS), att(m->AWS, m->AWS_bytes) {}
};

// The transformer blocks [first, first + count) of the global order (double blocks 0 .. depth_multimodal - 1, then single blocks)
// on the joint residual stream m->X, bf16 Linears.
static int mmdit_blocks_bf16(dk_mmdit* m, const bf16_t* mod_step, int first, int count, hipStream_t st) {
  const dk_mmdit_config& c = m->cfg;
  const int h = m->h(), B = m->B, S = m->S, S_t = m->S_t, S_i = m->S_i, r = c.mlp_ratio;
  const int mod_stride = m->mod_rows() * h;  // per batch row
  const float scale = 1.0f / sqrtf((float)m->D());
  // MultiModalTransformerBlock x depth_multimodal (mmdit.py:568-675).  The two streams run the same
  // Linear shapes on different weights; their GEMMs are issued as pairs so that the 256 x 256 kernel can
  // place the text tiles in the same wave as the image tiles (dk_launch_gemm_pair).
  bf16_t* XN_img = m->XN;
  bf16_t* XN_txt = m->XN + (size_t)B * S_i * h;
  bf16_t* HID_img = m->HID;
  const int ldh = m->ldh;
  bf16_t* HID_txt = m->HID + (size_t)B * S_i * ldh;
  bf16_t* X_img = m->X + (size_t)S_t * h;
  bf16_t* X_txt = m->X;
  const int Mi = B * S_i, Mt = B * S_t;
  for (int i = 0; i < c.depth_multimodal; ++i) {
    if (i < first || i >= first + count) continue;
    const bf16_t* mod_img = mod_step + (size_t)m->mod_offset(0, i) * h;
    const bf16_t* mod_txt = mod_step + (size_t)m->mod_offset(1, i) * h;
    const StreamW& wi = m->dimg[i];
    const StreamW& wt = m->dtxt[i];
    const bool txt_post = !m->txt_skipped(i);
    // pre_sdpa (mmdit.py:440-519): LN-modulate, q/k/v projection, QK-norm (+ RoPE)
    // (image and text stream of each elementwise stage in ONE launch: the 256 text rows do not run alone on the chip)
    DK_TRY(dk_launch_ln_modulate2(X_img, XN_img, Mi, mod_img, mod_img + h, S_i, X_txt, XN_txt, Mt, mod_txt, mod_txt + h, S_t, h, h, h,
                                  mod_stride, S, c.layer_norm_eps, st));
    GemmParams qkv_img = linear_params(XN_img, h, Mi, 0, wi.qkv_w, wi.qkv_b, m->QKV + (size_t)S_t * 3 * h, 3 * h, S_i, S, Mi, 3 * h, h, DK_EPI_BIAS,
                                       nullptr, 0, 0, nullptr, 0, 0, 0);
    GemmParams qkv_txt = linear_params(XN_txt, h, Mt, 0, wt.qkv_w, wt.qkv_b, m->QKV, 3 * h, S_t, S, Mt, 3 * h, h, DK_EPI_BIAS, nullptr, 0, 0,
                                       nullptr, 0, 0, 0);
    // QKNorm + RoPE: the keys in the projection's tail (or one pass over the buffer), the queries inside the attention kernel's Q load
    const bool kf = fuse_k(wi.kn) && fuse_k(wt.kn);
    if (kf) {
      set_key_norm(qkv_img, wi.kn, wi.qn, h, m->D(), c.use_rope ? m->rope : nullptr, S_t, S_i);
      set_key_norm(qkv_txt, wt.kn, wt.qn, h, m->D(), c.use_rope ? m->rope : nullptr, 0, S_t);
    }
    DK_TRY(dk_launch_gemm_pair(qkv_img, qkv_txt, st));
    if (!kf)
      DK_TRY(dk_launch_qk_norm_rope2(m->QKV + (size_t)S_t * 3 * h, Mi, wi.qn, wi.kn, S_i, S_t, m->QKV, Mt, wt.qn, wt.kn, S_t, 0, 3 * h, 0, h,
                                     c.num_heads, m->D(), 1e-6f, c.use_rope ? m->rope : nullptr, S, st, fuse_q()));
    AttnParams ap;
    ap.Q = m->QKV; ap.K = m->QKV + h; ap.V = m->QKV + 2 * h; ap.O = m->ATT;
    ap.B = B; ap.H = c.num_heads; ap.S = S; ap.D = m->D(); ap.ld = 3 * h; ap.ldo = h; ap.scale = scale;
    if (fuse_q() && !(kf && fuse_qg(wi.qn, false) && fuse_qg(wt.qn, false))) { ap.qn_a = wt.qn; ap.qn_b = wi.qn; ap.qn_split = S_t; ap.q_rope = c.use_rope ? m->rope : nullptr; }
    DK_TRY(dk_launch_attention(ap, st));
    // post_sdpa, sequential form (mmdit.py:537-548): residual += gate_attn * o_proj(attn)
    const GemmParams o_img = linear_params(m->ATT + (size_t)S_t * h, h, S_i, S, wi.o_w, wi.o_b, X_img, h, S_i, S, Mi, h, h, DK_EPI_GATE_RES,
                                           mod_img + 2 * h, S_i, mod_stride, X_img, h, S_i, S);
    if (txt_post) {
      DK_TRY(dk_launch_gemm_pair(o_img,
                                 linear_params(m->ATT, h, S_t, S, wt.o_w, wt.o_b, X_txt, h, S_t, S, Mt, h, h, DK_EPI_GATE_RES,
                                               mod_txt + 2 * h, S_t, mod_stride, X_txt, h, S_t, S),
                                 st));
    } else {
      DK_TRY(dk_launch_gemm(o_img, st));
    }
    // residual += gate_mlp * fc2(gelu(fc1(LN-mod(residual))))
    if (txt_post)
      DK_TRY(dk_launch_ln_modulate2(X_img, XN_img, Mi, mod_img + 3 * h, mod_img + 4 * h, S_i, X_txt, XN_txt, Mt, mod_txt + 3 * h,
                                    mod_txt + 4 * h, S_t, h, h, h, mod_stride, S, c.layer_norm_eps, st));
    else
      DK_TRY(dk_launch_ln_modulate(X_img, h, XN_img, h, Mi, h, mod_img + 3 * h, mod_img + 4 * h, mod_stride, S_i, S_i, S, c.layer_norm_eps, st));
    const GemmParams fc1_img = linear_params(XN_img, h, Mi, 0, wi.fc1_w, wi.fc1_b, HID_img, ldh, Mi, 0, Mi, r * h, h, DK_EPI_BIAS_GELU,
                                             nullptr, 0, 0, nullptr, 0, 0, 0);
    const GemmParams fc2_img = linear_params(HID_img, ldh, Mi, 0, wi.fc2_w, wi.fc2_b, X_img, h, S_i, S, Mi, h, r * h, DK_EPI_GATE_RES,
                                             mod_img + 5 * h, S_i, mod_stride, X_img, h, S_i, S, ldh);
    if (txt_post) {
      DK_TRY(dk_launch_gemm_pair(fc1_img,
                                 linear_params(XN_txt, h, Mt, 0, wt.fc1_w, wt.fc1_b, HID_txt, ldh, Mt, 0, Mt, r * h, h,
                                               DK_EPI_BIAS_GELU, nullptr, 0, 0, nullptr, 0, 0, 0),
                                 st));
      DK_TRY(dk_launch_gemm_pair(fc2_img,
                                 linear_params(HID_txt, ldh, Mt, 0, wt.fc2_w, wt.fc2_b, X_txt, h, S_t, S, Mt, h, r * h, DK_EPI_GATE_RES,
                                               mod_txt + 5 * h, S_t, mod_stride, X_txt, h, S_t, S, ldh),
                                 st));
    } else {
      DK_TRY(dk_launch_gemm(fc1_img, st));
      DK_TRY(dk_launch_gemm(fc2_img, st));
    }
  }

  // UnifiedTransformerBlock x depth_unified (mmdit.py:693-751), parallel attention + MLP
  for (int i = 0; i < c.depth_unified; ++i) {
    if (c.depth_multimodal + i < first || c.depth_multimodal + i >= first + count) continue;
    const StreamW& w = m->single[i];
    const bf16_t* mod = mod_step + (size_t)m->mod_offset(2, i) * h;
    const int M = B * S, ldcat = m->ldcat;
    DK_TRY(dk_launch_ln_modulate(m->X, h, m->XN, h, M, h, mod, mod + h, mod_stride, S, M, 0, c.layer_norm_eps, st));
    {  // linear1: [q|k|v] -> QKV, gelu(fc1) -> CAT[:, h:], one pass over the modulated activations
      GemmParams l1 = linear_params(m->XN, h, M, 0, w.qkv_w, w.qkv_b, m->QKV, 3 * h, M, 0, M, (3 + r) * h, h, DK_EPI_BIAS, nullptr, 0, 0,
                                    nullptr, 0, 0, 0);
      l1.n_split = 3 * h; l1.C2 = m->CAT + h; l1.ldc2 = ldcat; l1.epi2 = DK_EPI_BIAS_GELU;
      if (fuse_k(w.kn)) set_key_norm(l1, w.kn, w.qn, h, m->D(), c.use_rope ? m->rope : nullptr, 0, S);
      DK_TRY(dk_launch_gemm(l1, st));
    }
    if (!fuse_k(w.kn))
      DK_TRY(dk_launch_qk_norm_rope(m->QKV, 3 * h, 0, h, M, c.num_heads, m->D(), w.qn, w.kn, 1e-6f, c.use_rope ? m->rope : nullptr,
                                    S, S, 0, S, st, fuse_q()));
    AttnParams ap;
    ap.Q = m->QKV; ap.K = m->QKV + h; ap.V = m->QKV + 2 * h; ap.O = m->CAT;
    ap.B = B; ap.H = c.num_heads; ap.S = S; ap.D = m->D(); ap.ld = 3 * h; ap.ldo = ldcat; ap.scale = scale;
    if (fuse_q() && !(fuse_k(w.kn) && fuse_qg(w.qn, false))) { ap.qn_a = ap.qn_b = w.qn; ap.qn_split = 0; ap.q_rope = c.use_rope ? m->rope : nullptr; }
    DK_TRY(dk_launch_attention(ap, st));
    // x += gate * ([attn | gelu] @ [o_proj | fc2]^T + bias)   (one bias: quirk Q8)
    {  // (through linear_params: with this engine's split workspace -- below 1024 x 1024 the launch is a fraction of a round of the CUs and is cut along K)
      const GemmParams l2 = linear_params(m->CAT, ldcat, M, 0, w.l2_w, w.l2_b, m->X, h, M, 0, M, h, (1 + r) * h, DK_EPI_GATE_RES, mod + 2 * h, S, mod_stride,
                                          m->X, h, M, 0, ldcat);
      DK_TRY(dk_launch_gemm(l2, st));
    }
  }

  return 0;
}

extern "C" int dk_mmdit_forward(dk_mmdit* m, const void* tokens_in, const void* text, int32_t step_index, void* tokens_out,
                                void* stream) {
  DK_REQUIRE(m && m->prepared && m->mod_ready, "prepare + cache_modulation_params must precede forward");
  DK_REQUIRE(step_index >= 0 && step_index < m->n_t, "step index out of range");
  hipStream_t st = S_(stream);
  MmditCallScope scope(m);
  const dk_mmdit_config& c = m->cfg;
  const int h = m->h(), B = m->B, S = m->S, S_t = m->S_t, S_i = m->S_i, F = m->F();
  const bf16_t* mod_step = m->MOD + (size_t)step_index * B * m->mod_rows() * h;

  // context_embedder (mmdit.py:195): text rows of the joint stream -- recomputed from `text`, or copied from the
  // step-invariant result of dk_mmdit_cache_context when `text` is null
  if (text != nullptr) {
    DK_TRY(linear_call((const bf16_t*)text, c.token_level_text_embed_dim, B * S_t, 0, m->ctx_w, m->ctx_b, m->X, h, S_t, S, B * S_t, h,
                       c.token_level_text_embed_dim, DK_EPI_BIAS, nullptr, 0, 0, nullptr, 0, 0, 0, st));
  } else {
    DK_REQUIRE(m->ctx_ready, "forward without text needs dk_mmdit_cache_context first");
    DK_CHECK_HIP(hipMemcpy2DAsync(m->X, (size_t)S * h * 2, m->CTXE, (size_t)S_t * h * 2, (size_t)S_t * h * 2, B, hipMemcpyDeviceToDevice, st));
  }
  // x_embedder (+ learned positional embedding) (mmdit.py:197-206): image rows
  DK_TRY(linear_call((const bf16_t*)tokens_in, F, B * S_i, 0, m->xemb_w, m->xemb_b, m->X + (size_t)S_t * h, h, S_i, S, B * S_i, h, F,
                     c.use_pos_embed ? DK_EPI_RES : DK_EPI_BIAS, nullptr, 0, 0, c.use_pos_embed ? m->POS : nullptr, h, S_i, 0, st));
  const int n_blocks = c.depth_multimodal + c.depth_unified;
  DK_TRY(mmdit_blocks(m, mod_step, 0, n_blocks, st));
  return mmdit_final_layer(m, mod_step, (bf16_t*)tokens_out, st);
}

// Teacher-forced block range (include/dk_hip.h): x_in -> m->X, blocks [first, first + count), m->X -> x_out
extern "C" int dk_mmdit_run_blocks(dk_mmdit* m, const void* x_in, void* x_out, int32_t step_index, int32_t first_block, int32_t n_blocks,
                                   void* stream) {
  DK_REQUIRE(m && m->prepared && m->mod_ready, "prepare + cache_modulation_params must precede run_blocks");
  DK_REQUIRE(x_in && x_out, "null argument");
  DK_REQUIRE(step_index >= 0 && step_index < m->n_t, "step index out of range");
  const int total = m->cfg.depth_multimodal + m->cfg.depth_unified;
  DK_REQUIRE(first_block >= 0 && n_blocks >= 1 && first_block + n_blocks <= total, "block range outside the model");
  hipStream_t st = S_(stream);
  MmditCallScope scope(m);
  const size_t bytes = (size_t)m->B * m->S * m->h() * 2;
  const bf16_t* mod_step = m->MOD + (size_t)step_index * m->B * m->mod_rows() * m->h();
  DK_CHECK_HIP(hipMemcpyAsync(m->X, x_in, bytes, hipMemcpyDeviceToDevice, st));
  DK_TRY(mmdit_blocks(m, mod_step, first_block, n_blocks, st));
  DK_CHECK_HIP(hipMemcpyAsync(x_out, m->X, bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

extern "C" const void* dk_mmdit_debug_buffer(const dk_mmdit* m, int32_t which) {
  if (!m || !m->prepared) return nullptr;
  return which == 0 ? (const void*)m->X : which == 1 ? (const void*)m->MOD : nullptr;
}

// ---------------------------------------------------------------------------------------------
// VAE decoder engine (vae.py:336-401)
// ---------------------------------------------------------------------------------------------
struct dk_vae {
  dk_vae_config cfg;
  std::unordered_map<std::string, const void*> named;
  // workspace views
  bf16_t *bufA, *bufB, *T1, *Y, *SC, *LAT, *ZERO, *Qb, *Kb, *Vb, *Vt, *SCORES;
  float* gn;
  float *ss0, *ss1;  // GroupNorm (scale | shift) tables [B][2][C] of the fused norm -> silu -> conv stages
  void* GWS = nullptr;  // GEMM split workspace of this engine's launches (fp32 slabs + flags)
};

extern "C" int dk_vae_create(const dk_vae_config* cfg, dk_vae** out) {
  DK_REQUIRE(cfg && out, "null argument");
  DK_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= 4, "1..4 resolution levels");
  for (int i = 0; i < cfg->n_blocks; ++i)
    DK_REQUIRE(cfg->block_out_channels[i] % 64 == 0, "VAE channel counts must be multiples of 64");
  DK_REQUIRE(cfg->in_channels >= 1 && cfg->in_channels <= 64 && cfg->out_channels >= 1 && cfg->out_channels <= 64,
             "input / output channels of the VAE halves: 1..64");
  dk_vae* v = new dk_vae();
  v->cfg = *cfg;
  *out = v;
  return 0;
}
extern "C" void dk_vae_destroy(dk_vae* v) { delete v; }
extern "C" int dk_vae_bind(dk_vae* v, const char* name, const void* dev_ptr) {
  DK_REQUIRE(v && name && dev_ptr, "null argument");
  v->named[name] = dev_ptr;
  return 0;
}

static size_t vae_carve(dk_vae* v, Carver& c, int B, int h, int w) {
  const dk_vae_config& cf = v->cfg;
  // largest activation: walk the decoder (vae.py:386-401) and take max(B * H * W * C)
  size_t maxel = 0;
  {
    size_t H = h, W = w;
    int Cprev = cf.block_out_channels[cf.n_blocks - 1];
    maxel = (size_t)B * H * W * Cprev;
    for (int j = cf.n_blocks - 1; j >= 0; --j) {
      const int Cout = cf.block_out_channels[j];
      const size_t e = (size_t)B * H * W * (size_t)(Cprev > Cout ? Cprev : Cout);
      if (e > maxel) maxel = e;
      if (j > 0) {
        H *= 2;
        W *= 2;
        if ((size_t)B * H * W * Cout > maxel) maxel = (size_t)B * H * W * Cout;
      }
      Cprev = Cout;
    }
  }
  const size_t act = maxel * 2;
  v->bufA = (bf16_t*)c.take(act);
  v->bufB = (bf16_t*)c.take(act);
  v->T1 = (bf16_t*)c.take(act);
  v->Y = (bf16_t*)c.take(act);
  v->SC = (bf16_t*)c.take(act);
  v->LAT = (bf16_t*)c.take((size_t)B * h * w * 64 * 2);
  v->ZERO = (bf16_t*)c.take(256);
  const int Cm = cf.block_out_channels[cf.n_blocks - 1];
  const size_t tok = (size_t)h * w;
  v->Qb = (bf16_t*)c.take((size_t)B * tok * Cm * 2);
  v->Kb = (bf16_t*)c.take((size_t)B * tok * Cm * 2);
  v->Vb = (bf16_t*)c.take((size_t)B * tok * Cm * 2);
  v->Vt = (bf16_t*)c.take((size_t)B * align_up(tok, 64) * Cm * 2);  // (the flash form transposes every image's V up front)
  // the materialised score matrix of the general path; a 512-channel mid block runs the flash kernel (attention512.hip) and needs none
  v->SCORES = (bf16_t*)c.take(Cm == 512 ? 0 : tok * align_up(tok, 64) * 2);
  {
    // statistics scratch: up to 1024 chunk partials per batch row from the stand-alone pass, or one per 16 x 16 output tile of the
    // largest stage from the fused convs, + mean / rstd
    const size_t tiles = ((size_t)h << (cf.n_blocks - 1)) / 16 * (((size_t)w << (cf.n_blocks - 1)) / 16);
    const size_t npart = tiles > 1024 ? tiles : 1024;
    v->gn = (float*)c.take(((size_t)B * npart * 2 * cf.resnet_groups + (size_t)B * cf.resnet_groups * 2) * 4);
    int cmax = 0;
    for (int j = 0; j < cf.n_blocks; ++j) cmax = cf.block_out_channels[j] > cmax ? cf.block_out_channels[j] : cmax;
    v->ss0 = (float*)c.take((size_t)B * 2 * cmax * 4);
    v->ss1 = (float*)c.take((size_t)B * 2 * cmax * 4);
  }
  v->GWS = c.take(dk_gemm_split_workspace_bytes());
  return c.off;
}
extern "C" size_t dk_vae_workspace_bytes(const dk_vae* v, int32_t batch, int32_t latent_h, int32_t latent_w) {
  dk_vae tmp = *v;
  Carver c(nullptr, 0);
  return vae_carve(&tmp, c, batch, latent_h, latent_w) + 256;
}

struct VaeRun {
  dk_vae* v;
  hipStream_t st;
  int B;
  int rc = 0;
  const bf16_t* W(const std::string& name) {
    const bf16_t* p = nullptr;
    if (rc == 0) rc = need(v->named, name, &p);
    return p;
  }
  bool has(const std::string& name) const { return v->named.count(name) != 0; }
  int gn(const bf16_t* x, bf16_t* y, long HW, int C, const std::string& name, int silu) {
    const bf16_t *g = W(name + ".weight"), *b = W(name + ".bias");
    if (rc) return rc;
    return dk_groupnorm_bf16(x, y, B, HW, C, v->cfg.resnet_groups, g, b, v->cfg.group_norm_eps, silu, v->gn, st);
  }
  int conv(const bf16_t* x, bf16_t* y, int H, int Wd, int C, int O, const std::string& name, int ups, const bf16_t* res, int ldy) {
    dk_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.w = W(name + ".weight"); d.bias = W(name + ".bias"); d.y = y; d.res = res; d.zeros = v->ZERO;
    if (rc) return rc;
    d.B = B; d.H = H; d.W = Wd; d.C = C; d.O = O; d.ldy = ldy; d.ldr = O; d.upsample = ups;
    d.epilogue = res ? DK_EPI_RES : DK_EPI_BIAS;
    return conv3x3_launch(&d, v->GWS, st);
  }
  // ---- fused norm -> silu -> conv stages (conv_halo.hip) ----
  int n_part = 0;  // > 0: v->gn holds the output-statistics partials [B][n_part][G][2] of the tensor the last fused conv wrote
  bool halo_stage(int H, int Wd, int Cin, int Cout) const {
    if (g_dk_conv_halo == 0 || H % 16 != 0 || Wd % 16 != 0 || Cin % 64 != 0 || Cout % 128 != 0) return false;
    if ((size_t)H * Wd * (Cin > Cout ? Cin : Cout) * 2 >= (1ull << 31)) return false;
    // the fused stages always ask their conv for the output statistics of the next GroupNorm (dk_conv_halo_eligible: a channel
    // group must divide the 128-channel workgroup tile and span at most 64 channels) -- other group plans take gn() + conv()
    const int G = v->cfg.resnet_groups;
    if (G <= 0 || Cout % G != 0 || 128 % (Cout / G) != 0 || Cout / G > 64) return false;
    return g_dk_conv_halo != 2 || Cout < 256;
  }
  // the (scale | shift) table of GroupNorm `name` over tensor x: from the partials the producing conv left, or a statistics pass
  int gn_table(const bf16_t* x, long HW, int C, const std::string& name, float* ss) {
    const bf16_t *g = W(name + ".weight"), *b = W(name + ".bias");
    if (rc) return rc;
    const int np = n_part;
    n_part = 0;
    return dk_groupnorm_table_bf16(np > 0 ? nullptr : x, B, HW, C, v->cfg.resnet_groups, g, b, v->cfg.group_norm_eps, v->gn, np, ss, st);
  }
  // ResnetBlock2D (vae.py:60-101) in two launches + two statistics finalisations: x stays raw, both GroupNorm + SiLU are applied
  // on the way into the convs' LDS halo tiles, conv1 / conv2 leave the statistics of their outputs behind (stats_next: somebody
  // normalises `out` next), the 1x1 shortcut rides in conv2's reduction
  int resnet_fused(const bf16_t* x, bf16_t* out, int H, int Wd, int Cin, int Cout, const std::string& p, bool stats_next) {
    const long HW = (long)H * Wd;
    const int tiles = (H / 16) * (Wd / 16), G = v->cfg.resnet_groups;
    DK_TRY(gn_table(x, HW, Cin, p + ".norm1", v->ss0));
    ConvHaloParams c;
    memset(&c, 0, sizeof(c));
    c.x = x; c.w = W(p + ".conv1.weight"); c.bias = W(p + ".conv1.bias"); c.y = v->Y; c.gn_ss = v->ss0; c.gn_silu = 1;
    c.stats_out = v->gn; c.G_out = G;
    c.B = B; c.H = H; c.W = Wd; c.C = Cin; c.O = Cout; c.ldw = 9 * Cin; c.ldy = Cout;
    if (rc) return rc;
    DK_TRY(dk_launch_conv_halo(c, st));
    n_part = tiles;
    DK_TRY(gn_table(v->Y, HW, Cout, p + ".norm2", v->ss1));
    memset(&c, 0, sizeof(c));
    c.x = v->Y; c.bias = W(p + ".conv2.bias"); c.y = out; c.gn_ss = v->ss1; c.gn_silu = 1;
    c.B = B; c.H = H; c.W = Wd; c.C = Cout; c.O = Cout; c.ldy = Cout; c.ldr = Cout;
    if (has(p + ".conv_shortcut.weight")) {
      c.w = W(p + ".conv2_sc.weight");  // [conv2 | conv_shortcut] along the reduction (weights.pack_vae)
      c.ldw = 9 * Cout + Cin; c.x2 = x; c.C2 = Cin; c.bias2 = W(p + ".conv_shortcut.bias");
    } else {
      DK_REQUIRE(Cin == Cout, "resnet without shortcut must keep the channel count");
      c.w = W(p + ".conv2.weight"); c.ldw = 9 * Cout; c.res = x;
    }
    if (stats_next) { c.stats_out = v->gn; c.G_out = G; }
    if (rc) return rc;
    DK_TRY(dk_launch_conv_halo(c, st));
    n_part = stats_next ? tiles : 0;
    return 0;
  }
  // ResnetBlock2D (vae.py:60-101): x [B,H,W,Cin] -> out [B,H,W,Cout]
  int resnet(const bf16_t* x, bf16_t* out, int H, int Wd, int Cin, int Cout, const std::string& p, bool stats_next = false) {
    if (halo_stage(H, Wd, Cin, Cout) && (Cin == Cout || has(p + ".conv2_sc.weight"))) return resnet_fused(x, out, H, Wd, Cin, Cout, p, stats_next);
    n_part = 0;
    const long HW = (long)H * Wd;
    DK_TRY(gn(x, v->T1, HW, Cin, p + ".norm1", 1));
    DK_TRY(conv(v->T1, v->Y, H, Wd, Cin, Cout, p + ".conv1", 0, nullptr, Cout));
    DK_TRY(gn(v->Y, v->T1, HW, Cout, p + ".norm2", 1));
    const bf16_t* res = x;
    if (has(p + ".conv_shortcut.weight")) {
      const bf16_t *sw = W(p + ".conv_shortcut.weight"), *sb = W(p + ".conv_shortcut.bias");
      if (rc) return rc;
      DK_TRY(linear_plain(x, sw, sb, v->SC, (int)(B * HW), Cout, Cin, DK_EPI_BIAS, st));
      res = v->SC;
    } else {
      DK_REQUIRE(Cin == Cout, "resnet without shortcut must keep the channel count");
    }
    return conv(v->T1, out, H, Wd, Cout, Cout, p + ".conv2", 0, res, Cout);
  }
  // single-head attention (vae.py:28-57)
  int attention(const bf16_t* x, bf16_t* out, int H, int Wd, int C, const std::string& p) {
    const long HW = (long)H * Wd;
    const int T = (int)HW;
    DK_REQUIRE(T % 4 == 0, "VAE attention: even latent sides");
    const int Tp = (int)align_up((size_t)T, 64);  // K of the P.V product: zero-padded probability columns / V^T rows
    DK_TRY(gn(x, v->T1, HW, C, p + ".group_norm", 0));
    const bf16_t *qw = W(p + ".query_proj.weight"), *qb = W(p + ".query_proj.bias");
    const bf16_t *kw = W(p + ".key_proj.weight"), *kb = W(p + ".key_proj.bias");
    const bf16_t *vw = W(p + ".value_proj.weight"), *vb = W(p + ".value_proj.bias");
    const bf16_t *ow = W(p + ".out_proj.weight"), *ob = W(p + ".out_proj.bias");
    if (rc) return rc;
    DK_TRY(linear_plain(v->T1, qw, qb, v->Qb, B * T, C, C, DK_EPI_BIAS, st));
    DK_TRY(linear_plain(v->T1, kw, kb, v->Kb, B * T, C, C, DK_EPI_BIAS, st));
    DK_TRY(linear_plain(v->T1, vw, vb, v->Vb, B * T, C, C, DK_EPI_BIAS, st));
    const float scale = 1.0f / sqrtf((float)C);
    if (C == 512) {
      // flash form (attention512.hip): no [T, T] score matrix
      DK_TRY(attention_d512(v->Qb, v->Kb, v->Vb, v->Y, B, T, C, C, scale, v->Vt, st));
      return linear_call(v->Y, C, B * T, 0, ow, ob, out, C, B * T, 0, B * T, C, C, DK_EPI_RES, nullptr, 0, 0, x, C, B * T, 0, st);
    }
    for (int b = 0; b < B; ++b) {
      GemmParams g;
      memset(&g, 0, sizeof(g));
      g.A = v->Qb + (size_t)b * T * C; g.W = v->Kb + (size_t)b * T * C; g.C = v->SCORES;
      g.M = T; g.N = T; g.K = C; g.lda = C; g.ldc = Tp;
      g.a_seg_len = g.c_seg_len = g.r_seg_len = g.gate_seg_len = T;
      g.alpha = scale; g.epi = DK_EPI_BIAS;
      DK_TRY(dk_launch_gemm(g, st));
      DK_TRY(dk_launch_softmax_rows(v->SCORES, T, T, Tp, st));  // columns [T, Tp) come out as zeros
      DK_TRY(dk_launch_transpose(v->Vb + (size_t)b * T * C, v->Vt, T, C, st, Tp));
      // attn @ V: A = probs [T, Tp], W = V^T [C, Tp]; result into Y rows of this batch
      DK_TRY(linear_plain(v->SCORES, v->Vt, nullptr, v->Y + (size_t)b * T * C, T, C, Tp, DK_EPI_BIAS, st));
    }
    // out_proj + residual
    return linear_call(v->Y, C, B * T, 0, ow, ob, out, C, B * T, 0, B * T, C, C, DK_EPI_RES, nullptr, 0, 0, x, C, B * T, 0, st);
  }
};

extern "C" int dk_vae_decode(dk_vae* v, const float* latent, int32_t batch, int32_t latent_h, int32_t latent_w, float* image_f32,
                             uint8_t* image_u8, void* raw_bf16, void* workspace, size_t workspace_bytes, void* stream) {
  DK_REQUIRE(v && latent && workspace, "null argument");
  DK_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  Carver c(workspace, workspace_bytes);
  const size_t need_bytes = vae_carve(v, c, batch, latent_h, latent_w);
  DK_REQUIRE(need_bytes <= workspace_bytes, "workspace too small");
  const dk_vae_config& cf = v->cfg;
  VaeRun R{v, S_(stream), batch};
  hipStream_t st = R.st;
  // the flag region of the GEMM split workspace must be zero before the first launch (the kernels leave it zero)
  DK_CHECK_HIP(hipMemsetAsync((char*)v->GWS + dk_gemm_split_workspace_bytes() - 4096, 0, 4096, st));
  LinearWsScope ws_scope(v->GWS);
  DK_CHECK_HIP(hipMemsetAsync(v->ZERO, 0, 256, st));
  int H = latent_h, W = latent_w;
  const int Cm = cf.block_out_channels[cf.n_blocks - 1];
  DK_TRY(dk_launch_pad_channels(latent, v->LAT, (long)batch * H * W, cf.in_channels, 64, st));
  bf16_t *cur = v->bufA, *nxt = v->bufB;
  DK_TRY(R.conv(v->LAT, cur, H, W, 64, Cm, "conv_in", 0, nullptr, Cm));
  DK_TRY(R.resnet(cur, nxt, H, W, Cm, Cm, "mid_blocks.0")); std::swap(cur, nxt);
  DK_TRY(R.attention(cur, nxt, H, W, Cm, "mid_blocks.1")); std::swap(cur, nxt);
  DK_TRY(R.resnet(cur, nxt, H, W, Cm, Cm, "mid_blocks.2", true)); std::swap(cur, nxt);  // (the first up-block's norm1 reads its partials)
  int C = Cm;
  // up_blocks list index n-1 runs first (vae.py:379,393); index 0 has no upsample conv
  for (int j = cf.n_blocks - 1; j >= 0; --j) {
    const int Cout = cf.block_out_channels[j];
    for (int r = 0; r < cf.layers_per_block; ++r) {
      const std::string p = "up_blocks." + std::to_string(j) + ".resnets." + std::to_string(r);
      // (somebody normalises the output next: the following resnet, or conv_norm_out behind the last block)
      const bool gn_next = r + 1 < cf.layers_per_block || j == 0;
      DK_TRY(R.resnet(cur, nxt, H, W, r == 0 ? C : Cout, Cout, p, gn_next)); std::swap(cur, nxt);
    }
    C = Cout;
    if (j > 0) {
      H *= 2; W *= 2;
      R.n_part = 0;
      const std::string up = "up_blocks." + std::to_string(j) + ".upsample";
      if (g_dk_conv_halo != 3 && R.halo_stage(H, W, C, C)) {
        // the upsampling conv (vae.py:20-25,146) on the halo kernel too: nearest-x2 folded into the halo addressing, no norm in
        // front of it, and the statistics of its output for the next block's first GroupNorm
        ConvHaloParams c;
        memset(&c, 0, sizeof(c));
        c.x = cur; c.w = R.W(up + ".weight"); c.bias = R.W(up + ".bias"); c.y = nxt; c.stats_out = v->gn; c.G_out = cf.resnet_groups;
        c.B = batch; c.H = H; c.W = W; c.C = C; c.O = C; c.ups = 1; c.ldw = 9 * C; c.ldy = C;
        if (R.rc) return R.rc;
        DK_TRY(dk_launch_conv_halo(c, st));
        R.n_part = (H / 16) * (W / 16);
      } else {
        DK_TRY(R.conv(cur, nxt, H, W, C, C, up, 1, nullptr, C));
      }
      std::swap(cur, nxt);
    }
  }
  if (g_dk_conv_halo != 0 && H % 16 == 0 && W % 16 == 0 && C % 64 == 0 && cf.out_channels <= 4 && (size_t)H * W * C * 2 < (1ull << 31)) {
    // conv_norm_out -> silu -> conv_out -> clip / uint8 (vae.py:381,384,397-399; __init__.py:581-584,525-526) in one launch
    DK_TRY(R.gn_table(cur, (long)H * W, C, "conv_norm_out", v->ss0));
    ConvHaloParams c;
    memset(&c, 0, sizeof(c));
    c.x = cur; c.w = R.W("conv_out.weight"); c.bias = R.W("conv_out.bias"); c.gn_ss = v->ss0; c.gn_silu = 1;
    c.img = image_f32; c.u8 = image_u8; c.raw = raw_bf16 ? (bf16_t*)raw_bf16 : v->Y; c.out_channels = cf.out_channels;
    c.B = batch; c.H = H; c.W = W; c.C = C; c.O = cf.out_channels; c.ldw = 9 * C;
    if (R.rc) return R.rc;
    DK_TRY(dk_launch_conv_halo(c, st));
    return R.rc;
  }
  DK_TRY(R.gn(cur, v->T1, (long)H * W, C, "conv_norm_out", 1));
  bf16_t* raw = raw_bf16 ? (bf16_t*)raw_bf16 : v->Y;
  DK_TRY(R.conv(v->T1, raw, H, W, C, cf.out_channels, "conv_out", 0, nullptr, 4));
  DK_TRY(dk_launch_image_post(raw, 4, image_f32, image_u8, (long)batch * H * W, st));
  return R.rc;
}

// ---------------------------------------------------------------------------------------------
// VAE encoder engine (vae.py:404-467; img2img entry mlx/__init__.py:586-594).  Same handle type as the
// decoder: a dk_vae created with the encoder's config (in 3, out 32, layers_per_block 2) and bound to
// the encoder's module names (conv_in, down_blocks.{i}.resnets.{r}, down_blocks.{i}.downsample,
// mid_blocks.{0,1,2}, conv_norm_out, conv_out).
// ---------------------------------------------------------------------------------------------
static size_t vae_carve_encoder(dk_vae* v, Carver& c, int B, int H, int W) {
  const dk_vae_config& cf = v->cfg;
  size_t maxel = (size_t)B * H * W * 64;  // channel-padded input image
  {
    size_t h = H, w = W;
    int Cprev = cf.block_out_channels[0];
    for (int i = 0; i < cf.n_blocks; ++i) {
      const int Cout = cf.block_out_channels[i];
      const size_t e = (size_t)B * h * w * (size_t)(Cprev > Cout ? Cprev : Cout);
      if (e > maxel) maxel = e;
      if (i < cf.n_blocks - 1) { h /= 2; w /= 2; }
      Cprev = Cout;
    }
  }
  const size_t act = maxel * 2;
  v->bufA = (bf16_t*)c.take(act);
  v->bufB = (bf16_t*)c.take(act);
  v->T1 = (bf16_t*)c.take(act);
  v->Y = (bf16_t*)c.take(act);
  v->SC = (bf16_t*)c.take(act);
  v->LAT = (bf16_t*)c.take((size_t)B * H * W * 64 * 2);
  v->ZERO = (bf16_t*)c.take(256);
  const int Cm = cf.block_out_channels[cf.n_blocks - 1];
  const size_t tok = ((size_t)H >> (cf.n_blocks - 1)) * ((size_t)W >> (cf.n_blocks - 1));
  v->Qb = (bf16_t*)c.take((size_t)B * tok * Cm * 2);
  v->Kb = (bf16_t*)c.take((size_t)B * tok * Cm * 2);
  v->Vb = (bf16_t*)c.take((size_t)B * tok * Cm * 2);
  v->Vt = (bf16_t*)c.take((size_t)B * align_up(tok, 64) * Cm * 2);  // (the flash form transposes every image's V up front)
  // the materialised score matrix of the general path; a 512-channel mid block runs the flash kernel (attention512.hip) and needs none
  v->SCORES = (bf16_t*)c.take(Cm == 512 ? 0 : tok * align_up(tok, 64) * 2);
  {
    const size_t tiles = ((size_t)H / 16) * ((size_t)W / 16);
    const size_t npart = tiles > 1024 ? tiles : 1024;
    v->gn = (float*)c.take(((size_t)B * npart * 2 * cf.resnet_groups + (size_t)B * cf.resnet_groups * 2) * 4);
    int cmax = 0;
    for (int j = 0; j < cf.n_blocks; ++j) cmax = cf.block_out_channels[j] > cmax ? cf.block_out_channels[j] : cmax;
    v->ss0 = (float*)c.take((size_t)B * 2 * cmax * 4);
    v->ss1 = (float*)c.take((size_t)B * 2 * cmax * 4);
  }
  v->GWS = c.take(dk_gemm_split_workspace_bytes());
  return c.off;
}
extern "C" size_t dk_vae_encoder_workspace_bytes(const dk_vae* v, int32_t batch, int32_t image_h, int32_t image_w) {
  dk_vae tmp = *v;
  Carver c(nullptr, 0);
  return vae_carve_encoder(&tmp, c, batch, image_h, image_w) + 256;
}

extern "C" int dk_vae_encode(dk_vae* v, const float* image, int32_t batch, int32_t image_h, int32_t image_w, void* moments_bf16,
                             int32_t ldm, float* moments_f32, void* workspace, size_t workspace_bytes, void* stream) {
  DK_REQUIRE(v && image && workspace && (moments_bf16 || moments_f32), "null argument");
  DK_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  const dk_vae_config& cf = v->cfg;
  const int down = 1 << (cf.n_blocks - 1);
  DK_REQUIRE(image_h % down == 0 && image_w % down == 0, "image size must be a multiple of the VAE down-scaling factor");
  const int ldo = (cf.out_channels + 3) / 4 * 4;
  DK_REQUIRE(moments_bf16 == nullptr || ldm >= ldo, "moments leading dimension too small (multiple of 4 >= out_channels)");
  Carver c(workspace, workspace_bytes);
  const size_t need_bytes = vae_carve_encoder(v, c, batch, image_h, image_w);
  DK_REQUIRE(need_bytes <= workspace_bytes, "workspace too small");
  VaeRun R{v, S_(stream), batch};
  hipStream_t st = R.st;
  DK_CHECK_HIP(hipMemsetAsync((char*)v->GWS + dk_gemm_split_workspace_bytes() - 4096, 0, 4096, st));
  LinearWsScope ws_scope(v->GWS);
  DK_CHECK_HIP(hipMemsetAsync(v->ZERO, 0, 256, st));
  int H = image_h, W = image_w;
  DK_TRY(dk_launch_pad_channels(image, v->LAT, (long)batch * H * W, cf.in_channels, 64, st));
  bf16_t *cur = v->bufA, *nxt = v->bufB;
  int C = cf.block_out_channels[0];
  DK_TRY(R.conv(v->LAT, cur, H, W, 64, C, "conv_in", 0, nullptr, C));
  for (int i = 0; i < cf.n_blocks; ++i) {
    const int Cout = cf.block_out_channels[i];
    for (int r = 0; r < cf.layers_per_block; ++r) {
      const std::string p = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(r);
      DK_TRY(R.resnet(cur, nxt, H, W, r == 0 ? C : Cout, Cout, p)); std::swap(cur, nxt);
    }
    C = Cout;
    if (i < cf.n_blocks - 1) {  // pad (0,1),(0,1) + conv k3 s2 p0 (vae.py:141-143)
      H /= 2; W /= 2;
      DK_TRY(R.conv(cur, nxt, H, W, C, C, "down_blocks." + std::to_string(i) + ".downsample", 2, nullptr, C)); std::swap(cur, nxt);
    }
  }
  DK_TRY(R.resnet(cur, nxt, H, W, C, C, "mid_blocks.0")); std::swap(cur, nxt);
  DK_TRY(R.attention(cur, nxt, H, W, C, "mid_blocks.1")); std::swap(cur, nxt);
  DK_TRY(R.resnet(cur, nxt, H, W, C, C, "mid_blocks.2")); std::swap(cur, nxt);
  DK_TRY(R.gn(cur, v->T1, (long)H * W, C, "conv_norm_out", 1));
  bf16_t* mom = moments_bf16 ? (bf16_t*)moments_bf16 : v->Y;
  const int ld = moments_bf16 ? ldm : ldo;
  DK_TRY(R.conv(v->T1, mom, H, W, C, cf.out_channels, "conv_out", 0, nullptr, ld));
  if (moments_f32) DK_TRY(dk_launch_bf16_rows_to_f32(mom, ld, moments_f32, (long)batch * H * W, cf.out_channels, st));
  return R.rc;
}

extern "C" int dk_latent_sample_f32(const void* moments_bf16, int32_t ldm, const float* noise, float* latent, int64_t n_pixels,
                                    int32_t latent_channels, void* stream) {
  DK_REQUIRE(moments_bf16 && noise && latent && n_pixels > 0 && latent_channels > 0 && ldm >= 2 * latent_channels, "bad argument");
  return dk_launch_latent_sample((const bf16_t*)moments_bf16, ldm, noise, latent, (long)n_pixels, latent_channels, S_(stream));
}
