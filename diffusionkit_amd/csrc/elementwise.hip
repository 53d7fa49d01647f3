// HBM-bound kernels of the MMDiT step: adaptive LayerNorm modulation, per-head QK-RMSNorm + RoPE,
// timestep embedding / modulation-vector prep, RoPE table, patchify, and the fused
// x0-prediction + CFG + Euler update.  All bf16 traffic is moved as 16-byte vectors.
#include "dk_kernels.h"

// ---------------------------------------------------------------------------------------------
// AdaLN modulation: out = bf16( LN(x) * bf16(1 + scale[b]) + shift[b] )
// reference: affine_transform, python/src/diffusionkit/mlx/mmdit.py:958-972 (fused batch-1 form:
// mx.fast.layer_norm(x, 1 + scale, shift, eps)); LayerNorm :838-849 (no affine, eps 1e-6).
// One wave per row, row kept in registers (two-pass variance), 4 rows per workgroup.
// ---------------------------------------------------------------------------------------------
// One launch serves up to two jobs (the image and the text stream of a double block, mmdit.py:568-675: the text
// stream's 256 rows would otherwise run alone on the chip): blocks [0, blocks_a) belong to job a, the rest to job b.
struct LnJob {
  const bf16_t* x;
  bf16_t* out;
  const bf16_t *shift, *scale;
  int ldx, ldo, M, mod_stride, seg_len, x_seg_len, x_seg_stride;
};
// Every load of the row -- its NCH chunks, then the shift and scale chunks -- is issued unconditionally and back to back before
// the first use: one memory round trip per wave.  Guarding each chunk with `if (c < nchunks)` compiled to branch / load /
// s_waitcnt vmcnt(0) per chunk, twelve dependent round trips at h = 3072.  The loads and the store are buffer instructions
// whose resource spans exactly one row (h * 2 bytes from a scalar row base): lanes past the row end read zeros and their stores
// are dropped, one 32-bit offset register serves all three arrays, and the row stays in its packed bf16 form (unpacked again
// in each pass) so the kernel holds 12 * NCH data registers.
static_assert(sizeof(LnJob) % 8 == 0 && alignof(LnJob) == 8, "job b follows job a without padding in the kernarg segment");
template <int NCH>
__global__ __launch_bounds__(256) void dk_ln_modulate_kernel(LnJob ja, LnJob jb, int blocks_a, int h, float eps) {
  const bool first = (int)blockIdx.x < blocks_a;
  // one scalar base pointer into the kernarg segment (`first ? ja : jb` copies both jobs to scratch, see gemm256v3.hip)
  typedef const __attribute__((address_space(4))) LnJob karg_job_t;
  const __attribute__((address_space(4))) char* kbase = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  karg_job_t& j = *(karg_job_t*)(kbase + (first ? 0 : sizeof(LnJob)));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: row bases live in SGPRs)
  const int m = ((int)blockIdx.x - (first ? 0 : blocks_a)) * 4 + wave;
  if (m >= j.M) return;
  const size_t xrow = (size_t)((m / j.x_seg_len) * j.x_seg_stride + (m % j.x_seg_len)) * j.ldx;
  const size_t mod = (size_t)(m / j.seg_len) * j.mod_stride;
  const int nchunks = h >> 3;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(j.x + xrow), 0, h * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)(j.shift + mod), 0, h * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(j.scale + mod), 0, h * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(j.out + (size_t)m * j.ldo), 0, h * 2, 0x00020000);
  u32x4 raw[NCH], rs[NCH], rc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (lane + 64 * i) * 16, 0, 0);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    rs[i] = __builtin_amdgcn_raw_buffer_load_b128(rsh, (lane + 64 * i) * 16, 0, 0);
    rc[i] = __builtin_amdgcn_raw_buffer_load_b128(rsc, (lane + 64 * i) * 16, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);  // (without it the scheduler sinks the shift / scale loads below the two reductions)
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v0, v1;
      unpack2bf(raw[i][e], v0, v1);
      sum += v0 + v1;  // (zeros past the row end)
    }
  }
  const float mean = wave_sum(sum) / (float)h;
  __builtin_amdgcn_sched_barrier(0);  // (this one and the next two keep each pass's unpacked values out of the others' live ranges)
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const bool live = lane + 64 * i < nchunks;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v0, v1;
      asm volatile("" : "+v"(raw[i][e]));  // (opaque: else the unpacked row of pass 1 is kept live, 8 registers per chunk instead of 4)
      unpack2bf(raw[i][e], v0, v1);
      const float d0 = live ? v0 - mean : 0.f, d1 = live ? v1 - mean : 0.f;  // (masked before the product: sq += d * d stays one fma)
      sq += d0 * d0;
      sq += d1 * d1;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)h + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v0, v1, s0, s1, c0, c1;
      asm volatile("" : "+v"(raw[i][e]));  // (and v - mean of pass 2 likewise)
      unpack2bf(raw[i][e], v0, v1);
      unpack2bf(rs[i][e], s0, s1);
      unpack2bf(rc[i][e], c0, c1);
      const float y0 = (v0 - mean) * rstd * round_bf16(1.0f + c0) + s0;
      const float y1 = (v1 - mean) * rstd * round_bf16(1.0f + c1) + s1;
      o[e] = pack2bf(y0, y1);
    }
    __builtin_amdgcn_raw_buffer_store_b128(o, ro, (lane + 64 * i) * 16, 0, 0);
  }
}

static int launch_ln_jobs(const LnJob& a, const LnJob& b, int h, float eps, hipStream_t stream) {
  DK_REQUIRE(h % 8 == 0 && h <= 4096, "hidden size must be a multiple of 8 and <= 4096");
  for (const LnJob* j : {&a, &b})
    DK_REQUIRE(j->M == 0 || (j->ldx % 8 == 0 && j->ldo % 8 == 0 && j->mod_stride % 8 == 0), "strides must keep 16-byte alignment");
  const int blocks_a = (a.M + 3) / 4, blocks_b = (b.M + 3) / 4;
  dim3 grid(blocks_a + blocks_b), block(256);
  const int nch = (h / 8 + 63) / 64;
#define LN_CASE(N)                                                                                         \
  case N:                                                                                                  \
    hipLaunchKernelGGL(dk_ln_modulate_kernel<N>, grid, block, 0, stream, a, b, blocks_a, h, eps);          \
    break;
  switch (nch) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    default: DK_REQUIRE(false, "unsupported hidden size");
  }
#undef LN_CASE
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
static LnJob ln_job(const bf16_t* x, int ldx, bf16_t* out, int ldo, int M, const bf16_t* shift, const bf16_t* scale, int mod_stride,
                    int seg_len, int x_seg_len, int x_seg_stride) {
  LnJob j;
  j.x = x; j.out = out; j.shift = shift; j.scale = scale; j.ldx = ldx; j.ldo = ldo; j.M = M; j.mod_stride = mod_stride;
  j.seg_len = seg_len; j.x_seg_len = x_seg_len; j.x_seg_stride = x_seg_stride;
  return j;
}
int dk_launch_ln_modulate(const bf16_t* x, int ldx, bf16_t* out, int ldo, int M, int h, const bf16_t* shift,
                          const bf16_t* scale, int mod_stride, int seg_len, int x_seg_len, int x_seg_stride, float eps,
                          hipStream_t stream) {
  LnJob none = ln_job(nullptr, 8, nullptr, 8, 0, nullptr, nullptr, 8, 1, 1, 0);
  return launch_ln_jobs(ln_job(x, ldx, out, ldo, M, shift, scale, mod_stride, seg_len, x_seg_len, x_seg_stride), none, h, eps, stream);
}
// two row sets (image / text stream) of the same hidden size in one launch
int dk_launch_ln_modulate2(const bf16_t* x0, bf16_t* out0, int M0, const bf16_t* shift0, const bf16_t* scale0, int seg0, const bf16_t* x1,
                           bf16_t* out1, int M1, const bf16_t* shift1, const bf16_t* scale1, int seg1, int ldx, int ldo, int h,
                           int mod_stride, int x_seg_stride, float eps, hipStream_t stream) {
  return launch_ln_jobs(ln_job(x0, ldx, out0, ldo, M0, shift0, scale0, mod_stride, seg0, seg0, x_seg_stride),
                        ln_job(x1, ldx, out1, ldo, M1, shift1, scale1, mod_stride, seg1, seg1, x_seg_stride), h, eps, stream);
}

// ---------------------------------------------------------------------------------------------
// In-place per-head RMSNorm (learned weight) followed by RoPE on the q and k column groups of a
// token-major QKV buffer.  reference: QKNorm python/src/diffusionkit/mlx/mmdit.py:754-764
// (nn.RMSNorm eps 1e-6, one rounding), RoPE.apply :934-942 (adjacent pairs, fp32, one rounding).
// D/8 lanes cooperate on one (row, head, q|k) item; rope == nullptr skips the rotation,
// qw == nullptr skips the norm.
// ---------------------------------------------------------------------------------------------
struct QkJob {  // (two jobs per launch like LnJob)
  bf16_t* qkv;
  const bf16_t *qw, *kw;
  int rows, row_seg_len, row_seg_stride, pos_off;
};
template <int D>
__global__ __launch_bounds__(256) void dk_qk_norm_rope_kernel(QkJob ja, QkJob jb, int blocks_a, int ld, int q_off, int k_off, int H, float eps,
                                                              const float* __restrict__ rope, int k_only) {
  const bool first = (int)blockIdx.x < blocks_a;
  const QkJob& jj = first ? ja : jb;
  bf16_t* __restrict__ qkv = jj.qkv;
  const bf16_t* __restrict__ qw = jj.qw;
  const bf16_t* __restrict__ kw = jj.kw;
  const int rows = jj.rows, row_seg_len = jj.row_seg_len, row_seg_stride = jj.row_seg_stride, pos_off = jj.pos_off;
  constexpr int LPI = D / 8;  // lanes per item
  // 32-bit index arithmetic (rows * 2H * LPI < 2^31, checked by the launcher): 64-bit divisions cost more VALU
  // work than the 16 bytes this lane moves
  const unsigned gid = (blockIdx.x - (first ? 0u : (unsigned)blocks_a)) * 256u + threadIdx.x;
  const unsigned item = gid / LPI;
  const int sub = (int)(gid % LPI);
  // k_only: the queries are normalised / rotated inside the attention kernel's Q load, this pass touches the keys
  const unsigned per_row = (k_only ? 1u : 2u) * (unsigned)H;
  const unsigned nitems = (unsigned)rows * per_row;
  const bool active = item < nitems;
  const unsigned it = active ? item : nitems - 1;
  const int m = (int)(it / per_row);
  const int rem = (int)(it - (unsigned)m * per_row);
  const int which = k_only ? 1 : (rem >= H ? 1 : 0), head = k_only ? rem : rem - which * H;
  const int seg = m / row_seg_len, pos_in = m % row_seg_len;
  bf16_t* ptr = qkv + (size_t)(seg * row_seg_stride + pos_in) * ld + (which ? k_off : q_off) + head * D + sub * 8;
  const u32x4 raw = *(const u32x4*)ptr;
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) unpack2bf(raw[e], v[2 * e], v[2 * e + 1]);
  const bf16_t* w = which ? kw : qw;
  if (w != nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
#pragma unroll
    for (int o = LPI / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float r = rsqrtf(ss / (float)D + eps);
    const u32x4 wr = *(const u32x4*)(w + sub * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float w0, w1;
      unpack2bf(wr[e], w0, w1);
      v[2 * e] = round_bf16(v[2 * e] * r * w0);
      v[2 * e + 1] = round_bf16(v[2 * e + 1] * r * w1);
    }
  }
  if (rope != nullptr) {
    const float* tab = rope + ((size_t)(pos_off + pos_in) * (D / 2) + sub * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = tab[2 * e], s = tab[2 * e + 1];
      const float xe = v[2 * e], xo = v[2 * e + 1];
      v[2 * e] = c * xe - s * xo;
      v[2 * e + 1] = s * xe + c * xo;
    }
  }
  if (active) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
    *(u32x4*)ptr = o;
  }
}

static int launch_qk_jobs(const QkJob& a, const QkJob& b, int ld, int q_off, int k_off, int H, int D, float eps, const float* rope,
                          hipStream_t stream, int k_only = 0) {
  DK_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128");
  DK_REQUIRE(ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0, "alignment");
  const int per_row = (k_only ? 1 : 2) * H;
  const long ta = (long)a.rows * per_row * (D / 8), tb = (long)b.rows * per_row * (D / 8);
  DK_REQUIRE(ta + tb < (1L << 31) - 512, "qk_norm_rope: rows * 2H * D/8 must stay below 2^31");
  const int blocks_a = (int)((ta + 255) / 256), blocks_b = (int)((tb + 255) / 256);
  dim3 grid(blocks_a + blocks_b), block(256);
  if (D == 128)
    hipLaunchKernelGGL(dk_qk_norm_rope_kernel<128>, grid, block, 0, stream, a, b, blocks_a, ld, q_off, k_off, H, eps, rope, k_only);
  else
    hipLaunchKernelGGL(dk_qk_norm_rope_kernel<64>, grid, block, 0, stream, a, b, blocks_a, ld, q_off, k_off, H, eps, rope, k_only);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
int dk_launch_qk_norm_rope(bf16_t* qkv, int ld, int q_off, int k_off, int rows, int H, int D, const bf16_t* qw,
                           const bf16_t* kw, float eps, const float* rope, int row_seg_len, int row_seg_stride, int pos_off,
                           int S_pos, hipStream_t stream, int k_only) {
  (void)S_pos;
  if (qw == nullptr && rope == nullptr) return 0;
  QkJob a{qkv, qw, kw, rows, row_seg_len, row_seg_stride, pos_off}, none{nullptr, nullptr, nullptr, 0, 1, 0, 0};
  return launch_qk_jobs(a, none, ld, q_off, k_off, H, D, eps, rope, stream, k_only);
}
// the two streams of a double block (same buffer geometry, own weights / row segments / positions) in one launch
int dk_launch_qk_norm_rope2(bf16_t* qkv0, int rows0, const bf16_t* qw0, const bf16_t* kw0, int seg0, int pos0, bf16_t* qkv1, int rows1,
                            const bf16_t* qw1, const bf16_t* kw1, int seg1, int pos1, int ld, int q_off, int k_off, int H, int D,
                            float eps, const float* rope, int row_seg_stride, hipStream_t stream, int k_only) {
  if (qw0 == nullptr && rope == nullptr) return 0;
  QkJob a{qkv0, qw0, kw0, rows0, seg0, row_seg_stride, pos0}, b{qkv1, qw1, kw1, rows1, seg1, row_seg_stride, pos1};
  return launch_qk_jobs(a, b, ld, q_off, k_off, H, D, eps, rope, stream, k_only);
}

// ---------------------------------------------------------------------------------------------
// small elementwise helpers
// ---------------------------------------------------------------------------------------------
__global__ void dk_silu_kernel(const bf16_t* x, bf16_t* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = f2bf(silu_f(bf2f(x[i])));
}
int dk_launch_silu(const bf16_t* x, bf16_t* y, long n, hipStream_t stream) {
  hipLaunchKernelGGL(dk_silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// y[r, :] = bf16(a[r, :] + b[r / (rows / b_rows) , :])   (used for vec = y_embed[b] + t_embed[step])
__global__ void dk_add_kernel(const bf16_t* a, const bf16_t* b, int a_rows, int b_rows, bf16_t* y, int rows, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  // a is indexed by r % a_rows (batch row), b by r / a_rows (timestep row)
  y[i] = f2bf(bf2f(a[(size_t)(r % a_rows) * cols + c]) + bf2f(b[(size_t)min(r / a_rows, b_rows - 1) * cols + c]));
}
int dk_launch_add(const bf16_t* a, const bf16_t* b, int b_rows, bf16_t* y, int rows, int cols, hipStream_t stream) {
  const int a_rows = rows / b_rows;
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(dk_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, a_rows, b_rows, y, rows, cols);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

__device__ __forceinline__ float round_to(float v, int dt) {
  if (dt == 0) return round_bf16(v);
  if (dt == 1) return (float)(_Float16)v;
  return v;
}
// Sinusoidal timestep embedding evaluated in the reference's config.dtype
// (python/src/diffusionkit/mlx/mmdit.py:379-389, quirk Q2): out[i, :] = [cos(args), sin(args)].
__global__ void dk_timestep_embedding_kernel(const float* t, int n, int dim, float max_period, int dt, bf16_t* out) {
  const int i = blockIdx.x, j = threadIdx.x;
  const int half = dim / 2;
  if (i >= n || j >= half) return;
  const float ar = round_to((float)j, dt);
  const float freq = round_to(expf(-logf(max_period) * ar / (float)half), dt);
  const float arg = round_to(round_to(t[i], dt) * freq, dt);
  out[(size_t)i * dim + j] = f2bf(round_to(cosf(arg), dt));
  out[(size_t)i * dim + half + j] = f2bf(round_to(sinf(arg), dt));
}
int dk_launch_timestep_embedding(const float* t, int n, int rep, int dim, float max_period, int embed_dtype, bf16_t* out,
                                 hipStream_t stream) {
  (void)rep;
  DK_REQUIRE(dim % 2 == 0 && dim / 2 <= 1024, "frequency_embed_dim");
  hipLaunchKernelGGL(dk_timestep_embedding_kernel, dim3(n), dim3(dim / 2), 0, stream, t, n, dim, max_period, embed_dtype, out);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// RoPE cos/sin table [S, D/2, 2] for the joint [text, image] sequence
// (python/src/diffusionkit/mlx/mmdit.py:865-911, quirk Q14).
struct RopeAxes { int dim[4]; int n; };
__global__ void dk_rope_table_kernel(float* table, int S_txt, int gh, int gw, RopeAxes ax, float theta, int half) {
  const int s = blockIdx.x, pi = threadIdx.x;
  if (pi >= half) return;
  float pos[4] = {0.f, 0.f, 0.f, 0.f};
  if (s >= S_txt) {
    const int idx = s - S_txt;
    pos[1] = (float)(idx / gw);
    pos[2] = (float)(idx % gw);
  }
  int a = 0, local = pi;
  while (a < ax.n - 1 && local >= ax.dim[a] / 2) {
    local -= ax.dim[a] / 2;
    ++a;
  }
  const float scale = (float)(2 * local) / (float)ax.dim[a];
  const float omega = 1.0f / powf(theta, scale);
  const float ang = pos[a] * omega;
  table[((size_t)s * half + pi) * 2 + 0] = cosf(ang);
  table[((size_t)s * half + pi) * 2 + 1] = sinf(ang);
}
int dk_launch_rope_table(float* table, int S_txt, int gh, int gw, const int* axes, int n_axes, float theta, hipStream_t stream) {
  DK_REQUIRE(n_axes >= 1 && n_axes <= 4, "rope axes");
  RopeAxes ax;
  int half = 0;
  for (int i = 0; i < 4; ++i) ax.dim[i] = 0;
  for (int i = 0; i < n_axes; ++i) {
    ax.dim[i] = axes[i];
    half += axes[i] / 2;
  }
  ax.n = n_axes;
  const int S = S_txt + gh * gw;
  hipLaunchKernelGGL(dk_rope_table_kernel, dim3(S), dim3(((half + 63) / 64) * 64), 0, stream, table, S_txt, gh, gw, ax, theta, half);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void dk_f32_to_bf16_kernel(const float* x, bf16_t* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = f2bf(x[i]);
}
int dk_launch_f32_to_bf16(const float* x, bf16_t* y, long n, hipStream_t stream) {
  hipLaunchKernelGGL(dk_f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
__global__ void dk_affine_f32_kernel(const float* x, float* y, long n, float a, float b) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * a + b;
}
int dk_launch_affine_f32(const float* x, float* y, long n, float a, float b, hipStream_t stream) {
  hipLaunchKernelGGL(dk_affine_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n, a, b);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Patchify: latent [n_img, Hl, Wl, C] fp32 -> tokens [n_img*dup, S_i, p*p*C] bf16.
// reshape_order=1: FLUX space-to-depth, features (c, ph, pw) (mmdit.py:292-300);
// reshape_order=0: SD3 strided conv taps, features (ph, pw, c) (mmdit.py:285-290).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int patch_feature(int c, int ph, int pw, int C, int p, int reshape_order) {
  return reshape_order ? (c * p * p + ph * p + pw) : ((ph * p + pw) * C + c);
}
__global__ void dk_latent_to_tokens_kernel(const float* x, bf16_t* tok, int n_img, int dup, int Hl, int Wl, int C, int p,
                                           int reshape_order) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_img = (long)Hl * Wl * C;
  if (i >= per_img * n_img) return;
  const int img = (int)(i / per_img);
  long r = i % per_img;
  const int c = (int)(r % C);
  r /= C;
  const int xx = (int)(r % Wl), yy = (int)(r / Wl);
  const int gw = Wl / p, S_i = (Hl / p) * gw, F = p * p * C;
  const int t = (yy / p) * gw + (xx / p);
  const int f = patch_feature(c, yy % p, xx % p, C, p, reshape_order);
  const bf16_t v = f2bf(x[i]);
  for (int d = 0; d < dup; ++d) tok[((size_t)(d * n_img + img) * S_i + t) * F + f] = v;
}
int dk_launch_latent_to_tokens(const float* x, bf16_t* tok, int n_img, int dup, int Hl, int Wl, int C, int p, int reshape_order,
                               hipStream_t stream) {
  const long n = (long)n_img * Hl * Wl * C;
  hipLaunchKernelGGL(dk_latent_to_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, tok, n_img, dup, Hl,
                     Wl, C, p, reshape_order);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Fused step-loop tail: unpatchify + x0 prediction + CFG + Euler update + re-patchify.
// reference: CFGDenoiser.__call__ python/src/diffusionkit/mlx/__init__.py:691-719
// (x_bf16 - out*sigma in fp32; neg + w*(text - neg)), to_d :756, sample_euler :778-781;
// unpack/unpatchify mmdit.py:304-321, 975-988.  x stays fp32 (quirk Q6).
// ---------------------------------------------------------------------------------------------
__global__ void dk_euler_step_kernel(float* x, const bf16_t* model_out, int ld_out, bf16_t* tok, int n_img, int cfg_on, int Hl,
                                     int Wl, int C, int p, int reshape_order, float sigma, float sigma_next, float cfg_weight) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_img = (long)Hl * Wl * C;
  if (i >= per_img * n_img) return;
  const int img = (int)(i / per_img);
  long r = i % per_img;
  const int c = (int)(r % C);
  r /= C;
  const int xx = (int)(r % Wl), yy = (int)(r / Wl);
  const int gw = Wl / p, S_i = (Hl / p) * gw, F = p * p * C;
  const int t = (yy / p) * gw + (xx / p);
  const int f = patch_feature(c, yy % p, xx % p, C, p, reshape_order);
  const float xv = x[i];
  const float xb = round_bf16(xv);  // the value the denoiser saw
  const float o_text = bf2f(model_out[((size_t)img * S_i + t) * ld_out + f]);
  float den = xb - o_text * sigma;
  if (cfg_on) {
    const float o_neg = bf2f(model_out[((size_t)(n_img + img) * S_i + t) * ld_out + f]);
    const float den_neg = xb - o_neg * sigma;
    den = den_neg + cfg_weight * (den - den_neg);
  }
  const float d = (xv - den) / sigma;
  const float xn = xv + d * (sigma_next - sigma);
  x[i] = xn;
  const bf16_t nb = f2bf(xn);
  tok[((size_t)img * S_i + t) * F + f] = nb;
  if (cfg_on) tok[((size_t)(n_img + img) * S_i + t) * F + f] = nb;
}
int dk_launch_euler_step(float* x, const bf16_t* model_out, int ld_out, bf16_t* tok, int n_img, int cfg_on, int Hl, int Wl, int C,
                         int p, int reshape_order, float sigma, float sigma_next, float cfg_weight, hipStream_t stream) {
  const long n = (long)n_img * Hl * Wl * C;
  hipLaunchKernelGGL(dk_euler_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, model_out, ld_out, tok,
                     n_img, cfg_on, Hl, Wl, C, p, reshape_order, sigma, sigma_next, cfg_weight);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
