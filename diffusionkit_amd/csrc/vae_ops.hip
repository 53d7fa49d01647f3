// HBM-bound kernels of the latent-decode VAE (NHWC bf16): GroupNorm statistics / apply(+SiLU),
// row softmax for the single-head mid-block attention, 2-D transpose, channel padding and the
// final clip + uint8 conversion.
// reference: python/src/diffusionkit/mlx/vae.py:28-57 (Attention), :60-101 (ResnetBlock2D),
// :381,397-399 (conv_norm_out + silu), python/src/diffusionkit/mlx/__init__.py:581-584,525-526.
#include "dk_kernels.h"

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics (nn.GroupNorm(pytorch_compatible=True): per (batch, group) mean / variance
// over H*W*(C/G) elements, fp32).  Pass 1: each workgroup reduces a slab of pixels into per-group
// (sum, sumsq) partials; pass 2 combines the partials in double precision.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dk_gn_partial_kernel(const bf16_t* __restrict__ x, long HW, int C, int G, float* __restrict__ partial,
                                                            int nchunk) {
  // per-thread (sum, sumsq) of 8 channels, combined in a FIXED order (no atomics: the statistics,
  // and with them the decoded image, must be bit-reproducible for a fixed seed)
  __shared__ float red[256 * 16];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int tpp = C / 8;           // threads per pixel
  const int ppi = 256 / tpp;       // pixels per iteration
  const int cg = tid % tpp;        // channel chunk of this thread
  const int pl = tid / tpp;
  const long per = (HW + nchunk - 1) / nchunk;
  const long p0 = (long)chunk * per, p1 = min(HW, p0 + per);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  const bf16_t* base = x + (size_t)b * HW * C + cg * 8;
  auto accumulate = [&](const u32x4 raw) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a0, a1;
      unpack2bf(raw[e], a0, a1);
      s[2 * e] += a0; q[2 * e] += a0 * a0;
      s[2 * e + 1] += a1; q[2 * e + 1] += a1 * a1;
    }
  };
  long pix = p0 + pl;
  for (; pix + 3 * ppi < p1; pix += 4 * ppi) {  // four independent 16-byte loads in flight per lane
    const u32x4 r0 = *(const u32x4*)(base + (size_t)pix * C);
    const u32x4 r1 = *(const u32x4*)(base + (size_t)(pix + ppi) * C);
    const u32x4 r2 = *(const u32x4*)(base + (size_t)(pix + 2 * ppi) * C);
    const u32x4 r3 = *(const u32x4*)(base + (size_t)(pix + 3 * ppi) * C);
    accumulate(r0);
    accumulate(r1);
    accumulate(r2);
    accumulate(r3);
  }
  for (; pix < p1; pix += ppi) accumulate(*(const u32x4*)(base + (size_t)pix * C));
  // red[stat][pl][channel]: channel = cg*8 + e
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[(0 * ppi + pl) * C + cg * 8 + e] = s[e];
    red[(1 * ppi + pl) * C + cg * 8 + e] = q[e];
  }
  __syncthreads();
  if (tid < 2 * G) {
    const int g = tid >> 1, stat = tid & 1, cpg = C / G;
    float acc = 0.f;
    for (int pp = 0; pp < ppi; ++pp)
      for (int c = 0; c < cpg; ++c) acc += red[(stat * ppi + pp) * C + g * cpg + c];
    partial[((size_t)b * nchunk + chunk) * 2 * G + tid] = acc;
  }
}
// one workgroup of 256 threads per (batch, group): threads stride over the chunk partials (up to one per 16 x 16 output tile when a
// fused conv produced them: 4096 at 1024 x 1024), then a fixed-order tree (double) -- wave shuffles, four wave results through LDS
__global__ __launch_bounds__(256) void dk_gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int G, double count, float eps,
                                                             float* __restrict__ mean_rstd, const bf16_t* __restrict__ gamma,
                                                             const bf16_t* __restrict__ beta, int C, float* __restrict__ scale_shift) {
  __shared__ double red[8];
  const int b = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0, q = 0.0;
  for (int c = tid; c < nchunk; c += 256) {
    s += (double)partial[((size_t)b * nchunk + c) * 2 * G + 2 * g];
    q += (double)partial[((size_t)b * nchunk + c) * 2 * G + 2 * g + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) red[wave] = s, red[4 + wave] = q;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  q = (red[4] + red[5]) + (red[6] + red[7]);
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
  if (tid == 0) {
    mean_rstd[((size_t)b * G + g) * 2] = mf;
    mean_rstd[((size_t)b * G + g) * 2 + 1] = rf;
  }
  if (scale_shift != nullptr) {
    // the per-channel pair dk_gn_apply_kernel builds in LDS (same fp32 expressions), for consumers that apply the norm on load
    const int cpg = C / G;
    for (int c = tid; c < cpg; c += 256) {
      const int ch = g * cpg + c;
      const float sc = rf * bf2f(gamma[ch]);
      scale_shift[(size_t)b * 2 * C + ch] = sc;
      scale_shift[(size_t)b * 2 * C + C + ch] = bf2f(beta[ch]) - mf * sc;
    }
  }
}
int dk_launch_groupnorm_finalize(const float* partial, int nchunk, int B, int G, double count, float eps, float* mean_rstd,
                                 const bf16_t* gamma, const bf16_t* beta, int C, float* scale_shift, hipStream_t stream) {
  DK_REQUIRE(G >= 1 && nchunk >= 1 && (scale_shift == nullptr || (gamma && beta && C % G == 0)), "groupnorm finalize arguments");
  hipLaunchKernelGGL(dk_gn_finalize_kernel, dim3(B * G), dim3(256), 0, stream, partial, nchunk, G, count, eps, mean_rstd, gamma, beta, C,
                     scale_shift);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
// first pass alone: per (batch row, chunk, group) partial (sum, sum of squares)
int dk_launch_groupnorm_partials(const bf16_t* x, int B, long HW, int C, int G, float* partial, int nchunk, hipStream_t stream) {
  DK_REQUIRE(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "channels must be 8 * 2^k <= 2048");
  DK_REQUIRE(G <= 64 && C % G == 0, "groups");
  DK_REQUIRE(nchunk >= 1, "nchunk");
  hipLaunchKernelGGL(dk_gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, stream, x, HW, C, G, partial, nchunk);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
int dk_launch_groupnorm_stats(const bf16_t* x, int B, long HW, int C, int G, float* partial, int nchunk, float* mean_rstd,
                              float eps, hipStream_t stream) {
  const int rc = dk_launch_groupnorm_partials(x, B, HW, C, G, partial, nchunk, stream);
  if (rc) return rc;
  return dk_launch_groupnorm_finalize(partial, nchunk, B, G, (double)HW * (double)(C / G), eps, mean_rstd, nullptr, nullptr, C, nullptr, stream);
}

// y = [silu]( bf16( (x - mean) * rstd * gamma + beta ) ), evaluated as x * scale[c] + shift[c] with the per-channel
// fp32 pair (scale = rstd * gamma, shift = beta - mean * scale) built once per workgroup in LDS: the kernel is a
// pure stream (one 16-byte load and store per 8 values, no integer divisions in the element loop)
#define GN_APPLY_U 4
__global__ __launch_bounds__(256) void dk_gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long HW, int C, int G,
                                                          const float* __restrict__ mean_rstd, const bf16_t* __restrict__ gamma,
                                                          const bf16_t* __restrict__ beta, int do_silu) {
  __shared__ float tab[2 * 2048];  // scale[C] | shift[C]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  for (int c = tid; c < C; c += 256) {
    const float* mr = mean_rstd + ((size_t)b * G + c / cpg) * 2;
    const float sc = mr[1] * bf2f(gamma[c]);
    tab[c] = sc;
    tab[C + c] = bf2f(beta[c]) - mr[0] * sc;
  }
  __syncthreads();
  const unsigned cpr = (unsigned)(C / 8);                      // 16-byte chunks per pixel
  const unsigned total = (unsigned)HW * cpr;                   // chunks of this batch row (< 2^31, checked by the launcher)
  const bf16_t* xb = x + (size_t)b * HW * C;
  bf16_t* yb = y + (size_t)b * HW * C;
  const unsigned i0 = blockIdx.x * (256u * GN_APPLY_U) + tid;
  u32x4 raw[GN_APPLY_U];
#pragma unroll
  for (int u = 0; u < GN_APPLY_U; ++u) {
    const unsigned i = i0 + u * 256u;
    if (i < total) raw[u] = *(const u32x4*)(xb + (size_t)i * 8);
  }
#pragma unroll
  for (int u = 0; u < GN_APPLY_U; ++u) {
    const unsigned i = i0 + u * 256u;
    if (i >= total) continue;
    const unsigned c0 = (i & (cpr - 1)) * 8;  // cpr is a power of two (checked by the launcher)
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a0, a1;
      unpack2bf(raw[u][e], a0, a1);
      float y0 = round_bf16(a0 * tab[c0 + 2 * e] + tab[C + c0 + 2 * e]);
      float y1 = round_bf16(a1 * tab[c0 + 2 * e + 1] + tab[C + c0 + 2 * e + 1]);
      if (do_silu) {
        y0 = silu_f(y0);
        y1 = silu_f(y1);
      }
      o[e] = pack2bf(y0, y1);
    }
    *(u32x4*)(yb + (size_t)i * 8) = o;
  }
}
int dk_launch_groupnorm_apply(const bf16_t* x, bf16_t* y, int B, long HW, int C, int G, const float* mean_rstd,
                              const bf16_t* gamma, const bf16_t* beta, int do_silu, hipStream_t stream) {
  DK_REQUIRE(C <= 2048 && C % 8 == 0 && ((C / 8) & (C / 8 - 1)) == 0, "channels must be 8 * 2^k <= 2048");
  DK_REQUIRE((double)HW * (C / 8) < 2147483648.0, "one batch row must hold < 2^31 16-byte chunks");
  const long total = HW * (C / 8);
  const long per_wg = 256L * GN_APPLY_U;
  hipLaunchKernelGGL(dk_gn_apply_kernel, dim3((unsigned)((total + per_wg - 1) / per_wg), B), dim3(256), 0, stream, x, y, HW, C, G,
                     mean_rstd, gamma, beta, do_silu);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// In-place row softmax on a bf16 matrix (fp32 math), one workgroup per row, row cached in
// registers (ld <= 16384).  Columns [cols, ld) -- the padding up to the K-tile multiple the P.V GEMM needs -- are
// ignored on input (they may hold anything) and written as zeros.  reference: mx.softmax(scores) in vae.py:51.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dk_softmax_rows_kernel(bf16_t* __restrict__ x, int cols, int ld) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16_t* row = x + (size_t)blockIdx.x * ld;
  const int nchunks = ld >> 3;
  float v[8][8];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    if (c < nchunks) {
      const u32x4 raw = *(const u32x4*)(row + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) unpack2bf(raw[e], v[i][2 * e], v[i][2 * e + 1]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (c * 8 + e >= cols) v[i][e] = -3.0e38f;  // padding: by index, not by value
        mx = fmaxf(mx, v[i][e]);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    if (c < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = c * 8 + e < cols ? __expf(v[i][e] - mx) : 0.f;
        sum += v[i][e];
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    if (c < nchunks) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[i][2 * e] * inv, v[i][2 * e + 1] * inv);
      *(u32x4*)(row + c * 8) = o;
    }
  }
}
// the same for rows that do not fit the register cache (more than 16384 columns: latents beyond 128 x 128): three passes
// over the row through L2
__global__ __launch_bounds__(256) void dk_softmax_long_rows_kernel(bf16_t* __restrict__ x, int cols, int ld) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16_t* row = x + (size_t)blockIdx.x * ld;
  float mx = -3.0e38f;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, bf2f(row[c]));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += __expf(bf2f(row[c]) - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < ld; c += 256) row[c] = c < cols ? f2bf(__expf(bf2f(row[c]) - mx) * inv) : (bf16_t)0;
}
int dk_launch_softmax_rows(bf16_t* x, int rows, int cols, int ld, hipStream_t stream) {
  DK_REQUIRE(cols >= 1 && cols <= ld && ld % 8 == 0, "softmax: 1 <= cols <= ld, ld a multiple of 8");
  if (ld > 16384)
    hipLaunchKernelGGL(dk_softmax_long_rows_kernel, dim3(rows), dim3(256), 0, stream, x, cols, ld);
  else
    hipLaunchKernelGGL(dk_softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, cols, ld);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// y[c, r] = x[r, c] for r < R, zero for R <= r < ldy (row stride of y)
__global__ void dk_transpose_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int R, int Cc, int ldy) {
  __shared__ bf16_t tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    if (c < Cc) tile[j][tx] = r < R ? x[(size_t)r * Cc + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (r < ldy && c < Cc) y[(size_t)c * ldy + r] = tile[tx][j];
  }
}
int dk_launch_transpose(const bf16_t* x, bf16_t* y, int R, int Cc, hipStream_t stream, int ldy) {
  if (ldy <= 0) ldy = R;
  DK_REQUIRE(ldy >= R, "transpose: output row stride must cover the rows");
  hipLaunchKernelGGL(dk_transpose_kernel, dim3((Cc + 31) / 32, (ldy + 31) / 32), dim3(256), 0, stream, x, y, R, Cc, ldy);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// fp32 [npix, C] -> bf16 [npix, Cpad] with zero channel padding (latents enter the conv_in GEMM)
__global__ void dk_pad_channels_kernel(const float* x, bf16_t* y, long npix, int C, int Cpad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * Cpad) return;
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  y[i] = c < C ? f2bf(x[pix * C + c]) : (bf16_t)0;
}
int dk_launch_pad_channels(const float* x, bf16_t* y, long npix, int C, int Cpad, hipStream_t stream) {
  const long n = npix * Cpad;
  hipLaunchKernelGGL(dk_pad_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, npix, C, Cpad);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// decode_latents_to_image tail (mlx/__init__.py:581-584, 525-526) evaluated in the activation
// dtype like the reference: img = clip(bf16(x/2 + 0.5), 0, 1); u8 = uint8(bf16(img * 255)).
__global__ void dk_image_post_kernel(const bf16_t* x, int ldx, float* img, unsigned char* u8, long npix) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * 3) return;
  const int c = (int)(i % 3);
  const long pix = i / 3;
  float v = round_bf16(bf2f(x[pix * ldx + c]) * 0.5f + 0.5f);
  v = fminf(fmaxf(v, 0.f), 1.f);
  if (img) img[i] = v;
  if (u8) u8[i] = (unsigned char)round_bf16(v * 255.0f);
}
int dk_launch_image_post(const bf16_t* x, int ldx, float* img, unsigned char* u8, long npix, hipStream_t stream) {
  const long n = npix * 3;
  hipLaunchKernelGGL(dk_image_post_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ldx, img, u8, npix);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// encode_image_to_latents tail (mlx/__init__.py:586-594): moments [npix, ldm >= 2L] (mean | logvar) ->
// latent = mean + exp(0.5 * clip(logvar, -30, 20)) * noise, fp32 like the reference (its encoder output is fp32)
__global__ void dk_latent_sample_kernel(const bf16_t* mom, int ldm, const float* noise, float* out, long npix, int L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * L) return;
  const int c = (int)(i % L);
  const long pix = i / L;
  const float mean = bf2f(mom[pix * ldm + c]);
  const float logvar = fminf(fmaxf(bf2f(mom[pix * ldm + L + c]), -30.0f), 20.0f);
  out[i] = mean + expf(0.5f * logvar) * noise[i];
}
int dk_launch_latent_sample(const bf16_t* mom, int ldm, const float* noise, float* out, long npix, int L, hipStream_t stream) {
  const long n = npix * L;
  hipLaunchKernelGGL(dk_latent_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mom, ldm, noise, out, npix, L);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// bf16 [npix, ldx] -> fp32 [npix, C]
__global__ void dk_bf16_rows_to_f32_kernel(const bf16_t* x, int ldx, float* y, long npix, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * C) return;
  y[i] = bf2f(x[(i / C) * ldx + (i % C)]);
}
int dk_launch_bf16_rows_to_f32(const bf16_t* x, int ldx, float* y, long npix, int C, hipStream_t stream) {
  const long n = npix * C;
  hipLaunchKernelGGL(dk_bf16_rows_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ldx, y, npix, C);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
