// HBM-bound kernels of the text-conditioning step in front of the hot path: CLIP-L / CLIP-G and the
// T5-XXL encoder run once per prompt; their GEMMs and attention reuse the MMDiT kernels, these are the rest.
// reference: python/src/diffusionkit/mlx/clip.py:28-120, python/src/diffusionkit/mlx/t5.py:60-243.
#include "dk_kernels.h"

// nn.Embedding lookup (clip.py:96-97, t5.py:318): out[i, :] = table[ids[i], :] (+ pos[i % pos_rows, :] when pos != null)
__global__ __launch_bounds__(256) void dk_embedding_kernel(const bf16_t* __restrict__ table, const int* __restrict__ ids, const bf16_t* __restrict__ pos,
                                                          int pos_rows, bf16_t* __restrict__ out, float* __restrict__ out_f32, int n, int dim,
                                                          int vocab) {
  const int cpr = dim / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n * cpr) return;
  const int row = (int)(i / cpr), c = (int)(i % cpr);
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const u32x4 raw = *(const u32x4*)(table + (size_t)id * dim + c * 8);
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) unpack2bf(raw[e], v[2 * e], v[2 * e + 1]);
  if (pos != nullptr) {
    const u32x4 pr = *(const u32x4*)(pos + (size_t)(row % pos_rows) * dim + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p0, p1;
      unpack2bf(pr[e], p0, p1);
      v[2 * e] = round_bf16(v[2 * e] + p0);
      v[2 * e + 1] = round_bf16(v[2 * e + 1] + p1);
    }
  }
  if (out != nullptr) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
    *(u32x4*)(out + (size_t)row * dim + c * 8) = o;
  }
  if (out_f32 != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out_f32[(size_t)row * dim + c * 8 + e] = v[e];
  }
}
int dk_launch_embedding(const bf16_t* table, const int* ids, const bf16_t* pos, int pos_rows, bf16_t* out, float* out_f32, int n, int dim,
                        int vocab, hipStream_t stream) {
  DK_REQUIRE(dim % 8 == 0 && n > 0 && vocab > 0, "embedding: dim must be a multiple of 8");
  const long total = (long)n * (dim / 8);
  hipLaunchKernelGGL(dk_embedding_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, table, ids, pos, pos_rows > 0 ? pos_rows : 1,
                     out, out_f32, n, dim, vocab);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// Row normalisations with a learned weight, one wave per row, fp32 statistics, one output rounding:
//   MODE 0  nn.LayerNorm (clip.py:34-35,101): (x - mean) * rsqrt(var + eps) * w + b, x bf16
//   MODE 1  the reference's T5 RMSNorm (t5.py:131-151): w * cast(x * rsqrt(sum((x / sqrt(h))^2) + eps)), x fp32 (residual stream)
template <int MODE, typename TIN>
__global__ __launch_bounds__(256) void dk_rownorm_kernel(const TIN* __restrict__ x, bf16_t* __restrict__ out, int M, int h,
                                                        const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  const TIN* xr = x + (size_t)m * h;
  auto ld = [&](int i) -> float {
    if constexpr (sizeof(TIN) == 2)
      return bf2f(xr[i]);
    else
      return xr[i];
  };
  float s = 0.f, q = 0.f;
  for (int i = lane; i < h; i += 64) {
    const float v = ld(i);
    s += v;
    q += v * v;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  float mean = 0.f, r;
  if (MODE == 0) {
    mean = s / (float)h;
    float var = 0.f;  // second pass over the row (L2-resident) for the centred sum, like the oracle's two-pass form
    for (int i = lane; i < h; i += 64) {
      const float d = ld(i) - mean;
      var += d * d;
    }
    r = rsqrtf(wave_sum(var) / (float)h + eps);
  } else {
    r = rsqrtf(q / (float)h + eps);
  }
  bf16_t* orow = out + (size_t)m * h;
  for (int i = lane; i < h; i += 64) {
    const float v = ld(i);
    if (MODE == 0)
      orow[i] = f2bf((v - mean) * r * bf2f(w[i]) + (b ? bf2f(b[i]) : 0.f));
    else
      orow[i] = f2bf(bf2f(w[i]) * (v * r));  // x is fp32 there: the cast in between is a no-op in the reference
  }
}
int dk_launch_layernorm(const bf16_t* x, bf16_t* out, int M, int h, const bf16_t* w, const bf16_t* b, float eps, hipStream_t stream) {
  hipLaunchKernelGGL((dk_rownorm_kernel<0, bf16_t>), dim3((M + 3) / 4), dim3(256), 0, stream, x, out, M, h, w, b, eps);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
int dk_launch_t5_rmsnorm(const float* x, bf16_t* out, int M, int h, const bf16_t* w, float eps, hipStream_t stream) {
  hipLaunchKernelGGL((dk_rownorm_kernel<1, float>), dim3((M + 3) / 4), dim3(256), 0, stream, x, out, M, h, w, (const bf16_t*)nullptr, eps);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// elementwise helpers: op 0 quick_gelu x * sigmoid(1.702 x) (clip.py:11, nn.gelu_fast_approx); op 1 product a * b
// (t5.py:176-178 gated activation); op 2 fp32 residual r += bf16 y (t5.py:199-204: the T5 stream is fp32)
__global__ void dk_text_elementwise_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, float* r, long n, int op) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float av = bf2f(a[i]);
  if (op == 0)
    y[i] = f2bf(av * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * av)));
  else if (op == 1)
    y[i] = f2bf(av * bf2f(b[i]));
  else
    r[i] += av;
}
int dk_launch_text_elementwise(const bf16_t* a, const bf16_t* b, bf16_t* y, float* r, long n, int op, hipStream_t stream) {
  hipLaunchKernelGGL(dk_text_elementwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, y, r, n, op);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

// T5 relative-position bias table (t5.py:61-88): out[h, q, k] = emb[bucket[k - q + S - 1], h].  The bucket of every relative
// position (t5.py:14-58: a float32 log whose integer boundaries are implementation-dependent) is computed once on the host;
// columns k >= S up to ld (a multiple of 64) are left zero (the attention kernel masks those keys itself).
__global__ void dk_t5_bias_kernel(const bf16_t* __restrict__ emb, const int* __restrict__ rel_bucket, int H, int S, int ld,
                                  bf16_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * S * ld) return;
  const int k = (int)(i % ld), q = (int)((i / ld) % S), hd = (int)(i / ((long)ld * S));
  out[i] = k < S ? emb[(size_t)rel_bucket[k - q + S - 1] * H + hd] : (bf16_t)0;
}
int dk_launch_t5_bias(const bf16_t* emb, const int* rel_bucket, int H, int S, int ld, bf16_t* out, hipStream_t stream) {
  const long n = (long)H * S * ld;
  hipLaunchKernelGGL(dk_t5_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, emb, rel_bucket, H, S, ld, out);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
