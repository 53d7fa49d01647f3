// Shared device/host helpers for the gfx950 (CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 storage
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define DK_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even f32 -> bf16 through the hardware conversion (v_cvt_pk_bf16_f32 on gfx950: one
// instruction per PAIR of values instead of a 3-4 instruction integer sequence per value -- the GEMM tails were
// bound by that VALU work)
typedef __bf16 dk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float round_bf16(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const dk_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack2bf(uint32_t v, float& lo, float& hi) {
  lo = __uint_as_float(v << 16);
  hi = __uint_as_float(v & 0xffff0000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// nn.GELU() (exact erf form, mmdit.py:421): gelu(x) = x * Phi(x) with erfc(|x| / sqrt 2) from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 rounding that follows) -- 14 VALU instructions instead of ocml erff's 34; written
// on erfc so that the negative tail has no cancellation: gelu = 0.5 x erfc(-x / sqrt 2).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // erfc(|x| / sqrt 2)
  const float h = 0.5f * x * erfc_z;
  return x >= 0.f ? x - h : h;
}
// The same on two values at a time, written on 2-vectors so that the full-rate arithmetic becomes packed fp32 instructions
// (v_pk_fma / v_pk_mul: two lanes' worth per issue slot), and rearranged as gelu(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2) with the
// 0.5 |x| = z / sqrt 2 folded into the polynomial's coefficients: no compare / select, 6 packed + 2 x 2 transcendental instructions
// per pair instead of 2 x 17.  (GEMM tails: nothing else runs on the SIMD while a tile is written out, a GELU tile used to cost
// 5 us more than a bias-only one.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) {
  const f32x2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 d = z * 0.3275911f + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  // coefficients of the erfc polynomial times 1 / sqrt 2
  f32x2 p = t * 0.75052697643f + -1.02753365239f;
  p = p * t + 1.00509129513f;
  p = p * t + -0.20116957125f;
  p = p * t + 0.18019173255f;
  const f32x2 a = z * z * -1.4426950408889634f;
  const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2 w = p * t * z;
  return __builtin_elementwise_max(x, f32x2{0.f, 0.f}) - w * e;
}

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- host side -----------------------------------------------------------------------
#include <string>
void dk_set_error(const std::string& msg);
#define DK_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      dk_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
      return -2;                                                                        \
    }                                                                                   \
  } while (0)
#define DK_REQUIRE(cond, msg)                                                           \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      dk_set_error(std::string("requirement failed: ") + #cond + " -- " + (msg));       \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

// ---- MX-fp8 activations (OCP e4m3 elements, one E8M0 scale per row and 32 consecutive columns) -----------------------
// Position of the scale byte of (physical row r, 32-column block kb) in the side array of an activation buffer with
// n_blk128 = ceil(rows / 128) + 1 row blocks: [kb / 4][r / 128][kb % 4][r % 16][(r / 16) % 8].  The fp8 GEMM
// (gemm256f8.hip) reads, per wave and 128-column K-tile, the 512 contiguous bytes of its 128-row block: lane
// (kb % 4) * 16 + r % 16 gets the 8 bytes of its eight 16-row fragments.
__host__ __device__ __forceinline__ size_t dk_mx_scale_index(unsigned r, unsigned kb, unsigned n_blk128) {
  return ((((size_t)(kb >> 2) * n_blk128 + (r >> 7)) * 64 + (kb & 3) * 16 + (r & 15)) << 3) + ((r >> 4) & 7);
}
// Quantise the 8 values this lane holds of a 32-element block shared by 4 ADJACENT lanes (lane & ~3 .. + 3): returns the 8
// e4m3 bytes (element 0 in the low byte of .x) and the block's E8M0 scale byte.  Scale = the smallest power of two s with
// amax / s <= 448 (e4m3's largest finite value), clamped to [2^-126, 2^127]; elements = RNE(v / s), clamped to +-448
// (the oracle restates exactly this: oracle/fp8.py).
// maximum over the 4 lanes of a quad (lane & ~3 .. + 3) through DPP quad permutes: two VALU instructions, no LDS crossbar
__device__ __forceinline__ float dk_quad_max(float v) {
  int x = __float_as_int(v);
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false)));  // quad_perm [1,0,3,2]: lane ^ 1
  x = __float_as_int(v);
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)));  // quad_perm [2,3,0,1]: lane ^ 2
}
__device__ __forceinline__ uint2 dk_mx8_quantize8(const float* v, unsigned& e8m0) {
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
  amax = dk_quad_max(amax);
  const float t = amax * (1.0f / 448.0f);
  unsigned e = (__float_as_uint(t) + 0x7FFFFFu) >> 23;  // ceil(log2 t) + 127
  e = e < 1u ? 1u : (e > 254u ? 254u : e);
  const float inv = __uint_as_float((254u - e) << 23);  // 2^(127 - e)
  // |v * inv| <= 448 (1 + 2^-23): the rounding of t can leave the block maximum a last-place unit above 448, which still rounds
  // to 448 (v_cvt_pk_fp8_f32 overflows to NaN only beyond 464, profiles/r02_fp8_probe.log) -- no clamp needed, except when the
  // scale itself was clamped (e = 254: inv = 0, everything becomes 0; e = 1: amax <= 448 * 2^-126, far inside the range)
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, w1, true);
  e8m0 = e;
  return make_uint2((unsigned)w0, (unsigned)w1);
}

// Launch-site setup that HIP keeps PER DEVICE (hipFuncSetAttribute of the dynamic-LDS limit): `static DkDeviceOnce once; if (once.first())
// { ...; once.mark(); }` runs the block the first time each device of the process meets the call site (a plain static bool would set
// the attribute on the first GPU only and launch with the 64 KiB default on the next).  Two threads racing through first() both
// set the attribute: idempotent.
#include <atomic>
struct DkDeviceOnce {
  std::atomic<unsigned long long> done{0ull};
  static int device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 ? dev : -1;
  }
  bool first() const {
    const int dev = device();
    return dev < 0 || (done.load(std::memory_order_acquire) & (1ull << dev)) == 0ull;
  }
  void mark() {
    const int dev = device();
    if (dev >= 0) done.fetch_or(1ull << dev, std::memory_order_release);
  }
};

// Compute units of the CURRENT device, cached per device (the kernel choosers, the pair test and the launchers must agree on one
// number; a single static shared by all devices, or a literal 256, lets the chooser's tile height and the launcher's disagree on another part).
// Without a usable device (CPU-side symbol checks) it answers 256, the MI355X's count.
inline int dk_device_cu_count() {
  static std::atomic<int> cache[64];
  const int dev = DkDeviceOnce::device();
  if (dev < 0) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  cache[dev].store(n, std::memory_order_relaxed);
  return n;
}
