// Producers of MX-fp8 activations for the fp8 GEMM (gemm256f8.hip): OCP e4m3 elements with one E8M0 scale per row and 32
// consecutive columns, scale bytes in the side-array layout of dk_mx_scale_index (dk_common.h).
//   dk_ln_modulate_mx8_kernel  adaptive-LayerNorm modulation (mmdit.py:958-972, dk_ln_modulate_kernel's arithmetic and bf16
//                              rounding point) whose output row leaves as MX-fp8: the input of the q/k/v, fc1 and linear1 GEMMs
//   dk_quantize_mx8_kernel     bf16 rows -> MX-fp8 rows: the attention output in front of o_proj / linear2
// Both keep a row's 32-element block in 4 adjacent lanes (8 consecutive elements each), so the block maximum is two
// cross-lane exchanges inside a quad.  HBM-bound: one read of the bf16 row, one write of half as many bytes.
#include "dk_kernels.h"

struct Mx8Job {
  const bf16_t* x;
  const bf16_t *shift, *scale;  // null: plain quantisation (no LayerNorm)
  int ldx, M, mod_stride, seg_len, x_seg_len, x_seg_stride;
  Mx8Out o;
};

// Memory schedule as in dk_ln_modulate_kernel (elementwise.hip): every load of the row -- and of its shift / scale chunks -- goes
// out back to back ahead of the first use, through buffer resources that span exactly one row (lanes past the row end read
// zeros, their element stores are dropped), and the row stays packed between the passes.
static_assert(sizeof(Mx8Job) % 8 == 0 && alignof(Mx8Job) == 8, "job b follows job a without padding in the kernarg segment");
template <int NCH, bool LN>
__global__ __launch_bounds__(256) void dk_rows_to_mx8_kernel(Mx8Job ja, Mx8Job jb, int blocks_a, int h, float eps) {
  const bool first = (int)blockIdx.x < blocks_a;
  typedef const __attribute__((address_space(4))) Mx8Job karg_job_t;  // one scalar base pointer, see gemm256v3.hip
  const __attribute__((address_space(4))) char* kbase = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  karg_job_t& j = *(karg_job_t*)(kbase + (first ? 0 : sizeof(Mx8Job)));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = ((int)blockIdx.x - (first ? 0 : blocks_a)) * 4 + wave;
  if (m >= j.M) return;
  const size_t xrow = (size_t)((m / j.x_seg_len) * j.x_seg_stride + (m % j.x_seg_len)) * j.ldx;
  const int nchunks = h >> 3;  // a multiple of 4: every quad of lanes holds whole 32-element blocks
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(j.x + xrow), 0, h * 2, 0x00020000);
  u32x4 raw[NCH], rs[NCH], rc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (lane + 64 * i) * 16, 0, 0);
  if (LN) {
    const size_t mod = (size_t)(m / j.seg_len) * j.mod_stride;
    const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)(j.shift + mod), 0, h * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(j.scale + mod), 0, h * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      rs[i] = __builtin_amdgcn_raw_buffer_load_b128(rsh, (lane + 64 * i) * 16, 0, 0);
      rc[i] = __builtin_amdgcn_raw_buffer_load_b128(rsc, (lane + 64 * i) * 16, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // (keeps the shift / scale loads above the reductions)
  float mean = 0.f, rstd = 0.f;
  if (LN) {
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
    mean = wave_sum(sum) / (float)h;
    __builtin_amdgcn_sched_barrier(0);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const bool live = lane + 64 * i < nchunks;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v0, v1;
        asm volatile("" : "+v"(raw[i][e]));  // (opaque: the unpacked row of the first pass must not stay live)
        unpack2bf(raw[i][e], v0, v1);
        const float d0 = live ? v0 - mean : 0.f, d1 = live ? v1 - mean : 0.f;
        sq += d0 * d0;
        sq += d1 * d1;
      }
    }
    rstd = rsqrtf(wave_sum(sq) / (float)h + eps);
  }
  const unsigned orow = (unsigned)(j.o.row0 + (m / j.o.seg_len) * j.o.seg_stride + (m % j.o.seg_len));
  const unsigned col0 = (unsigned)j.o.col0, n_blk128 = (unsigned)j.o.n_blk128;
  unsigned char* __restrict__ scales = j.o.scales;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(j.o.out + (size_t)orow * j.o.ldo + col0), 0, h, 0x00020000);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    const int c = lane + 64 * i;  // (the exchanges inside dk_mx8_quantize8 run for every lane of the wave: no divergence above it)
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (LN) asm volatile("" : "+v"(raw[i][e]));
      unpack2bf(raw[i][e], v[2 * e], v[2 * e + 1]);
      if (LN) {
        float s0, s1, c0, c1;
        unpack2bf(rs[i][e], s0, s1);
        unpack2bf(rc[i][e], c0, c1);
        // the bf16 value dk_ln_modulate_kernel would have stored (same expression, same rounding), then quantised
        v[2 * e] = round_bf16((v[2 * e] - mean) * rstd * round_bf16(1.0f + c0) + s0);
        v[2 * e + 1] = round_bf16((v[2 * e + 1] - mean) * rstd * round_bf16(1.0f + c1) + s1);
      }
    }
    unsigned e8;
    const uint2 q8 = dk_mx8_quantize8(v, e8);
    u32x2 q;
    q[0] = q8.x; q[1] = q8.y;
    __builtin_amdgcn_raw_buffer_store_b64(q, ro, c * 8, 0, 0);
    if (c < nchunks && (lane & 3) == 0) scales[dk_mx_scale_index(orow, (col0 >> 5) + (unsigned)(c >> 2), n_blk128)] = (unsigned char)e8;
  }
}

static int launch_mx8_jobs(const Mx8Job& a, const Mx8Job& b, int h, float eps, bool ln, hipStream_t stream) {
  DK_REQUIRE(h % 32 == 0 && h <= 4096, "row length must be a multiple of 32 (one MX block) and <= 4096");
  for (const Mx8Job* j : {&a, &b}) {
    if (j->M == 0) continue;
    DK_REQUIRE(j->x && j->o.out && j->o.scales, "null pointer");
    DK_REQUIRE(j->ldx % 8 == 0 && (!ln || j->mod_stride % 8 == 0), "bf16 strides must keep 16-byte alignment");
    DK_REQUIRE(j->o.ldo % 8 == 0 && j->o.col0 % 32 == 0 && j->o.col0 + h <= j->o.ldo && ((uintptr_t)j->o.out & 7) == 0,
               "fp8 output: pitch a multiple of 8 bytes, column offset a multiple of 32");
    DK_REQUIRE(j->o.seg_len > 0 && j->x_seg_len > 0 && (!ln || j->seg_len > 0), "segment lengths must be positive");
    const long last = (long)j->o.row0 + (long)((j->M - 1) / j->o.seg_len) * j->o.seg_stride + (j->M - 1) % j->o.seg_len;
    DK_REQUIRE(last < (long)j->o.n_blk128 * 128, "fp8 output rows exceed the scale array");
  }
  const int blocks_a = (a.M + 3) / 4, blocks_b = (b.M + 3) / 4;
  dim3 grid(blocks_a + blocks_b), block(256);
  const int nch = (h / 8 + 63) / 64;
#define MX_CASE(N)                                                                                                   \
  case N:                                                                                                            \
    if (ln) hipLaunchKernelGGL((dk_rows_to_mx8_kernel<N, true>), grid, block, 0, stream, a, b, blocks_a, h, eps);    \
    else hipLaunchKernelGGL((dk_rows_to_mx8_kernel<N, false>), grid, block, 0, stream, a, b, blocks_a, h, eps);      \
    break;
  switch (nch) {
    MX_CASE(1) MX_CASE(2) MX_CASE(3) MX_CASE(4) MX_CASE(5) MX_CASE(6) MX_CASE(7) MX_CASE(8)
    default: DK_REQUIRE(false, "unsupported row length");
  }
#undef MX_CASE
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

static Mx8Job mx8_job(const bf16_t* x, int ldx, int M, const bf16_t* shift, const bf16_t* scale, int mod_stride, int seg_len, int x_seg_len,
                      int x_seg_stride, const Mx8Out& o) {
  Mx8Job j;
  j.x = x; j.shift = shift; j.scale = scale; j.ldx = ldx; j.M = M; j.mod_stride = mod_stride; j.seg_len = seg_len;
  j.x_seg_len = x_seg_len; j.x_seg_stride = x_seg_stride; j.o = o;
  return j;
}
static Mx8Job mx8_none() {
  Mx8Out o;
  o.out = nullptr; o.scales = nullptr; o.ldo = 8; o.n_blk128 = 1; o.row0 = 0; o.seg_len = 1; o.seg_stride = 0; o.col0 = 0;
  return mx8_job(nullptr, 8, 0, nullptr, nullptr, 8, 1, 1, 0, o);
}

int dk_launch_quantize_mx8(const bf16_t* x, int ldx, int x_seg_len, int x_seg_stride, int M, int h, const Mx8Out& o, hipStream_t stream) {
  return launch_mx8_jobs(mx8_job(x, ldx, M, nullptr, nullptr, 8, 1, x_seg_len > 0 ? x_seg_len : M, x_seg_stride, o), mx8_none(), h, 0.f, false, stream);
}
int dk_launch_ln_modulate_mx8(const bf16_t* x, int ldx, int M, int h, const bf16_t* shift, const bf16_t* scale, int mod_stride, int seg_len,
                              int x_seg_len, int x_seg_stride, float eps, const Mx8Out& o, hipStream_t stream) {
  DK_REQUIRE(shift && scale, "shift / scale missing");
  return launch_mx8_jobs(mx8_job(x, ldx, M, shift, scale, mod_stride, seg_len, x_seg_len, x_seg_stride, o), mx8_none(), h, eps, true, stream);
}
int dk_launch_ln_modulate2_mx8(const bf16_t* x0, int M0, const bf16_t* shift0, const bf16_t* scale0, int seg0, const Mx8Out& o0,
                               const bf16_t* x1, int M1, const bf16_t* shift1, const bf16_t* scale1, int seg1, const Mx8Out& o1, int ldx, int h,
                               int mod_stride, int x_seg_stride, float eps, hipStream_t stream) {
  DK_REQUIRE(shift0 && scale0 && shift1 && scale1, "shift / scale missing");
  return launch_mx8_jobs(mx8_job(x0, ldx, M0, shift0, scale0, mod_stride, seg0, seg0, x_seg_stride, o0),
                         mx8_job(x1, ldx, M1, shift1, scale1, mod_stride, seg1, seg1, x_seg_stride, o1), h, eps, true, stream);
}
