// Issue-overlap probe (lab only): how much VALU / transcendental / LDS-read work fits UNDER a stream of 32x32x16 bf16 MFMAs on a
// gfx950 SIMD, in a fixed (asm volatile) instruction order, with one or two waves per SIMD.  One workgroup per CU; every wave loops
// over a body of 8 independent MFMAs (8 accumulators: the matrix pipe never waits for a dependency) and, spread evenly between
// them, NV v_fma_f32, NE v_exp_f32 and NL ds_read_b128 (conflict-free, 1 KB each) per body.  Printed: cycles per MFMA per SIMD
// against the bare MFMA loop of the same launch shape, i.e. what the extra instruction streams cost.  The attention kernels'
// steady state is (NV, NE, NL) = (40, 8, 12) per 8 MFMAs (profiles/r02_pmc_attention.md: 5.5 VALU and 1.9 LDS instructions per MFMA).
//   hipcc --offload-arch=gfx950 -O3 scripts/issue_probe.hip -o build_lab/issue_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

template <int N>
__device__ __forceinline__ constexpr int share(int i) {  // how many of N instructions go behind MFMA i of 8
  return ((i + 1) * N) / 8 - (i * N) / 8;
}

template <int NV, int NE, int NL, int NT, int NTR = 0, int NX = 0, int KIND = 0>
__global__ __launch_bounds__(NT, 1) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, unsigned lds_base) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  bf16x8 a = in[tid], b = in[tid + NT];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float x[8], ex[8];
  f32x4 l[4];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 l2[8];
  unsigned xi[8];
  f32x2 xp[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) xi[i] = (unsigned)tid + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) xp[i] = f32x2{1.0f + i, 0.5f};
#pragma unroll
  for (int i = 0; i < 8; ++i) l2[i] = f32x2{0.f, 0.f};
  const unsigned laddr2 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(tid & 63) * 8u + (unsigned)(tid >> 6) * 2048u + lds_base;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (float)a[i], ex[i] = (float)b[i] * 1e-3f;
#pragma unroll
  for (int i = 0; i < 4; ++i) l[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c1 = 0.999f, c2 = 1e-3f;
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(tid & 63) * 16u +
                         (unsigned)(tid >> 6) * 2048u + lds_base;
  for (int i = tid; i < 24 * 1024; i += NT) ((float*)smem)[i] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    int nv = 0, ne = 0, nl = 0, ntr = 0, nx = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < share<NL>(i); ++j, ++nl)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[nl & 3]) : "v"(laddr), "n"(1024 * (0 + 0)));
#pragma unroll
      for (int j = 0; j < share<NTR>(i); ++j, ++ntr)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(l2[ntr & 7]) : "v"(laddr2), "n"(0));
#pragma unroll
      for (int j = 0; j < share<NE>(i); ++j, ++ne) asm volatile("v_exp_f32 %0, %0" : "+v"(ex[ne & 7]));
#pragma unroll
      for (int j = 0; j < share<NX>(i); ++j, ++nx) {
        if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(xi[nx & 7]) : "v"(x[nx & 7]), "v"(x[(nx + 1) & 7]));
        if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[nx & 7]) : "v"(c1), "v"(c2));
        if (KIND == 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(xi[nx & 7]), "+v"(xi[(nx + 4) & 7]));
        if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[nx & 7]) : "v"(c2));
        if (KIND == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(xi[nx & 7]) : "v"(xi[7]));
        if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[nx & 3]) : "v"(xp[3]));
        if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xp[nx & 3]) : "v"(xp[3]));
      }
#pragma unroll
      for (int j = 0; j < share<NV>(i); ++j, ++nv) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[nv & 7]) : "v"(c1), "v"(c2));
    }
    if (NL > 0 || NTR > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
    s += x[i] + ex[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s += l[i][0] + l[i][1] + l[i][2] + l[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += l2[i][0] + l2[i][1] + (float)xi[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += xp[i][0] + xp[i][1];
  out[blockIdx.x * NT + tid] = s;
  if (tid == 0 && blockIdx.x == 0) ((long long*)(out + gridDim.x * NT))[0] = t1 - t0;
}

static float* d_out;
static bf16x8* d_in;
static double base_us[3];
static unsigned g_lds_base = 0;

template <int NV, int NE, int NL, int NT, int NTR = 0, int NX = 0, int KIND = 0>
static void run(const char* what) {
  const int iters = 4000, grid = 256;
  const size_t lds = 160 * 1024;
  CK(hipFuncSetAttribute((const void*)probe<NV, NE, NL, NT, NTR, NX, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  probe<NV, NE, NL, NT, NTR, NX, KIND><<<grid, NT, lds>>>(d_in, d_out, 200, g_lds_base);
  double best = 1e30;
  long long cyc = 0;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    probe<NV, NE, NL, NT, NTR, NX, KIND><<<grid, NT, lds>>>(d_in, d_out, iters, g_lds_base);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms * 1e3 < best) best = ms * 1e3;
    CK(hipMemcpy(&cyc, d_out + grid * NT, 8, hipMemcpyDeviceToHost));
  }
  const int wps = NT / 256;
  const double mfma_per_simd = (double)iters * 8 * wps;
  const double tf = 2.0 * 32 * 32 * 16 * mfma_per_simd * 1024 / (best * 1e-6) / 1e12;
  if (NV == 0 && NE == 0 && NL == 0 && NTR == 0 && NX == 0) base_us[wps] = best;
  printf("%-34s %d wave/SIMD  per 8 MFMA: %2d fma %2d exp %2d ds_read_b128 %2d ds_read_b64_tr %2d x kind %d | %8.1f us  %7.1f TF  x%.3f of bare MFMA  (%.1f counter ticks / MFMA / SIMD)\n",
         what, wps, NV, NE, NL, NTR, NX, KIND, best, tf, best / base_us[wps], (double)cyc / mfma_per_simd);
}

#define BOTH(NV, NE, NL, WHAT)  \
  run<NV, NE, NL, 256>(WHAT);   \
  run<NV, NE, NL, 512>(WHAT)

int main(int argc, char** argv) {
  const bool lds_sweep = argc > 1;
  CK(hipMalloc(&d_out, (256 * 512 + 16) * sizeof(float)));
  CK(hipMalloc(&d_in, 1024 * sizeof(bf16x8)));
  CK(hipMemset(d_in, 0x3c, 1024 * sizeof(bf16x8)));
  BOTH(0, 0, 0, "bare MFMA");
  if (lds_sweep) {
    for (unsigned kb : {0u, 16u, 32u, 40u, 48u, 56u, 64u, 80u, 96u, 112u, 128u, 143u}) {
      g_lds_base = kb * 1024u;
      printf("---- LDS base %u KB\n", kb);
      run<0, 0, 16, 256>("b128 reads");
      run<0, 0, 16, 512>("b128 reads");
      run<0, 0, 0, 256, 32>("tr reads");
      run<0, 0, 0, 512, 32>("tr reads");
    }
    return 0;
  }
  BOTH(16, 0, 0, "VALU");
  BOTH(32, 0, 0, "VALU");
  BOTH(48, 0, 0, "VALU");
  BOTH(56, 0, 0, "VALU");
  BOTH(64, 0, 0, "VALU");
  BOTH(0, 4, 0, "exp");
  BOTH(0, 8, 0, "exp");
  BOTH(0, 16, 0, "exp");
  BOTH(0, 0, 4, "LDS");
  BOTH(0, 0, 8, "LDS");
  BOTH(0, 0, 12, "LDS");
  BOTH(0, 0, 16, "LDS");
  BOTH(40, 8, 0, "attention VALU mix");
  BOTH(40, 8, 6, "attention mix, half the LDS reads");
  BOTH(40, 8, 12, "attention mix");
  BOTH(24, 8, 12, "attention mix, lean VALU");
  BOTH(8, 0, 10, "GEMM-like (reads + addressing)");
#define BOTH5(NV, NE, NL, NTR, WHAT)  \
  run<NV, NE, NL, 256, NTR>(WHAT);    \
  run<NV, NE, NL, 512, NTR>(WHAT)
  BOTH5(0, 0, 0, 8, "transpose reads");
  BOTH5(0, 0, 0, 16, "transpose reads");
  BOTH5(0, 0, 0, 24, "transpose reads");
  BOTH5(0, 0, 0, 32, "transpose reads");
  BOTH5(0, 0, 4, 8, "attention M phase (K b128 + V tr)");
  BOTH5(40, 8, 4, 8, "attention mix with tr reads");
  BOTH5(0, 0, 8, 16, "2x attention M phase reads");
#define BOTHX(NE, NX, KIND, WHAT)           \
  run<0, NE, 0, 256, 0, NX, KIND>(WHAT);    \
  run<0, NE, 0, 512, 0, NX, KIND>(WHAT)
  BOTHX(24, 0, 0, "exp");
  BOTHX(32, 0, 0, "exp");
  BOTHX(48, 0, 0, "exp");
  BOTHX(64, 0, 0, "exp");
  BOTHX(0, 16, 0, "cvt_pk_bf16");
  BOTHX(0, 32, 0, "cvt_pk_bf16");
  BOTHX(0, 48, 0, "cvt_pk_bf16");
  BOTHX(0, 64, 0, "cvt_pk_bf16");
  BOTHX(0, 32, 1, "max3");
  BOTHX(0, 48, 1, "max3");
  BOTHX(0, 64, 1, "max3");
  BOTHX(0, 16, 2, "permlane32_swap");
  BOTHX(0, 32, 2, "permlane32_swap");
  BOTHX(0, 48, 3, "add_f32");
  BOTHX(0, 64, 3, "add_f32");
  BOTHX(0, 48, 4, "xor_b32");
  BOTHX(0, 64, 4, "xor_b32");
  BOTHX(0, 24, 5, "pk_add_f32");
  BOTHX(0, 32, 5, "pk_add_f32");
  BOTHX(0, 48, 5, "pk_add_f32");
  BOTHX(0, 24, 6, "pk_fma_f32");
  BOTHX(0, 32, 6, "pk_fma_f32");
  BOTHX(0, 48, 6, "pk_fma_f32");
  return 0;
}
