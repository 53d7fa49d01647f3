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

template <int NV, int NE, int NL, int NT>
__global__ __launch_bounds__(NT, 1) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
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
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (float)a[i], ex[i] = (float)b[i] * 1e-3f;
#pragma unroll
  for (int i = 0; i < 4; ++i) l[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c1 = 0.999f, c2 = 1e-3f;
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(tid & 63) * 16u +
                         (unsigned)(tid >> 6) * 8192u;
  for (int i = tid; i < 24 * 1024; i += NT) ((float*)smem)[i] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    int nv = 0, ne = 0, nl = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < share<NL>(i); ++j, ++nl)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[nl & 3]) : "v"(laddr), "n"(1024 * (0 + 0)));
#pragma unroll
      for (int j = 0; j < share<NE>(i); ++j, ++ne) asm volatile("v_exp_f32 %0, %0" : "+v"(ex[ne & 7]));
#pragma unroll
      for (int j = 0; j < share<NV>(i); ++j, ++nv) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[nv & 7]) : "v"(c1), "v"(c2));
    }
    if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
  out[blockIdx.x * NT + tid] = s;
  if (tid == 0 && blockIdx.x == 0) ((long long*)(out + gridDim.x * NT))[0] = t1 - t0;
}

static float* d_out;
static bf16x8* d_in;
static double base_us[3];

template <int NV, int NE, int NL, int NT>
static void run(const char* what) {
  const int iters = 4000, grid = 256;
  const size_t lds = 100 * 1024;
  CK(hipFuncSetAttribute((const void*)probe<NV, NE, NL, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  probe<NV, NE, NL, NT><<<grid, NT, lds>>>(d_in, d_out, 200);
  double best = 1e30;
  long long cyc = 0;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    probe<NV, NE, NL, NT><<<grid, NT, lds>>>(d_in, d_out, iters);
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
  if (NV == 0 && NE == 0 && NL == 0) base_us[wps] = best;
  printf("%-34s %d wave/SIMD  per 8 MFMA: %2d fma %2d exp %2d ds_read_b128 | %8.1f us  %7.1f TF  x%.3f of bare MFMA  (%.1f counter ticks / MFMA / SIMD)\n",
         what, wps, NV, NE, NL, best, tf, best / base_us[wps], (double)cyc / mfma_per_simd);
}

#define BOTH(NV, NE, NL, WHAT)  \
  run<NV, NE, NL, 256>(WHAT);   \
  run<NV, NE, NL, 512>(WHAT)

int main() {
  CK(hipMalloc(&d_out, (256 * 512 + 16) * sizeof(float)));
  CK(hipMalloc(&d_in, 1024 * sizeof(bf16x8)));
  CK(hipMemset(d_in, 0x3c, 1024 * sizeof(bf16x8)));
  BOTH(0, 0, 0, "bare MFMA");
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
  return 0;
}
