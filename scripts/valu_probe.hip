// lab: VALU / transcendental issue costs on gfx950 without MFMAs: per loop trip NE v_exp_f32 (finite inputs, distinct destination
// registers) and NF independent v_fma_f32 (and variants), interleaved; 1 or 2 waves per SIMD.  Do exponentials run beside the FMAs?
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_probe.hip -o build_lab/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NE, int NF, int KIND, int NT>
__global__ __launch_bounds__(NT, 1) void probe(float* out, int iters, float seed) {
  float x[8], e[8], src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = seed + i, e[i] = 0.f, src[i] = -0.5f * (float)(i + 1) + seed * 1e-3f * (float)threadIdx.x;
  const float c1 = 0.999f, c2 = 1e-3f;
  constexpr int TOT = NE > NF ? NE : NF;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < TOT; ++i) {
      // spread both streams evenly over the trip
      if ((i * NE) / TOT != ((i + 1) * NE) / TOT) {
        if (KIND == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i & 7]) : "v"(src[i & 7]));
        if (KIND == 1) asm volatile("v_log_f32 %0, %1" : "=v"(e[i & 7]) : "v"(x[i & 7]));
        if (KIND == 2) asm volatile("v_rcp_f32 %0, %1" : "=v"(e[i & 7]) : "v"(x[i & 7]));
      }
      if ((i * NF) / TOT != ((i + 1) * NF) / TOT) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(c1), "v"(c2));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + e[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

static float* d_out;
template <int NE, int NF, int KIND, int NT>
static void run(const char* what) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  probe<NE, NF, KIND, NT><<<256, NT>>>(d_out, 100, 1.0f);
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    probe<NE, NF, KIND, NT><<<256, NT>>>(d_out, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms * 1e3 < best) best = ms * 1e3;
  }
  // cycles per trip per SIMD at a nominal 2.4 GHz (no MFMA load: the clock sits near its maximum)
  const double ns_trip = best * 1e3 / iters;
  printf("%-22s %d wave/SIMD: %2d trans + %3d fma per trip | %8.1f us  %7.2f ns / trip  = %6.1f cycles at 2.4 GHz\n", what, NT / 256, NE, NF, best, ns_trip,
         ns_trip * 2.4);
}
#define BOTH(NE, NF, KIND, WHAT)   \
  run<NE, NF, KIND, 256>(WHAT);    \
  run<NE, NF, KIND, 512>(WHAT)
int main() {
  CK(hipMalloc(&d_out, 256 * 512 * 4));
  BOTH(0, 128, 0, "fma only");
  BOTH(32, 0, 0, "exp only");
  BOTH(32, 128, 0, "exp + fma");
  BOTH(32, 64, 0, "exp + fma");
  BOTH(32, 32, 0, "exp + fma");
  BOTH(16, 128, 0, "exp + fma");
  BOTH(32, 0, 1, "log only");
  BOTH(32, 0, 2, "rcp only");
  return 0;
}
