// MFMA issue-rate probe (lab only): 256 workgroups x 8 waves, each wave loops over 8 independent
// 32x32x16 bf16 accumulators.  Variants: accumulators in arch VGPRs (builtin) vs AGPRs (inline asm
// "+a"), 1 or 2 waves per SIMD, random vs zero operands.  Prints TFLOP/s and cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o build_lab/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 4; ++i) b[i] = in[(blockIdx.x * 512 + tid) * 6 + 2 + i];
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (MODE == 0) {
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
          } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ni][mi]) : "v"(a[ni]), "v"(b[mi]));
          }
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 512 + tid] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// same FLOPs per wave and iteration with 16x16x32 MFMAs: 32 accumulators of 4 registers (8 x 4 tiles of 16x16)
__global__ __launch_bounds__(512, 2) void probe16(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[2], b[4];  // same operand set as the 32x32x16 probe (2 + 4 distinct fragments)
  for (int i = 0; i < 2; ++i) a[i] = in[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 4; ++i) b[i] = in[(blockIdx.x * 512 + tid) * 6 + 2 + i];
  f32x4 acc[2][4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)  // 8 x (2 x 4) x 16x16x32 = 64 MFMAs = same FLOPs as 32 x 32x32x16
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  out[blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
  const int iters = 4096;
  const int nblk = 256;
  std::vector<unsigned short> h((size_t)nblk * 512 * 6 * 8);
  void *din, *dout;
  CK(hipMalloc(&din, h.size() * 2));
  CK(hipMalloc(&dout, (size_t)nblk * 512 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero) {
    unsigned st = 1234;
    for (auto& v : h) {
      st = st * 1664525u + 1013904223u;
      // random bf16 in roughly [-1, 1): sign + exponent 0x3f.. + mantissa
      v = zero ? 0 : (unsigned short)(((st >> 16) & 0x8000) | 0x3f00 | ((st >> 8) & 0xff));
    }
    CK(hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode)
      for (int threads = 256; threads <= 512; threads += 256) {
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
          CK(hipEventRecord(e0));
          if (mode == 0)
            hipLaunchKernelGGL(probe<0>, dim3(nblk), dim3(threads), 0, 0, (const bf16x8*)din, (float*)dout, iters);
          else
            hipLaunchKernelGGL(probe<1>, dim3(nblk), dim3(threads), 0, 0, (const bf16x8*)din, (float*)dout, iters);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        if (mode == 0 && threads == 512) {
          float b16 = 1e30f;
          for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe16, dim3(nblk), dim3(512), 0, 0, (const bf16x8*)din, (float*)dout, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < b16) b16 = ms;
          }
          printf("%s operands, 16x16x32 MFMA, 8 waves/CU: %.1f TFLOP/s  (%.3f ms)\n", zero ? "zero  " : "random",
                 (double)nblk * 8 * iters * 32 * 32768.0 / (b16 * 1e-3) / 1e12, b16);
        }
        const double n_mfma = (double)nblk * (threads / 64) * iters * 32;
        const double tf = n_mfma * 32768.0 / (best * 1e-3) / 1e12;
        printf("%s operands, acc in %s, %d waves/CU: %.1f TFLOP/s  (%.3f ms)\n", zero ? "zero  " : "random", mode ? "AGPR" : "VGPR",
               threads / 64, tf, best);
      }
  }
  return 0;
}
