// L2 -> CU load-path probe (lab only): 256 workgroups x 8 waves stream an L2-resident buffer, 64 KiB per
// workgroup and iteration (the per-K-tile volume of the 256^2 GEMM), through
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction) into a 2 x 64 KiB LDS ring
//   mode 1: global_load_dwordx4 into registers (8 x 16 B per lane in flight), xor-reduced
//   mode 2: half of the pieces through each path
// and prints bytes per clock and CU (clock from s_memtime deltas is not needed: wall time at the reported
// average clock is enough to compare the paths).
//   hipcc --offload-arch=gfx950 -O3 scripts/dma_probe.hip -o build_lab/dma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ src, size_t span, unsigned* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every workgroup walks the buffer from its own start so that neighbouring CUs hit different lines
  size_t pos = ((size_t)blockIdx.x * 65536 * 7) % span;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    const char* base = src + pos + (size_t)wave * 8192;  // 8 KiB per wave and iteration
    const unsigned ring = (it & 1) * 65536 + wave * 8192;
    u32x4 r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool dma = MODE == 0 || (MODE == 2 && (j & 1) == 0);
      if (dma)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + j * 1024 + lane * 16), (lds_ptr_t)((lds_char*)smem + ring + j * 1024), 16, 0, 0);
      else
        r[j] = *(const u32x4*)(base + j * 1024 + lane * 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool dma = MODE == 0 || (MODE == 2 && (j & 1) == 0);
      if (!dma) acc ^= r[j];
    }
    // one iteration stays in flight (the GEMM keeps two K-tiles in its ring)
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    pos += 65536;
    if (pos + 65536 > span) pos = 0;
  }
  if (MODE != 1) acc[0] ^= *(const unsigned*)(smem + tid * 4);
  out[blockIdx.x * 512 + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main(int argc, char** argv) {
  const int iters = 4096, nblk = 256;
  void *src, *out;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipMalloc(&out, (size_t)nblk * 512 * 4));
  for (size_t span_mb : {2, 16, 256, 2048}) {
    const size_t span = span_mb << 20;
    CK(hipMalloc(&src, span));
    CK(hipMemset(src, 1, span));
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(nblk), dim3(512), 131072, 0, (const char*)src, span, (unsigned*)out, iters);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(nblk), dim3(512), 131072, 0, (const char*)src, span, (unsigned*)out, iters);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(nblk), dim3(512), 131072, 0, (const char*)src, span, (unsigned*)out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double bytes = (double)nblk * iters * 65536.0;
      printf("span %5zu MiB  %-28s %7.2f TB/s  (%.1f GB/s per CU, %.1f B/clk/CU at 2.1 GHz)  %.3f ms\n", span_mb,
             mode == 0 ? "global_load_lds (LDS-DMA)" : mode == 1 ? "global_load_dwordx4 (regs)" : "half / half", bytes / (best * 1e-3) / 1e12,
             bytes / nblk / (best * 1e-3) / 1e9, bytes / nblk / (best * 1e-3) / 2.1e9, best);
    }
    CK(hipFree(src));
  }
  return 0;
}
