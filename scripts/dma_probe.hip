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

// GEMM-like sharing: the 32 workgroups of an XCD form a 4 x 8 tile block; the 8 workgroups of a block row read the
// same "A panel" K-slab (32 KiB per iteration), the 4 of a block column the same "W panel" K-slab, all in step
// (STAG = 0) or with their K position rotated by STAG * (index inside the sharing group) iterations.
template <int STAG>
__global__ __launch_bounds__(512, 2) void probe_shared(const char* __restrict__ src, unsigned* __restrict__ out, int iters, int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;  // 32 workgroups per XCD
  const int row = cu >> 3, col = cu & 7;                  // 4 x 8 block
  const size_t panel = (size_t)nk * 32768;                // one panel = nk K-slabs of 32 KiB
  const char* pa = src + (size_t)(xcd * 12 + row) * panel;
  const char* pw = src + (size_t)(xcd * 12 + 4 + col) * panel;
  for (int it = 0; it < iters; ++it) {
    const int ka = (it + STAG * col) % nk, kw = (it + STAG * row) % nk;
    const unsigned ring = (it & 1) * 65536 + wave * 8192;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pa + (size_t)ka * 32768 + wave * 4096 + j * 1024 + lane * 16),
                                       (lds_ptr_t)((lds_char*)smem + ring + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pw + (size_t)kw * 32768 + wave * 4096 + j * 1024 + lane * 16),
                                       (lds_ptr_t)((lds_char*)smem + ring + 4096 + j * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the workgroups of a block stay in step like the GEMM's do
  }
  out[blockIdx.x * 512 + tid] = *(const unsigned*)(smem + tid * 4);
}

// The same sharing, with the GEMM's real addressing: a panel is 256 (A) / 256 (W) rows at a row stride of `ld` bytes and a
// K-tile is 128 bytes of every row (each DMA instruction = 8 rows x 128 B), instead of one contiguous 32 KiB slab.
__global__ __launch_bounds__(512, 2) void probe_shared_rows(const char* __restrict__ src, unsigned* __restrict__ out, int iters, int nk, size_t ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const int row = cu >> 3, col = cu & 7;
  const size_t panel = (size_t)256 * ld;  // 256 rows
  const char* pa = src + (size_t)(xcd * 12 + row) * panel;
  const char* pw = src + (size_t)(xcd * 12 + 4 + col) * panel;
  const size_t lane_off = (size_t)(lane >> 3) * ld + (lane & 7) * 16;  // 8 rows x 8 chunks of 16 B
  for (int it = 0; it < iters; ++it) {
    const int k = it % nk;
    const unsigned ring = (it & 1) * 65536 + wave * 8192;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t r0 = (size_t)(wave * 32 + j * 8) * ld + (size_t)k * 128;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pa + r0 + lane_off), (lds_ptr_t)((lds_char*)smem + ring + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pw + r0 + lane_off), (lds_ptr_t)((lds_char*)smem + ring + 4096 + j * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  out[blockIdx.x * 512 + tid] = *(const unsigned*)(smem + tid * 4);
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
  {
    const int nk = 48, it2 = 48 * 64;
    const size_t span = (size_t)8 * 12 * nk * 32768;  // 8 XCDs x 12 panels
    CK(hipMalloc(&src, span));
    CK(hipMemset(src, 1, span));
    for (int stag = 0; stag < 3; ++stag) {
      float best = 1e30f;
      for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        if (stag == 0) hipLaunchKernelGGL(probe_shared<0>, dim3(nblk), dim3(512), 131072, 0, (const char*)src, (unsigned*)out, it2, nk);
        if (stag == 1) hipLaunchKernelGGL(probe_shared<1>, dim3(nblk), dim3(512), 131072, 0, (const char*)src, (unsigned*)out, it2, nk);
        if (stag == 2) hipLaunchKernelGGL(probe_shared<3>, dim3(nblk), dim3(512), 131072, 0, (const char*)src, (unsigned*)out, it2, nk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double bytes = (double)nblk * it2 * 65536.0;
      printf("GEMM-like sharing (4 x 8 tile block per XCD, panels of %d K-slabs), K rotation %d: %7.2f TB/s  (%.1f B/clk/CU at 2.1 GHz, %.2f us per 64 KiB K-tile)\n",
             nk, stag == 2 ? 3 : stag, bytes / (best * 1e-3) / 1e12, bytes / nblk / (best * 1e-3) / 2.1e9, best * 1e3 / it2);
    }
    CK(hipFree(src));
    for (size_t ld : {(size_t)6144, (size_t)6144 + 256, (size_t)30720, (size_t)128 * 48}) {
      const size_t span2 = (size_t)8 * 12 * 256 * ld;
      CK(hipMalloc(&src, span2));
      CK(hipMemset(src, 1, span2));
      const int nk2 = (int)(ld / 128) < 48 ? (int)(ld / 128) : 48;
      float best = 1e30f;
      for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe_shared_rows, dim3(nblk), dim3(512), 131072, 0, (const char*)src, (unsigned*)out, it2, nk2, ld);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double bytes = (double)nblk * it2 * 65536.0;
      printf("GEMM-like sharing, row-strided K-tiles (row stride %zu B): %7.2f TB/s  (%.1f B/clk/CU at 2.1 GHz, %.2f us per 64 KiB K-tile)\n", ld,
             bytes / (best * 1e-3) / 1e12, bytes / nblk / (best * 1e-3) / 2.1e9, best * 1e3 / it2);
      CK(hipFree(src));
    }
  }
  return 0;
}
