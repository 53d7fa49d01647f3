// Lab probe: what does `buffer_load_dwordx4 ... lds` write for lanes whose offset is out of the descriptor's range?
// LDS is pre-filled with 0xAB; even lanes load in range, odd lanes use voffset 0x80000000 (>= num_records).
//   hipcc --offload-arch=gfx950 -O3 scripts/oob_probe.hip -o build_lab/oob_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) char lds_char;

__global__ void probe(const unsigned char* src, unsigned nbytes, unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) smem[i] = 0xAB;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes, 0x00020000);
  const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((lds_char*)smem), 16, (int)voff, 0, 0, 0);
  // second piece: in-range lane offsets but a scalar offset -- is soffset part of the range check?
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((lds_char*)smem + 1024), 16, (int)(lane * 16u), (int)(nbytes - 512u), 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 2048; i += 64) out[i] = smem[i];
}

int main() {
  const unsigned n = 4096;
  std::vector<unsigned char> h(n);
  for (unsigned i = 0; i < n; ++i) h[i] = (unsigned char)(1 + i % 200);
  unsigned char *d, *o;
  hipMalloc(&d, n);
  hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, o);
  std::vector<unsigned char> r(2048);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  int in_ok = 0, oob_zero = 0, oob_stale = 0, oob_other = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int b = 0; b < 16; ++b) {
      const unsigned char v = r[lane * 16 + b];
      if (!(lane & 1)) in_ok += v == h[lane * 16 + b];
      else if (v == 0) ++oob_zero;
      else if (v == 0xAB) ++oob_stale;
      else ++oob_other;
    }
  printf("piece 1: in-range bytes correct %d / 512; out-of-range lanes: zero %d, stale %d, other %d (of 512)\n", in_ok, oob_zero, oob_stale, oob_other);
  int s_ok = 0, s_zero = 0, s_stale = 0;
  for (int i = 0; i < 1024; ++i) {
    const unsigned src_i = n - 512 + i;
    const unsigned char v = r[1024 + i];
    if (src_i < n && v == h[src_i]) ++s_ok;
    else if (v == 0) ++s_zero;
    else if (v == 0xAB) ++s_stale;
  }
  printf("piece 2 (soffset = n - 512, lane offsets 0..1008): correct %d, zero %d, stale %d (512 bytes lie past the end)\n", s_ok, s_zero, s_stale);
  return 0;
}
