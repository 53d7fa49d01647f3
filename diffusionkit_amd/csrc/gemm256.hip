// 256 x 256 x 64 bf16 MFMA GEMM for gfx950 (large-M linears of the MMDiT blocks).
//
// Same contract as dk_gemm_bf16_kernel<0> (gemm.hip): C = epi(alpha * A . W^T + bias), both
// operands K-major, segment-mapped rows, fused bias / GELU-erf / SiLU / gate*x+residual epilogues.
// Replaces nn.Linear at python/src/diffusionkit/mlx/mmdit.py:821-832 (q/k/v/o, fc1/fc2) and the
// fused linear1/linear2 of the single-stream blocks (:693-751) where M, N are large.
//
// Workgroup = 8 waves (2 along M x 4 along N), one workgroup per CU (128 KiB LDS); every wave owns
// a 128 x 64 output block = 2 x 4 accumulators of v_mfma_f32_32x32x16_bf16 (128 registers).
// A K-tile is four 16 KiB half-tiles {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255},
// each a lane-linear [128][64] bf16 image written by global_load_lds_dwordx4 (2 per thread), with
// the 16-byte chunk index XOR-swizzled by (row>>1)&7 on the SOURCE address and again on the
// ds_read_b128 (guide rule 21).  A wave reads exactly one A half-tile and one W half-tile.
//
// Schedule: the K loop is software-pipelined over a 2-deep ring of K-tiles (8 half-tile slots).
// Each K-tile is computed in 4 quadrant phases (64 rows x 32 cols x K=64 = 8 MFMAs each) in snake
// order so every phase reads one new operand; the refill of the ring (one half-tile per phase) is
// issued as soon as the last reader of that slot has passed a barrier, and is waited for with a
// COUNTED s_waitcnt vmcnt(N) -- loads stay in flight across barriers.  The MFMA is issued with
// swapped operands (W fragment as A) so a lane owns one output row: the epilogue works on 8-byte
// column runs.
#include "dk_kernels.h"

#define T256 256
#define BK 64
#define HALF_BYTES (128 * BK * 2)  // 16 KiB
#define KT_BYTES (4 * HALF_BYTES)  // 64 KiB: A0 A1 W0 W1
#define LDS_BYTES (2 * KT_BYTES)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ int swz_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void dk_gemm256_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- workgroup -> tile: XCD-contiguous chunks, then groups of GROUP tile rows ----
  const int nbm = (p.M + T256 - 1) / T256, nbn = (p.N + T256 - 1) / T256;
  const int nwg = nbm * nbn;
  int t;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int GROUP = 4;
  const int tpg = GROUP * nbn;
  const int g = t / tpg;
  const int first_m = g * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int tm = first_m + (t % tpg) % gsz;
  const int tn = (t % tpg) / gsz;
  const int m0 = tm * T256, n0 = tn * T256;

  // ---- DMA source pointers: half-tile h (0,1 = A; 2,3 = W), 2 instructions j per thread ----
  // instruction j of wave w covers rows w*16 + j*8 + (lane>>3) of the 128-row half-tile
  const bf16_t* src[4][2];
  {
    const int srow = lane >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 16 + j * 8 + srow;
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int m = min(m0 + hh * 128 + r, p.M - 1);
        const int phys = (m / p.a_seg_len) * p.a_seg_stride + (m % p.a_seg_len);
        src[hh][j] = p.A + (size_t)phys * p.lda + chunk * 8;
        const int n = min(n0 + hh * 128 + r, p.N - 1);
        src[2 + hh][j] = p.W + (size_t)n * p.K + chunk * 8;
      }
    }
  }
  const int nk = p.K / BK;

  // issue the DMA of half-tile hh of K-tile kt into ring slot (kt & 1)
  auto issue_half = [&](int kt, int hh) {
    char* dst = smem + (kt & 1) * KT_BYTES + hh * HALF_BYTES + (wave * 16) * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[hh][j] + (size_t)kt * BK), (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (VARIANT == 0) {
    // ---- simple schedule: whole K-tile per barrier, DMA of tile kt+1 under the MFMAs of kt ----
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) issue_half(0, hh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) issue_half(kt + 1, hh);
      }
      const char* As = smem + (kt & 1) * KT_BYTES + wm * HALF_BYTES;
      const char* Ws = smem + (kt & 1) * KT_BYTES + (2 + (wn >> 1)) * HALF_BYTES;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = kk * 2 + hi;
        bf16x8 wf[2], xf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = *(const bf16x8*)(Ws + swz_off((wn & 1) * 64 + i * 32 + l31, c));
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(As + swz_off(i * 32 + l31, c));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // ---- phased schedule -----------------------------------------------------------------
    // Per K-tile kt (ring slot s = kt & 1) a wave reads A half wm and W half 2 + (wn>>1).
    // Quadrant phases in snake order:  P0 = (a-lo, w-lo)  P1 = (a-lo, w-hi)  P2 = (a-hi, w-hi)
    // P3 = (a-hi, w-lo).  All waves pass the same barriers; after the barrier that ends phase Pq
    // of tile kt nobody reads ... (see the slot-release table below).
    //
    // Refill order: the half-tiles of tile kt+2 go into the slots of tile kt.  A slot is free once
    // every wave has finished its last ds_read of it.  All four phases of tile kt read both the A
    // and the W half of every wave, so slot set (kt & 1) is entirely free only after the barrier
    // that closes P3(kt).  The refill of tile kt+2 is therefore issued during the phases of tile
    // kt+1 (one half-tile per phase), i.e. it has a whole K-tile (4 phases ~ 1000+ cycles) to land
    // before tile kt+2 starts; the wait is a counted vmcnt at P3 of tile kt+1.
    //
    // Prologue: tiles 0 and 1 fully issued; wait for tile 0 only (vmcnt(8) leaves tile 1 in flight).
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) issue_half(0, hh);
    if (nk > 1) {
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) issue_half(1, hh);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nk; ++kt) {
      const char* As = smem + (kt & 1) * KT_BYTES + wm * HALF_BYTES;
      const char* Ws = smem + (kt & 1) * KT_BYTES + (2 + (wn >> 1)) * HALF_BYTES + (wn & 1) * 64 * 128;
      // tile kt-1's slots (== tile kt+1's slots) were released by the barrier that closed tile kt-1;
      // tile kt+1 was issued during tile kt-1 (or in the prologue).  During THIS tile we issue
      // tile kt+2?  No: its slots are the ones being read now.  So: issue nothing new here except
      // what is already in flight; tile kt+2 is issued at the END of this tile (after the closing
      // barrier), split across the phases of tile kt+1.
      const bool refill = (kt >= 1) && (kt + 1 < nk);  // refill slot set (kt+1)&1 ... see below
      (void)refill;
      bf16x8 wlo[4], whi[4], xa[2][4];
      // ---- P0: read a-lo (8) + w-lo (4) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wlo[kk] = *(const bf16x8*)(Ws + swz_off(l31, kk * 2 + hi));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xa[i][kk] = *(const bf16x8*)(As + swz_off(i * 32 + l31, kk * 2 + hi));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[kk], xa[i][kk], acc[0][i], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      // ---- P1: read w-hi (4) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) whi[kk] = *(const bf16x8*)(Ws + swz_off(32 + l31, kk * 2 + hi));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[kk], xa[i][kk], acc[1][i], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      // ---- P2: read a-hi (8) ----
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xa[i][kk] = *(const bf16x8*)(As + swz_off(64 + i * 32 + l31, kk * 2 + hi));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[1][2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[kk], xa[i][kk], acc[1][2 + i], 0, 0, 0);
      // ---- P3: w-lo still in registers ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[0][2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[kk], xa[i][kk], acc[0][2 + i], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      // all ds_reads of this tile's slots are complete (their data was consumed by MFMAs above)
      // -> after the barrier the slots may be refilled with tile kt+2
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) issue_half(kt + 2, hh);
        // tile kt+1 (issued one iteration ago) must have landed: leave only tile kt+2's 8 loads in flight
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue: lane owns row m (per mi) and columns nb + {0..3} per (ni, g4) ----
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 128 + mi * 32 + l31;
    if (m >= p.M) continue;
    const size_t crow = (size_t)((m / p.c_seg_len) * p.c_seg_stride + (m % p.c_seg_len)) * p.ldc;
    size_t rrow = 0;
    const bf16_t* gate = nullptr;
    if (p.epi == DK_EPI_GATE_RES || p.epi == DK_EPI_RES)
      rrow = (size_t)((m / p.r_seg_len) * p.r_seg_stride + (m % p.r_seg_len)) * p.ldr;
    if (p.epi == DK_EPI_GATE_RES) gate = p.gate + (size_t)(m / p.gate_seg_len) * p.gate_stride;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int nb = n0 + wn * 64 + ni * 32 + 8 * g4 + 4 * hi;
        if (nb >= p.N) continue;  // N % 4 == 0 is required by the launcher
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g4 + e] * p.alpha;
        if (p.bias) {
          const uint2 bb = *(const uint2*)(p.bias + nb);
          float b0, b1, b2, b3;
          unpack2bf(bb.x, b0, b1);
          unpack2bf(bb.y, b2, b3);
          v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]);
        if (p.epi == DK_EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
        } else if (p.epi == DK_EPI_BIAS_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        } else if (p.epi == DK_EPI_GATE_RES || p.epi == DK_EPI_RES) {
          const uint2 rr = *(const uint2*)(p.res + rrow + nb);
          float r0, r1, r2, r3;
          unpack2bf(rr.x, r0, r1);
          unpack2bf(rr.y, r2, r3);
          if (p.epi == DK_EPI_GATE_RES) {
            const uint2 gg = *(const uint2*)(gate + nb);
            float g0, g1, g2, g3;
            unpack2bf(gg.x, g0, g1);
            unpack2bf(gg.y, g2, g3);
            v[0] = r0 + round_bf16(g0 * v[0]);
            v[1] = r1 + round_bf16(g1 * v[1]);
            v[2] = r2 + round_bf16(g2 * v[2]);
            v[3] = r3 + round_bf16(g3 * v[3]);
          } else {
            v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
          }
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *(uint2*)(p.C + crow + nb) = o;
      }
    }
  }
}

static int g_gemm256_variant = -1;

int dk_launch_gemm256(const GemmParams& p, hipStream_t stream) {
  DK_REQUIRE(!p.conv, "gemm256 is a plain GEMM");
  DK_REQUIRE(p.K % BK == 0 && p.N % 4 == 0 && p.ldc % 4 == 0 && p.lda % 8 == 0, "gemm256 alignment");
  static bool attr_set = false;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const char* e = getenv("DK_GEMM256_VARIANT");
    g_gemm256_variant = e ? atoi(e) : 1;
    attr_set = true;
  }
  const int nbm = (p.M + T256 - 1) / T256, nbn = (p.N + T256 - 1) / T256;
  dim3 grid(nbm * nbn), block(512);
  dk_prof_begin(0, 2.0 * (double)p.M * (double)p.N * (double)p.K, stream);
  if (g_gemm256_variant == 0)
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<0>, grid, block, LDS_BYTES, stream, p);
  else
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<1>, grid, block, LDS_BYTES, stream, p);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
