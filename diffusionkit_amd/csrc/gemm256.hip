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

// Epilogue shared by the data-parallel and the stream-K kernel: lane owns row m (per mi) and the
// columns nb + {0..3} per (ni, g4) -- 8-byte runs of bias / GELU / gate*x+residual outputs.
__device__ __forceinline__ void dk_epilogue256(const GemmParams& p, const f32x16 (&acc)[2][4], int m0, int n0, int wm, int wn,
                                               int hi, int l31) {
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

// ABL (ablation builds for the kernel lab only): bit 0 = no DMA in the main loop, bit 1 = no
// ds_reads in the main loop, bit 2 = no MFMAs.  ABL = 0 is the product kernel.
template <int VARIANT, int ABL = 0>
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
        src[2 + hh][j] = p.W + (size_t)n * p.ldw + chunk * 8;
      }
    }
  }
  const int nk = p.K / BK;

  // issue the DMA of half-tile hh of K-tile kt into ring slot (kt & 1)
  auto issue_half = [&](int kt, int hh) {
    if ((ABL & 1) && kt >= 2) return;
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

  if (VARIANT == 0 || VARIANT == 2) {
    // ---- simple schedule: whole K-tile per barrier, DMA of tile kt+1 under the MFMAs of kt ----
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) issue_half(0, hh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 wf[2], xf[4];
    for (int kt = 0; kt < nk; ++kt) {
      if (VARIANT == 0 && kt + 1 < nk) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) issue_half(kt + 1, hh);
      }
      const char* As = smem + (kt & 1) * KT_BYTES + wm * HALF_BYTES;
      const char* Ws = smem + (kt & 1) * KT_BYTES + (2 + (wn >> 1)) * HALF_BYTES;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = kk * 2 + hi;
        // VARIANT 2: one half-tile of DMA per kk-step, so that a wave blocked on VMEM issue overlaps
        // the MFMAs of the other wave on its SIMD instead of both issuing DMA right after the barrier
        if (VARIANT == 2 && kt + 1 < nk) issue_half(kt + 1, kk);
        if (!(ABL & 2) || kt == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) wf[i] = *(const bf16x8*)(Ws + swz_off((wn & 1) * 64 + i * 32 + l31, c));
#pragma unroll
          for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(As + swz_off(i * 32 + l31, c));
        }
        if (ABL & 4) {
          asm volatile("" ::"v"(wf[0]), "v"(wf[1]), "v"(xf[0]), "v"(xf[1]), "v"(xf[2]), "v"(xf[3]));
        } else {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }
      if (!(ABL & 8)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  } else if (VARIANT == 3) {
    // ---- register-pipelined schedule ---------------------------------------------------------
    // hipcc leaves "6 ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 8 MFMA" per k-step of the plain loop:
    // the LDS latency of every k-step is exposed.  Here the fragments of k-step kk+1 are read into
    // a second register set BEFORE the MFMAs of k-step kk are issued (the waits become counted
    // lgkmcnt(6)), and the tile barrier sits between k-steps 2 and 3 so that the first fragments of
    // the NEXT tile are also in flight under the last 8 MFMAs of this one:
    //   S0: read(kt,1) | mfma(kt,0)   S1: read(kt,2) | mfma(kt,1)   S2: read(kt,3) | mfma(kt,2)
    //   vmcnt(0) lgkmcnt(0) barrier   -> tile kt+1 landed; every wave has finished reading tile kt
    //   S3: read(kt+1,0), DMA(kt+2 -> slot of kt) | mfma(kt,3)
#define DK_RD(SET, BUF, KK)                                                                                   \
  do {                                                                                                        \
    const char* As_ = smem + (BUF) * KT_BYTES + wm * HALF_BYTES;                                              \
    const char* Ws_ = smem + (BUF) * KT_BYTES + (2 + (wn >> 1)) * HALF_BYTES;                                 \
    _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) wf##SET[a_] =                                            \
        *(const bf16x8*)(Ws_ + swz_off((wn & 1) * 64 + a_ * 32 + l31, (KK) * 2 + hi));                        \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_) xf##SET[a_] =                                            \
        *(const bf16x8*)(As_ + swz_off(a_ * 32 + l31, (KK) * 2 + hi));                                        \
  } while (0)
#define DK_MM(SET)                                                                                            \
  do {                                                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)         \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf##SET[ni], xf##SET[mi], acc[ni][mi], 0, 0, 0); \
  } while (0)
    bf16x8 wf0[2], xf0[4], wf1[2], xf1[4];
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
    asm volatile("" ::: "memory");
    DK_RD(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int b = kt & 1;
      DK_RD(1, b, 1);
      __builtin_amdgcn_sched_barrier(0);
      DK_MM(0);
      __builtin_amdgcn_sched_barrier(0);
      DK_RD(0, b, 2);
      __builtin_amdgcn_sched_barrier(0);
      DK_MM(1);
      __builtin_amdgcn_sched_barrier(0);
      DK_RD(1, b, 3);
      __builtin_amdgcn_sched_barrier(0);
      DK_MM(0);
      __builtin_amdgcn_sched_barrier(0);
      // tile kt+1 (DMA issued one tile ago) landed; all my reads of tile kt done (kk3 fragments arrived)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 1 < nk) DK_RD(0, b ^ 1, 0);
      if (kt + 2 < nk) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) issue_half(kt + 2, hh);
      }
      __builtin_amdgcn_sched_barrier(0);
      DK_MM(1);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef DK_RD
#undef DK_MM
  } else if (VARIANT == 4) {
    // ---- register-pipelined schedule with hand-placed LDS waits -------------------------------
    // Same pipeline as VARIANT 3, but the fragment reads are inline-asm ds_read_b128 so that hipcc's
    // waitcnt pass does not see them (it answers a pending LDS read with s_waitcnt lgkmcnt(0) at
    // every second k-step, exposing the LDS latency).  The waits are counted by hand: when the MFMAs
    // of k-step kk start, the 6 reads of k-step kk+1 are the only younger LDS operations of the wave
    // -> s_waitcnt lgkmcnt(6).  The wait statement names the fragment registers it covers as "+v"
    // operands, so no MFMA that consumes them can be scheduled above it (guide 5.7, form ii).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned offk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offk[kk] = (unsigned)swz_off(l31, kk * 2 + hi);
    const unsigned sA = lds0 + wm * HALF_BYTES;
    const unsigned sW = lds0 + (2 + (wn >> 1)) * HALF_BYTES + (wn & 1) * 64 * 128;
#define DK_LDS_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR))
#define DK_RD(SET, BUFOFF, KK)                              \
  do {                                                      \
    const unsigned aA_ = offk[KK] + sA + (BUFOFF);          \
    const unsigned aW_ = offk[KK] + sW + (BUFOFF);          \
    DK_LDS_RD(wf##SET[0], aW_, 0);                          \
    DK_LDS_RD(wf##SET[1], aW_, 4096);                       \
    DK_LDS_RD(xf##SET[0], aA_, 0);                          \
    DK_LDS_RD(xf##SET[1], aA_, 4096);                       \
    DK_LDS_RD(xf##SET[2], aA_, 8192);                       \
    DK_LDS_RD(xf##SET[3], aA_, 12288);                      \
  } while (0)
#define DK_WAIT(N, SET)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                  \
               : "+v"(wf##SET[0]), "+v"(wf##SET[1]), "+v"(xf##SET[0]), "+v"(xf##SET[1]), "+v"(xf##SET[2]), \
                 "+v"(xf##SET[3]))
#define DK_MM(SET)                                                                                            \
  do {                                                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)         \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf##SET[ni], xf##SET[mi], acc[ni][mi], 0, 0, 0); \
  } while (0)
    bf16x8 wf0[2], xf0[4], wf1[2], xf1[4];
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
    asm volatile("" ::: "memory");
    DK_RD(0, 0u, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned bo = (kt & 1) * KT_BYTES;
      DK_RD(1, bo, 1);
      DK_WAIT(6, 0);
      DK_MM(0);
      DK_RD(0, bo, 2);
      DK_WAIT(6, 1);
      DK_MM(1);
      DK_RD(1, bo, 3);
      DK_WAIT(6, 0);
      DK_MM(0);
      __builtin_amdgcn_sched_barrier(0);  // keep these MFMAs in front of the wait: they run under it
      // tile kt+1 (DMA issued one tile ago) has landed; my reads of tile kt are complete
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                   : "+v"(wf1[0]), "+v"(wf1[1]), "+v"(xf1[0]), "+v"(xf1[1]), "+v"(xf1[2]), "+v"(xf1[3])
                   :
                   : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 1 < nk) DK_RD(0, bo ^ KT_BYTES, 0);
      if (kt + 2 < nk) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) issue_half(kt + 2, hh);
      }
      DK_MM(1);
    }
#undef DK_LDS_RD
#undef DK_RD
#undef DK_WAIT
#undef DK_MM
  } else {
    // ---- staggered 4-section schedule ------------------------------------------------------
    // Wave group g0 = waves 0-3 (wm = 0, A half 0), g1 = waves 4-7 (wm = 1, A half 1); every SIMD
    // hosts one wave of each group.  A K-tile is processed in four sections separated by
    // workgroup barriers:  R_A (16 ds_read_b128: both W sub-blocks + A rows 0-63)  M_A (16 MFMA)
    // R_B (8 ds_read_b128: A rows 64-127)  M_B (16 MFMA).  g1 runs ONE barrier behind g0, so on
    // every SIMD one wave issues MFMAs while the other one reads LDS / issues DMA.
    // Global barrier numbering B(j); g0 executes its j-th barrier as B(j), g1 as B(j+1):
    //   g0:        R_A(kt) B(4kt)  M_A(kt) B(4kt+1) R_B(kt) B(4kt+2) M_B(kt) B(4kt+3)
    //   g1: B(4kt) R_A(kt) B(4kt+1) M_A(kt) B(4kt+2) R_B(kt) B(4kt+3) M_B(kt)
    // Ring-slot release (last ds_read of buffer kt&1, lgkmcnt(0) precedes every barrier):
    //   W halves: B(4kt+1)   A half 0: B(4kt+2)   A half 1: B(4kt+3)
    // so tile kt+2 is DMA'd into buffer kt&1 as  W(kt+2) after B(4kt+1), A0(kt+2) after B(4kt+2),
    // A1(kt+2) after B(4kt+3)  (2 + 2 + ... = 8 global_load_lds per thread and tile), and tile kt+1
    // must be complete before B(4kt+3): at that point the younger loads in flight are W(kt+2) and
    // A0(kt+2) = 6 instructions -> s_waitcnt vmcnt(6); loads stay in flight across barriers.
#define DK_BAR()                                          \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)
#define DK_READ_A(BASE, ROW0)                                                                       \
  if (!(ABL & 2) || kt == 0)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)   \
      xa[i][kk] = *(const bf16x8*)((BASE) + swz_off((ROW0) + i * 32 + l31, kk * 2 + hi));
#define DK_READ_W(BASE)                                                                             \
  if (!(ABL & 2) || kt == 0)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)   \
      wf[i][kk] = *(const bf16x8*)((BASE) + swz_off(i * 32 + l31, kk * 2 + hi));
#define DK_MFMA(MI0)                                                                                \
  do {                                                                                              \
    if (ABL & 4) {                                                                                  \
      _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                            \
        asm volatile("" ::"v"(wf[0][kk]), "v"(wf[1][kk]), "v"(xa[0][kk]), "v"(xa[1][kk]));          \
      }                                                                                             \
      break;                                                                                        \
    }                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[ni][(MI0) + i] =                          \
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni][kk], xa[i][kk], acc[ni][(MI0) + i], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                  \
  } while (0)

    // DMA is issued only in R sections (while the partner wave of the SIMD runs MFMAs):
    //   R_A(kt): A halves of tile kt+1 (their slots were released by B(4kt-2) / B(4kt-1))
    //   R_B(kt): W halves of tile kt+2 (released by B(4kt+1))
    // Tile kt+1 = {W(kt+1) from R_B(kt-1), A(kt+1) from R_A(kt)} must have landed before B(4kt+3); the
    // only younger loads of a wave at that point are W(kt+2) = 4 instructions -> vmcnt(4).
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) issue_half(0, hh);
    if (nk > 1) {
      issue_half(1, 2);
      issue_half(1, 3);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DK_BAR();  // tile 0 visible to every wave

    bf16x8 wf[2][4], xa[2][4];
    const int w_off = (2 + (wn >> 1)) * HALF_BYTES + (wn & 1) * 64 * 128;
    if (wm == 0) {
      for (int kt = 0; kt < nk; ++kt) {
        const char* buf = smem + (kt & 1) * KT_BYTES;
        if (kt + 1 < nk) { issue_half(kt + 1, 0); issue_half(kt + 1, 1); }
        DK_READ_W(buf + w_off)
        DK_READ_A(buf, 0)
        DK_BAR();  // B(4kt)
        DK_MFMA(0);
        DK_BAR();  // B(4kt+1)
        if (kt + 2 < nk) { issue_half(kt + 2, 2); issue_half(kt + 2, 3); }
        DK_READ_A(buf, 64)
        DK_BAR();  // B(4kt+2)
        DK_MFMA(2);
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DK_BAR();  // B(4kt+3)
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        const char* buf = smem + (kt & 1) * KT_BYTES;
        DK_BAR();  // B(4kt)
        if (kt + 1 < nk) { issue_half(kt + 1, 0); issue_half(kt + 1, 1); }
        DK_READ_W(buf + w_off)
        DK_READ_A(buf + HALF_BYTES, 0)
        DK_BAR();  // B(4kt+1)
        DK_MFMA(0);
        DK_BAR();  // B(4kt+2)
        if (kt + 2 < nk) { issue_half(kt + 2, 2); issue_half(kt + 2, 3); }
        DK_READ_A(buf + HALF_BYTES, 64)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DK_BAR();  // B(4kt+3)
        DK_MFMA(2);
      }
    }
#undef DK_BAR
#undef DK_READ_A
#undef DK_READ_W
#undef DK_MFMA
  }

  dk_epilogue256(p, acc, m0, n0, wm, wn, hi, l31);
}

int dk_launch_gemm256(const GemmParams& p, int variant, hipStream_t stream) {
  DK_REQUIRE(!p.conv, "gemm256 is a plain GEMM");
  DK_REQUIRE(p.K % BK == 0 && p.N % 4 == 0 && p.ldc % 4 == 0 && p.lda % 8 == 0, "gemm256 alignment");
  static bool attr_set = false;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#ifdef DK_LAB_ABLATIONS
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<0, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256_bf16_kernel<1, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#endif
    attr_set = true;
  }
  const int nbm = (p.M + T256 - 1) / T256, nbn = (p.N + T256 - 1) / T256;
  dim3 grid(nbm * nbn), block(512);
  dk_prof_begin(0, 2.0 * (double)p.M * (double)p.N * (double)p.K, stream);
  if (variant == 0)
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<0>, grid, block, LDS_BYTES, stream, p);
  else if (variant == 2)
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<2>, grid, block, LDS_BYTES, stream, p);
  else if (variant == 4)
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<3>, grid, block, LDS_BYTES, stream, p);
  else if (variant == 5)
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<4>, grid, block, LDS_BYTES, stream, p);
#ifdef DK_LAB_ABLATIONS
  else if (variant == 21) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<0, 1>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 22) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<0, 2>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 23) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<0, 3>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 24) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<0, 4>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 31) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<0, 11>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 11) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<1, 1>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 12) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<1, 2>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 13) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<1, 3>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 14) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<1, 4>), grid, block, LDS_BYTES, stream, p);
  else if (variant == 16) hipLaunchKernelGGL((dk_gemm256_bf16_kernel<1, 6>), grid, block, LDS_BYTES, stream, p);
#endif
  else
    hipLaunchKernelGGL(dk_gemm256_bf16_kernel<1>, grid, block, LDS_BYTES, stream, p);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
