// 256 x 256 x 64 bf16 MFMA GEMM, second generation: register-pipelined K loop + LDS-staged tail, in
// a tile-parallel and a stream-K (persistent) form.  Same contract as dk_gemm_bf16_kernel<0>
// (gemm.hip): C = epi(alpha * A . W^T + bias) for nn.Linear call sites
// python/src/diffusionkit/mlx/mmdit.py:821-832 (q/k/v/o, fc1/fc2) and the fused linear1 / linear2 of
// the single-stream blocks (:693-751).
//
// Restrictions (checked by the launcher; other shapes use the 128^2 kernel): M % 256 == 0,
// N % 256 == 0, K % 64 == 0, and every row-segment length (a/c/r/gate) is a multiple of 256 or >= M,
// so that a tile never straddles a segment: the segment maps are evaluated once per tile on the
// scalar unit.
//
// K loop: 8 waves (2 x 4), wave tile 128 x 64, v_mfma_f32_32x32x16_bf16 with swapped operands;
// global_load_lds_dwordx4 into a 2-deep ring of K-tiles (XOR-swizzled on the source side), fragments
// double-buffered in registers with inline-asm ds_read_b128 and hand-counted s_waitcnt lgkmcnt(6)
// (see gemm256.hip VARIANT 4 for why).
//
// Tail: the accumulators go through LDS once (per wave a private, XOR-swizzled [128][32] fp32 image per
// 32-column half), and are read back row-major, so that every global access of the tail -- residual
// loads, bf16 stores, stream-K slab traffic -- is a 64 / 128-byte contiguous run per row instead of
// 8 bytes per lane at a row stride (the row-per-lane MFMA layout), and the tail needs a handful of
// registers instead of keeping 128 accumulators live across three different consumers.
//
// Stream-K: a persistent grid of G = #CU workgroups splits the T * nk K-tile iterations evenly.
// A workgroup walks its contiguous range tile by tile; a segment that starts inside a tile (always
// the FIRST segment of a workgroup) writes its fp32 partial tile to the workgroup's slab and
// publishes a flag; the segment that starts a tile is the tile's finisher: it adds the slabs of the
// following workgroups and runs the epilogue.  Producers never wait and a finisher only waits for
// segments that other workgroups compute first, so there is no circular wait while all G
// workgroups are resident (one per CU by LDS).  Hand-off per guide G16 (R1): write-through (sc1) slab
// stores, every wave s_waitcnt vmcnt(0), barrier, one relaxed agent-scope flag store -- deferred to the
// first tile barrier of the workgroup's next segment so that the store drain hides under its DMA;
// consumer: relaxed poll, one agent-scope acquire, barrier, plain loads.  Flags are reset by their consumer.
#include <cstring>

#include "dk_kernels.h"

#define T256 256
#define BK 64
#define HALF_BYTES (128 * BK * 2)
#define KT_BYTES (4 * HALF_BYTES)
#define LDS_BYTES (2 * KT_BYTES)
#define SLAB_FLOATS (256 * 256)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) char lds_char;

struct SkArgs {
  float* slabs;          // [G][256*256] fp32, row-major tile images
  unsigned* flags;       // [G]
  unsigned* error_word;  // set to 1 when a bounded spin gave up
  int G;
};

// 16-byte write-through (sc1) store: the slab reaches memory without an agent-scope release fence
// (buffer_wbl2 would write back every dirty line of the XCD's L2, other workgroups' C tiles included)
__device__ __forceinline__ void store_sc1_b128(float* ptr, f32x4 v) {
#ifdef DK_SK_PLAINST  /* lab: timing with plain write-back stores (hand-off not guaranteed) */
  *(f32x4*)ptr = v;
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#endif
}

__device__ __forceinline__ int swz128(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// Grouped form: the launch covers the tiles of problem `pa` followed by the tiles of an optional second
// problem `pb` with the same N, K and epilogue kind but its own operands, row maps and M (the text
// stream of a double block next to its image stream: mmdit.py:568-675 runs the same Linear shapes on
// both).  The 12 + 192 tiles of the two streams then share one wave of the 256 CUs instead of the
// text stream running alone on 48 small tiles.
// SCHED 0: the 8 DMA instructions of a K-tile are issued together after the tile barrier.
// SCHED 1: they are spread, one in front of every second MFMA, over the two k-steps that follow the
// barrier, and the two wave groups of a SIMD (wm = 0 / 1) place them on opposite MFMA slots, so that
// while one wave is busy issuing a global_load_lds (~100 cycles) its partner keeps the MFMA pipe fed.
template <bool STREAMK, int SCHED>
__global__ __launch_bounds__(512, 2) void dk_gemm256v2_kernel(GemmParams pa, GemmParams pb, int tiles_a, int tiles_b, SkArgs sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nk = pa.K / BK;
  const long total = (long)(tiles_a + tiles_b) * nk;
  const int G = STREAMK ? sk.G : tiles_a + tiles_b;
  int v;  // XCD-contiguous workgroup index: neighbouring tiles share an L2
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = G >> 3, r = G & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  auto wg_start = [&](int u) { return STREAMK ? (long)u * total / G : (long)u * nk; };
  long it = wg_start(v);
  const long it_end = wg_start(v + 1);

  // ---- lane-constant parts of the LDS fragment addresses ----
  const int srow = lane >> 3;
  unsigned offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = (unsigned)swz128(l31, kk * 2 + hi);
  const unsigned sA = wm * HALF_BYTES;
  const unsigned sW = (2 + (wn >> 1)) * HALF_BYTES + (wn & 1) * 64 * 128;

  bool publish_pending = false;  // slab stores issued, flag not yet published
  while (it < it_end) {
    const int tile = (int)(it / nk);
    const bool second = tile >= tiles_a;
    const GemmParams& p = second ? pb : pa;
    const int tl = second ? tile - tiles_a : tile;  // tile index inside its problem
    const int nbm = p.M / T256, nbn = p.N / T256;
    unsigned la[2], lw[2];  // lane part of the A / W source byte offset for DMA instruction j
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j);  // = (lane&7) ^ (((wave*16 + j*8 + srow) >> 1) & 7)
      la[j] = ((unsigned)srow * (unsigned)p.lda + chunk * 8) * 2u;
      lw[j] = ((unsigned)srow * (unsigned)p.ldw + chunk * 8) * 2u;
    }
    const int kb = (int)(it - (long)tile * nk);
    const int ke = (int)min((long)nk, kb + (it_end - it));
    const int nseg = ke - kb;
    const int GROUP = 4;
    const int tpg = GROUP * nbn;
    const int g = tl / tpg;
    const int first_m = g * GROUP;
    const int gsz = min(nbm - first_m, GROUP);
    const int tm = first_m + (tl % tpg) % gsz;
    const int tn = (tl % tpg) / gsz;
    const int m0 = tm * T256, n0 = tn * T256;

    // tile-uniform source bases (bytes): rows m0 + hh*128 + wave*16 + j*8 (+ srow in the lane part)
    const size_t physA0 = (size_t)((m0 / p.a_seg_len) * p.a_seg_stride + (m0 % p.a_seg_len));
    const char* gA = (const char*)p.A + ((physA0 + wave * 16) * (size_t)p.lda + (size_t)kb * BK) * 2;
    const char* gW = (const char*)p.W + (((size_t)n0 + wave * 16) * (size_t)p.ldw + (size_t)kb * BK) * 2;
    const size_t a128 = (size_t)128 * p.lda * 2, a8 = (size_t)8 * p.lda * 2;
    const size_t w128 = (size_t)128 * p.ldw * 2, w8 = (size_t)8 * p.ldw * 2;
    auto issue_tile = [&](int i) {  // i-th K-tile of this segment -> ring slot i & 1
      const unsigned dst0 = (i & 1) * KT_BYTES + (wave * 16) * 128;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gA + hh * a128 + j * a8 + (size_t)i * (BK * 2) + la[j]),
                                           (lds_ptr_t)((lds_char*)0 + dst0 + hh * HALF_BYTES + j * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gW + hh * w128 + j * w8 + (size_t)i * (BK * 2) + lw[j]),
                                           (lds_ptr_t)((lds_char*)0 + dst0 + (2 + hh) * HALF_BYTES + j * 1024), 16, 0, 0);
        }
    };

    auto issue_piece = [&](int i, int gidx) {  // one of the 8 DMA instructions of K-tile i: (operand, half, j)
      const int op = gidx & 1, hh = (gidx >> 1) & 1, j = gidx >> 2;
      const unsigned dst0 = (i & 1) * KT_BYTES + (wave * 16) * 128;
      if (op == 0)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gA + hh * a128 + j * a8 + (size_t)i * (BK * 2) + la[j]),
                                         (lds_ptr_t)((lds_char*)0 + dst0 + hh * HALF_BYTES + j * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gW + hh * w128 + j * w8 + (size_t)i * (BK * 2) + lw[j]),
                                         (lds_ptr_t)((lds_char*)0 + dst0 + (2 + hh) * HALF_BYTES + j * 1024), 16, 0, 0);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---------------- K loop: register-pipelined, hand-counted LDS waits ----------------
#define DK_LDS_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR))
#define DK_RD(SET, BUFOFF, KK)                 \
  do {                                         \
    const unsigned aA_ = offk[KK] + sA + (BUFOFF); \
    const unsigned aW_ = offk[KK] + sW + (BUFOFF); \
    DK_LDS_RD(wf##SET[0], aW_, 0);             \
    DK_LDS_RD(wf##SET[1], aW_, 4096);          \
    DK_LDS_RD(xf##SET[0], aA_, 0);             \
    DK_LDS_RD(xf##SET[1], aA_, 4096);          \
    DK_LDS_RD(xf##SET[2], aA_, 8192);          \
    DK_LDS_RD(xf##SET[3], aA_, 12288);         \
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
// 8 MFMAs with NG DMA pieces (TILE, G0..G0+NG-1) in front of MFMA slots PH, PH+2, ... when ON
#define DK_MMG(SET, TILE, G0, NG, PH, ON)                                                                       \
  do {                                                                                                          \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                                          \
      if ((ON) && e_ >= (PH) && ((e_ - (PH)) & 1) == 0 && ((e_ - (PH)) >> 1) < (NG))                            \
        issue_piece((TILE), (G0) + ((e_ - (PH)) >> 1));                                                         \
      acc[e_ >> 2][e_ & 3] =                                                                                    \
          __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf##SET[e_ >> 2], xf##SET[e_ & 3], acc[e_ >> 2][e_ & 3], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
  } while (0)
#define DK_LOOP_SCHED1(PH, N3, N0, N1)                                                                          \
  for (int i = 0; i < nseg; ++i) {                                                                              \
    const unsigned bo = (i & 1) * KT_BYTES;                                                                     \
    const bool on1 = i >= 1 && i + 1 < nseg; /* second half of tile i+1 (first half went out in S3 of i-1) */    \
    DK_RD(1, bo, 1);                                                                                            \
    DK_WAIT(6, 0);                                                                                              \
    DK_MMG(0, i + 1, N3, N0, PH, on1);                                                                          \
    DK_RD(0, bo, 2);                                                                                            \
    DK_WAIT(6, 1);                                                                                              \
    DK_MMG(1, i + 1, N3 + N0, N1, PH, on1);                                                                     \
    DK_RD(1, bo, 3);                                                                                            \
    DK_WAIT(6, 0);                                                                                              \
    DK_MM(0);                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"                                                                \
                 : "+v"(wf1[0]), "+v"(wf1[1]), "+v"(xf1[0]), "+v"(xf1[1]), "+v"(xf1[2]), "+v"(xf1[3])           \
                 :                                                                                              \
                 : "memory");                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    if (STREAMK && publish_pending) {                                                                           \
      if (tid == 0) __hip_atomic_store(sk.flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           \
      publish_pending = false;                                                                                  \
    }                                                                                                           \
    if (i + 1 < nseg) DK_RD(0, bo ^ KT_BYTES, 0);                                                               \
    DK_MMG(1, i + 2, 0, N3, PH, i + 2 < nseg);                                                                  \
  }
    if (SCHED >= 1) {
      bf16x8 wf0[2], xf0[4], wf1[2], xf1[4];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_tile(0);
      if (nseg > 1) {
        issue_tile(1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // (the first fragment reads sit INSIDE the branches: an inline-asm load that is still in flight must not
      //  be live across a compiler-visible branch -- hipcc spilled it right after the ds_read, before the data
      //  had landed, and the wm = 1 waves computed k-step 0 of every tile from garbage)
      if (SCHED == 1) {  // 4 pieces behind the barrier (S3), 4 in the next k-step (S0)
        if (wm == 0) {
          DK_RD(0, 0u, 0);
          DK_LOOP_SCHED1(0, 4, 4, 0)
        } else {
          DK_RD(0, 0u, 0);
          DK_LOOP_SCHED1(1, 4, 4, 0)
        }
      } else {  // SCHED 2: 3 + 3 + 2 over S3, S0, S1
        if (wm == 0) {
          DK_RD(0, 0u, 0);
          DK_LOOP_SCHED1(0, 3, 3, 2)
        } else {
          DK_RD(0, 0u, 0);
          DK_LOOP_SCHED1(1, 3, 3, 2)
        }
      }
    } else {
      bf16x8 wf0[2], xf0[4], wf1[2], xf1[4];
      // every wave has finished the tail of the previous segment (it reads the ring region)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_tile(0);
      if (nseg > 1) {
        issue_tile(1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      DK_RD(0, 0u, 0);
      for (int i = 0; i < nseg; ++i) {
        const unsigned bo = (i & 1) * KT_BYTES;
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(wf1[0]), "+v"(wf1[1]), "+v"(xf1[0]), "+v"(xf1[1]), "+v"(xf1[2]), "+v"(xf1[3])
                     :
                     : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (STREAMK && publish_pending) {
          // the vmcnt(0) above also drained this wave's sc1 slab stores of the previous segment and
          // every wave has passed the barrier: the slab is complete in memory -> publish the flag
          if (tid == 0) __hip_atomic_store(sk.flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          publish_pending = false;
        }
        if (i + 1 < nseg) DK_RD(0, bo ^ KT_BYTES, 0);
        if (i + 2 < nseg) issue_tile(i + 2);
        DK_MM(1);
      }
    }
#undef DK_LDS_RD
#undef DK_RD
#undef DK_WAIT
#undef DK_MM
#undef DK_MMG
#undef DK_LOOP_SCHED1

    // ---------------- tail: accumulators -> LDS (wave-private image) -> row-major ----------------
    // All waves passed the last loop barrier after their final ds_read, so the ring is free.
    const bool producer = STREAMK && kb > 0;
    const bool out2 = p.n_split > 0 && n0 >= p.n_split;  // tile-uniform: second output of a column-split GEMM
    bf16_t* const Cb = out2 ? p.C2 : p.C;
    const int ldcb = out2 ? p.ldc2 : p.ldc;
    const int epi = out2 ? p.epi2 : p.epi;
    const int ncol0 = out2 ? n0 - p.n_split : n0;
    const size_t physC0 = (size_t)((m0 / p.c_seg_len) * p.c_seg_stride + (m0 % p.c_seg_len)) + wm * 128;
    const size_t physR0 = (size_t)((m0 / p.r_seg_len) * p.r_seg_stride + (m0 % p.r_seg_len)) + wm * 128;
    const bf16_t* gate_row = p.gate ? p.gate + (size_t)(m0 / p.gate_seg_len) * p.gate_stride : nullptr;
    const long tile_end = (long)(tile + 1) * nk;
    const unsigned reg0 = (unsigned)wave * 16384u;  // this wave's 16 KiB staging image
    const int rrow = lane >> 3, rchunk = lane & 7;   // read-back: 8 rows x 8 chunks of 16 B per instruction

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      // stage: lane owns row mi*32 + l31, columns 8*g4 + 4*hi + {0..3} of this 32-column half
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int row = mi * 32 + l31;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[ni][mi][4 * g4 + e];
          *(__attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((2 * g4 + hi) ^ (row & 7)) << 4)) = o;
        }
      // (same wave writes and reads the image: program order + the compiler's lgkmcnt suffice)
      const int col = n0 + wn * 64 + ni * 32 + rchunk * 4;       // column of the GEMM (bias, gate, residual)
      const int ocol = ncol0 + wn * 64 + ni * 32 + rchunk * 4;   // column inside the output it goes to
      float bias4[4] = {0.f, 0.f, 0.f, 0.f}, gate4[4] = {0.f, 0.f, 0.f, 0.f};
      if (!producer) {
        if (p.bias) {
          const uint2 bb = *(const uint2*)(p.bias + col);
          unpack2bf(bb.x, bias4[0], bias4[1]);
          unpack2bf(bb.y, bias4[2], bias4[3]);
        }
        if (epi == DK_EPI_GATE_RES) {
          const uint2 gg = *(const uint2*)(gate_row + col);
          unpack2bf(gg.x, gate4[0], gate4[1]);
          unpack2bf(gg.y, gate4[2], gate4[3]);
        }
      }
      if (!producer && STREAMK && ke < nk && ni == 0) {
        // finisher: wait (once per tile) for every contributing workgroup
        for (int u = v + 1; u < G && wg_start(u) < tile_end; ++u) {
          if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(sk.flags + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > (1u << 24)) {
                __hip_atomic_store(sk.error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
          }
        }
#ifndef DK_SK_NOACQ  /* lab: timing without the agent-scope acquire (results may be stale) */
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        __syncthreads();
      }
#pragma unroll 4
      for (int itr = 0; itr < 16; ++itr) {
        const int row = itr * 8 + rrow;  // row inside the wave's 128-row block
        f32x4 a = *(const __attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + ((rchunk ^ (row & 7)) << 4));
        const size_t slab_idx = (size_t)(wm * 128 + row) * 256 + wn * 64 + ni * 32 + rchunk * 4;
        if (producer) {
          store_sc1_b128(sk.slabs + (size_t)v * SLAB_FLOATS + slab_idx, a);
          continue;
        }
        if (STREAMK && ke < nk) {
          for (int u = v + 1; u < G && wg_start(u) < tile_end; ++u) {
            const f32x4 o = *(const f32x4*)(sk.slabs + (size_t)u * SLAB_FLOATS + slab_idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += o[e];
          }
        }
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = round_bf16(a[e] * p.alpha + bias4[e]);
        if (epi == DK_EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = gelu_erf_f(vv[e]);
        } else if (epi == DK_EPI_BIAS_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = silu_f(vv[e]);
        } else if (epi == DK_EPI_GATE_RES || epi == DK_EPI_RES) {
          const uint2 rr = *(const uint2*)(p.res + (physR0 + row) * (size_t)p.ldr + col);
          float r4[4];
          unpack2bf(rr.x, r4[0], r4[1]);
          unpack2bf(rr.y, r4[2], r4[3]);
          if (epi == DK_EPI_GATE_RES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = r4[e] + round_bf16(gate4[e] * vv[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] += r4[e];
          }
        }
        uint2 o2;
        o2.x = pack2bf(vv[0], vv[1]);
        o2.y = pack2bf(vv[2], vv[3]);
        *(uint2*)(Cb + (physC0 + row) * (size_t)ldcb + ocol) = o2;
      }
    }

    if (producer) {
      if (it + nseg < it_end) {
        publish_pending = true;  // published after the first tile barrier of the next segment (stores drain under its DMA)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have completed
        __syncthreads();
        if (tid == 0) __hip_atomic_store(sk.flags + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (STREAMK && ke < nk) {
      __syncthreads();  // every wave has read the slabs
      if (tid == 0)
        for (int u = v + 1; u < G && wg_start(u) < tile_end; ++u)
          __hip_atomic_store(sk.flags + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    it += nseg;
  }
}

int g_dk_v2_sched = 3;  // dk_tune_set("gemm_sched", v): 0 grouped DMA issue, 1 / 2 interleaved anti-phase (32x32x16 MFMA), 3 = gemm256v3.hip (16x16x32 MFMA, default)

bool dk_gemm256v2_eligible(const GemmParams& p) {
  auto seg_ok = [&](int len) { return len >= p.M || len % 256 == 0; };
  return !p.conv && p.n_split % 256 == 0 && (p.n_split == 0 || (p.C2 != nullptr && p.ldc2 % 4 == 0)) && p.M % 256 == 0 && p.N % 256 == 0 && p.K % BK == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 && p.ldc % 4 == 0 &&
         seg_ok(p.a_seg_len) && seg_ok(p.c_seg_len) && (p.res == nullptr || (seg_ok(p.r_seg_len) && p.ldr % 4 == 0)) &&
         (p.gate == nullptr || seg_ok(p.gate_seg_len)) && (size_t)p.lda * 2 * 8 < (1ull << 31) && (size_t)p.ldw * 2 * 8 < (1ull << 31);
}

size_t dk_streamk_workspace_bytes() { return (size_t)256 * SLAB_FLOATS * 4 + 4096; }

int dk_launch_gemm256v2(const GemmParams& p, const GemmParams* p2, bool streamk, hipStream_t stream) {
  DK_REQUIRE(dk_gemm256v2_eligible(p), "gemm256v2: shape / segment map not eligible");
  if (p2) {
    DK_REQUIRE(dk_gemm256v2_eligible(*p2), "gemm256v2: second problem not eligible");
    DK_REQUIRE(p2->N == p.N && p2->K == p.K && p2->epi == p.epi && p2->alpha == p.alpha, "grouped GEMM: N, K, epilogue must match");
  }
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v2_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v2_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v2_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v2_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    int dev = 0;
    DK_CHECK_HIP(hipGetDevice(&dev));
    DK_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_set = true;
  }
  SkArgs sk;
  memset(&sk, 0, sizeof(sk));
  const int tiles_a = (p.M / T256) * (p.N / T256);
  const int tiles_b = p2 ? (p2->M / T256) * (p2->N / T256) : 0;
  int grid = tiles_a + tiles_b;
  if (streamk) {
    DK_REQUIRE(p.workspace != nullptr && p.workspace_bytes >= dk_streamk_workspace_bytes(), "stream-K workspace missing or too small");
    DK_REQUIRE(((uintptr_t)p.workspace & 255) == 0, "stream-K workspace must be 256-byte aligned");
    sk.slabs = (float*)p.workspace;
    sk.flags = (unsigned*)((char*)p.workspace + (size_t)256 * SLAB_FLOATS * 4);
    sk.error_word = sk.flags + 512;
    const long total = (long)grid * (p.K / BK);
    int G = n_cu < 256 ? n_cu : 256;
    if ((long)G > total) G = (int)total;
    sk.G = G;
    grid = G;
  }
  double work = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  if (p2) work += 2.0 * (double)p2->M * (double)p2->N * (double)p2->K;
  dk_prof_begin(0, work, stream);
  const GemmParams& pb = p2 ? *p2 : p;
  const bool v3_ok = dk_gemm256v3_eligible(p) && (!p2 || dk_gemm256v3_eligible(*p2));  // (16-byte aligned outputs)
  if (!streamk && g_dk_v2_sched == 3 && v3_ok)
    dk_launch_gemm256v3_raw(p, pb, tiles_a, tiles_b, stream);
  else if (streamk)
    hipLaunchKernelGGL((dk_gemm256v2_kernel<true, 0>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sk);
  else if (g_dk_v2_sched == 1 || g_dk_v2_sched == 3)
    hipLaunchKernelGGL((dk_gemm256v2_kernel<false, 1>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sk);
  else if (g_dk_v2_sched == 2)
    hipLaunchKernelGGL((dk_gemm256v2_kernel<false, 2>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sk);
  else
    hipLaunchKernelGGL((dk_gemm256v2_kernel<false, 0>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sk);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
