// 256 x 256 x 128 fp8 (OCP e4m3) GEMM on the block-scaled MFMA of gfx950: fp8 weights / fp8-quantised activations, fp32
// accumulation, bf16 (or MX-fp8) output -- the Linear layers of the MMDiT blocks for BASELINE.json configs[3]
// ("FLUX.1-dev, fp8 weights / bf16 activations, CDNA4 fp8 MFMA").  Call sites: every nn.Linear of the transformer blocks,
// python/src/diffusionkit/mlx/mmdit.py:821-832 and the fused linear1 / linear2 of the single-stream blocks (:693-751);
// the reference's own quantised path is MLX's 4-bit nn.QuantizedLinear (mlx/model_io.py:728-734,772-775) -- this is the
// MI355X-native counterpart, not a restatement of it.
//
// Number formats
//   W   [N, K] e4m3 + one fp32 scale per output channel (wscale[n], applied in the tail); the hardware scale of the
//       weight operand is the constant 2^0.
//   A   [M, K] e4m3 + one E8M0 scale per (row, 32 consecutive K) -- the OCP MX block format.  The hardware applies it:
//       v_mfma_scale_f32_16x16x128_f8f6f4 takes, per lane, the scale of the lane's (row, 32-K block) from a byte of a VGPR.
//       Scale bytes live in a side array laid out so that ONE coalesced 8-byte load per lane and K-tile brings a wave the
//       scales of its 128 rows (dk_mx_scale_index below); producers: dk_ln_modulate_mx8, dk_quantize_mx8 (fp8_ops.hip)
//       and this kernel's own tail (fc1 + GELU -> fc2's input).
//   C   bf16 like the bf16 kernels, or MX-fp8 (same format as A) when the consumer is another fp8 GEMM.
//
// Kernel shape = the bf16 kernel's (gemm256v3.hip): 8 waves (2 x 4), wave tile 128 (m) x 64 (n), LDS-DMA
// (buffer_load_dwordx4 ... lds) into a ring of two activation + three weight slots of 32 KiB (see LDS_BYTES), hand-counted LDS waits, LDS-staged row-major tail.
// A K-tile is 128 fp8 = 128 bytes per row -- byte-for-byte the bf16 kernel's 64-element tile, so DMA, ring and bank
// behaviour carry over -- and ONE MFMA per (16 x 16) fragment pair consumes it whole: 32 MFMAs of 2 x 16 x 16 x 128 flop per
// wave and K-tile, each twice as long as a bf16 16x16x32 one, i.e. the same MFMA time per K-tile for twice the K depth.
//
// Operand fragment (pinned on hardware by scripts/fp8_probe.hip, profiles/r02_fp8_probe.log): lane (l15 = lane & 15,
// q = lane >> 4) holds row l15; its first four registers are K-elements [16 q, 16 q + 16), its last four [64 + 16 q, 64 + 16 q + 16)
// -- the instruction is two K = 64 halves -- and the scale register byte of lane (l15, g) is the E8M0 scale of row l15, K-elements
// [32 g, 32 g + 32).  Read j of a fragment therefore takes the 16-byte chunk 4 j + q of the row: exactly the bf16 kernel's
// conflict-free pattern (chunk (4 j + q) ^ swz(row)), and the 8 scale bytes a lane loads per K-tile are those of block q.
//
// K-tile schedule (4 steps of 8 MFMAs; A fragments in two sets of two, W fragments in ONE set of four that is reloaded
// fragment by fragment behind its last use):
//   S0  A1 <- m-frags 2,3     MFMA (n 0..3) x (m 0,1), n-major, waiting for W[n] one fragment at a time   + DMA of tile i+1
//   S1  A0 <- m-frags 4,5     MFMA x (m 2,3)
//   S2  A1 <- m-frags 6,7     MFMA x (m 4,5)
//       wait: own DMA pieces + scale load of tile i+1 landed, own reads done; tile barrier
//   S3  A0 <- tile i+1 m 0,1  MFMA x (m 6,7), n-major; W[n] <- tile i+1 right behind its two MFMAs  + DMA of tile i+2
#include <cstring>
#include <type_traits>

#include "dk_kernels.h"

#ifndef DK_F8_ABL
#define DK_F8_ABL 0  // lab only (timing ablations, results wrong): 1 no scale loads in the K loop, 2 no DMA in it, 4 no fragment reads in it
#endif
#ifndef DK_F8_SAFE
#define DK_F8_SAFE 0  // lab / debugging: every LDS wait is lgkmcnt(0)
#endif

#define T256 256
#define BKB 128                      // bytes (= fp8 elements) of a row per K-tile
#define HALF_BYTES (128 * BKB)       // 128 rows
#define OP_BYTES (2 * HALF_BYTES)    // one operand of one K-tile: rows 0-127, rows 128-255
// LDS ring: two slots of the activation operand, THREE of the weight operand (all 160 KiB): weights stream from HBM (every
// block of the model has its own), activations come out of L2 / Infinity Cache -- so the weight pieces of K-tile i+2 are issued
// 1.75 K-tiles ahead of their first read, the activation pieces 1.0 ahead (the in-order vmcnt lets the newest four stay in flight)
#define W_BASE (2 * OP_BYTES)
#define LDS_BYTES (5 * OP_BYTES)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// position of block `bid` inside the XCD-contiguous order of `count` blocks (hardware places block b on XCD b & 7)
__device__ __forceinline__ int f8_xcd_contiguous(int bid, int count) {
  const int x = bid & 7;
  int start = 0;
  for (int y = 0; y < x; ++y) start += y < count ? (count - y + 7) >> 3 : 0;
  return start + (bid >> 3);
}

// K split of a launch that is at most half a round of tiles (round 6; gemm256v3.hip's SplitArgs for the all-remainder case): every tile is cut into S
// equal K ranges, piece 0 = [0, ks) is the tile's finisher, pieces 1 .. S-1 share [ks, nk) and are producers (raw fp32 accumulators -> slab, flag).
// Block order = dispatch order: the finishers first, then the producers; tiles * S <= #CU, so every workgroup is resident and a finisher's wait ends.
struct F8Split {
  float* slabs;     // [tiles * (S - 1)][512 threads x 32 accumulator quads] fp32, thread-linear (16 bytes per thread and quad: coalesced)
  unsigned* flags;  // [tiles * (S - 1)], zero between launches (reset by the finisher)
  unsigned* error_word;
  int S;            // pieces per tile (0 / 1: no split)
  int ks;           // K-tiles of the finisher piece
};
#define F8_SLAB_FLOATS (256 * 256)

__global__ __launch_bounds__(512, 2) void dk_gemm256f8_kernel(GemmF8Params pa, GemmF8Params pb, int tiles_a, int tiles_b, F8Split sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, q = lane >> 4;

  const int n_tiles = tiles_a + tiles_b;
  const int piece = sp.S > 1 ? (int)blockIdx.x / n_tiles : -1;  // -1 whole tile, 0 finisher, >= 1 producer
  const int tile = f8_xcd_contiguous(sp.S > 1 ? (int)blockIdx.x - piece * n_tiles : (int)blockIdx.x, n_tiles);
  const bool second = tile >= tiles_a;
  // ONE scalar base into the kernel-argument segment for this tile's parameter block (round 4; gemm256v3.hip).  Written as
  // `second ? pb : pa` the compiler kept both blocks' M .. c_seg_len in SCRATCH and read them back with dynamically indexed scratch_load_dword
  // (105 sites, vector registers, memory latency at the head of every prologue and tail) and selected the rest field by field: 106 SGPRs + lane
  // spills.  Now 72 SGPRs, 232 VGPRs, no scratch: +7-12 % per launch on the FLUX shapes, FLUX.1-dev fp8 44.2 -> 42.7 ms per step
  // (profiles/r04_gemm_fp8_scalar_params.log, r04_gemm_kernarg_pointer.log).
  static_assert(sizeof(GemmF8Params) % 8 == 0 && alignof(GemmF8Params) == 8, "pb follows pa without padding");
  typedef const __attribute__((address_space(4))) GemmF8Params karg_params_t;
  const __attribute__((address_space(4))) char* kbase = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  karg_params_t& p = *(karg_params_t*)(kbase + (second ? sizeof(GemmF8Params) : 0));
  // (the twelve ints prologue and tail share, as named scalars)
  const int p_M = p.M;
  const int p_N = p.N;
  const int p_K = p.K;
  const int p_lda = p.lda;
  const int p_ldw = p.ldw;
  const int p_ldc = p.ldc;
  const int p_ldr = p.ldr;
  const int p_a_seg_len = p.a_seg_len;
  const int p_a_seg_stride = p.a_seg_stride;
  const int p_a_row0 = p.a_row0;
  const int p_sa_nblk = p.sa_nblk;
  const int p_c_seg_len = p.c_seg_len;
  const int tl = second ? tile - tiles_a : tile;
  const int nk_full = p_K / BKB;
  int k0 = 0, nk = nk_full;  // this workgroup's K-tile range [k0, k0 + nk)
  if (piece == 0) {
    nk = sp.ks;
  } else if (piece > 0) {
    const int rest = nk_full - sp.ks, np = sp.S - 1;
    k0 = sp.ks + rest * (piece - 1) / np;
    nk = sp.ks + rest * piece / np - k0;
  }
  const int nbm = (p_M + T256 - 1) / T256, nbn = p_N / T256;

  // lane-constant parts of the LDS fragment addresses: row l15 (+ 16 * fragment), read j = position (4 j + q) ^ swz(row)
  unsigned offk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) offk[j] = (unsigned)(l15 * 128 + (((j * 4 + q) ^ (l15 >> 1)) << 4));
  const unsigned sA = wm * HALF_BYTES;
  const unsigned sW = W_BASE + (wn >> 1) * HALF_BYTES + (wn & 1) * 64 * 128;

  const int srow = lane >> 3;
  const int GROUP = 4;
  const int tpg = GROUP * nbn;
  const int g = tl / tpg;
  const int first_m = g * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int tm = first_m + (tl % tpg) % gsz;
  const int tn = (tl % tpg) / gsz;
  const int m0 = tm * T256, n0 = tn * T256;

  // DMA sources: LDS position (lane & 7) of a row takes global chunk (lane & 7) ^ swz(row) (the read side XORs it back)
  unsigned la[2][2], lw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j);  // = (lane & 7) ^ swz(row), row = wave * 16 + j * 8 + srow
    lw[j] = (unsigned)srow * (unsigned)p_ldw + chunk * 16;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int m = min(m0 + hh * 128 + wave * 16 + j * 8 + srow, p_M - 1);
      const unsigned phys = (unsigned)((m / p_a_seg_len) * p_a_seg_stride + (m % p_a_seg_len));
      la[hh][j] = phys * (unsigned)p_lda + chunk * 16;
    }
  }
  const char* gA = (const char*)p.A + (size_t)k0 * BKB;
  const char* gW = (const char*)p.W + ((size_t)n0 + wave * 16) * (size_t)p_ldw + (size_t)k0 * BKB;
  const size_t w128 = (size_t)128 * p_ldw, w8 = (size_t)8 * p_ldw;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)gA, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)gW, 0, -1, 0x00020000);
  // the four DMA instructions (k: half, j) of one operand of K-tile i; `slot` = byte offset of the ring slot it goes to
  auto issue_a = [&](int i, int k) {
    const int hh = k & 1, j = k >> 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)((lds_char*)0 + (i & 1) * OP_BYTES + (wave * 16) * 128 + hh * HALF_BYTES + j * 1024),
                                             16, (int)la[hh][j], i * BKB, 0, 0);
  };
  auto issue_w = [&](int i, unsigned slot, int k) {
    const int hh = k & 1, j = k >> 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)((lds_char*)0 + W_BASE + slot + (wave * 16) * 128 + hh * HALF_BYTES + j * 1024), 16,
                                             (int)lw[j], (int)(hh * w128 + j * w8) + i * BKB, 0, 0);
  };

  // E8M0 scales of this wave's 128 A rows for one K-tile: 8 bytes per lane (byte mf = row mf * 16 + l15, block q), one
  // coalesced 512-byte read per wave (layout: dk_mx_scale_index).  The wave's rows must be one aligned 128-row block of the
  // physical buffer (checked by the launcher: segments and offsets are multiples of 128 rows).
  // (clamped to the last row block of M: the second wave row of a tile whose rows end at m0 + 128 owns no rows -- its results are
  //  masked -- and must not read scale bytes past the caller's side array)
  const int mrow0 = m0 + wm * 128;
  const int mrow_sc = min(mrow0, (p_M - 1) & ~127);
  const unsigned a_blk = (unsigned)(((mrow_sc / p_a_seg_len) * p_a_seg_stride + (mrow_sc % p_a_seg_len) + p_a_row0) >> 7);
  const size_t sa_step = (size_t)p_sa_nblk * 512;  // bytes between K-tiles
  const unsigned char* sa_ptr = p.SA + ((size_t)a_blk * 64 + lane) * 8 + (size_t)k0 * sa_step;
  u32x2 sa_cur, sa_nxt;

  f32x4 acc[4][8];  // [nf][mf]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#define F8_RD(DST, ADDR, OFF)                                                                                        \
  do {                                                                                                               \
    if (!(DK_F8_ABL & 4) || !in_loop) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR));       \
  } while (0)
// the two m-fragments (MFA, MFA + 1) of a set: 4 reads
#define F8_RDA(SET, BUFOFF, OFFA, OFFB)                 \
  do {                                                  \
    const unsigned a0_ = offk[0] + sA + (BUFOFF);       \
    const unsigned a1_ = offk[1] + sA + (BUFOFF);       \
    F8_RD(SET##lo[0], a0_, OFFA);                       \
    F8_RD(SET##hi[0], a1_, OFFA);                       \
    F8_RD(SET##lo[1], a0_, OFFB);                       \
    F8_RD(SET##hi[1], a1_, OFFB);                       \
  } while (0)
#define F8_RDW(NF, BUFOFF, OFF)                         \
  do {                                                  \
    F8_RD(wlo[NF], offk[0] + sW + (BUFOFF), OFF);       \
    F8_RD(whi[NF], offk[1] + sW + (BUFOFF), OFF);       \
  } while (0)
#if DK_F8_SAFE
#define F8_CNT(N) "0"
#else
#define F8_CNT(N) #N
#endif
#define F8_WAIT2(N, A, B) asm volatile("s_waitcnt lgkmcnt(" F8_CNT(N) ")" : "+v"(A), "+v"(B))
#define F8_WAIT4(N, A, B, C_, D_) asm volatile("s_waitcnt lgkmcnt(" F8_CNT(N) ")" : "+v"(A), "+v"(B), "+v"(C_), "+v"(D_))
#define F8_WAIT6(N, A, B, C_, D_, E_, F_) \
  asm volatile("s_waitcnt lgkmcnt(" F8_CNT(N) ")" : "+v"(A), "+v"(B), "+v"(C_), "+v"(D_), "+v"(E_), "+v"(F_))
// the wait in front of the tile barrier: own fragment reads, own DMA pieces of the next K-tile and its scale load; the four
// weight pieces of K-tile i+2 (the newest four memory operations) stay in flight
#define F8_TILE_WAIT(KEEP, A, B, C_, D_, S_) \
  asm volatile("s_waitcnt vmcnt(" #KEEP ") lgkmcnt(0)" : "+v"(A), "+v"(B), "+v"(C_), "+v"(D_), "+v"(S_)::"memory")
#define F8_FRAG(LO, HI) __builtin_shufflevector(LO, HI, 0, 1, 2, 3, 4, 5, 6, 7)
// one MFMA: acc[NF][MF] += W[NF] . A(SET)[AI]; scale of the activation rows = byte (MF & 3) of sa_cur[MF >> 2]
#define F8_MM(NF, SET, AI, MF)                                                                                              \
  do {                                                                                                                      \
    acc[NF][MF] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(F8_FRAG(wlo[NF], whi[NF]), F8_FRAG(SET##lo[AI], SET##hi[AI]), \
                                                                   acc[NF][MF], 0, 0, 0, 0x7F7F7F7F, (MF) & 3, (int)sa_cur[(MF) >> 2]); \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
  } while (0)
// a DMA piece in front of MFMA slot SLOT of a step, when this wave group's phase PH puts it there (W: weight pieces, A: activation)
#define F8_PIECE_ON(ON, SLOT, PH) ((ON) && !(DK_F8_ABL & 2) && (SLOT) >= (PH) && (((SLOT) - (PH)) & 1) == 0 && (((SLOT) - (PH)) >> 1) < 4)
#define F8_PIECE_W(ON, TILE, SLOT, PH)                                             \
  do {                                                                             \
    if (F8_PIECE_ON(ON, SLOT, PH)) issue_w((TILE), wo_nn, ((SLOT) - (PH)) >> 1);   \
  } while (0)
#define F8_PIECE_A(ON, TILE, SLOT, PH)                                             \
  do {                                                                             \
    if (F8_PIECE_ON(ON, SLOT, PH)) issue_a((TILE), ((SLOT) - (PH)) >> 1);          \
  } while (0)
// One K-tile.  ON1 / ON2 (compile-time): whether the scales of K-tile i+1 / the DMA pieces of K-tile i+2 (weights in S0,
// activations in S3 -- in that program order, see F8_TILE_WAIT) are issued -- false only in the last two K-tiles, so that the
// steady-state loop carries no branches around them.  wo_cur / wo_nxt / wo_nn: ring slots of the weights of K-tiles i, i+1, i+2.
#define F8_ITER(PH, ON1, ON2)                                                                                       \
  {                                                                                                                 \
    constexpr bool in_loop = true;                                                                                  \
    const unsigned bo = (i & 1) * OP_BYTES;                                                                         \
    /* ---- S0: (n 0..3) x (m 0,1) ---- */                                                                         \
    F8_WAIT6(6, a0lo[0], a0hi[0], a0lo[1], a0hi[1], wlo[0], whi[0]);                                                \
    F8_RDA(a1, bo, 4096, 6144);                                                                                     \
    if ((ON1) && !(DK_F8_ABL & 1)) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sa_nxt) : "v"(sa_ptr + (size_t)(i + 1) * sa_step) : "memory"); \
    F8_PIECE_W(ON2, i + 2, 0, PH); F8_MM(0, a0, 0, 0);                                                             \
    F8_PIECE_W(ON2, i + 2, 1, PH); F8_MM(0, a0, 1, 1);                                                             \
    F8_WAIT2(8, wlo[1], whi[1]);                                                                                    \
    F8_PIECE_W(ON2, i + 2, 2, PH); F8_MM(1, a0, 0, 0);                                                             \
    F8_PIECE_W(ON2, i + 2, 3, PH); F8_MM(1, a0, 1, 1);                                                             \
    F8_WAIT2(6, wlo[2], whi[2]);                                                                                    \
    F8_PIECE_W(ON2, i + 2, 4, PH); F8_MM(2, a0, 0, 0);                                                             \
    F8_PIECE_W(ON2, i + 2, 5, PH); F8_MM(2, a0, 1, 1);                                                             \
    F8_WAIT2(4, wlo[3], whi[3]);                                                                                    \
    F8_PIECE_W(ON2, i + 2, 6, PH); F8_MM(3, a0, 0, 0);                                                             \
    F8_PIECE_W(ON2, i + 2, 7, PH); F8_MM(3, a0, 1, 1);                                                             \
    /* ---- S1: x (m 2,3) ---- */                                                                                   \
    F8_RDA(a0, bo, 8192, 10240);                                                                                    \
    F8_WAIT4(4, a1lo[0], a1hi[0], a1lo[1], a1hi[1]);                                                                \
    F8_MM(0, a1, 0, 2); F8_MM(0, a1, 1, 3); F8_MM(1, a1, 0, 2); F8_MM(1, a1, 1, 3);                                 \
    F8_MM(2, a1, 0, 2); F8_MM(2, a1, 1, 3); F8_MM(3, a1, 0, 2); F8_MM(3, a1, 1, 3);                                 \
    /* ---- S2: x (m 4,5) ---- */                                                                                   \
    F8_RDA(a1, bo, 12288, 14336);                                                                                   \
    F8_WAIT4(4, a0lo[0], a0hi[0], a0lo[1], a0hi[1]);                                                                \
    F8_MM(0, a0, 0, 4); F8_MM(0, a0, 1, 5); F8_MM(1, a0, 0, 4); F8_MM(1, a0, 1, 5);                                 \
    F8_MM(2, a0, 0, 4); F8_MM(2, a0, 1, 5); F8_MM(3, a0, 0, 4); F8_MM(3, a0, 1, 5);                                 \
    if (ON2) F8_TILE_WAIT(4, a1lo[0], a1hi[0], a1lo[1], a1hi[1], sa_nxt);                                           \
    else F8_TILE_WAIT(0, a1lo[0], a1hi[0], a1lo[1], a1hi[1], sa_nxt);                                               \
    __builtin_amdgcn_s_barrier();                                                                                   \
    asm volatile("" ::: "memory");                                                                                  \
    /* ---- S3: x (m 6,7); next tile's first A set and, fragment by fragment, its W set ---- */                     \
    F8_RDA(a0, bo ^ OP_BYTES, 0, 2048);  /* unconditional: after the last tile these read stale ring data nobody uses */ \
    F8_PIECE_A(ON2, i + 2, 0, PH); F8_MM(0, a1, 0, 6);                                                             \
    F8_PIECE_A(ON2, i + 2, 1, PH); F8_MM(0, a1, 1, 7);                                                             \
    F8_RDW(0, wo_nxt, 0);                                                                                       \
    F8_PIECE_A(ON2, i + 2, 2, PH); F8_MM(1, a1, 0, 6);                                                             \
    F8_PIECE_A(ON2, i + 2, 3, PH); F8_MM(1, a1, 1, 7);                                                             \
    F8_RDW(1, wo_nxt, 2048);                                                                                    \
    F8_PIECE_A(ON2, i + 2, 4, PH); F8_MM(2, a1, 0, 6);                                                             \
    F8_PIECE_A(ON2, i + 2, 5, PH); F8_MM(2, a1, 1, 7);                                                             \
    F8_RDW(2, wo_nxt, 4096);                                                                                    \
    F8_PIECE_A(ON2, i + 2, 6, PH); F8_MM(3, a1, 0, 6);                                                             \
    F8_PIECE_A(ON2, i + 2, 7, PH); F8_MM(3, a1, 1, 7);                                                             \
    F8_RDW(3, wo_nxt, 6144);                                                                                    \
    sa_cur = sa_nxt;                                                                                                \
    { const unsigned t_ = wo_cur; wo_cur = wo_nxt; wo_nxt = wo_nn; wo_nn = t_; }                                    \
  }
// everything in flight at a section boundary is waited for there (an inline-asm load must not be live across a
// compiler-visible merge)
#define F8_DRAIN()                                                                                                  \
  do {                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0lo[0]), "+v"(a0hi[0]), "+v"(a0lo[1]), "+v"(a0hi[1]));              \
    asm volatile("" : "+v"(wlo[0]), "+v"(whi[0]), "+v"(wlo[1]), "+v"(whi[1]));                                      \
    asm volatile("" : "+v"(wlo[2]), "+v"(whi[2]), "+v"(wlo[3]), "+v"(whi[3]));                                      \
  } while (0)
#define F8_DRIVE(PH)                                     \
  {                                                      \
    int i = 0;                                           \
    for (; i + 2 < nk; ++i) F8_ITER(PH, true, true)      \
    F8_DRAIN();                                          \
    if (i + 1 < nk) {                                    \
      F8_ITER(PH, true, false)                           \
      ++i;                                               \
      F8_DRAIN();                                        \
    }                                                    \
    F8_ITER(PH, false, false)                            \
    F8_DRAIN();                                          \
  }

  {
    i32x4 wlo[4], whi[4], a0lo[2], a0hi[2], a1lo[2], a1hi[2];
    // prologue: scales of K-tile 0, K-tile 0 completely, then K-tile 1 (the loop's first wait lets only its own four weight
    // pieces, of K-tile 2, stay in flight)
    // (the asm load's destination must not be read -- not even copied -- before its wait: no branch-dependent statement names it,
    //  so that no merge of two definitions makes the compiler copy the register while the load is in flight; a first build did)
    unsigned wo_cur = 0u, wo_nxt = OP_BYTES, wo_nn = 2u * OP_BYTES;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sa_cur) : "v"(sa_ptr) : "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) issue_w(0, wo_cur, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) issue_a(0, k);
    if (nk > 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) issue_a(1, k);
#pragma unroll
      for (int k = 0; k < 4; ++k) issue_w(1, wo_nxt, k);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("" : "+v"(sa_cur));  // from here on the scales of K-tile 0 are there in either case
    sa_nxt = u32x2{0u, 0u};
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (the first fragment reads sit INSIDE the branches: an inline-asm load that is still in flight must not be live across
    //  a compiler-visible branch)
    constexpr bool in_loop = false;
    if (wm == 0) {
      F8_RDA(a0, 0u, 0, 2048);
      F8_RDW(0, wo_cur, 0); F8_RDW(1, wo_cur, 2048); F8_RDW(2, wo_cur, 4096); F8_RDW(3, wo_cur, 6144);
      F8_DRIVE(0)
    } else {
      F8_RDA(a0, 0u, 0, 2048);
      F8_RDW(0, wo_cur, 0); F8_RDW(1, wo_cur, 2048); F8_RDW(2, wo_cur, 4096); F8_RDW(3, wo_cur, 6144);
      F8_DRIVE(1)
    }
  }
#undef F8_RD
#undef F8_RDA
#undef F8_RDW
#undef F8_WAIT2
#undef F8_WAIT4
#undef F8_WAIT6
#undef F8_TILE_WAIT
#undef F8_MM
#undef F8_PIECE_ON
#undef F8_PIECE_W
#undef F8_PIECE_A
#undef F8_ITER
#undef F8_DRAIN
#undef F8_DRIVE

  // ---------------- K split: producers hand their raw accumulators to the tile's finisher (gemm256v3.hip's protocol: write-through slab stores,
  // vmcnt(0) in every wave, barrier, one relaxed agent-scope flag store; finisher: relaxed poll, one agent-scope acquire, barrier, plain loads).
  // The MX scales were applied inside the MFMAs, the weight scale and the bias are applied once, by the finisher's tail: partial sums simply add.
  if (piece >= 0) {
    const int n_prod = sp.S - 1;
    if (piece >= 1) {
      float* const slab = sp.slabs + (size_t)(tile * n_prod + piece - 1) * F8_SLAB_FLOATS + (size_t)tid * 4;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf)
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slab + (size_t)(nf * 8 + mf) * 2048), "v"(acc[nf][mf]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have completed
      __syncthreads();
      if (tid == 0) __hip_atomic_store(sp.flags + tile * n_prod + piece - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      for (int pp = 0; pp < n_prod; ++pp) {
        unsigned spins = 0;
        while (__hip_atomic_load(sp.flags + tile * n_prod + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 24)) {
            __hip_atomic_store(sp.error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    for (int pp = 0; pp < n_prod; ++pp) {
      const float* const slab = sp.slabs + (size_t)(tile * n_prod + pp) * F8_SLAB_FLOATS + (size_t)tid * 4;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
          const f32x4 o = *(const f32x4*)(slab + (size_t)(nf * 8 + mf) * 2048);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nf][mf][e] += o[e];
        }
    }
    __syncthreads();  // every wave has read the slabs
    if (tid == 0)
      for (int pp = 0; pp < n_prod; ++pp) __hip_atomic_store(sp.flags + tile * n_prod + pp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---------------- tail: accumulators -> LDS (wave-private image) -> row-major ----------------
  // All waves passed the last loop barrier after their final ds_read of live data, so the ring is free.
  const bool out2 = p.n_split > 0 && n0 >= p.n_split;  // tile-uniform: second output of a column-split GEMM
  void* const Cb = out2 ? p.C2 : p.C;
  const int ldcb = out2 ? p.ldc2 : p_ldc;
  const int epi = out2 ? p.epi2 : p.epi;
  const bool out_mx8 = out2 ? p.c2_mx8 != 0 : p.c_mx8 != 0;
  const int ncol0 = out2 ? n0 - p.n_split : n0;
  const bool has_res = epi == DK_EPI_GATE_RES || epi == DK_EPI_RES;
  auto inside = [&](int len) { return m0 / len == (m0 + T256 - 1) / len; };
  const bool fast = m0 + T256 <= p_M && inside(p_c_seg_len) && (!has_res || inside(p.r_seg_len)) &&
                    (epi != DK_EPI_GATE_RES || inside(p.gate_seg_len));
  const size_t physC0 = (size_t)((m0 / p_c_seg_len) * p.c_seg_stride + (m0 % p_c_seg_len)) + wm * 128;
  const size_t physR0 = has_res ? (size_t)((m0 / p.r_seg_len) * p.r_seg_stride + (m0 % p.r_seg_len)) + wm * 128 : 0;
  const bf16_t* gate_row = epi == DK_EPI_GATE_RES ? p.gate + (size_t)(m0 / p.gate_seg_len) * p.gate_stride : nullptr;
  const unsigned reg0 = (unsigned)wave * 16384u;  // this wave's 16 KiB staging image
  const int rrow = lane >> 2, rc2 = (lane & 3) * 2;

  auto unpack8 = [](const uint4 v, float* f) {
    unpack2bf(v.x, f[0], f[1]);
    unpack2bf(v.y, f[2], f[3]);
    unpack2bf(v.z, f[4], f[5]);
    unpack2bf(v.w, f[6], f[7]);
  };

  // the weight scales and the bias of this lane's 16 columns (4 per 16-column fragment), all loads up front
  f32x4 ws_q[4];
  u32x2 bias_q[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}};
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    ws_q[nf] = *(const f32x4*)(p.wscale + n0 + wn * 64 + nf * 16 + 4 * q);
    if (p.bias) bias_q[nf] = *(const u32x2*)(p.bias + n0 + wn * 64 + nf * 16 + 4 * q);
  }
  // Key tile of a q / k / v projection with the fused QKNorm + RoPE (gemm256v3.hip has the bf16 twin): the sum of squares of every
  // row over its head's columns -- this wave's 64 columns from the accumulators, for 128-column heads plus the partner wave's
  // through LDS (behind the staging images)
  const bool qtile = p.qn_w != nullptr && !out2 && n0 >= p.qn_col0 && n0 < p.qn_col1;  // query tile (round 4)
  const bool kfuse = p.kn_w != nullptr && !out2 && ((n0 >= p.kn_col0 && n0 < p.kn_col1) || qtile);  // tile-uniform
  const bf16_t* const nw = qtile ? p.qn_w : p.kn_w;
  constexpr unsigned XCH_OFF = 8u * 16384u;
  if (kfuse) {
#pragma unroll
    for (int mf = 0; mf < 8; ++mf) {
      float ss = 0.f;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        float b4[4];
        unpack2bf(bias_q[nf][0], b4[0], b4[1]);
        unpack2bf(bias_q[nf][1], b4[2], b4[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = round_bf16(acc[nf][mf][e] * ws_q[nf][e] + b4[e]);
          ss += v * v;
        }
      }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (q == 0) *(__attribute__((address_space(3))) float*)((lds_char*)0 + XCH_OFF + (wave * 128 + mf * 16 + l15) * 4) = ss;
    }
    __syncthreads();
  }
  // the two 32-column halves of the wave tile (compile-time index: the accumulators must stay in registers)
  auto tail_pass = [&](auto ni_c) {
    constexpr int ni = decltype(ni_c)::value;
    // stage round_bf16(wscale * acc + bias) -- what every epilogue starts from -- as bf16: lane owns row mf*16 + l15, columns
    // (nf & 1)*16 + 4*q + {0..3} of this 32-column half; 64-byte rows, 16-byte chunk c at position c ^ ((row >> 2) & 3)
    // (conflict-free for these 8-byte writes and the 16-byte read-back)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      float b4[4];
      unpack2bf(bias_q[ni * 2 + nf][0], b4[0], b4[1]);
      unpack2bf(bias_q[ni * 2 + nf][1], b4[2], b4[3]);
      const f32x4 w4 = ws_q[ni * 2 + nf];
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) {
        const int row = mf * 16 + l15;
        const f32x4 a = acc[ni * 2 + nf][mf];
        const unsigned lo = pack2bf(a[0] * w4[0] + b4[0], a[1] * w4[1] + b4[1]), hi = pack2bf(a[2] * w4[2] + b4[2], a[3] * w4[3] + b4[3]);
        *(__attribute__((address_space(3))) u32x2*)((lds_char*)0 + reg0 + row * 64 + (((nf * 2 + (q >> 1)) ^ ((row >> 2) & 3)) << 4) + (q & 1) * 8) = u32x2{lo, hi};
      }
    }
    const int col = n0 + wn * 64 + ni * 32 + rc2 * 4;      // first of this lane's 8 columns of the GEMM (gate, residual)
    const int ocol = ncol0 + wn * 64 + ni * 32 + rc2 * 4;  // the same inside the output it goes to
    float gate8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto rows = [&](auto fast_c, auto ek_c, auto mx_c, auto kf_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      constexpr bool KF = decltype(kf_c)::value;  // key tile with the fused QKNorm + RoPE (bias-only epilogue, bf16 out)
      float kw8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int kcol = KF ? col % p.kn_D : 0;  // first of this lane's 8 columns inside its head (the ranges start at multiples of 256)
      if (KF) unpack8(*(const uint4*)(nw + kcol), kw8);
      constexpr int EK = decltype(ek_c)::value;  // the epilogue as a compile-time constant (straight-line row loop), or -1: `epi`
      constexpr int MX = decltype(mx_c)::value;  // output: 1 MX-fp8, 0 bf16, -1: `out_mx8`
      const int ep = EK >= 0 ? EK : epi;
      const bool hres = EK >= 0 ? (EK == DK_EPI_GATE_RES || EK == DK_EPI_RES) : has_res;
      const bool omx = MX >= 0 ? MX == 1 : out_mx8;
      if (FAST && ep == DK_EPI_GATE_RES) unpack8(*(const uint4*)(gate_row + col), gate8);
      int c_seg = 0, c_rem = 0, r_seg = 0, r_rem = 0, g_seg = 0, g_rem = 0;
      if (!FAST) {
        const int ms = mrow0 + rrow;
        c_seg = ms / p_c_seg_len, c_rem = ms % p_c_seg_len;
        if (hres) r_seg = ms / p.r_seg_len, r_rem = ms % p.r_seg_len;
        if (ep == DK_EPI_GATE_RES) g_seg = ms / p.gate_seg_len, g_rem = ms % p.gate_seg_len;
      }
      // MX-fp8 output on a tile-uniform row map: a lane's eight rows (itr) are the eight 16-row fragments of its 128-row block, and
      // their scale bytes are the eight consecutive bytes of ONE word of the side array (dk_mx_scale_index: byte (r / 16) % 8) --
      // collected in two registers over the (fully unrolled) loop, stored once per lane quad and 32-column half
      unsigned sc_lo = 0u, sc_hi = 0u;
      // FAST tiles: the residual rows (updated in place: the compiler may not move a row's load above the previous row's store) and the
      // cos / sin entries of a key tile are fetched BEFORE the row loop -- one memory round trip per pass instead of one per row (round 4,
      // gemm256v3.hip)
      constexpr bool PRE_RES = FAST && (EK == DK_EPI_GATE_RES || EK == DK_EPI_RES);
      constexpr bool PRE_ROPE = FAST && KF;
      uint4 res_pre[PRE_RES ? 8 : 1];
      f32x4 rope_pre[PRE_ROPE ? 16 : 1];
      if (PRE_RES) {
#pragma unroll
        for (int itr = 0; itr < 8; ++itr) res_pre[itr] = *(const uint4*)(p.res + (physR0 + itr * 16 + rrow) * (size_t)p_ldr + col);
      }
      if (PRE_ROPE) {
        if (p.kn_rope != nullptr) {
#pragma unroll
          for (int itr = 0; itr < 8; ++itr) {
            const int kpos_ = (mrow0 + itr * 16 + rrow) % p.kn_seg_len;
            const float* tab = p.kn_rope + ((size_t)(p.kn_pos_off + kpos_) * (size_t)(p.kn_D / 2) + (size_t)(kcol >> 1)) * 2;
            rope_pre[2 * itr] = *(const f32x4*)tab, rope_pre[2 * itr + 1] = *(const f32x4*)(tab + 4);
          }
        }
      }
#pragma unroll
      for (int itr = 0; itr < 8; ++itr) {
        const int row = itr * 16 + rrow;  // row inside the wave's 128-row block
        size_t crow = physC0 + row, rrow_phys = physR0 + row;
        bool valid = true;
        if (!FAST) {
          valid = mrow0 + row < p_M;
          crow = (size_t)c_seg * p.c_seg_stride + c_rem;
          rrow_phys = (size_t)r_seg * p.r_seg_stride + r_rem;
          if (ep == DK_EPI_GATE_RES && valid) unpack8(*(const uint4*)(p.gate + (size_t)g_seg * p.gate_stride + col), gate8);
          for (c_rem += 16; c_rem >= p_c_seg_len; c_rem -= p_c_seg_len) ++c_seg;
          if (hres)
            for (r_rem += 16; r_rem >= p.r_seg_len; r_rem -= p.r_seg_len) ++r_seg;
          if (ep == DK_EPI_GATE_RES)
            for (g_rem += 16; g_rem >= p.gate_seg_len; g_rem -= p.gate_seg_len) ++g_seg;
        }
        const u32x4 sv = *(const __attribute__((address_space(3))) u32x4*)((lds_char*)0 + reg0 + row * 64 + ((((unsigned)rc2 >> 1) ^ ((unsigned)(row >> 2) & 3u)) << 4));
        if (EK == DK_EPI_BIAS && MX == 0 && !KF) {  // bias only, bf16 out: the staged values ARE the output
          if (FAST || valid) *(u32x4*)((bf16_t*)Cb + crow * (size_t)ldcb + ocol) = sv;
          continue;
        }
        float vv[8];
        unpack8(make_uint4(sv[0], sv[1], sv[2], sv[3]), vv);
        if (KF) {
          const int kpos = (mrow0 + row) % p.kn_seg_len;  // the row's position inside its sequence
          float ss = *(const __attribute__((address_space(3))) float*)((lds_char*)0 + XCH_OFF + (wave * 128 + row) * 4);
          if (p.kn_D == 128) ss += *(const __attribute__((address_space(3))) float*)((lds_char*)0 + XCH_OFF + ((wave ^ 1) * 128 + row) * 4);
          const float r = rsqrtf(ss / (float)p.kn_D + p.kn_eps);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = round_bf16(vv[e] * r * kw8[e]);
          if (p.kn_rope != nullptr) {
            const float* tab = p.kn_rope + ((size_t)(p.kn_pos_off + kpos) * (size_t)(p.kn_D / 2) + (size_t)(kcol >> 1)) * 2;
            f32x4 t0 = {1.f, 0.f, 1.f, 0.f}, t1 = {1.f, 0.f, 1.f, 0.f};
            if (PRE_ROPE) t0 = rope_pre[2 * itr], t1 = rope_pre[2 * itr + 1];
            else if (valid) t0 = *(const f32x4*)tab, t1 = *(const f32x4*)(tab + 4);
            const float cs[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float c = cs[2 * i], sn = cs[2 * i + 1], xe = vv[2 * i], xo = vv[2 * i + 1];
              vv[2 * i] = c * xe - sn * xo;
              vv[2 * i + 1] = sn * xe + c * xo;
            }
          }
        }
        if (ep == DK_EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2 g2 = gelu_erf_f2(f32x2{vv[e], vv[e + 1]});
            vv[e] = g2[0], vv[e + 1] = g2[1];
          }
        } else if (ep == DK_EPI_BIAS_SILU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = silu_f(vv[e]);
        } else if (hres) {
          uint4 rr = make_uint4(0u, 0u, 0u, 0u);
          if (PRE_RES) rr = res_pre[itr];
          else if (FAST || valid) rr = *(const uint4*)(p.res + rrow_phys * (size_t)p_ldr + col);
          float r8[8];
          unpack8(rr, r8);
          if (ep == DK_EPI_GATE_RES) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = r8[e] + round_bf16(gate8[e] * vv[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] += r8[e];
          }
        }
        if (omx) {
          // MX-fp8 output: the four lanes of a row hold one 32-column block; values are rounded to bf16 first (what the bf16
          // path would have stored), then quantised -- 8 bytes per lane, one scale byte per block
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = round_bf16(vv[e]);
          unsigned e8;
          const uint2 q8 = dk_mx8_quantize8(vv, e8);
          if (FAST || valid) *(uint2*)((unsigned char*)Cb + crow * (size_t)ldcb + ocol) = q8;
          if (FAST) {
            if (itr < 4) sc_lo |= e8 << (8 * (itr & 3));
            else sc_hi |= e8 << (8 * (itr & 3));
          } else if (valid && (lane & 3) == 0) {
            p.SC[dk_mx_scale_index((unsigned)crow + (unsigned)p.c_row0, (unsigned)(p.sc_kb0 + (ocol >> 5)), (unsigned)p.sc_nblk)] = (unsigned char)e8;
          }
        } else {
          uint4 o4;
          o4.x = pack2bf(vv[0], vv[1]);
          o4.y = pack2bf(vv[2], vv[3]);
          o4.z = pack2bf(vv[4], vv[5]);
          o4.w = pack2bf(vv[6], vv[7]);
          if (FAST || valid) *(uint4*)((bf16_t*)Cb + crow * (size_t)ldcb + ocol) = o4;
        }
      }
      if (FAST && omx && (lane & 3) == 0) {
        // rows physC0 + c_row0 + rrow + 16 * {0..7}: one aligned 128-row block (the launcher checks the alignment), bytes 0..7
        const unsigned r0 = (unsigned)physC0 + (unsigned)p.c_row0 + (unsigned)rrow;
        const unsigned kb = (unsigned)(p.sc_kb0 + (ocol >> 5));
        *(uint2*)(p.SC + dk_mx_scale_index(r0, kb, (unsigned)p.sc_nblk)) = make_uint2(sc_lo, sc_hi);
      }
    };
    using Run = std::integral_constant<int, -1>;
    using No = std::false_type;
    using Yes = std::true_type;
    if (kfuse) {  // (bias-only epilogue, bf16 out -- checked by the launcher)
      if (fast)
        rows(Yes{}, std::integral_constant<int, DK_EPI_BIAS>{}, std::integral_constant<int, 0>{}, Yes{});
      else
        rows(No{}, std::integral_constant<int, DK_EPI_BIAS>{}, std::integral_constant<int, 0>{}, Yes{});
    } else if (fast) {
      // the model's combinations with epilogue and output kind folded at compile time (no scalar branches inside the row loop)
      if (epi == DK_EPI_BIAS && !out_mx8)
        rows(Yes{}, std::integral_constant<int, DK_EPI_BIAS>{}, std::integral_constant<int, 0>{}, No{});
      else if (epi == DK_EPI_BIAS_GELU && out_mx8)
        rows(Yes{}, std::integral_constant<int, DK_EPI_BIAS_GELU>{}, std::integral_constant<int, 1>{}, No{});
      else if (epi == DK_EPI_GATE_RES && !out_mx8)
        rows(Yes{}, std::integral_constant<int, DK_EPI_GATE_RES>{}, std::integral_constant<int, 0>{}, No{});
      else
        rows(Yes{}, Run{}, Run{}, No{});
    } else {
      rows(No{}, Run{}, Run{}, No{});
    }
  };
  tail_pass(std::integral_constant<int, 0>{});
  tail_pass(std::integral_constant<int, 1>{});
}

bool dk_gemm256f8_eligible(const GemmF8Params& p) {
  if (p.M <= 0 || p.N % 256 != 0 || p.K % BKB != 0 || p.lda % 16 != 0 || p.ldw % 16 != 0 || p.lda < p.K || p.ldw < p.K) return false;
  if (p.A == nullptr || p.W == nullptr || p.SA == nullptr || p.wscale == nullptr || p.C == nullptr) return false;
  if (p.n_split % 256 != 0 || (p.n_split > 0 && (p.C2 == nullptr || p.n_split >= p.N))) return false;
  if (p.a_seg_len <= 0 || p.c_seg_len <= 0) return false;
  // scale reads: every wave's 128 rows are one aligned 128-row block of the A buffer
  if (p.a_seg_len % 128 != 0 || p.a_seg_stride % 128 != 0 || p.a_row0 % 128 != 0 || p.sa_nblk <= 0) return false;
  const bool res1 = p.epi == DK_EPI_GATE_RES || p.epi == DK_EPI_RES;
  const bool res2 = p.n_split > 0 && (p.epi2 == DK_EPI_GATE_RES || p.epi2 == DK_EPI_RES);
  if ((res1 || res2) && (p.res == nullptr || p.r_seg_len <= 0 || p.ldr % 8 != 0)) return false;
  if ((p.epi == DK_EPI_GATE_RES || (p.n_split > 0 && p.epi2 == DK_EPI_GATE_RES)) && (p.gate == nullptr || p.gate_seg_len <= 0)) return false;
  if (p.kn_w != nullptr) {  // fused key QKNorm + RoPE: whole 256-column tiles of 128- or 64-column heads, bias-only bf16 first output
    if (p.epi != DK_EPI_BIAS || p.c_mx8 || (p.kn_D != 128 && p.kn_D != 64) || p.kn_seg_len <= 0 || p.kn_col0 % 256 != 0 || p.kn_col1 % 256 != 0 ||
        p.kn_col0 >= p.kn_col1 || p.kn_col1 > (p.n_split > 0 ? p.n_split : p.N) || ((uintptr_t)p.kn_w & 15) != 0 || ((uintptr_t)p.kn_rope & 15) != 0)
      return false;
    if (p.qn_w != nullptr && (p.qn_col0 % 256 != 0 || p.qn_col1 % 256 != 0 || p.qn_col0 >= p.qn_col1 || p.qn_col1 > (p.n_split > 0 ? p.n_split : p.N) ||
                              (p.qn_col0 < p.kn_col1 && p.kn_col0 < p.qn_col1) || ((uintptr_t)p.qn_w & 15) != 0))
      return false;
  } else if (p.qn_w != nullptr) {
    return false;  // the query side rides on the key side's machinery
  }
  // outputs: bf16 rows of 16-byte stores, or MX-fp8 rows of 8-byte stores with the scale side array
  auto al = [](const void* q, int a) { return ((uintptr_t)q & (uintptr_t)(a - 1)) == 0; };
  if (p.c_mx8 ? (p.ldc % 8 != 0 || !al(p.C, 8)) : (p.ldc % 8 != 0 || !al(p.C, 16))) return false;
  if (p.n_split > 0 && (p.c2_mx8 ? (p.ldc2 % 8 != 0 || !al(p.C2, 8)) : (p.ldc2 % 8 != 0 || !al(p.C2, 16)))) return false;
  if ((p.c_mx8 || (p.n_split > 0 && p.c2_mx8)) &&
      (p.SC == nullptr || p.sc_nblk <= 0 || p.c_seg_len % 128 != 0 || p.c_seg_stride % 128 != 0 || p.c_row0 % 128 != 0))
    return false;  // (the tail stores a 128-row block's eight scale bytes as one word)
  if (!al(p.res, 16) || !al(p.bias, 16) || !al(p.gate, 16) || !al(p.wscale, 16) || (p.gate != nullptr && p.gate_stride % 8 != 0)) return false;
  // 32-bit byte offsets on the DMA side
  const size_t a_rows = (size_t)((p.M - 1) / p.a_seg_len) * p.a_seg_stride + (size_t)((p.M - 1) % p.a_seg_len) + 1;
  return a_rows * (size_t)p.lda < (1ull << 32) && (size_t)p.ldw * 8 < (1ull << 31);
}

// tile-parallel launch of `p` and, optionally, a second problem `p2` with the same N, K and epilogues
int dk_launch_gemm256f8(const GemmF8Params& p, const GemmF8Params* p2, hipStream_t stream) {
  DK_REQUIRE(dk_gemm256f8_eligible(p), "gemm256f8: shape / strides / scale layout not eligible");
  if (p2) {
    DK_REQUIRE(dk_gemm256f8_eligible(*p2), "gemm256f8: second problem not eligible");
    DK_REQUIRE(p2->N == p.N && p2->K == p.K && p2->epi == p.epi && p2->n_split == p.n_split && p2->c_mx8 == p.c_mx8 &&
                   (p.n_split == 0 || (p2->epi2 == p.epi2 && p2->c2_mx8 == p.c2_mx8)),
               "grouped fp8 GEMM: N, K, epilogue must match");
  }
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256f8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_once.mark();
  }
  const int tiles_a = ((p.M + T256 - 1) / T256) * (p.N / T256);
  const int tiles_b = p2 ? ((p2->M + T256 - 1) / T256) * (p2->N / T256) : 0;
  // K split (round 6): a launch of at most half a round of tiles whose reduction is long enough -- the bf16 rule (gemm256v3.hip: plan_split, measured
  // break-even around 32 saved K-tile steps; an fp8 K-tile of 128 elements takes as long as a bf16 one of 64) -- FLUX's fc2 / linear2 below 1024 x 1024.
  // Never with the fused key QKNorm (a cut tile's finisher has no second pass over its row sums) and never without the caller's workspace.
  F8Split sp;
  memset(&sp, 0, sizeof(sp));
  {
    const int n_cu = dk_device_cu_count(), G = n_cu & ~7, tiles = tiles_a + tiles_b, nk = p.K / BKB;
    const bool have_ws = p.workspace != nullptr && p.workspace_bytes >= dk_gemm_split_workspace_bytes() && ((uintptr_t)p.workspace & 255) == 0 &&
                         p.kn_w == nullptr && (p2 == nullptr || p2->kn_w == nullptr);
    if (have_ws && g_dk_v3_split != 0 && tiles > 0 && tiles * 2 <= G) {
      const int S = G / tiles < 4 ? G / tiles : 4;
      const int ks = (nk + S - 1) / S;
      const int min_saved = g_dk_v3_split > 0 ? 1 : (g_dk_v3_split_min >= 0 ? g_dk_v3_split_min : 32);
      if (S >= 2 && nk - ks >= S - 1 && nk - ks >= min_saved && tiles * (S - 1) <= 256) {
        sp.S = S; sp.ks = ks;
        sp.slabs = (float*)p.workspace;
        sp.flags = (unsigned*)((char*)p.workspace + (size_t)256 * F8_SLAB_FLOATS * 4);
        sp.error_word = sp.flags + 512;
      }
    }
  }
  const int grid = (tiles_a + tiles_b) * (sp.S > 1 ? sp.S : 1);
  double work = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  if (p2) work += 2.0 * (double)p2->M * (double)p2->N * (double)p2->K;
  dk_prof_begin(3, work, stream);
  hipLaunchKernelGGL(dk_gemm256f8_kernel, dim3(grid), dim3(512), LDS_BYTES, stream, p, p2 ? *p2 : p, tiles_a, tiles_b, sp);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
