// 256 x 256 x 64 bf16 MFMA GEMM, third generation: the second-generation kernel (gemm256sk.hip, tile-parallel
// form) with the K loop rebuilt on v_mfma_f32_16x16x32_bf16.  Same contract, same grouped (two-problem) launch and
// column split; nn.Linear call sites python/src/diffusionkit/mlx/mmdit.py:821-832 and the fused linear1 / linear2
// of the single-stream blocks (:693-751).
//
// Shapes: N % 128 == 0 and K % 64 == 0; M is free.  (N % 256 == 128 -- SD3.5-large's h = 38 * 64 = 2432, config.py:72-74: the last
// column tile is half a tile: its upper 128 weight rows are fetched as a second copy of the lower 128, so that every DMA piece stays
// inside W, and the two waves columns that own them skip the tail.)  Rows go through the segment maps per lane on the load side
// (clamped to the last row when M is ragged) and, in the tail, per tile when the tile lies inside one segment
// (the common case) or per row when it straddles a segment boundary or the end of M -- the text stream of the
// SD3 double blocks (B = 2 segments of 589 rows, mmdit.py:608-625) rides in the image stream's launch that way.
//
// Why: the chip is power-limited on real (random) bf16 data, and the 16x16x32 instruction does the same FLOPs
// for less power than 32x32x16 -- scripts/mfma_probe.hip measures 2.21-2.24 PFLOP/s against 1.90-1.93 PFLOP/s
// for the same operand fragments (both reach 2.49 PFLOP/s on zeros).
//
// K loop: 8 waves (2 x 4), wave tile 128 (m) x 64 (n); per K = 32 slice the wave needs 4 W fragments and
// 8 A fragments (ds_read_b128 each: row = base + (lane & 15), 16-byte chunk = 4 * kk + (lane >> 4)) for
// 32 MFMAs.  A K-tile is 4 steps of 16 MFMAs:  (kk0, m 0-63), (kk0, m 64-127), (kk1, m 0-63), (kk1, m 64-127),
// with two W register sets and two A register sets reloaded one step ahead (inline-asm ds_read_b128,
// hand-counted s_waitcnt lgkmcnt), the tile barrier between steps 2 and 3, and the 8 LDS-DMA instructions
// (buffer_load_dwordx4 ... lds: SGPR resource + 32-bit lane offset) of a K-tile spread over steps 3 and 0, one in
// front of every fourth MFMA, on opposite slots for the two wave groups of a SIMD.  The second wave of each SIMD issues
// its fragment reads in the middle of a step instead of in front of it (DK_ITER_SKEW), and the steady state carries no
// branches: the last two K-tiles, which issue less DMA, are peeled (DK_DRIVE).
//
// Tail: accumulators -> wave-private LDS image (fp32, XOR-swizzled) -> row-major read-back, 8 columns per lane, one
// 16-byte store per lane and row; bias / GELU (erfc polynomial) / SiLU / gate * x + residual on the way.
// Remainder waves can be cut along K (SplitArgs, plan_split): finisher + producer pieces through fp32 slabs.
//
// Tile height: template parameter MF = 16-row fragments per wave along m, 8 (256-row tiles) or 7 (224-row tiles).  The hot
// shapes have M = 4352 / 4608 / 8192 rows and N / 256 = 12 .. 48 column tiles: with 256-row tiles the last round of the 256 CUs is
// 20 % empty (N = 3072: 204 tiles, one round), with 224-row tiles the same work is 240 / 252 tiles of 7/8 the size -- one round
// of 0.875 tile-times instead of 1.0.  The launcher takes the height that minimises rounds x height (dk_tune_set
// ("gemm_mf", 7 | 8) forces one).  The LDS image keeps its two 128-row A slots; a 224-row tile uses 112 rows of each.
//
// C / D layout of the swapped-operand MFMA (A-operand = W fragment, B-operand = activation fragment):
// lane holds output row m = mf*16 + (lane & 15), columns n = nf*16 + 4*(lane >> 4) + {0..3}.
#include <cstring>
#include <type_traits>

#include "dk_kernels.h"

#ifndef DK_V3_ABL
#define DK_V3_ABL 0  // lab only (scripts/build_lab.sh ABL=n), bit mask: 1 no DMA inside the K loop, 2 no fragment reads inside it, 4 no tile
                     // barrier, 8 producers do not store, 16 finishers neither wait nor read, 32 no C stores, 64 no tail (run-time false),
                     // 128 every K-tile's DMA re-reads K-tile 0 (same bytes into the LDS, all of them L2 hits)
#endif

// placement of the 4 DMA pieces inside a 16-MFMA step: in front of MFMA slots PH, PH + STR, ... (PH0 / PH1 for the
// two wave groups of a SIMD)
#ifndef DK_V3_PH0
#define DK_V3_PH0 0
#define DK_V3_PH1 2
#define DK_V3_STR 4
#endif
#ifndef DK_V3_NT_STORE
#define DK_V3_NT_STORE 0  // lab: C leaves through non-temporal stores
#endif
#ifndef DK_V3_SKEW
#define DK_V3_SKEW 1
#endif
#ifndef DK_V3_SKEW_R
#define DK_V3_SKEW_R 8  // MFMA slot of a step behind which the skewed wave group issues its fragment reads
#endif

#define T256 256
#define BK 64
#define HALF_BYTES (128 * BK * 2)
#define OP_BYTES (2 * HALF_BYTES)  // one operand of one K-tile: rows 0-127, rows 128-255
// LDS ring: two slots of the activation operand, THREE of the weight operand (all 160 KiB).  Weights stream from HBM (every
// block of the model has its own, 5-38 GB per step in total), activations come out of L2 / Infinity Cache: the four weight
// pieces of K-tile i+2 are issued in the first step of K-tile i, 1.75 K-tiles ahead of their first read (a 2-slot ring gave
// them 0.5-1.0), the activation pieces behind the tile barrier, 1.0 ahead; the in-order vmcnt lets the newest four -- the
// weight pieces -- stay in flight across the barrier.  Cold-weight launches: within 1-6 % of warm ones (2-slot ring: 10-24 %).
#define W_BASE (2 * OP_BYTES)
#define LDS_BYTES (5 * OP_BYTES)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) char lds_char;

// Remainder split (dk_launch_gemm256v3): the tiles beyond the last full wave of the CUs -- n_rem < #CU of them --
// are cut along K into S pieces with the SAME cut points for every tile (so that the workgroups that run at the
// same time still walk K in step and share A / W panels in L2): piece 0 = [0, ks) is the tile's finisher, pieces
// 1 .. S-1 share [ks, nk) and are producers (fp32 partial tile -> slab, flag).  Block order = dispatch order:
// full tiles, then the n_rem finishers, then the producers; a finisher only waits at its very end, and at least
// #CU - n_rem CUs are never held by finishers, so producers always get to run.
struct SplitArgs {
  float* slabs;     // [n_rem * (S - 1)][256 * 256] fp32 row-major tile images
  unsigned* flags;  // [n_rem * (S - 1)], zero between launches (reset by the finisher)
  unsigned* error_word;
  int n_dp;         // full tiles (multiple of 8); 0 <= n_dp <= tiles
  int n_rem;        // split tiles = tiles - n_dp (0: no split)
  int S;            // pieces per split tile
  int ks;           // K-tiles of the finisher piece
};
#define SLAB_FLOATS (256 * 256)

// 16-byte write-through (sc1) store: the slab reaches memory without an agent-scope release fence
__device__ __forceinline__ void v3_store_sc1_b128(float* ptr, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

// position of block `bid` inside the XCD-contiguous order of the `count` blocks that start at block `base`
// (hardware places block b on XCD b & 7): neighbouring positions share an XCD, hence an L2
__device__ __forceinline__ int xcd_contiguous(int bid, int base, int count) {
  const int x = bid & 7;
  int start = 0;
  for (int y = 0; y < x; ++y) {
    const int first = (y - base) & 7;  // offset of XCD y's first block inside the group
    start += first < count ? (count - first + 7) >> 3 : 0;
  }
  return start + ((bid - base) >> 3);
}

// CONV: the activation operand is the im2col view of an NHWC tensor (3x3, pad 1, stride 1, optionally over the nearest-x2
// upsampling of the stored tensor -- vae.py:20-25,73,79,134): GEMM row m = output pixel (b, y, x), K-tile = 64 channels of one
// tap.  Per DMA piece the lane offset is recomputed from the pixel and the tap (a dozen VALU instructions); padding taps take
// an offset beyond the buffer descriptor's range, for which the LDS-DMA writes zeros (scripts/oob_probe.hip).
template <int MF, bool CONV>
__global__ __launch_bounds__(512, 2) void dk_gemm256v3_kernel(GemmParams pa, GemmParams pb, int tiles_a, int tiles_b, SplitArgs sp) {
  static_assert(MF == 8 || MF == 7, "wave tile: 8 or 7 fragments of 16 rows");
  constexpr int BM = 32 * MF;     // tile rows (two wave rows)
  constexpr int HROWS = 16 * MF;  // rows of one wave row = rows used of a 128-row LDS slot
  constexpr int NHI = MF - 4;     // fragments of the second ("hi") m-group of a wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, q = lane >> 4;

  const int nk_full = pa.K / BK;
  int tile;       // XCD-contiguous tile index: neighbouring tiles share an L2
  int k0 = 0, nk = nk_full;  // this workgroup's K-tile range [k0, k0 + nk)
  int piece = -1;            // -1 full tile, 0 finisher of a split tile, >= 1 producer
  int rt = 0;                // index of the split tile
  {
    const int bid = blockIdx.x;
    if (bid < sp.n_dp || sp.n_rem == 0) {
      tile = xcd_contiguous(bid, 0, sp.n_rem == 0 ? tiles_a + tiles_b : sp.n_dp);
    } else {
      const int j = bid - sp.n_dp;
      piece = j / sp.n_rem;
      const int base = sp.n_dp + piece * sp.n_rem;
      rt = xcd_contiguous(bid, base, sp.n_rem);
      tile = sp.n_dp + rt;
      if (piece == 0) {
        nk = sp.ks;
      } else {
        const int rest = nk_full - sp.ks, np = sp.S - 1;
        k0 = sp.ks + rest * (piece - 1) / np;
        nk = sp.ks + rest * piece / np - k0;
      }
    }
  }
  const bool second = tile >= tiles_a;
  // ONE scalar base into the kernel-argument segment for this tile's parameter block (round 4).  Written as `second ? pb : pa` the compiler
  // loads BOTH blocks and s_cselects field by field (36 selects and 30 SGPRs spilled to VGPR lanes in the prologue, 2039 v_readlane over the
  // tail's code paths); with a base pointer every field is one s_load at its use: 100-104 -> 82-104 SGPRs, no lane spills, +0.5 % in the model
  // (profiles/r04_gemm_kernarg_pointer.log).  Layout: by-value struct arguments sit back to back from offset 0 of the segment.
  static_assert(sizeof(GemmParams) % 8 == 0 && alignof(GemmParams) == 8, "pb follows pa without padding");
  typedef const __attribute__((address_space(4))) GemmParams karg_params_t;
  const __attribute__((address_space(4))) char* kbase = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  karg_params_t& p = *(karg_params_t*)(kbase + (second ? sizeof(GemmParams) : 0));
  const int tl = second ? tile - tiles_a : tile;  // tile index inside its problem
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + T256 - 1) / T256;

  // ---- lane-constant parts of the LDS fragment addresses: row l15 (+ 16 * fragment), chunk 4*kk + q ----
  unsigned offk[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) offk[kk] = (unsigned)(l15 * 128 + (((kk * 4 + q) ^ (l15 >> 1)) << 4));
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
  const int m0 = tm * BM, n0 = tn * T256;

  // DMA sources: A rows through the segment map per lane (32-bit byte offsets from p.A; rows beyond M - 1 re-read
  // the last row, their results are never stored), W rows from a tile-uniform base + lane part
  unsigned la[2][2], lw[2];  // (CONV: la = the row's pixel, b << 24 | y << 12 | x)
  unsigned lch[2] = {0u, 0u};  // CONV: byte offset of the lane's 16-byte chunk inside a 128-byte K-tile row
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j);  // = (lane&7) ^ (((wave*16 + j*8 + srow) >> 1) & 7)
    lw[j] = ((unsigned)srow * (unsigned)p.ldw + chunk * 8) * 2u;
    lch[j] = (unsigned)chunk * 16u;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      // (MF = 7: wave 7's rows lie beyond the 112 rows a wave row uses -- it fetches duplicates of other rows into LDS rows
      //  nobody reads, so that every wave issues the same 8 pieces per K-tile)
      const int m = min(m0 + hh * HROWS + wave * 16 + j * 8 + srow, p.M - 1);
      if (CONV) {
        const int hw = p.cH * p.cW;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.cW, x = rem - y * p.cW;
        la[hh][j] = ((unsigned)b << 24) | ((unsigned)y << 12) | (unsigned)x;
      } else {
        la[hh][j] = (unsigned)m;  // (mapped through the row segments below)
      }
    }
  }
  if (!CONV) {
    // a tile inside one row segment (every tile of an image stream) maps its rows with one scalar division instead of four per lane
    // (round 5: the per-lane integer divisions were ~ 0.3 us in front of the first DMA piece of every tile)
    const int a_seg0 = m0 / p.a_seg_len;
    const bool a_uniform = min(m0 + BM - 1, p.M - 1) / p.a_seg_len == a_seg0;
    const int a_base = a_seg0 * p.a_seg_stride - a_seg0 * p.a_seg_len;
    if (a_uniform) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) la[hh][j] += (unsigned)a_base;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) la[hh][j] = (unsigned)(((int)la[hh][j] / p.a_seg_len) * p.a_seg_stride + ((int)la[hh][j] % p.a_seg_len));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) la[hh][j] = (la[hh][j] * (unsigned)p.lda + (lch[j] >> 1)) * 2u;
  }
  // CONV: tap (dy, dx in -1..1) and channel byte offset of the NEXT activation K-tile to be issued; stored tensor [cB, Hs, Ws, cC]
  const int cv_ush = CONV && p.ups == 1 ? 1 : 0;
  const int cv_Hs = p.cH >> cv_ush, cv_Ws = p.cW >> cv_ush;
  const int cv_C2 = p.cC * 2;
  int cv_dy = -1, cv_dx = -1, cv_cb = 0;
  if (CONV) {
    const int cpt = p.cC / BK;
    const int tap = k0 / cpt;
    cv_cb = (k0 - tap * cpt) * (BK * 2);
    cv_dy = tap / 3 - 1;
    cv_dx = tap - (tap / 3) * 3 - 1;
  }
  const char* gA = (const char*)p.A + (CONV ? (size_t)0 : (size_t)k0 * (BK * 2));
  const char* gW = (const char*)p.W + ((size_t)n0 + wave * 16) * (size_t)p.ldw * 2 + (size_t)k0 * (BK * 2);
  // (half a column tile at the end of N: rows 128-255 of the weight slot are a second copy of rows 0-127)
  const size_t w128 = n0 + T256 > p.N ? (size_t)0 : (size_t)128 * p.ldw * 2, w8 = (size_t)8 * p.ldw * 2;
  const bool col_ok = n0 + wn * 64 < p.N;  // wave-uniform: this wave's 64 columns exist

  // LDS-DMA in the buffer form: SGPR resource (base, 4 GiB range) + 32-bit lane offset + scalar offset -- no 64-bit
  // per-lane address and no VALU per piece
  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc((void*)gA, 0, CONV ? p.cB * cv_Hs * cv_Ws * cv_C2 : -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)gW, 0, -1, 0x00020000);
  // ring slots (byte offsets from W_BASE) of the weights of K-tiles i, i+1, i+2 of the loop below
  unsigned wo_cur = 0u, wo_nxt = OP_BYTES, wo_nn = 2u * OP_BYTES;
  // one of the 8 DMA instructions of K-tile i: gidx 0..3 the activation pieces (half, j), 4..7 the weight pieces into slot `wslot`
  auto issue_piece_to = [&](int i, int gidx, unsigned wslot) {
    const int hh = gidx & 1, j = (gidx >> 1) & 1;
    if (gidx < 4 && CONV) {
      const unsigned pc = la[hh][j];
      const int iy = (int)((pc >> 12) & 0xFFFu) + cv_dy, ix = (int)(pc & 0xFFFu) + cv_dx;
      const bool ok = (unsigned)iy < (unsigned)p.cH && (unsigned)ix < (unsigned)p.cW;
      const int spix = ((int)(pc >> 24) * cv_Hs + (iy >> cv_ush)) * cv_Ws + (ix >> cv_ush);
      const unsigned voff = ok ? (unsigned)spix * (unsigned)cv_C2 + lch[j] : 0x80000000u;  // padding tap: out of range -> zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)((lds_char*)0 + (i & 1) * OP_BYTES + (wave * 16) * 128 + hh * HALF_BYTES + j * 1024),
                                               16, (int)voff, cv_cb, 0, 0);
    } else if (gidx < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)((lds_char*)0 + (i & 1) * OP_BYTES + (wave * 16) * 128 + hh * HALF_BYTES + j * 1024),
                                               16, (int)la[hh][j], ((DK_V3_ABL & 128) ? 0 : i) * (BK * 2), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)((lds_char*)0 + W_BASE + wslot + (wave * 16) * 128 + hh * HALF_BYTES + j * 1024),
                                               16, (int)lw[j], (int)(hh * w128 + j * w8) + ((DK_V3_ABL & 128) ? 0 : i) * (BK * 2), 0, 0);
  };
  auto issue_piece = [&](int i, int gidx) { issue_piece_to(i, gidx, wo_nn); };  // the loop only ever issues K-tile i+2
  auto conv_advance = [&]() {  // CONV: the activation pieces of one K-tile are out -- on to the next 64 channels / the next tap
    if (CONV) {
      cv_cb += BK * 2;
      if (cv_cb == cv_C2) {
        cv_cb = 0;
        if (++cv_dx == 2) cv_dx = -1, ++cv_dy;
      }
    }
  };

  f32x4 acc[4][MF];  // [nf][mf]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // ---------------- K loop: register-pipelined, hand-counted LDS waits ----------------
#define DK_LDS_RD(DST, ADDR, OFF)                                                              \
  do {                                                                                        \
    if (!(DK_V3_ABL & 2) || !in_loop) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR)); \
  } while (0)
#define DK_RDW(SET, BUFOFF, KK)                    \
  do {                                             \
    const unsigned aW_ = offk[KK] + sW + (BUFOFF); \
    DK_LDS_RD(wf##SET[0], aW_, 0);                 \
    DK_LDS_RD(wf##SET[1], aW_, 2048);              \
    DK_LDS_RD(wf##SET[2], aW_, 4096);              \
    DK_LDS_RD(wf##SET[3], aW_, 6144);              \
  } while (0)
#define DK_RDA_LO(SET, BUFOFF, KK)                 \
  do {                                             \
    const unsigned aA_ = offk[KK] + sA + (BUFOFF); \
    DK_LDS_RD(xf##SET[0], aA_, 0);                 \
    DK_LDS_RD(xf##SET[1], aA_, 2048);              \
    DK_LDS_RD(xf##SET[2], aA_, 4096);              \
    DK_LDS_RD(xf##SET[3], aA_, 6144);              \
  } while (0)
#define DK_RDA_HI(SET, BUFOFF, KK)                 \
  do {                                             \
    const unsigned aA_ = offk[KK] + sA + (BUFOFF); \
    DK_LDS_RD(xf##SET[0], aA_, 8192);              \
    DK_LDS_RD(xf##SET[1], aA_, 10240);             \
    DK_LDS_RD(xf##SET[2], aA_, 12288);             \
    if (NHI > 3) DK_LDS_RD(xf##SET[3], aA_, 14336); \
  } while (0)
// the wait in front of the tile barrier: own fragment reads and own DMA pieces of the next K-tile; the four weight pieces of
// K-tile i+2 (issued last) stay in flight when there are any (KEEP)
#define DK_TILE_WAIT(V, KEEP)                                                                                            \
  do {                                                                                                                   \
    if (KEEP)                                                                                                            \
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3])::"memory");          \
    else                                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3])::"memory");          \
  } while (0)
#define DK_WAIT4(N, V) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]))
// everything but the NHI hi-fragment reads issued last has landed
#define DK_WAIT8_HI(V, U)                        \
  do {                                           \
    if constexpr (NHI == 4) DK_WAIT8(4, V, U);   \
    else DK_WAIT8(3, V, U);                      \
  } while (0)
#define DK_WAIT8(N, V, U)                  \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" \
               : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]))
// 16 MFMAs acc[nf][MB + mf] += W[nf] . A[mf], with NG DMA pieces (TILE, G0 ..) in front of slots PH, PH+4, ... when ON
// (NM = m-fragments of the group: 4, or NHI for the hi group; a step is 4 * NM MFMAs, MFMA e_ = (nf = e_ / NM, mf = e_ % NM); the
//  pieces sit NM slots apart so that four of them fit any step)
#define DK_MMG(WSET, ASET, MB, NM, TILE, G0, NG, PH, ON) DK_MMGR(WSET, ASET, MB, NM, TILE, G0, NG, PH, ON, 0, 4 * (NM))
#define DK_MMGR(WSET, ASET, MB, NM, TILE, G0, NG, PH, ON, E0, E1)                                                 \
  do {                                                                                                            \
    _Pragma("unroll") for (int e_ = (E0); e_ < (E1); ++e_) {                                                           \
      if (!(DK_V3_ABL & 1) && (NG) > 0 && (ON) && e_ >= (PH) && ((e_ - (PH)) % (NM)) == 0 && ((e_ - (PH)) / (NM)) < (NG)) \
        issue_piece((TILE), (G0) + ((e_ - (PH)) / (NM)));                                                         \
      acc[e_ / (NM)][(MB) + (e_ % (NM))] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf##WSET[e_ / (NM)], xf##ASET[e_ % (NM)], \
                                                                                   acc[e_ / (NM)][(MB) + (e_ % (NM))], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                          \
    }                                                                                                             \
  } while (0)
// One K-tile for the wave group that reads its fragments in FRONT of every 16-MFMA step.  ON2 (compile-time): whether the
// DMA pieces of K-tile i+2 are issued (weights in the first step, activations behind the barrier -- in that program order,
// see DK_TILE_WAIT) -- false only in the last two K-tiles, so that the steady-state loop carries no branches around the pieces.
#define DK_ITER(PH, ON1, ON2)                                                                                    \
  {                                                                                                              \
    constexpr bool in_loop = true;                                                                               \
    const unsigned bo = (i & 1) * OP_BYTES;                                                                      \
    DK_RDA_HI(1, bo, 0);                                                                                         \
    DK_WAIT8_HI(wf0, xf0);                                                                                       \
    DK_MMG(0, 0, 0, 4, i + 2, 4, 4, PH, ON2);                                                                    \
    DK_RDW(1, wo_cur, 1);                                                                                        \
    DK_RDA_LO(0, bo, 1);                                                                                         \
    DK_WAIT4(8, xf1);                                                                                            \
    DK_MMG(0, 1, 4, NHI, i + 2, 0, 0, PH, false);                                                                \
    DK_RDA_HI(1, bo, 1);                                                                                         \
    DK_WAIT8_HI(wf1, xf0);                                                                                       \
    DK_MMG(1, 0, 0, 4, i + 2, 0, 0, PH, false);                                                                  \
    DK_TILE_WAIT(xf1, ON2);                                                                                      \
    if (!(DK_V3_ABL & 4)) __builtin_amdgcn_s_barrier();                                                          \
    asm volatile("" ::: "memory");                                                                               \
    DK_RDW(0, wo_nxt, 0); /* unconditional: after the last tile these read stale ring data that */               \
    DK_RDA_LO(0, bo ^ OP_BYTES, 0); /* nobody uses; they are waited for behind the loop          */               \
    DK_MMG(1, 1, 4, NHI, i + 2, 0, 4, PH, ON2);                                                                  \
    DK_ROTATE_W();                                                                                               \
  }
// The same K-tile for the second wave of each SIMD with its fragment reads behind MFMA slot R of every step instead of
// in front of it, so that the two waves of a SIMD do not run their read / wait sections at the same time (+1.5-2 %).
#define DK_ITER_SKEW(PH, R, ON1, ON2)                                                                            \
  {                                                                                                              \
    constexpr bool in_loop = true;                                                                               \
    const unsigned bo = (i & 1) * OP_BYTES;                                                                      \
    constexpr int RH = (R) * NHI / 4; /* the same relative slot inside a hi step of 4 * NHI MFMAs */            \
    DK_WAIT8(0, wf0, xf0);                                                                                       \
    DK_MMGR(0, 0, 0, 4, i + 2, 4, 4, PH, ON2, 0, R);                                                             \
    DK_RDA_HI(1, bo, 0);                                                                                         \
    DK_MMGR(0, 0, 0, 4, i + 2, 4, 4, PH, ON2, R, 16);                                                            \
    DK_WAIT4(0, xf1);                                                                                            \
    DK_MMGR(0, 1, 4, NHI, i + 2, 0, 0, 0, false, 0, RH);                                                         \
    DK_RDW(1, wo_cur, 1);                                                                                        \
    DK_RDA_LO(0, bo, 1);                                                                                         \
    DK_MMGR(0, 1, 4, NHI, i + 2, 0, 0, 0, false, RH, 4 * NHI);                                                   \
    DK_WAIT8(0, wf1, xf0);                                                                                       \
    DK_MMGR(1, 0, 0, 4, i + 2, 0, 0, 0, false, 0, R);                                                            \
    DK_RDA_HI(1, bo, 1);                                                                                         \
    DK_MMGR(1, 0, 0, 4, i + 2, 0, 0, 0, false, R, 16);                                                           \
    DK_TILE_WAIT(xf1, ON2);                                                                                      \
    if (!(DK_V3_ABL & 4)) __builtin_amdgcn_s_barrier();                                                          \
    asm volatile("" ::: "memory");                                                                               \
    DK_MMGR(1, 1, 4, NHI, i + 2, 0, 4, PH, ON2, 0, RH);                                                          \
    DK_RDW(0, wo_nxt, 0);                                                                                        \
    DK_RDA_LO(0, bo ^ OP_BYTES, 0);                                                                              \
    DK_MMGR(1, 1, 4, NHI, i + 2, 0, 4, PH, ON2, RH, 4 * NHI);                                                    \
    DK_ROTATE_W();                                                                                               \
  }
// all K-tiles of this workgroup: branch-free steady state, then the two tiles that issue less.  The fragments in flight
// at a section boundary are waited for there (an inline-asm load must not be in flight across a compiler-visible merge).
#define DK_ROTATE_W()           \
  do {                          \
    const unsigned t_ = wo_cur; \
    wo_cur = wo_nxt;            \
    wo_nxt = wo_nn;             \
    wo_nn = t_;                 \
    conv_advance();             \
  } while (0)
#define DK_DRIVE(ITER, ...)                                                                                      \
  {                                                                                                              \
    int i = 0;                                                                                                   \
    for (; i + 2 < nk; ++i) ITER(__VA_ARGS__, true, true)                                                        \
    DK_WAIT8(0, wf0, xf0);                                                                                       \
    if (i + 1 < nk) {                                                                                            \
      ITER(__VA_ARGS__, true, false)                                                                             \
      ++i;                                                                                                       \
      DK_WAIT8(0, wf0, xf0);                                                                                     \
    }                                                                                                            \
    ITER(__VA_ARGS__, false, false)                                                                              \
    DK_WAIT8(0, wf0, xf0);                                                                                       \
  }

  {
    bf16x8 wf0[4], wf1[4], xf0[4], xf1[4];
    // prologue: K-tile 0, then K-tile 1 (the loop's first wait lets only its own four weight pieces, of K-tile 2, stay in flight)
#pragma unroll
    for (int gidx = 0; gidx < 8; ++gidx) issue_piece_to(0, gidx, wo_cur);
    conv_advance();
    if (nk > 1) {
#pragma unroll
      for (int gidx = 0; gidx < 8; ++gidx) issue_piece_to(1, gidx, wo_nxt);
      conv_advance();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (the first fragment reads sit INSIDE the branches: an inline-asm load that is still in flight must not be
    //  live across a compiler-visible branch, see gemm256sk.hip)
    constexpr bool in_loop = false;
    if (wm == 0) {
      DK_RDW(0, wo_cur, 0);
      DK_RDA_LO(0, 0u, 0);
      DK_DRIVE(DK_ITER, DK_V3_PH0)
    } else {
      DK_RDW(0, wo_cur, 0);
      DK_RDA_LO(0, 0u, 0);
#if DK_V3_SKEW
      DK_DRIVE(DK_ITER_SKEW, DK_V3_PH1, DK_V3_SKEW_R)
#else
      DK_DRIVE(DK_ITER, DK_V3_PH1)
#endif
    }
  }
#undef DK_LDS_RD
#undef DK_RDW
#undef DK_RDA_LO
#undef DK_RDA_HI
#undef DK_WAIT4
#undef DK_TILE_WAIT
#undef DK_WAIT8
#undef DK_WAIT8_HI
#undef DK_MMG
#undef DK_MMGR
#undef DK_ITER
#undef DK_ITER_SKEW
#undef DK_DRIVE
#undef DK_ROTATE_W

  // ---------------- K split, round 6: the pieces of a cut tile exchange their RAW accumulators (registers -> slab -> registers) ----------------
  // Until round 6 a cut tile went through a second tail: fp32 staging images (two passes over the LDS), row-major fp32 slabs written from the read-back,
  // the finisher adding them row by row inside its epilogue loop.  The accumulators are in registers on both sides: a producer now stores them as they are
  // (thread-linear: one coalesced 16-byte write-through store per thread and accumulator quad), the finisher adds them in front of the ordinary
  // whole-tile tail (bias before staging, bf16 image, one LDS round trip) -- gemm256f8.hip's exchange, which was written first.  Same hand-off protocol
  // (guide G16): write-through stores, vmcnt(0) in every wave, barrier, one relaxed agent-scope flag store; finisher: relaxed poll, one agent-scope
  // acquire, barrier, plain loads.  Summation order of a cut tile: finisher's K range first, then the producers' in piece order (as before).
  // (Linears only: the conv form of the 256-row kernel sits at exactly 256 registers and keeps the fp32-image path below -- this block cost it 7 spills)
  if (!CONV && piece >= 0) {
    const int n_prod_x = sp.S - 1;
    // (thread index from a FRESH lane id + the scalar wave id: a value carried across the K loop costs a register there -- the conv form of the
    //  256-row kernel has none to spare)
    int lane_x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_x));
    const unsigned toff = (unsigned)(wave * 64 + lane_x) * 4u;  // floats
    if (piece >= 1) {
      float* const slab = sp.slabs + (size_t)(rt * n_prod_x + piece - 1) * SLAB_FLOATS;  // (uniform base: scalar registers)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) v3_store_sc1_b128(slab + (size_t)(nf * 8 + mf) * 2048 + toff, acc[nf][mf]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have completed
      __syncthreads();
      if (wave == 0 && lane_x == 0) __hip_atomic_store(sp.flags + rt * n_prod_x + piece - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (wave == 0 && lane_x == 0) {
      for (int pp = 0; pp < n_prod_x; ++pp) {
        unsigned spins = 0;
        while (__hip_atomic_load(sp.flags + rt * n_prod_x + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
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
    for (int pp = 0; pp < n_prod_x; ++pp) {
      const float* const slab = sp.slabs + (size_t)(rt * n_prod_x + pp) * SLAB_FLOATS;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const f32x4 o = *(const f32x4*)(slab + (size_t)(nf * 8 + mf) * 2048 + toff);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nf][mf][e] += o[e];
        }
    }
    __syncthreads();  // every wave has read the slabs
    if (wave == 0 && lane_x == 0)
      for (int pp = 0; pp < n_prod_x; ++pp) __hip_atomic_store(sp.flags + rt * n_prod_x + pp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    piece = -1;  // from here on the finisher is a whole tile
  }
  if (!CONV) __builtin_assume(piece < 0);  // (no Linear tile reaches the fp32-image tail path below any more; convolutions still do)

  // ---------------- tail: accumulators -> LDS (wave-private image) -> row-major ----------------
  // All waves passed the last loop barrier after their final ds_read, so the ring is free.
  const bool out2 = p.n_split > 0 && n0 >= p.n_split;  // tile-uniform: second output of a column-split GEMM
  bf16_t* const Cb = out2 ? p.C2 : p.C;
  const int ldcb = out2 ? p.ldc2 : p.ldc;
  const int epi = out2 ? p.epi2 : p.epi;
  const int ncol0 = out2 ? n0 - p.n_split : n0;
  const bool has_res = epi == DK_EPI_GATE_RES || epi == DK_EPI_RES;
  // a tile that lies inside one row segment of every map and inside M evaluates the maps once (scalar unit);
  // otherwise each lane walks its rows through the maps (fast == false)
  auto inside = [&](int len) { return m0 / len == (m0 + BM - 1) / len; };
  const bool fast = m0 + BM <= p.M && inside(p.c_seg_len) && (!has_res || inside(p.r_seg_len)) &&
                    (epi != DK_EPI_GATE_RES || inside(p.gate_seg_len));
  const int mrow0 = m0 + wm * HROWS;  // first GEMM row of this wave's block
  const size_t physC0 = (size_t)((m0 / p.c_seg_len) * p.c_seg_stride + (m0 % p.c_seg_len)) + wm * HROWS;
  const size_t physR0 = has_res ? (size_t)((m0 / p.r_seg_len) * p.r_seg_stride + (m0 % p.r_seg_len)) + wm * HROWS : 0;
  const bf16_t* gate_row = epi == DK_EPI_GATE_RES ? p.gate + (size_t)(m0 / p.gate_seg_len) * p.gate_stride : nullptr;
  // read-back: a lane takes 8 consecutive columns (two 16-byte chunks) of one row, 4 lanes a 32-column row of the
  // image, 16 rows per step -- one 16-byte global store per lane and step (8-byte stores are issue-bound: half as
  // many instructions, guide T21).  Image swizzle chunk ^ ((row >> 1) & 7): conflict-free for the staging writes
  // (16 rows x one chunk per 16 lanes) and for these reads (4 rows x 4 even / odd chunks per 16 lanes).
  const int rrow = lane >> 2, rc2 = (lane & 3) * 2;

  // split tile: a producer stores its fp32 partial tile to its slab; the finisher first waits for every producer of
  // the tile (hand-off per guide G16: write-through slab stores, vmcnt(0) in every wave, barrier, one relaxed
  // agent-scope flag store; consumer: relaxed poll, one agent-scope acquire, barrier, plain loads)
  const int n_prod = sp.S - 1;
  float* const my_slab = piece >= 1 ? sp.slabs + (size_t)(rt * n_prod + piece - 1) * SLAB_FLOATS : nullptr;
  if (piece == 0 && !(DK_V3_ABL & 16)) {
    if (tid == 0) {
      for (int pp = 0; pp < n_prod; ++pp) {
        unsigned spins = 0;
        while (__hip_atomic_load(sp.flags + rt * n_prod + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
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
  }

  auto unpack8 = [](const uint4 v, float* f) {
    unpack2bf(v.x, f[0], f[1]);
    unpack2bf(v.y, f[2], f[3]);
    unpack2bf(v.z, f[4], f[5]);
    unpack2bf(v.w, f[6], f[7]);
  };

  // whole tiles: the bias of this lane's 16 columns (4 per 16-column fragment), all loads up front -- one latency, not one per pass
  u32x2 bias_q[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}};
  if (piece < 0 && p.bias && col_ok) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) bias_q[nf] = *(const u32x2*)(p.bias + n0 + wn * 64 + nf * 16 + 4 * q);
  }
  // Key tile of a q / k / v projection with the fused QKNorm + RoPE: the sum of squares of every row over its head's columns --
  // this wave's 64 columns from the accumulators (same bf16-rounded values that get staged), for 128-column heads plus the
  // partner wave's 64 through LDS (behind the staging images)
  const bool qtile = !CONV && p.qn_w != nullptr && n0 >= p.qn_col0 && n0 < p.qn_col1;  // query tile (round 4): same treatment, own weight
  const bool kfuse = !CONV && p.kn_w != nullptr && piece < 0 && ((n0 >= p.kn_col0 && n0 < p.kn_col1) || qtile);  // tile-uniform
  const bf16_t* const nw = qtile ? p.qn_w : p.kn_w;
  constexpr unsigned XCH_OFF = 8u * 16384u;
  if (kfuse) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      float ss = 0.f;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        float b4[4];
        unpack2bf(bias_q[nf][0], b4[0], b4[1]);
        unpack2bf(bias_q[nf][1], b4[2], b4[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = round_bf16(acc[nf][mf][e] * p.alpha + b4[e]);
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
  // (STAGE / EMIT: a whole tile stages BOTH halves -- two 8 KiB bf16 images per wave -- before it reads the first one back: one
  //  LDS round trip of latency per tile instead of two; a split tile's 16 KiB fp32 image holds one half at a time)
  auto tail_pass = [&](auto ni_c, auto stage_c, auto emit_c) {
    constexpr int ni = decltype(ni_c)::value;
    constexpr bool STAGE = decltype(stage_c)::value, EMIT = decltype(emit_c)::value;
    const unsigned reg0 = (unsigned)wave * 16384u + ((STAGE && EMIT) ? 0u : (unsigned)ni * 8192u);  // this pass's image
    // stage: lane owns row mf*16 + l15, columns (nf & 1)*16 + 4*q + {0..3} of this 32-column half.  A whole tile (no K split)
    // stages round_bf16(alpha * acc + bias) -- what every epilogue starts from -- as bf16: half the LDS bytes of the fp32 image
    // (64-byte rows, 16-byte chunk c at position c ^ ((row >> 2) & 3): conflict-free for these 8-byte writes and the 16-byte
    // read-back); split tiles stage the fp32 partial sums (128-byte rows)
    const bool bf_stage = piece < 0;
    if (!STAGE) {
    } else if (bf_stage) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        float b4[4];
        unpack2bf(bias_q[ni * 2 + nf][0], b4[0], b4[1]);
        unpack2bf(bias_q[ni * 2 + nf][1], b4[2], b4[3]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int row = mf * 16 + l15;
          const f32x4 a = acc[ni * 2 + nf][mf];
          uint2 w;
          w.x = pack2bf(a[0] * p.alpha + b4[0], a[1] * p.alpha + b4[1]);
          w.y = pack2bf(a[2] * p.alpha + b4[2], a[3] * p.alpha + b4[3]);
          *(__attribute__((address_space(3))) u32x2*)((lds_char*)0 + reg0 + row * 64 + (((nf * 2 + (q >> 1)) ^ ((row >> 2) & 3)) << 4) + (q & 1) * 8) = u32x2{w.x, w.y};
        }
      }
    } else {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int row = mf * 16 + l15;
          *(__attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((nf * 4 + q) ^ ((row >> 1) & 7)) << 4)) = acc[ni * 2 + nf][mf];
        }
    }
    // (same wave writes and reads the image: program order + the compiler's lgkmcnt suffice)
    if (!EMIT || !col_ok) return;
    const int col = n0 + wn * 64 + ni * 32 + rc2 * 4;      // first of this lane's 8 columns of the GEMM (bias, gate, residual)
    const int ocol = ncol0 + wn * 64 + ni * 32 + rc2 * 4;  // the same inside the output it goes to
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gate8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && piece == 0) unpack8(*(const uint4*)(p.bias + col), bias8);  // (whole tiles: the bias went in before staging)
    // the row loop is instantiated twice -- tile-uniform maps (FAST) or a per-lane walk through the maps -- so that
    // the common case keeps its short body (one v_add per address, batched loads)
    auto rows = [&](auto fast_c, auto bf_c, auto ek_c, auto kf_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      constexpr bool KF = decltype(kf_c)::value;  // key tile with the fused QKNorm + RoPE (bf16 image, bias-only epilogue)
      constexpr bool BF = decltype(bf_c)::value;  // bf16 image of a whole tile / fp32 image of a split tile
      constexpr int EK = decltype(ek_c)::value;   // the epilogue as a compile-time constant (straight-line row loop), or -1: `epi`
      constexpr bool PLAIN = BF && EK == DK_EPI_BIAS && !KF;  // a bias-only epilogue on a bf16 image: the staged values ARE the output
      float kw8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int kcol = KF ? col % p.kn_D : 0;  // first of this lane's 8 columns inside its head (the ranges start at multiples of 256)
      if (KF) unpack8(*(const uint4*)(nw + kcol), kw8);
      const int ep = EK >= 0 ? EK : epi;
      const bool hres = EK >= 0 ? (EK == DK_EPI_GATE_RES || EK == DK_EPI_RES) : has_res;
      if (FAST && ep == DK_EPI_GATE_RES) unpack8(*(const uint4*)(gate_row + col), gate8);
      // row walk of the slow path: (segment, row inside it) of this lane's current row in each map; 16 rows per step
      int c_seg = 0, c_rem = 0, r_seg = 0, r_rem = 0, g_seg = 0, g_rem = 0;
      if (!FAST) {
        const int ms = mrow0 + rrow;
        c_seg = ms / p.c_seg_len, c_rem = ms % p.c_seg_len;
        if (hres) r_seg = ms / p.r_seg_len, r_rem = ms % p.r_seg_len;
        if (ep == DK_EPI_GATE_RES) g_seg = ms / p.gate_seg_len, g_rem = ms % p.gate_seg_len;
      }
      // FAST tiles: everything the rows read from memory is fetched BEFORE the row loop (round 4).  The residual is updated in place
      // (C == res), so the compiler may not move a row's load above the previous row's store: as written before, every row of every
      // pass was its own dependent memory round trip (16 per wave and tile; with the fused key norm two more loads per row for the cos /
      // sin entries) -- the tail's latency chain.  One round trip per pass now; values and their order unchanged.
      constexpr bool PRE_RES = FAST && BF && (EK == DK_EPI_GATE_RES || EK == DK_EPI_RES);
      constexpr bool PRE_ROPE = FAST && KF;
      uint4 res_pre[PRE_RES ? MF : 1];
      f32x4 rope_pre[PRE_ROPE ? 2 * MF : 1];
      if (PRE_RES) {
#pragma unroll
        for (int itr = 0; itr < MF; ++itr) res_pre[itr] = *(const uint4*)(p.res + (physR0 + itr * 16 + rrow) * (size_t)p.ldr + col);
      }
      if (PRE_ROPE) {
        if (p.kn_rope != nullptr) {
#pragma unroll
          for (int itr = 0; itr < MF; ++itr) {
            const int kpos_ = (mrow0 + itr * 16 + rrow) % p.kn_seg_len;
            const float* tab = p.kn_rope + ((size_t)(p.kn_pos_off + kpos_) * (size_t)(p.kn_D / 2) + (size_t)(kcol >> 1)) * 2;
            rope_pre[2 * itr] = *(const f32x4*)tab, rope_pre[2 * itr + 1] = *(const f32x4*)(tab + 4);
          }
        }
      }
#pragma unroll
      for (int itr = 0; itr < MF; ++itr) {
        const int row = itr * 16 + rrow;  // row inside the wave's block of HROWS rows
        size_t crow = physC0 + row, rrow_phys = physR0 + row;
        bool valid = true;
        const int kpos = KF ? (mrow0 + row) % p.kn_seg_len : 0;  // the row's position inside its sequence
        if (!FAST) {
          valid = mrow0 + row < p.M;
          crow = (size_t)c_seg * p.c_seg_stride + c_rem;
          rrow_phys = (size_t)r_seg * p.r_seg_stride + r_rem;
          if (ep == DK_EPI_GATE_RES && valid) unpack8(*(const uint4*)(p.gate + (size_t)g_seg * p.gate_stride + col), gate8);
          for (c_rem += 16; c_rem >= p.c_seg_len; c_rem -= p.c_seg_len) ++c_seg;
          if (hres)
            for (r_rem += 16; r_rem >= p.r_seg_len; r_rem -= p.r_seg_len) ++r_seg;
          if (ep == DK_EPI_GATE_RES)
            for (g_rem += 16; g_rem >= p.gate_seg_len; g_rem -= p.gate_seg_len) ++g_seg;
        }
        float vv[8];
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (BF) {
          const u32x4 sv = *(const __attribute__((address_space(3))) u32x4*)((lds_char*)0 + reg0 + row * 64 + ((((unsigned)rc2 >> 1) ^ ((unsigned)(row >> 2) & 3u)) << 4));
          if (PLAIN) {
            if (((DK_V3_ABL & 32) ? p.alpha == -1234.5f : true) && (FAST || valid)) *(u32x4*)(Cb + crow * (size_t)ldcb + ocol) = sv;
            continue;
          }
          unpack8(make_uint4(sv[0], sv[1], sv[2], sv[3]), vv);
        } else {
          const unsigned sw = (unsigned)((row >> 1) & 7);
          a0 = *(const __attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((unsigned)rc2 ^ sw) << 4));
          a1 = *(const __attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((unsigned)(rc2 + 1) ^ sw) << 4));
        }
        if (!BF) {  // split tile
          const size_t slab_idx = (size_t)(wm * HROWS + row) * 256 + wn * 64 + ni * 32 + rc2 * 4;
          if (piece >= 1) {
            if (!(DK_V3_ABL & 8)) {  // (lab: 8 = producers do not store)
              v3_store_sc1_b128(my_slab + slab_idx, a0);
              v3_store_sc1_b128(my_slab + slab_idx + 4, a1);
            }
            continue;
          }
          for (int pp = 0; pp < ((DK_V3_ABL & 16) ? 0 : n_prod); ++pp) {  // (lab: 16 = finishers neither wait nor read)
            const float* sl = sp.slabs + (size_t)(rt * n_prod + pp) * SLAB_FLOATS + slab_idx;
            const f32x4 o0 = *(const f32x4*)sl, o1 = *(const f32x4*)(sl + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) a0[e] += o0[e], a1[e] += o1[e];
          }
        }
        if (!BF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vv[e] = round_bf16(a0[e] * p.alpha + bias8[e]);
            vv[4 + e] = round_bf16(a1[e] * p.alpha + bias8[4 + e]);
          }
        }
        if (KF) {
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
          else if (FAST || valid) rr = *(const uint4*)(p.res + rrow_phys * (size_t)p.ldr + col);
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
        uint4 o4;
        o4.x = pack2bf(vv[0], vv[1]);
        o4.y = pack2bf(vv[2], vv[3]);
        o4.z = pack2bf(vv[4], vv[5]);
        o4.w = pack2bf(vv[6], vv[7]);
        // (lab: 32 = the C stores sit behind a condition that is false at run time -- the work stays, the traffic goes)
        if (((DK_V3_ABL & 32) ? p.alpha == -1234.5f : true) && (FAST || valid)) {
#if DK_V3_NT_STORE
          const u32x4 ov = {o4.x, o4.y, o4.z, o4.w};
          __builtin_nontemporal_store(ov, (u32x4*)(Cb + crow * (size_t)ldcb + ocol));
#else
          *(uint4*)(Cb + crow * (size_t)ldcb + ocol) = o4;
#endif
        }
      }
    };
    using EkRun = std::integral_constant<int, -1>;
    using No = std::false_type;
    using Yes = std::true_type;
    if (kfuse) {  // (whole tile, bias-only epilogue -- checked by the launcher)
      if (fast)
        rows(Yes{}, Yes{}, std::integral_constant<int, DK_EPI_BIAS>{}, Yes{});
      else
        rows(No{}, Yes{}, std::integral_constant<int, DK_EPI_BIAS>{}, Yes{});
    } else if (bf_stage && fast) {
      // the common epilogues with the epilogue folded at compile time (no scalar branches inside the row loop)
      if (epi == DK_EPI_BIAS)
        rows(Yes{}, Yes{}, std::integral_constant<int, DK_EPI_BIAS>{}, No{});
      else if (epi == DK_EPI_BIAS_GELU)
        rows(Yes{}, Yes{}, std::integral_constant<int, DK_EPI_BIAS_GELU>{}, No{});
      else if (epi == DK_EPI_GATE_RES)
        rows(Yes{}, Yes{}, std::integral_constant<int, DK_EPI_GATE_RES>{}, No{});
      else if (epi == DK_EPI_RES)
        rows(Yes{}, Yes{}, std::integral_constant<int, DK_EPI_RES>{}, No{});
      else
        rows(Yes{}, Yes{}, EkRun{}, No{});
    } else if (bf_stage) {
      rows(No{}, Yes{}, EkRun{}, No{});
    } else {
      if (fast)
        rows(Yes{}, No{}, EkRun{}, No{});
      else
        rows(No{}, No{}, EkRun{}, No{});
    }
  };
  if (!((DK_V3_ABL & 64) && p.alpha != -1234.5f)) {  // (lab: 64 = no tail at run time)
    if (piece < 0) {
      tail_pass(std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{});
      tail_pass(std::integral_constant<int, 1>{}, std::true_type{}, std::false_type{});
      tail_pass(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
      tail_pass(std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{});
    } else {
      tail_pass(std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
      tail_pass(std::integral_constant<int, 1>{}, std::true_type{}, std::true_type{});
    }
  }
  if (piece >= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have completed
    __syncthreads();
    if (tid == 0) __hip_atomic_store(sp.flags + rt * n_prod + piece - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (piece == 0) {
    __syncthreads();  // every wave has read the slabs
    if (tid == 0)
      for (int pp = 0; pp < n_prod; ++pp) __hip_atomic_store(sp.flags + rt * n_prod + pp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

bool dk_gemm256v3_eligible(const GemmParams& p) {
  if (p.M <= 0 || p.N % 128 != 0 || p.K % BK != 0 || (!p.conv && p.lda % 8 != 0) || p.ldw % 8 != 0 || p.ldc % 8 != 0) return false;
  if (p.conv) {  // 3x3 / pad 1 / stride 1 (optionally over the nearest-x2 view); pixel packed as b:8 | y:12 | x:12, 31-bit byte offsets
    if (p.N % 256 != 0) return false;  // (a half column tile would waste half the MFMAs of a 128-channel stage: the 128^2 kernel's)
    if (p.ups < 0 || p.ups > 1 || p.cC % BK != 0 || p.K != 9 * p.cC || p.M != p.cB * p.cH * p.cW || p.n_split != 0) return false;
    if (p.cB > 256 || p.cH > 4096 || p.cW > 4096 || (p.ups == 1 && (p.cH % 2 != 0 || p.cW % 2 != 0))) return false;
    if ((size_t)p.cB * (p.cH >> p.ups) * (p.cW >> p.ups) * p.cC * 2 >= (1ull << 31) || ((uintptr_t)p.A & 15) != 0) return false;
  }
  if (p.n_split % 256 != 0 || (p.n_split > 0 && (p.C2 == nullptr || p.ldc2 % 8 != 0 || p.n_split >= p.N))) return false;
  if (p.a_seg_len <= 0 || p.c_seg_len <= 0) return false;
  const bool res1 = p.epi == DK_EPI_GATE_RES || p.epi == DK_EPI_RES;
  const bool res2 = p.n_split > 0 && (p.epi2 == DK_EPI_GATE_RES || p.epi2 == DK_EPI_RES);
  if ((res1 || res2) && (p.res == nullptr || p.r_seg_len <= 0 || p.ldr % 8 != 0)) return false;
  if ((p.epi == DK_EPI_GATE_RES || (p.n_split > 0 && p.epi2 == DK_EPI_GATE_RES)) && (p.gate == nullptr || p.gate_seg_len <= 0)) return false;
  if (p.kn_w != nullptr) {  // fused key QKNorm + RoPE: whole 256-column tiles of 128- or 64-column heads, bias-only first output
    if (p.conv || p.epi != DK_EPI_BIAS || (p.kn_D != 128 && p.kn_D != 64) || p.kn_seg_len <= 0 || p.kn_col0 % 256 != 0 || p.kn_col1 % 256 != 0 || p.kn_col0 >= p.kn_col1 ||
        p.kn_col1 > (p.n_split > 0 ? p.n_split : p.N) || ((uintptr_t)p.kn_w & 15) != 0 || ((uintptr_t)p.kn_rope & 15) != 0)
      return false;
    if (p.qn_w != nullptr && (p.qn_col0 % 256 != 0 || p.qn_col1 % 256 != 0 || p.qn_col0 >= p.qn_col1 || p.qn_col1 > (p.n_split > 0 ? p.n_split : p.N) ||
                              (p.qn_col0 < p.kn_col1 && p.kn_col0 < p.qn_col1) || ((uintptr_t)p.qn_w & 15) != 0))
      return false;
  } else if (p.qn_w != nullptr) {
    return false;  // the query side rides on the key side's machinery
  }
  // 16-byte accesses in the tail
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!al16(p.C) || !al16(p.C2) || !al16(p.res) || !al16(p.bias) || !al16(p.gate) || (p.gate != nullptr && p.gate_stride % 8 != 0)) return false;
  // 32-bit byte offsets on the DMA side: the A rows this problem touches and 8 rows of W
  const size_t a_rows = (size_t)((p.M - 1) / p.a_seg_len) * p.a_seg_stride + (size_t)((p.M - 1) % p.a_seg_len) + 1;
  return (p.conv || a_rows * (size_t)p.lda * 2 < (1ull << 32)) && (size_t)p.ldw * 2 * 8 < (1ull << 31);
}

// dk_tune_set("gemm_split", v): -1 (default) split a remainder wave of at most half the CUs into equal pieces when the
// cost model below says it pays, 0 never, 1 whenever possible.  Kernel lab (profiles/archive/r01_gemm_lab.md): every workgroup
// carries ~18 us of fixed cost (launch, first DMA, tail) and a CU runs its K-tiles ~20 % slower when all 256 CUs are busy
// than when 192 are, so a remainder of MORE than half the CUs (finisher + several producer pieces in turn on the
// spare CUs) loses on every shape but the longest-K one and is only taken when forced.
int g_dk_v3_split = -1;
int g_dk_v3_split_min = -1;  // dk_tune_set("gemm_split_min", v): saved K-tile steps below which a Linear that is ALL remainder stays whole; -1: 32 (see plan_split)

// 256 fp32 tile images + 4 KiB of flags (and the error word)
size_t dk_gemm_split_workspace_bytes() { return (size_t)256 * SLAB_FLOATS * 4 + 4096; }

// How the tiles beyond the last full wave of the CUs are cut along K (see SplitArgs).  n_rem == 0: no split.
struct SplitPlan {
  int n_dp, n_rem, S, ks;
};
static SplitPlan plan_split(int tiles, int nk, bool have_ws, int n_cu, bool linear = false) {
  SplitPlan none{tiles, 0, 1, nk};
  if (!have_ws || g_dk_v3_split == 0 || n_cu < 16) return none;
  const int G = n_cu & ~7;
  const int T = tiles % G;
  if (T == 0) return none;
  const int E = G - T;
  int S, ks, t_steps;  // t_steps: K-tile steps until the split wave is done
  if (T > G / 2) {  // one producer piece per tile, c = ceil(T / E) of them in turn on each of the E spare CUs
    if (g_dk_v3_split < 0) return none;
    S = 2;
    const int c = (T + E - 1) / E;
    ks = (nk * c + c) / (c + 1);  // ~ nk * c / (c + 1), rounded up: the finishers must not end before the producers
    if (ks > nk - 1) ks = nk - 1;
    t_steps = ks > c * (nk - ks) ? ks : c * (nk - ks);
  } else {  // S equal pieces per tile, one CU each
    S = G / T < 4 ? G / T : 4;
    ks = (nk + S - 1) / S;
    t_steps = ks;
  }
  if (S < 2 || ks < 1 || nk - ks < S - 1 || T * (S - 1) > 256) return none;
  // a K-tile step costs about 1.45 us; splitting costs a slab write + read and a flag round trip per tile
  if (g_dk_v3_split < 0 && (nk - t_steps) * 1.45 < 25.0) return none;
  // a Linear that is ALL remainder (round 6: FLUX / SD3 below 1024 x 1024, 60 - 120 tiles): every tile pays S - 1 slab round trips at once, and the few
  // busy CUs run their K-tiles in ~1 us.  Measured (profiles/r06_gemm_small_m.log, r06_res_sweep.md): with the fp32-image exchange of rounds 1-5 FLUX's o_proj at
  // 512 x 512 (K = 3072, four ranges: 36 steps saved) lost 16 us with the split and fc2 / linear2 (144 / 180 saved) gained 60 - 90; with the raw-accumulator
  // exchange (above, round 6) the in-model sweep is flat from 0 to 48 saved steps and 1 % better at 24 - 36 than at 48: the rule cuts from 32
  if (g_dk_v3_split < 0 && linear && tiles < G && nk - t_steps < (g_dk_v3_split_min >= 0 ? g_dk_v3_split_min : 32)) return none;
  return SplitPlan{tiles - T, T, S, ks};
}

// dk_tune_set("gemm_mf", v): wave-tile height in 16-row fragments; -1 (default) = the height with the fewest rounds x height, 8 / 7 forced
int g_dk_v3_mf = -1;

// Tile height for a launch.  Model: rounds of the CUs x rows per tile, over both problems of a grouped launch (same N).  Measured
// (profiles/r02_gemm_tile_height.log): on the FLUX shapes 224-row tiles save 2.7 % of the GEMM time -- far less than the 12.5 %
// the model promises, because a K-tile runs slower the more CUs are busy (the chip is power / fabric bound) -- and on
// the short-K SD3 shapes the 15 % extra tiles (each with its fixed prologue + tail) cost more than the fuller round gives back.
// So: 224-row tiles only for long reductions, and only when the model predicts at least 10 %.
static long v3_tiles(const GemmParams& p, const GemmParams* p2, int bm) {
  long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + T256 - 1) / T256);
  if (p2) tiles += (long)((p2->M + bm - 1) / bm) * ((p2->N + T256 - 1) / T256);
  return tiles;
}
static bool v3_have_ws(const GemmParams& p, const GemmParams* p2) {
  // (a launch with the fused key QKNorm is never split: a split tile's finisher has no second pass over its row sums)
  return p.workspace != nullptr && p.workspace_bytes >= dk_gemm_split_workspace_bytes() && ((uintptr_t)p.workspace & 255) == 0 &&
         p.kn_w == nullptr && (p2 == nullptr || p2->kn_w == nullptr);
}
static int pick_mf(const GemmParams& p, const GemmParams* p2, int n_cu, bool have_ws) {
  if (g_dk_v3_mf == 7 || g_dk_v3_mf == 8) return g_dk_v3_mf;
  if (p.K < 2048) return 8;
  // Small launches (round 6; the reference CLI's 512 x 512 default: FLUX's o_proj / fc2 / linear2 are 60 - 84 tiles for 256 CUs): when both
  // heights leave at least half the CUs idle the launch is one split "remainder", and what a CU runs is the finisher piece: ks K-tiles of bm rows
  const long t7 = v3_tiles(p, p2, 224), t8 = v3_tiles(p, p2, 256);
  if (have_ws && !p.conv && (t7 < t8 ? t7 : t8) * 2 <= n_cu && t7 <= n_cu && t8 <= n_cu) {
    const int nk = p.K / BK;
    const SplitPlan s7 = plan_split((int)t7, nk, true, n_cu, true), s8 = plan_split((int)t8, nk, true, n_cu, true);
    if (s7.n_rem > 0 || s8.n_rem > 0) return 224L * (s7.n_rem > 0 ? s7.ks : nk) < 256L * (s8.n_rem > 0 ? s8.ks : nk) ? 7 : 8;
  }
  long cost[2];
  for (int mf = 7; mf <= 8; ++mf) {
    const int bm = 32 * mf;
    cost[mf - 7] = ((v3_tiles(p, p2, bm) + n_cu - 1) / n_cu) * bm;
  }
  return cost[0] * 10 <= cost[1] * 9 ? 7 : 8;
}

// The whole launch is at most half a round of the CUs and this kernel would cut every tile along K (dk_use_v4 in gemm.hip leaves such
// launches here: the one-wave-per-SIMD kernel has no K split, and 60 tiles on 256 CUs waste three quarters of the chip)
bool dk_gemm256v3_splits_whole_launch(const GemmParams& p, const GemmParams* p2) {
  if (p.conv || !dk_gemm256v3_eligible(p) || (p2 != nullptr && !dk_gemm256v3_eligible(*p2))) return false;
  const int n_cu = dk_device_cu_count();
  const bool have_ws = v3_have_ws(p, p2);
  const int bm = 32 * pick_mf(p, p2, n_cu, have_ws);
  const long tiles = v3_tiles(p, p2, bm);
  if (tiles * 2 > n_cu) return false;
  const SplitPlan pl = plan_split((int)tiles, p.K / BK, have_ws, n_cu, true);
  return pl.n_rem > 0 && pl.n_dp == 0;
}

// `p2` null: one problem.  (tiles_a / tiles_b of older callers are recomputed here: they depend on the tile height)
int dk_launch_gemm256v3_raw(const GemmParams& p, const GemmParams& pb, int /*tiles_a*/, int tiles_b_in, hipStream_t stream) {
  const int n_cu = dk_device_cu_count();
  const bool two = tiles_b_in > 0;
  const bool have_ws = v3_have_ws(p, two ? &pb : nullptr);
  const int mf = pick_mf(p, two ? &pb : nullptr, n_cu, have_ws);
  const int bm = 32 * mf;
  const int tiles_a = ((p.M + bm - 1) / bm) * ((p.N + T256 - 1) / T256);
  const int tiles_b = two ? ((pb.M + bm - 1) / bm) * ((pb.N + T256 - 1) / T256) : 0;
  const SplitPlan pl = plan_split(tiles_a + tiles_b, p.K / BK, have_ws, n_cu, !p.conv);
  if (g_dk_gemm_plan != nullptr) {
    DkGemmPlan& gp = *g_dk_gemm_plan;
    gp.kernel = 3; gp.tile_rows = bm; gp.tiles = tiles_a + tiles_b; gp.workgroups = pl.n_dp + pl.n_rem * pl.S; gp.split_tiles = pl.n_rem;
    gp.k_pieces = pl.n_rem > 0 ? pl.S : 1; gp.ks = pl.n_rem > 0 ? pl.ks : p.K / BK; gp.n_cu = n_cu; gp.launches += 1;
    return 0;
  }
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v3_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v3_kernel<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v3_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v3_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_once.mark();
  }
  SplitArgs sp;
  memset(&sp, 0, sizeof(sp));
  sp.n_dp = pl.n_dp; sp.n_rem = pl.n_rem; sp.S = pl.S; sp.ks = pl.ks;
  if (pl.n_rem > 0) {
    sp.slabs = (float*)p.workspace;
    sp.flags = (unsigned*)((char*)p.workspace + (size_t)256 * SLAB_FLOATS * 4);
    sp.error_word = sp.flags + 512;
  }
  const int grid = pl.n_dp + pl.n_rem * pl.S;
  if (p.conv) {
    if (mf == 8)
      hipLaunchKernelGGL((dk_gemm256v3_kernel<8, true>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sp);
    else
      hipLaunchKernelGGL((dk_gemm256v3_kernel<7, true>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sp);
  } else if (mf == 8)
    hipLaunchKernelGGL((dk_gemm256v3_kernel<8, false>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sp);
  else
    hipLaunchKernelGGL((dk_gemm256v3_kernel<7, false>), dim3(grid), dim3(512), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sp);
  return 0;
}

// tile-parallel launch of `p` and, optionally, a second problem `p2` with the same N, K, alpha and epilogue
int dk_launch_gemm256v3(const GemmParams& p, const GemmParams* p2, hipStream_t stream) {
  DK_REQUIRE(dk_gemm256v3_eligible(p), "gemm256v3: shape / strides not eligible");
  if (p2) {
    DK_REQUIRE(!p.conv && !p2->conv, "gemm256v3: no grouped convolutions");
    DK_REQUIRE(dk_gemm256v3_eligible(*p2), "gemm256v3: second problem not eligible");
    DK_REQUIRE(p2->N == p.N && p2->K == p.K && p2->epi == p.epi && p2->alpha == p.alpha && p2->n_split == p.n_split &&
                   (p.n_split == 0 || p2->epi2 == p.epi2),
               "grouped GEMM: N, K, epilogue must match");
  }
  double work = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  if (p2) work += 2.0 * (double)p2->M * (double)p2->N * (double)p2->K;
  if (g_dk_gemm_plan != nullptr) return dk_launch_gemm256v3_raw(p, p2 ? *p2 : p, 0, p2 ? 1 : 0, stream);  // (plan mode: no device call)
  dk_prof_begin(p.conv ? 1 : 0, work, stream);
  const int rc = dk_launch_gemm256v3_raw(p, p2 ? *p2 : p, 0, p2 ? 1 : 0, stream);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return rc;
}
