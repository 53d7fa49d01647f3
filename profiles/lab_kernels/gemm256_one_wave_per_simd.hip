// LAB KERNEL, not part of libdk_hip.so (see DESIGN.md section 7.1 and profiles/archive/r01_gemm_lab_v4_*.log): the contract, tile order,
// LDS-DMA ring, tail and remainder split of diffusionkit_amd/csrc/gemm256v3.hip with ONE wave per SIMD -- 4 waves (2 x 2),
// wave tile 128 x 128, accumulators pinned by register class through inline-asm MFMAs (192 AGPR + 64 VGPR) -- and the K-loop
// schedule the measurements point to: the fragments of a whole K = 32 slice of both operands are read one slice ahead, an
// operand's half of a ring slot is released (barrier) as soon as every wave has read its second slice and the DMA of K-tile
// i+2 goes out right behind it, a third barrier with vmcnt(16) marks K-tile i+1 as landed.  Round-1 result on MI355X: the
// steady-state loop is ~2 % faster per K-tile than v3's (K = 16384: 1373 vs 1366 TF), the 4-wave tail costs 5 us more per
// tile (19.4 vs 14.5 us fixed cost), so every K <= 3072 shape loses 8-15 %.  To build it into the lab library: copy it to
// diffusionkit_amd/csrc/gemm256v4.hip, add it to the Makefile / scripts/build_lab.sh and forward dk_launch_gemm256v3_raw to
// dk_launch_gemm256v4_raw.
#include <cstring>
#include <type_traits>

#include "dk_kernels.h"

#define T256 256
#define BK 64
#define HALF_BYTES (128 * BK * 2)
#define KT_BYTES (4 * HALF_BYTES)
#define LDS_BYTES (2 * KT_BYTES)

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
__device__ __forceinline__ void v4_store_sc1_b128(float* ptr, f32x4 v) {
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

__global__ __launch_bounds__(256, 1) void dk_gemm256v4_kernel(GemmParams pa, GemmParams pb, int tiles_a, int tiles_b, SplitArgs sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  const GemmParams& p = second ? pb : pa;
  const int tl = second ? tile - tiles_a : tile;  // tile index inside its problem
  const int nbm = (p.M + T256 - 1) / T256, nbn = p.N / T256;

  // ---- lane-constant parts of the LDS fragment addresses: row l15 (+ 16 * fragment), chunk 4*kk + q ----
  unsigned offk[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) offk[kk] = (unsigned)(l15 * 128 + (((kk * 4 + q) ^ (l15 >> 1)) << 4));
  const unsigned sA = wm * HALF_BYTES;
  const unsigned sW = (2 + wn) * HALF_BYTES;

  const int srow = lane >> 3;
  const int GROUP = 4;
  const int tpg = GROUP * nbn;
  const int g = tl / tpg;
  const int first_m = g * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int tm = first_m + (tl % tpg) % gsz;
  const int tn = (tl % tpg) / gsz;
  const int m0 = tm * T256, n0 = tn * T256;

  // DMA sources: A rows through the segment map per lane (32-bit byte offsets from p.A; rows beyond M - 1 re-read
  // the last row, their results are never stored), W rows from a tile-uniform base + lane part
  unsigned la[2][4], lw[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * (j & 1));  // = (lane&7) ^ (((wave*32 + j*8 + srow) >> 1) & 7)
    if (j < 2) lw[j] = ((unsigned)srow * (unsigned)p.ldw + chunk * 8) * 2u;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int m = min(m0 + hh * 128 + wave * 32 + j * 8 + srow, p.M - 1);
      const unsigned phys = (unsigned)((m / p.a_seg_len) * p.a_seg_stride + (m % p.a_seg_len));
      la[hh][j] = (phys * (unsigned)p.lda + chunk * 8) * 2u;
    }
  }
  const char* gA = (const char*)p.A + (size_t)k0 * (BK * 2);
  const char* gW = (const char*)p.W + ((size_t)n0 + wave * 32) * (size_t)p.ldw * 2 + (size_t)k0 * (BK * 2);
  const size_t w128 = (size_t)128 * p.ldw * 2, w8 = (size_t)8 * p.ldw * 2;

  // LDS-DMA in the buffer form: SGPR resource (base, 4 GiB range) + 32-bit lane offset + scalar offset
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)gA, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)gW, 0, -1, 0x00020000);
  auto issue_piece = [&](int i, int gidx) {  // one of the 16 DMA instructions of K-tile i: (operand, half, j)
    const int op = gidx & 1, hh = (gidx >> 1) & 1, j = gidx >> 2;
    const unsigned dst0 = (i & 1) * KT_BYTES + (wave * 32) * 128;
    if (op == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)((lds_char*)0 + dst0 + hh * HALF_BYTES + j * 1024), 16, (int)la[hh][j],
                                               i * (BK * 2), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)((lds_char*)0 + dst0 + (2 + hh) * HALF_BYTES + j * 1024), 16, (int)lw[j & 1],
                                               (int)(hh * w128 + j * w8) + i * (BK * 2), 0, 0);
  };
  auto issue_tile = [&](int i) {
#pragma unroll
    for (int gidx = 0; gidx < 16; ++gidx) issue_piece(i, gidx);
  };

  f32x4 acc[8][8];  // [nf][mf]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#define V4_NF_AGPR 6
  // ---------------- K loop: whole-slice fragment prefetch, per-operand slot release, three barriers per K-tile ----------------
#define V5_RD1(DST, ADDR, F)                                                                           \
  do {                                                                                                 \
    if ((F) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(DST) : "v"(ADDR));                          \
    else if ((F) == 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(DST) : "v"(ADDR));         \
    else if ((F) == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(DST) : "v"(ADDR));         \
    else if ((F) == 3) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(DST) : "v"(ADDR));         \
    else if ((F) == 4) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(DST) : "v"(ADDR));         \
    else if ((F) == 5) asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(DST) : "v"(ADDR));        \
    else if ((F) == 6) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(DST) : "v"(ADDR));        \
    else asm volatile("ds_read_b128 %0, %1 offset:14336" : "=v"(DST) : "v"(ADDR));                      \
  } while (0)
#define V5_LGKM0(V)                                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                        \
               : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7])::"memory")
#define V5_MFMA(E, WSET, XSET)                                                                          \
  do {                                                                                                  \
    if (((E) >> 3) < V4_NF_AGPR)                                                                        \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(E) >> 3][(E) & 7]) : "v"(WSET[(E) >> 3]), "v"(XSET[(E) & 7])); \
    else                                                                                                \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(E) >> 3][(E) & 7]) : "v"(WSET[(E) >> 3]), "v"(XSET[(E) & 7])); \
  } while (0)
  // piece numbering of issue_piece: even = A, odd = W; (hh, j) = ((g >> 1) & 1, g >> 2)
  {
    bf16x8 w0[8], x0[8], w1[8], x1[8];  // fragments of slice kk = 0 / kk = 1 of the current K-tile (nf / mf = 0..7)
    // prologue: K-tiles 0 and 1 into the two slots, K-tile 0 landed, its first slice read
    issue_tile(0);
    issue_tile(nk > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) V5_RD1(w0[f], offk[0] + sW, f);
#pragma unroll
    for (int f = 0; f < 8; ++f) V5_RD1(x0[f], offk[0] + sA, f);
    V5_LGKM0(w0);
    V5_LGKM0(x0);
    for (int i = 0; i < nk; ++i) {
      const unsigned bo = (i & 1) * KT_BYTES;
      const unsigned aW1 = offk[1] + sW + bo, aA1 = offk[1] + sA + bo;
      const unsigned aW0n = offk[0] + sW + (bo ^ KT_BYTES), aA0n = offk[0] + sA + (bo ^ KT_BYTES);
      const int tn = min(i + 2, nk - 1);  // K-tile whose DMA goes into this tile's slot (clamped: the last two re-fetch the last tile)
      // ---- slice 0: 64 MFMAs from (w0, x0); the second slice of W, then of A, is read meanwhile
#define V5_S0(E0)                                                                                         \
  _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                     \
    const int e = (E0) + e_;                                                                              \
    if (e < 16 && (e & 1) == 0) V5_RD1(w1[e >> 1], aW1, e >> 1);                                          \
    if (e == 20) { /* every wave has read all of this slot's W rows: their DMA for K-tile i+2 may go */   \
      V5_LGKM0(w1);                                                                                       \
      __builtin_amdgcn_s_barrier();                                                                       \
      asm volatile("" ::: "memory");                                                                      \
    }                                                                                                     \
    if (e >= 21 && e < 37 && (e & 1) == 1) issue_piece(tn, 2 * ((e - 21) >> 1) + 1);                      \
    if (e >= 22 && e < 38 && (e & 1) == 0) V5_RD1(x1[(e - 22) >> 1], aA1, (e - 22) >> 1);                 \
    if (e == 44) { /* ... and all of its A rows */                                                        \
      V5_LGKM0(x1);                                                                                       \
      __builtin_amdgcn_s_barrier();                                                                       \
      asm volatile("" ::: "memory");                                                                      \
    }                                                                                                     \
    if (e >= 45 && e < 61 && (e & 1) == 1) issue_piece(tn, 2 * ((e - 45) >> 1));                          \
    V5_MFMA(e, w0, x0);                                                                                   \
  }
      V5_S0(0)
      V5_S0(16)
      V5_S0(32)
      V5_S0(48)
#undef V5_S0
      // ---- slice 1: 64 MFMAs from (w1, x1); K-tile i+1 has landed, its first slice is read meanwhile
#define V5_S1(E0)                                                                                         \
  _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                     \
    const int e = (E0) + e_;                                                                              \
    if (e == 6) {                                                                                         \
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); /* the 16 pieces issued above may fly; K-tile i+1 has landed */ \
      __builtin_amdgcn_s_barrier();                                                                       \
      asm volatile("" ::: "memory");                                                                      \
    }                                                                                                     \
    if (e >= 8 && e < 56 && (e % 3) == 2) {                                                               \
      const int r = (e - 8) / 3; /* 0..15 */                                                              \
      if (r < 8) V5_RD1(w0[r & 7], aW0n, r);                                                              \
      else V5_RD1(x0[(r - 8) & 7], aA0n, r - 8);                                                          \
    }                                                                                                     \
    V5_MFMA(e, w1, x1);                                                                                   \
  }
      V5_S1(0)
      V5_S1(16)
      V5_S1(32)
      V5_S1(48)
#undef V5_S1
      V5_LGKM0(w0);
      V5_LGKM0(x0);
    }
    // the DMA of the clamped extra tiles must not land in the tail's staging image; MFMA -> VALU read distance (inline asm)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef V5_RD1
#undef V5_LGKM0
#undef V5_MFMA

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
  auto inside = [&](int len) { return m0 / len == (m0 + T256 - 1) / len; };
  const bool fast = m0 + T256 <= p.M && inside(p.c_seg_len) && (!has_res || inside(p.r_seg_len)) &&
                    (epi != DK_EPI_GATE_RES || inside(p.gate_seg_len));
  const int mrow0 = m0 + wm * 128;  // first GEMM row of this wave's block
  const size_t physC0 = (size_t)((m0 / p.c_seg_len) * p.c_seg_stride + (m0 % p.c_seg_len)) + wm * 128;
  const size_t physR0 = has_res ? (size_t)((m0 / p.r_seg_len) * p.r_seg_stride + (m0 % p.r_seg_len)) + wm * 128 : 0;
  const bf16_t* gate_row = epi == DK_EPI_GATE_RES ? p.gate + (size_t)(m0 / p.gate_seg_len) * p.gate_stride : nullptr;
  const unsigned reg0 = (unsigned)wave * 16384u;  // this wave's 16 KiB staging image
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
  if (piece == 0) {
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

#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    // stage: lane owns row mf*16 + l15, columns (nf & 1)*16 + 4*q + {0..3} of this 32-column half
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < 8; ++mf) {
        const int row = mf * 16 + l15;
        *(__attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((nf * 4 + q) ^ ((row >> 1) & 7)) << 4)) = acc[ni * 2 + nf][mf];
      }
    // (same wave writes and reads the image: program order + the compiler's lgkmcnt suffice)
    const int col = n0 + wn * 128 + ni * 32 + rc2 * 4;      // first of this lane's 8 columns of the GEMM (bias, gate, residual)
    const int ocol = ncol0 + wn * 128 + ni * 32 + rc2 * 4;  // the same inside the output it goes to
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gate8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && piece < 1) unpack8(*(const uint4*)(p.bias + col), bias8);
    // the row loop is instantiated twice -- tile-uniform maps (FAST) or a per-lane walk through the maps -- so that
    // the common case keeps its short body (one v_add per address, batched loads)
    auto rows = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      if (FAST && epi == DK_EPI_GATE_RES) unpack8(*(const uint4*)(gate_row + col), gate8);
      // row walk of the slow path: (segment, row inside it) of this lane's current row in each map; 16 rows per step
      int c_seg = 0, c_rem = 0, r_seg = 0, r_rem = 0, g_seg = 0, g_rem = 0;
      if (!FAST) {
        const int ms = mrow0 + rrow;
        c_seg = ms / p.c_seg_len, c_rem = ms % p.c_seg_len;
        if (has_res) r_seg = ms / p.r_seg_len, r_rem = ms % p.r_seg_len;
        if (epi == DK_EPI_GATE_RES) g_seg = ms / p.gate_seg_len, g_rem = ms % p.gate_seg_len;
      }
#pragma unroll 4
      for (int itr = 0; itr < 8; ++itr) {
        const int row = itr * 16 + rrow;  // row inside the wave's 128-row block
        size_t crow = physC0 + row, rrow_phys = physR0 + row;
        bool valid = true;
        if (!FAST) {
          valid = mrow0 + row < p.M;
          crow = (size_t)c_seg * p.c_seg_stride + c_rem;
          rrow_phys = (size_t)r_seg * p.r_seg_stride + r_rem;
          if (epi == DK_EPI_GATE_RES && valid) unpack8(*(const uint4*)(p.gate + (size_t)g_seg * p.gate_stride + col), gate8);
          for (c_rem += 16; c_rem >= p.c_seg_len; c_rem -= p.c_seg_len) ++c_seg;
          if (has_res)
            for (r_rem += 16; r_rem >= p.r_seg_len; r_rem -= p.r_seg_len) ++r_seg;
          if (epi == DK_EPI_GATE_RES)
            for (g_rem += 16; g_rem >= p.gate_seg_len; g_rem -= p.gate_seg_len) ++g_seg;
        }
        const unsigned sw = (unsigned)((row >> 1) & 7);
        f32x4 a0 = *(const __attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((unsigned)rc2 ^ sw) << 4));
        f32x4 a1 = *(const __attribute__((address_space(3))) f32x4*)((lds_char*)0 + reg0 + row * 128 + (((unsigned)(rc2 + 1) ^ sw) << 4));
        if (piece >= 0) {  // split tile
          const size_t slab_idx = (size_t)(wm * 128 + row) * 256 + wn * 128 + ni * 32 + rc2 * 4;
          if (piece >= 1) {
            v4_store_sc1_b128(my_slab + slab_idx, a0);
            v4_store_sc1_b128(my_slab + slab_idx + 4, a1);
            continue;
          }
          for (int pp = 0; pp < n_prod; ++pp) {
            const float* sl = sp.slabs + (size_t)(rt * n_prod + pp) * SLAB_FLOATS + slab_idx;
            const f32x4 o0 = *(const f32x4*)sl, o1 = *(const f32x4*)(sl + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) a0[e] += o0[e], a1[e] += o1[e];
          }
        }
        float vv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vv[e] = round_bf16(a0[e] * p.alpha + bias8[e]);
          vv[4 + e] = round_bf16(a1[e] * p.alpha + bias8[4 + e]);
        }
        if (epi == DK_EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = gelu_erf_f(vv[e]);
        } else if (epi == DK_EPI_BIAS_SILU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = silu_f(vv[e]);
        } else if (has_res) {
          uint4 rr = make_uint4(0u, 0u, 0u, 0u);
          if (FAST || valid) rr = *(const uint4*)(p.res + rrow_phys * (size_t)p.ldr + col);
          float r8[8];
          unpack8(rr, r8);
          if (epi == DK_EPI_GATE_RES) {
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
        if (FAST || valid) *(uint4*)(Cb + crow * (size_t)ldcb + ocol) = o4;
      }
    };
    if (fast)
      rows(std::true_type{});
    else
      rows(std::false_type{});
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

// remainder split: the plan of gemm256v3.hip (same fixed costs per workgroup, same tuning knob)
// How the tiles beyond the last full wave of the CUs are cut along K (see SplitArgs).  n_rem == 0: no split.
struct SplitPlan {
  int n_dp, n_rem, S, ks;
};
static SplitPlan plan_split(int tiles, int nk, bool have_ws, int n_cu) {
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
  return SplitPlan{tiles - T, T, S, ks};
}

int dk_launch_gemm256v4_raw(const GemmParams& p, const GemmParams& pb, int tiles_a, int tiles_b, hipStream_t stream) {
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    int dev = 0;
    DK_CHECK_HIP(hipGetDevice(&dev));
    DK_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_set = true;
  }
  const bool have_ws = p.workspace != nullptr && p.workspace_bytes >= dk_streamk_workspace_bytes() && ((uintptr_t)p.workspace & 255) == 0;
  const SplitPlan pl = plan_split(tiles_a + tiles_b, p.K / BK, have_ws, n_cu);
  SplitArgs sp;
  memset(&sp, 0, sizeof(sp));
  sp.n_dp = pl.n_dp; sp.n_rem = pl.n_rem; sp.S = pl.S; sp.ks = pl.ks;
  if (pl.n_rem > 0) {
    sp.slabs = (float*)p.workspace;
    sp.flags = (unsigned*)((char*)p.workspace + (size_t)256 * SLAB_FLOATS * 4);
    sp.error_word = sp.flags + 512;
  }
  const int grid = pl.n_dp + pl.n_rem * pl.S;
  hipLaunchKernelGGL(dk_gemm256v4_kernel, dim3(grid), dim3(256), LDS_BYTES, stream, p, pb, tiles_a, tiles_b, sp);
  return 0;
}
