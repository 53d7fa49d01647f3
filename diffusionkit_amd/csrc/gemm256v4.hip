// 256 x 256 x 64 bf16 MFMA GEMM, fourth generation: ONE wave per SIMD (256 threads), a 128 x 128 block of the tile per wave with its
// 256 accumulators in AGPRs, and the whole main body -- prologue DMA, K loop, the two peeled last K-tiles, the drain of the accumulators
// into the bf16 staging image -- as ONE hand-scheduled inline-asm block with the register file addressed by hand
// (gemm256v4_asm.inc, written and CPU-checked by scripts/gen_gemm256v4.py: an instruction-level emulator runs the same instruction
// list with every DMA piece and LDS read completing as late, or as early, as its waits allow).  Same contract, grouped launch, column
// split and epilogues as gemm256v3.hip; nn.Linear call sites python/src/diffusionkit/mlx/mmdit.py:821-832 and the fused
// linear1 / linear2 of the single-stream blocks (:693-751).
//
// Why (NOTES_r03 open item 1): in the 8-wave frame of gemm256v3.hip a wave holds 128 accumulators and re-reads its
// fragments from LDS for every 32 MFMAs; the lone wave of this frame reads a whole K = 32 slice of BOTH operands (64 registers) one
// slice ahead of the 64 MFMAs that use it -- a third fewer LDS reads per MAC, one non-MFMA instruction between two MFMAs, and each
// operand's ring slot is released by its own barrier as soon as its second slice has been read, so the LDS-DMA pieces of K-tile i + 2
// have 1.25 - 1.5 K-tiles to land in a TWO-slot ring (128 KiB, the rest of the LDS stays free).  The schedule follows the shape of the
// vendor library's MT256x256x64 kernel as read from its disassembly (profiles/NOTES_r05.md); nothing of it is linked or copied.
//
// Shapes: N % 256 == 0, K % 64 == 0, any M, any row-segment maps; no K split, no half tiles, no convolution form (gemm256v3.hip keeps
// those).  C / D layout as in gemm256v3.hip: lane holds row m = mf*16 + (lane & 15), columns n = nf*16 + 4*(lane >> 4) + {0..3}.
//
// Tail: the asm block leaves round_bf16(alpha * acc + bias) in the LDS image gemm256v3.hip's tail stages (eight wave-private 16 KiB
// regions; a wave of this kernel owns two of them), so the read-back below is that kernel's: 8 columns per lane, one 16-byte store
// per lane and row, epilogues folded at compile time.  The fused QKNorm + RoPE of key / query tiles takes its row sums from the
// staged image (a wave's 128 columns hold whole heads: no exchange between waves).
#include <cstring>
#include <type_traits>

#include "dk_kernels.h"

#define V4_T 256
#define V4_BK 64
#define V4_LDS_BYTES (4 * 32768)

typedef __attribute__((address_space(3))) char v4_lds_char;
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// position of block `bid` inside the XCD-contiguous order of `count` blocks (hardware places block b on XCD b & 7)
__device__ __forceinline__ int v4_xcd_contiguous(int bid, int count) {
  const int x = bid & 7;
  int start = 0;
  for (int y = 0; y < x; ++y) start += y < count ? (count - y + 7) >> 3 : 0;
  return start + (bid >> 3);
}

#ifndef V4_NT
#define V4_NT 0  // lab: cache policy of the C stores -- 0 plain (L2 write-back), 1 non-temporal (nt), 2 write-through to memory (sc0 sc1)
#endif
__device__ __forceinline__ void v4_store16(bf16_t* ptr, u32x4 v) {
#if V4_NT == 1
  __builtin_nontemporal_store(v, (u32x4*)ptr);
#elif V4_NT == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(v) : "memory");
#else
  *(u32x4*)ptr = v;
#endif
}

struct V4Tail {
  const GemmParams __attribute__((address_space(4))) * p;
  bf16_t* Cb;
  int ldcb, epi, n0, ncol0, m0, wm, wn2, lane;
  bool has_res;
  const bf16_t* nw;  // QKNorm weight of this tile (key or query), or null
};

// read-back of the staged tile: see gemm256v3.hip (rows lambda); FAST: tile-uniform row maps; EK: epilogue folded at compile time (-1: run time);
// KF: key / query tile with the fused QKNorm + RoPE
// CUT (with FAST): the tile lies inside one segment of every map but M ends inside it (the last row tile: 4352 rows in 224-row tiles) -- the
// tile-uniform addressing of FAST with a row limit: rows past it are neither stored nor (residual) loaded from their own address
template <bool FAST, int EK, bool KF, int MF, bool CUT = false>
__device__ __forceinline__ void v4_rows(const V4Tail& t) {
  static_assert(FAST || !CUT, "CUT is a form of FAST");
  constexpr int HROWS = 16 * MF;  // rows of a wave's block (the LDS image keeps 128-row regions; a 224-row tile uses 112 of each)
  const auto& p = *t.p;
  constexpr bool PLAIN = EK == DK_EPI_BIAS && !KF;
  const int lane = t.lane;
  const int rrow = lane >> 2, rc2 = (lane & 3) * 2;
  const int mrow0 = t.m0 + t.wm * HROWS;
  const int row_lim = CUT ? t.p->M - 1 - mrow0 : HROWS;  // last valid row of this wave's block
  if (CUT && row_lim < 0) return;                        // (the whole block lies past M)
  const int ep = EK >= 0 ? EK : t.epi;
  const bool hres = EK >= 0 ? (EK == DK_EPI_GATE_RES || EK == DK_EPI_RES) : t.has_res;
  const size_t physC0 = (size_t)((t.m0 / p.c_seg_len) * p.c_seg_stride + (t.m0 % p.c_seg_len)) + t.wm * HROWS;
  const size_t physR0 = hres ? (size_t)((t.m0 / p.r_seg_len) * p.r_seg_stride + (t.m0 % p.r_seg_len)) + t.wm * HROWS : 0;
  const bf16_t* gate_row = ep == DK_EPI_GATE_RES ? p.gate + (size_t)(t.m0 / p.gate_seg_len) * p.gate_stride : nullptr;
  auto unpack8 = [](const uint4 v, float* f) {
    unpack2bf(v.x, f[0], f[1]);
    unpack2bf(v.y, f[2], f[3]);
    unpack2bf(v.z, f[4], f[5]);
    unpack2bf(v.w, f[6], f[7]);
  };
  // KF: sum of squares of every row over its head's columns, from the staged (bf16-rounded) values: this lane's 8 rows, per head
  // (kn_D = 128: the wave's 128 columns are one head; 64: two heads = the two virtual waves)
  float ss[2][MF];
  if (KF) {
#pragma unroll
    for (int vwn = 0; vwn < 2; ++vwn) {
#pragma unroll
      for (int itr = 0; itr < MF; ++itr) ss[vwn][itr] = 0.f;
      for (int ni = 0; ni < 2; ++ni) {
        const unsigned reg0 = (unsigned)((t.wm * 4 + 2 * t.wn2 + vwn) * 16384 + ni * 8192);
#pragma unroll
        for (int itr = 0; itr < MF; ++itr) {
          const int row = itr * 16 + rrow;
          const u32x4 sv = *(const __attribute__((address_space(3))) u32x4*)((v4_lds_char*)0 + reg0 + row * 64 + ((((unsigned)rc2 >> 1) ^ ((unsigned)(row >> 2) & 3u)) << 4));
          float vv[8];
          unpack8(make_uint4(sv[0], sv[1], sv[2], sv[3]), vv);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss[vwn][itr] += vv[e] * vv[e];
        }
      }
#pragma unroll
      for (int itr = 0; itr < MF; ++itr) {
        ss[vwn][itr] += __shfl_xor(ss[vwn][itr], 1, 64);
        ss[vwn][itr] += __shfl_xor(ss[vwn][itr], 2, 64);
      }
    }
    if (p.kn_D == 128) {
#pragma unroll
      for (int itr = 0; itr < MF; ++itr) ss[0][itr] = ss[1][itr] = ss[0][itr] + ss[1][itr];
    }
  }
  // FAST tiles: everything the rows read from memory is fetched before the first store -- for all four passes at once (the residual is
  // updated in place, C == res: behind a store the compiler may not move a load up, and a lone wave has nothing else to cover the
  // round trip; gemm256v3.hip round 4 did this per pass)
  constexpr bool PRE_RES = FAST && (EK == DK_EPI_GATE_RES || EK == DK_EPI_RES);
  constexpr bool PRE_ROPE = FAST && KF;
  uint4 res_pre[PRE_RES ? 4 : 1][PRE_RES ? MF : 1];
  f32x4 rope_pre[PRE_ROPE ? 4 : 1][PRE_ROPE ? 2 * MF : 1];
  float gate_pre[(FAST && EK == DK_EPI_GATE_RES) ? 4 : 1][8];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int wn = 2 * t.wn2 + (pass >> 1), ni = pass & 1;
    const int col = t.n0 + wn * 64 + ni * 32 + rc2 * 4;
    if (PRE_RES) {
#pragma unroll
      for (int itr = 0; itr < MF; ++itr) res_pre[pass][itr] = *(const uint4*)(p.res + (physR0 + (CUT ? min(itr * 16 + rrow, row_lim) : itr * 16 + rrow)) * (size_t)p.ldr + col);
    }
    if (FAST && EK == DK_EPI_GATE_RES) unpack8(*(const uint4*)(gate_row + col), gate_pre[pass]);
    if (PRE_ROPE) {
      if (p.kn_rope != nullptr) {
        const int kcol = col % p.kn_D;
#pragma unroll
        for (int itr = 0; itr < MF; ++itr) {
          const int kpos_ = (mrow0 + itr * 16 + rrow) % p.kn_seg_len;
          const float* tab = p.kn_rope + ((size_t)(p.kn_pos_off + kpos_) * (size_t)(p.kn_D / 2) + (size_t)(kcol >> 1)) * 2;
          rope_pre[pass][2 * itr] = *(const f32x4*)tab, rope_pre[pass][2 * itr + 1] = *(const f32x4*)(tab + 4);
        }
      }
    }
  }
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int vwn = pass >> 1, ni = pass & 1;
    const int wn = 2 * t.wn2 + vwn;
    const unsigned reg0 = (unsigned)((t.wm * 4 + wn) * 16384 + ni * 8192);
    const int col = t.n0 + wn * 64 + ni * 32 + rc2 * 4;      // first of this lane's 8 columns of the GEMM (gate, residual)
    const int ocol = t.ncol0 + wn * 64 + ni * 32 + rc2 * 4;  // the same inside the output it goes to
    float gate8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float kw8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int kcol = KF ? col % p.kn_D : 0;
    if (KF) unpack8(*(const uint4*)(t.nw + kcol), kw8);
    if (FAST && EK == DK_EPI_GATE_RES) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gate8[e] = gate_pre[pass][e];
    } else if (FAST && ep == DK_EPI_GATE_RES) {
      unpack8(*(const uint4*)(gate_row + col), gate8);
    }
    int c_seg = 0, c_rem = 0, r_seg = 0, r_rem = 0, g_seg = 0, g_rem = 0;
    if (!FAST) {
      const int ms = mrow0 + rrow;
      c_seg = ms / p.c_seg_len, c_rem = ms % p.c_seg_len;
      if (hres) r_seg = ms / p.r_seg_len, r_rem = ms % p.r_seg_len;
      if (ep == DK_EPI_GATE_RES) g_seg = ms / p.gate_seg_len, g_rem = ms % p.gate_seg_len;
    }
#pragma unroll
    for (int itr = 0; itr < MF; ++itr) {
      const int row = itr * 16 + rrow;  // row inside the wave's block of 128 rows
      size_t crow = physC0 + row, rrow_phys = physR0 + row;
      bool valid = !CUT || row <= row_lim;
      const int kpos = KF ? (mrow0 + row) % p.kn_seg_len : 0;
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
      const u32x4 sv = *(const __attribute__((address_space(3))) u32x4*)((v4_lds_char*)0 + reg0 + row * 64 + ((((unsigned)rc2 >> 1) ^ ((unsigned)(row >> 2) & 3u)) << 4));
      if (PLAIN) {
        if ((FAST && !CUT) || valid) v4_store16(t.Cb + crow * (size_t)t.ldcb + ocol, sv);
        continue;
      }
      float vv[8];
      unpack8(make_uint4(sv[0], sv[1], sv[2], sv[3]), vv);
      if (KF) {
        const float r = rsqrtf(ss[vwn][itr] / (float)p.kn_D + p.kn_eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = round_bf16(vv[e] * r * kw8[e]);
        if (p.kn_rope != nullptr) {
          const float* tab = p.kn_rope + ((size_t)(p.kn_pos_off + kpos) * (size_t)(p.kn_D / 2) + (size_t)(kcol >> 1)) * 2;
          f32x4 t0 = {1.f, 0.f, 1.f, 0.f}, t1 = {1.f, 0.f, 1.f, 0.f};
          if (PRE_ROPE) t0 = rope_pre[pass][2 * itr], t1 = rope_pre[pass][2 * itr + 1];
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
        if (PRE_RES) rr = res_pre[pass][itr];
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
      if ((FAST && !CUT) || valid) v4_store16(t.Cb + crow * (size_t)t.ldcb + ocol, u32x4{o4.x, o4.y, o4.z, o4.w});
    }
  }
}

// MF: 16-row activation fragments per wave: 8 = 256-row tiles, 7 = 224-row tiles (the launcher takes the height with the fewest rounds x height,
// gemm256v3.hip's rule: 4352 rows x 12 column tiles are 204 tiles of 256 rows -- 80 % of the CUs -- or 240 of 224)
template <int MF>
__global__ __launch_bounds__(256, 1) void dk_gemm256v4_kernel(GemmParams pa, GemmParams pb, int tiles_a, int tiles_b, int skew) {
  constexpr int BM = 32 * MF, HROWS = 16 * MF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(v4_lds_char*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  // Start skew (multi-round launches): the first round's workgroups start up to `skew` x 0.25 us apart (by their position inside their
  // XCD), and every later round inherits the spread -- without it all 256 CUs reach their tails at the same moment and 256 x 128 KiB of
  // C leave in one burst per round while the matrix pipes idle (profiles/r03_gemm_k_sweep_ablations.log: 2.4 us of a 9.8 us fixed cost)
  if (skew > 0 && blockIdx.x < 256) {
    const int n = (int)((blockIdx.x >> 3) & 31) * skew / 32;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn2 = wave & 1;
  const int l15 = lane & 15, q = lane >> 4;

  const int tile = v4_xcd_contiguous(blockIdx.x, tiles_a + tiles_b);
  const bool second = tile >= tiles_a;
  // one scalar base into the kernel-argument segment for this tile's parameter block (gemm256v3.hip, round 4)
  static_assert(sizeof(GemmParams) % 8 == 0 && alignof(GemmParams) == 8, "pb follows pa without padding");
  typedef const __attribute__((address_space(4))) GemmParams karg_params_t;
  const __attribute__((address_space(4))) char* kbase = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  karg_params_t& p = *(karg_params_t*)(kbase + (second ? sizeof(GemmParams) : 0));
  const int tl = second ? tile - tiles_a : tile;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / V4_T;
  const int GROUP = 4;
  const int tpg = GROUP * nbn;
  const int g = tl / tpg;
  const int first_m = g * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int tm = first_m + (tl % tpg) % gsz;
  const int tn = (tl % tpg) / gsz;
  const int m0 = tm * BM, n0 = tn * V4_T;

  // ---- DMA piece offsets: piece gg of a wave covers rows hh*128 + (wave*2 + u)*16 + j*8 + (lane >> 3) of the operand's 256-row K-tile,
  // 16-byte chunk (lane & 7) of LDS row r holds global chunk (lane & 7) ^ ((r >> 1) & 7) (conflict-free ds_read_b128, gemm256v3.hip)
  const int srow = lane >> 3;
  u32x8 voX, voW;
  // (a tile that lies inside one row segment -- every tile of the image stream -- maps its rows with one scalar division instead of eight
  //  per lane: the integer divisions were ~ 0.4 us in front of the first DMA piece of every tile)
  const int a_seg0 = m0 / p.a_seg_len;
  const bool a_uniform = min(m0 + BM - 1, p.M - 1) / p.a_seg_len == a_seg0;
  const int a_base = a_seg0 * p.a_seg_stride - a_seg0 * p.a_seg_len;
#pragma unroll
  for (int gg = 0; gg < 8; ++gg) {
    const int hh = gg & 1, j = (gg >> 1) & 1, u = gg >> 2;
    const int rih = (wave * 2 + u) * 16 + j * 8 + srow;  // row inside the 128-row LDS half
    const int row = hh * 128 + rih;
    const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j);
    // rows beyond M - 1 re-read the last row, their results are never stored; MF = 7: LDS rows 112 .. 127 of a half hold duplicates nobody reads
    voX[gg] = (unsigned)min(m0 + hh * HROWS + min(rih, HROWS - 1), p.M - 1);
    voW[gg] = ((unsigned)row * (unsigned)p.ldw + chunk * 8) * 2u;
  }
  if (a_uniform) {  // (a scalar branch: the divisions below are not executed)
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) voX[gg] += (unsigned)a_base;
  } else {
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) voX[gg] = (unsigned)(((int)voX[gg] / p.a_seg_len) * p.a_seg_stride + ((int)voX[gg] % p.a_seg_len));
  }
#pragma unroll
  for (int gg = 0; gg < 8; ++gg) {
    const int j = (gg >> 1) & 1;
    const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j);
    voX[gg] = (voX[gg] * (unsigned)p.lda + chunk * 8) * 2u;
  }
  // fragment read addresses (X kk0, X kk1, W kk0, W kk1) and the two drain addresses
  u32x4 rd;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const unsigned offk = (unsigned)(l15 * 128 + (((kk * 4 + q) ^ (l15 >> 1)) << 4));
    rd[kk] = wm * 16384 + offk;
    rd[2 + kk] = 65536 + wn2 * 16384 + offk;
  }
  u32x2 dr;
  dr[0] = (unsigned)((wm * 4 + 2 * wn2) * 16384 + l15 * 64 + (q & 1) * 8 + (((q >> 1) ^ ((l15 >> 2) & 3)) << 4));
  dr[1] = dr[0] ^ 32u;
  // bias of this lane's 32 columns (4 per 16-column fragment)
  f32x8 bq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) bq[i][e] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int nf8 = 0; nf8 < 8; ++nf8) {
      const u32x2 b2 = *(const u32x2*)(p.bias + n0 + wn2 * 128 + nf8 * 16 + 4 * q);
      float b0, b1, b2f, b3;
      unpack2bf(b2[0], b0, b1);
      unpack2bf(b2[1], b2f, b3);
      bq[nf8 >> 1][(nf8 & 1) * 4 + 0] = b0;
      bq[nf8 >> 1][(nf8 & 1) * 4 + 1] = b1;
      bq[nf8 >> 1][(nf8 & 1) * 4 + 2] = b2f;
      bq[nf8 >> 1][(nf8 & 1) * 4 + 3] = b3;
    }
  }
  const char* gA = (const char*)p.A;
  const char* gW = (const char*)p.W + (size_t)n0 * (size_t)p.ldw * 2;
  const u32x4 rX = {(unsigned)(size_t)gA, (unsigned)((size_t)gA >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  const u32x4 rW = {(unsigned)(size_t)gW, (unsigned)((size_t)gW >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  const int nk = p.K / V4_BK;
  const float alpha = p.alpha;

#ifdef V4_TRACE  // lab (scripts/build_lab.sh TRACE=1): time stamps of workgroups 0 and gridDim.x - 1, wave 0, into p.workspace
  unsigned long long t_entry = __builtin_readcyclecounter(), ts0, ts1, ts2, ts3;
  asm volatile(
#include "../../profiles/lab_kernels/gemm256v4_variants/gemm256v4_asm_trace.inc"
      : "={s[74:75]}"(ts0), "={s[76:77]}"(ts1), "={s[78:79]}"(ts2), "={s[80:81]}"(ts3)
      : [koff] "s"(0), [nk] "s"(nk), [dstx] "s"(wave * 4096), [alpha] "s"(alpha), [wave] "s"(wave), "{v[0:7]}"(voX), "{v[8:15]}"(voW), "{v[16:19]}"(rd),
        "{v[24:25]}"(dr), "{v[160:167]}"(bq[0]), "{v[168:175]}"(bq[1]), "{v[176:183]}"(bq[2]), "{v[184:191]}"(bq[3]), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
      :
#include "gemm256v4_clobbers.inc"
  );
#else
  if constexpr (MF == 7) {
    asm volatile(
#include "gemm256v4_asm7.inc"
        :
        : [koff] "s"(0), [nk] "s"(nk), [dstx] "s"(wave * 4096), [alpha] "s"(alpha), [wave] "s"(wave), "{v[0:7]}"(voX), "{v[8:15]}"(voW), "{v[16:19]}"(rd),
          "{v[24:25]}"(dr), "{v[160:167]}"(bq[0]), "{v[168:175]}"(bq[1]), "{v[176:183]}"(bq[2]), "{v[184:191]}"(bq[3]), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
        :
#include "gemm256v4_clobbers.inc"
    );
  } else
  asm volatile(
#include "gemm256v4_asm.inc"
      :
      : [koff] "s"(0), [nk] "s"(nk), [dstx] "s"(wave * 4096), [alpha] "s"(alpha), [wave] "s"(wave), "{v[0:7]}"(voX), "{v[8:15]}"(voW), "{v[16:19]}"(rd),
        "{v[24:25]}"(dr), "{v[160:167]}"(bq[0]), "{v[168:175]}"(bq[1]), "{v[176:183]}"(bq[2]), "{v[184:191]}"(bq[3]), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
      :
#include "gemm256v4_clobbers.inc"
  );
#endif

  // ---------------- tail: staged bf16 image -> row-major, epilogues on the way ----------------
  const bool out2 = p.n_split > 0 && n0 >= p.n_split;  // tile-uniform: second output of a column-split GEMM
  V4Tail t;
  t.p = &p;
  t.Cb = out2 ? p.C2 : p.C;
  t.ldcb = out2 ? p.ldc2 : p.ldc;
  t.epi = out2 ? p.epi2 : p.epi;
  t.n0 = n0;
  t.ncol0 = out2 ? n0 - p.n_split : n0;
  t.m0 = m0, t.wm = wm, t.wn2 = wn2, t.lane = lane;
  t.has_res = t.epi == DK_EPI_GATE_RES || t.epi == DK_EPI_RES;
  auto inside = [&](int len) { return m0 / len == (m0 + BM - 1) / len; };
  const bool fast = m0 + BM <= p.M && inside(p.c_seg_len) && (!t.has_res || inside(p.r_seg_len)) && (t.epi != DK_EPI_GATE_RES || inside(p.gate_seg_len));
  const bool qtile = p.qn_w != nullptr && n0 >= p.qn_col0 && n0 < p.qn_col1;
  const bool kfuse = p.kn_w != nullptr && ((n0 >= p.kn_col0 && n0 < p.kn_col1) || qtile);
  t.nw = qtile ? p.qn_w : p.kn_w;
  // the last row tile of a single-segment problem: tile-uniform maps, M ends inside it
  auto inside_cut = [&](int len) { return m0 / len == (p.M - 1) / len; };
  const bool cut = !fast && m0 + BM > p.M && inside_cut(p.c_seg_len) && (!t.has_res || inside_cut(p.r_seg_len)) &&
                   (t.epi != DK_EPI_GATE_RES || inside_cut(p.gate_seg_len));
  if (kfuse) {  // (bias-only epilogue -- checked by the launcher)
    if (fast) v4_rows<true, DK_EPI_BIAS, true, MF>(t);
    else if (cut) v4_rows<true, DK_EPI_BIAS, true, MF, true>(t);
    else v4_rows<false, DK_EPI_BIAS, true, MF>(t);
  } else if (cut) {
    if (t.epi == DK_EPI_BIAS) v4_rows<true, DK_EPI_BIAS, false, MF, true>(t);
    else if (t.epi == DK_EPI_BIAS_GELU) v4_rows<true, DK_EPI_BIAS_GELU, false, MF, true>(t);
    else if (t.epi == DK_EPI_GATE_RES) v4_rows<true, DK_EPI_GATE_RES, false, MF, true>(t);
    else v4_rows<false, -1, false, MF>(t);
  } else if (fast) {
    if (t.epi == DK_EPI_BIAS) v4_rows<true, DK_EPI_BIAS, false, MF>(t);
    else if (t.epi == DK_EPI_BIAS_GELU) v4_rows<true, DK_EPI_BIAS_GELU, false, MF>(t);
    else if (t.epi == DK_EPI_GATE_RES) v4_rows<true, DK_EPI_GATE_RES, false, MF>(t);
    else if (t.epi == DK_EPI_RES) v4_rows<true, DK_EPI_RES, false, MF>(t);
    else v4_rows<true, -1, false, MF>(t);
  } else {
    v4_rows<false, -1, false, MF>(t);
  }
#ifdef V4_TRACE
  if (p.workspace != nullptr && wave == 0 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tail's stores have left)
    unsigned long long* o = (unsigned long long*)p.workspace + (blockIdx.x == 0 ? 0 : 8);
    o[0] = t_entry; o[1] = ts0; o[2] = ts1; o[3] = ts2; o[4] = ts3; o[5] = __builtin_readcyclecounter();
  }
#endif
}

// Tile height of a launch: gemm256v3.hip's rule (rounds of the CUs x rows per tile over both problems; 224-row tiles only for long reductions and
// only when the model predicts at least 10 %; dk_tune_set("gemm_mf", 7 | 8) forces one)
// every row tile of height bm lies inside one segment of every row map (tile-uniform tail paths: FAST, or CUT for the last one)
bool dk_gemm256v4_uniform_tiles(const GemmParams& p, int bm) {
  auto ok = [&](int len) { return len >= p.M || len % bm == 0; };
  const bool res = p.epi == DK_EPI_GATE_RES || p.epi == DK_EPI_RES || (p.n_split > 0 && (p.epi2 == DK_EPI_GATE_RES || p.epi2 == DK_EPI_RES));
  const bool gate = p.epi == DK_EPI_GATE_RES || (p.n_split > 0 && p.epi2 == DK_EPI_GATE_RES);
  return ok(p.a_seg_len) && ok(p.c_seg_len) && (!res || ok(p.r_seg_len)) && (!gate || ok(p.gate_seg_len)) && (p.kn_w == nullptr || ok(p.kn_seg_len));
}

int dk_gemm256v4_pick_mf(const GemmParams& p, const GemmParams* p2, int n_cu) {
  if (g_dk_v3_mf == 7 || g_dk_v3_mf == 8) return g_dk_v3_mf;
  if (p.K < 2048) return 8;
  if (!dk_gemm256v4_uniform_tiles(p, 224) || (p2 && !dk_gemm256v4_uniform_tiles(*p2, 224))) return 8;  // (the per-row tail path is slow here)
  long cost[2];
  for (int mf = 7; mf <= 8; ++mf) {
    const int bm = 32 * mf;
    long tiles = (long)((p.M + bm - 1) / bm) * (p.N / 256);
    if (p2) tiles += (long)((p2->M + bm - 1) / bm) * (p2->N / 256);
    cost[mf - 7] = ((tiles + n_cu - 1) / n_cu) * bm;
  }
  return cost[0] * 10 <= cost[1] * 9 ? 7 : 8;
}

int g_dk_v4_skew = -1;  // dk_tune_set("gemm_skew", v): start skew of multi-round launches in 0.25 us steps; -1 (default): none

// dk_tune_set("gemm", 10) forces this kernel on every shape it accepts; -1 (automatic): see dk_launch_gemm / dk_launch_gemm_pair
bool dk_gemm256v4_eligible(const GemmParams& p) {
  if (p.conv || !dk_gemm256v3_eligible(p)) return false;
  if (p.N % 256 != 0) return false;
  // 32-bit byte offsets of the W pieces: 255 rows
  return (size_t)p.ldw * 2 * 256 < (1ull << 31);
}

int dk_launch_gemm256v4(const GemmParams& p, const GemmParams* p2, hipStream_t stream) {
  DK_REQUIRE(dk_gemm256v4_eligible(p), "gemm256v4: shape / strides not eligible");
  if (p2) {
    DK_REQUIRE(dk_gemm256v4_eligible(*p2), "gemm256v4: second problem not eligible");
    DK_REQUIRE(p2->N == p.N && p2->K == p.K && p2->epi == p.epi && p2->alpha == p.alpha && p2->n_split == p.n_split &&
                   (p.n_split == 0 || p2->epi2 == p.epi2),
               "grouped GEMM: N, K, epilogue must match");
  }
  const int n_cu = dk_device_cu_count();
  const int mf = dk_gemm256v4_pick_mf(p, p2, n_cu);
  const int bm = 32 * mf;
  const int tiles_a = ((p.M + bm - 1) / bm) * (p.N / 256);
  const int tiles_b = p2 ? ((p2->M + bm - 1) / bm) * (p2->N / 256) : 0;
  if (g_dk_gemm_plan != nullptr) {
    DkGemmPlan& pl = *g_dk_gemm_plan;
    pl.kernel = 4; pl.tile_rows = bm; pl.tiles = pl.workgroups = tiles_a + tiles_b; pl.split_tiles = 0; pl.k_pieces = 1; pl.ks = p.K / V4_BK;
    pl.n_cu = n_cu; pl.launches += 1;
    return 0;
  }
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v4_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm256v4_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS_BYTES));
    attr_once.mark();
  }
  double work = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  if (p2) work += 2.0 * (double)p2->M * (double)p2->N * (double)p2->K;
  dk_prof_begin(0, work, stream);
  // ... in the lab with warm weights.  Inside the model (weights from HBM) the start skew is flat: 58.78 / 58.81 against 58.86 / 58.94 ms per FLUX
  // step (profiles/r05_gemm_v4_start_skew.log) -- the automatic choice keeps it off; dk_tune_set("gemm_skew", n) turns it on
  const int skew = tiles_a + tiles_b <= n_cu || g_dk_v4_skew < 0 ? 0 : g_dk_v4_skew;
  if (mf == 7)
    hipLaunchKernelGGL(dk_gemm256v4_kernel<7>, dim3(tiles_a + tiles_b), dim3(256), V4_LDS_BYTES, stream, p, p2 ? *p2 : p, tiles_a, tiles_b, skew);
  else
    hipLaunchKernelGGL(dk_gemm256v4_kernel<8>, dim3(tiles_a + tiles_b), dim3(256), V4_LDS_BYTES, stream, p, p2 ? *p2 : p, tiles_a, tiles_b, skew);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
