// 3x3 convolution of the VAE decoder's 256- and 512-channel stages in the one-wave-per-SIMD frame of gemm256v4.hip: a workgroup of 256
// threads owns 16 x 16 pixels x 256 output channels, a wave 128 pixels x 128 channels with its 256 accumulators in AGPRs, and the whole
// main body is ONE hand-scheduled inline-asm block (conv256v4_asm_x.inc / _asm_p.inc, written and CPU-checked by scripts/gen_conv256v4.py).
// reference: python/src/diffusionkit/mlx/vae.py:60-101 (ResnetBlock2D: norm -> silu -> conv), :160-176 (nearest x2 upsample -> conv).
//
// Why (profiles/r05_vae_by_stage.md, r03_conv_halo_ablations.md): in conv_halo.hip's 8-wave frame a wave owns 64 x 64 of a
// 256 x 128 tile: per MFMA it reads as many LDS bytes as the matrix pipe can absorb, the two workgroups of a CU run the same phase of the
// same K-tile, and "the MFMA time and everything else add up" (1000 TFLOP/s).  Here a wave's block is 128 x 128 -- half the LDS reads per
// MAC -- a halo and a weight K-tile serve 256 output channels instead of 128, the weights come through an LDS-DMA ring (no register round
// trip), and every non-MFMA instruction -- fragment reads, DMA pieces, the halo loads of the NEXT 64-channel chunk, their GroupNorm-apply +
// SiLU transform (conv_halo.hip's arithmetic, instruction for instruction) and their LDS stores -- sits in an MFMA gap of the schedule.
//
// Data flow per workgroup: per 64-channel chunk the 18 x 18 pixel halo is loaded ONCE (11 x 16-byte items per thread), transformed in
// registers and stored to one of two LDS halo slots (324 rows x 144 B: conflict-free ds_read_b128 for 16 consecutive rows); the nine taps
// of the chunk read their shifted windows from it (one base register, an immediate offset per tap and pixel row).  Weights: [O, ldw],
// column tap * C + c: K-tile (chunk, tap) is 64 contiguous columns of 256 rows -> 8 LDS-DMA pieces per wave into a two-slot ring.
// The 1x1 shortcut extension, the image tail and 128-column tiles stay with conv_halo.hip.
//
// Tail: the asm block leaves round_bf16(acc + bias) in gemm256v4's staging image; the read-back adds the residual, stores 16 bytes per lane,
// and sums the GroupNorm statistics of the stored values per channel group in a fixed order (no atomics).
#include "dk_kernels.h"

typedef __attribute__((address_space(3))) char c4_lds_char;
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));

#define C4_LDS_BYTES 163840
#define C4_ROWB 144
#define C4_HSLOT (324 * C4_ROWB)
#define C4_BIAS_LDS (2 * C4_HSLOT)
#define C4_W_BASE 98304
#define C4_RED_LDS 131072  // statistics exchange of the tail (the second weight slot: dead behind the K loop)

__device__ __forceinline__ int c4_xcd_contiguous(int bid, int count) {
  const int x = bid & 7;
  int start = 0;
  for (int y = 0; y < x; ++y) start += y < count ? (count - y + 7) >> 3 : 0;
  return start + (bid >> 3);
}

// NT = 256: waves 2 x 2 of 128 pixels x 128 channels (conv256v4_asm_x / _p.inc); NT = 128 (the 128-channel stages): waves 4 x 1 of 64 pixels x 128
// channels, 64 MFMAs per K-tile, weight K-tiles of 128 rows in a three-slot ring (conv256v4_asm_x128 / _p128.inc)
template <bool XFORM, int NT>
__global__ __launch_bounds__(256, 1) void dk_conv256v4_kernel(ConvHaloParams p) {
  constexpr bool N128 = NT == 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(c4_lds_char*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  c4_lds_char* const lds = (c4_lds_char*)0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = N128 ? wave : wave >> 1, wn2 = N128 ? 0 : wave & 1;  // wave's pixel-row block (N128: 4 rows, else 8), column half
  constexpr int PR = N128 ? 4 : 8;                                    // pixel rows (16-pixel fragments) per wave
  const int l15 = lane & 15, q = lane >> 4;

  // ---- this workgroup's tile: workgroups that follow each other on an XCD share the pixel tile (its halo stays in that XCD's L2)
  const int tiles_n = p.O / NT, tiles_x = p.W >> 4, tiles_y = p.H >> 4;
  const int pos = c4_xcd_contiguous(blockIdx.x, gridDim.x);
  const int nt = pos % tiles_n, pt = pos / tiles_n;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, b = pt / (tiles_x * tiles_y);
  const int n0 = nt * NT;
  const int Hs = p.H >> p.ups, Ws = p.W >> p.ups;
  const int px0 = tx * 16, py0 = ty * 16;

  // ---- halo items of this thread: item i = halo row (tid >> 3) + 32 i, 16-byte channel chunk tid & 7
  const int c8 = tid & 7;
  u32x16 va;
  unsigned okmask = 0u;
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    const int id = tid + 256 * i;
    const int hrow = id >> 3;
    const int hy = hrow / 18, hx = hrow - hy * 18;
    const int y = py0 - 1 + hy, x = px0 - 1 + hx;
    const bool ok = id < 324 * 8 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    va[i] = ok ? ((unsigned)((y >> p.ups) * Ws + (x >> p.ups)) * (unsigned)p.C + c8 * 8u) * 2u : 0x80000000u;  // padding: out of range -> zeros
    okmask |= (ok ? 1u : 0u) << i;
  }
  va[11] = okmask;
  // ---- weight pieces: piece g of a wave covers rows hh*128 + (wave*2 + u)*16 + j*8 + (lane >> 3) of the 256-row K-tile (gemm256v4.hip)
  const int srow = lane >> 3;
  u32x4 vb = {0u, 0u, 0u, 0u};
  if constexpr (N128) {  // piece g of a wave: rows (wave*4 + g)*8 + (lane >> 3) of the 128-row K-tile
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = (wave * 4 + g) * 8 + srow;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      va[12 + g] = ((unsigned)row * (unsigned)p.ldw + chunk * 8) * 2u;
    }
  } else {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int hh = g & 1, j = (g >> 1) & 1, u = g >> 2;
      const int row = hh * 128 + (wave * 2 + u) * 16 + j * 8 + srow;
      const int chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j);
      const unsigned o = ((unsigned)row * (unsigned)p.ldw + chunk * 8) * 2u;
      if (g < 4) va[12 + g] = o;
      else vb[g - 4] = o;
    }
  }
  // ---- read / write bases
  u32x4 rd;  // halo window base, weight fragment addresses kk0 / kk1, halo write base
  rd[0] = (unsigned)((wm * PR * 18 + l15) * C4_ROWB + q * 16);
  rd[1] = (unsigned)(C4_W_BASE + wn2 * 16384 + l15 * 128 + (((0 * 4 + q) ^ (l15 >> 1)) << 4));
  rd[2] = (unsigned)(C4_W_BASE + wn2 * 16384 + l15 * 128 + (((1 * 4 + q) ^ (l15 >> 1)) << 4));
  rd[3] = (unsigned)((tid >> 3) * C4_ROWB + c8 * 16);
  u32x4 dr;  // drain addresses (gemm256v4.hip), scale / shift table offsets
  dr[0] = (unsigned)((N128 ? 2 * wave : wm * 4 + 2 * wn2) * 16384 + l15 * 64 + (q & 1) * 8 + (((q >> 1) ^ ((l15 >> 2) & 3)) << 4));
  dr[1] = dr[0] ^ 32u;
  dr[2] = (unsigned)(c8 * 32);
  dr[3] = (unsigned)(c8 * 32 + p.C * 4);
  const unsigned ba = (unsigned)(C4_BIAS_LDS + (wn2 * 128 + 4 * q) * 4);
  // ---- bias of the tile's 256 channels as fp32 into the LDS table the drain reads (conv_halo.hip: b4 = bias + bias2 in fp32)
  if (tid < NT) {
    float bv = bf2f(p.bias[n0 + tid]);
    if (p.bias2) bv += bf2f(p.bias2[n0 + tid]);
    *(__attribute__((address_space(3))) float*)(lds + C4_BIAS_LDS + tid * 4) = bv;
  }
  const char* gX = (const char*)(p.x + (size_t)b * Hs * Ws * p.C);
  const char* gW = (const char*)(p.w + (size_t)n0 * (size_t)p.ldw);
  const char* gG = p.gn_ss ? (const char*)(p.gn_ss + (size_t)b * 2 * p.C) : (const char*)p.w;
  const u32x4 rX = {(unsigned)(size_t)gX, (unsigned)((size_t)gX >> 32) & 0xffffu, (unsigned)(Hs * Ws * p.C * 2), 0x00020000u};
  const u32x4 rW = {(unsigned)(size_t)gW, (unsigned)((size_t)gW >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  const u32x4 rG = {(unsigned)(size_t)gG, (unsigned)((size_t)gG >> 32) & 0xffffu, (unsigned)(2 * p.C * 4), 0x00020000u};
  const int nloop = (p.C >> 6) - 1;
  const int m10 = __builtin_amdgcn_readfirstlane(wave == 0 ? -1 : 0);  // lanes 0-31 of wave 0 hold an eleventh halo item

  if constexpr (XFORM && N128) {
    asm volatile(
#include "conv256v4_asm_x128.inc"
        :
        : [nloop] "s"(nloop), [dstw] "s"(C4_W_BASE + wave * 4096), [tapb] "s"(2 * p.C), [wrapb] "s"(128 - 16 * p.C), [m10] "s"(m10),
          "{v[0:15]}"(va), "{v[16:19]}"(vb), "{v[240:243]}"(rd), "{v[244:247]}"(dr), "{v248}"(ba), "{s[56:59]}"(rG), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
        :
#include "conv256v4_clobbers.inc"
    );
  } else if constexpr (N128) {
    asm volatile(
#include "conv256v4_asm_p128.inc"
        :
        : [nloop] "s"(nloop), [dstw] "s"(C4_W_BASE + wave * 4096), [tapb] "s"(2 * p.C), [wrapb] "s"(128 - 16 * p.C), [m10] "s"(m10),
          "{v[0:15]}"(va), "{v[16:19]}"(vb), "{v[240:243]}"(rd), "{v[244:247]}"(dr), "{v248}"(ba), "{s[56:59]}"(rG), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
        :
#include "conv256v4_clobbers.inc"
    );
  } else if constexpr (XFORM) {
    asm volatile(
#include "conv256v4_asm_x.inc"
        :
        : [nloop] "s"(nloop), [dstw] "s"(C4_W_BASE + wave * 4096), [tapb] "s"(2 * p.C), [wrapb] "s"(128 - 16 * p.C), [m10] "s"(m10),
          "{v[0:15]}"(va), "{v[16:19]}"(vb), "{v[240:243]}"(rd), "{v[244:247]}"(dr), "{v248}"(ba), "{s[56:59]}"(rG), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
        :
#include "conv256v4_clobbers.inc"
    );
  } else {
    asm volatile(
#include "conv256v4_asm_p.inc"
        :
        : [nloop] "s"(nloop), [dstw] "s"(C4_W_BASE + wave * 4096), [tapb] "s"(2 * p.C), [wrapb] "s"(128 - 16 * p.C), [m10] "s"(m10),
          "{v[0:15]}"(va), "{v[16:19]}"(vb), "{v[240:243]}"(rd), "{v[244:247]}"(dr), "{v248}"(ba), "{s[56:59]}"(rG), "{s[60:63]}"(rX), "{s[64:67]}"(rW)
        :
#include "conv256v4_clobbers.inc"
    );
  }

  // ---------------- tail: staged bf16 image -> NHWC rows (+ residual), statistics of the stored values ----------------
  // lane: pixel x = lane >> 2 of pixel row itr, 16-byte chunk lane & 3 of the pass's 32 columns (4 passes: 2 column halves x 2)
  // (NT = 128: a wave's 64 pixels x 128 columns sit in the staging regions 2 wave, 2 wave + 1: 64 rows of each)
  const int rrow = lane >> 2, rc = lane & 3;
  u32x4 resv[4][PR];
  if (p.res) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int col = n0 + (2 * wn2 + (pass >> 1)) * 64 + (pass & 1) * 32 + rc * 8;
#pragma unroll
      for (int itr = 0; itr < PR; ++itr) {
        const size_t pix = ((size_t)b * p.H + py0 + wm * PR + itr) * p.W + px0 + rrow;
        resv[pass][itr] = *(const u32x4*)(p.res + pix * (size_t)p.ldr + col);
      }
    }
  }
  float ssum[4][8], ssq[4][8];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int wn = 2 * wn2 + (pass >> 1), ni = pass & 1;
    const unsigned reg0 = (unsigned)((N128 ? 2 * wave + (pass >> 1) : wm * 4 + wn) * 16384 + ni * 8192);
    const int col = n0 + wn * 64 + ni * 32 + rc * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[pass][e] = ssq[pass][e] = 0.f;
#pragma unroll
    for (int itr = 0; itr < PR; ++itr) {
      const int row = itr * 16 + rrow;
      const size_t pix = ((size_t)b * p.H + py0 + wm * PR + itr) * p.W + px0 + rrow;
      u32x4 sv = *(const __attribute__((address_space(3))) u32x4*)(lds + reg0 + row * 64 + (((unsigned)rc ^ ((unsigned)(row >> 2) & 3u)) << 4));
      if (p.res) {
        const u32x4 rv = resv[pass][itr];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0, v1, r0, r1;
          unpack2bf(sv[e], v0, v1);
          unpack2bf(rv[e], r0, r1);
          sv[e] = pack2bf(v0 + r0, v1 + r1);
        }
      }
      *(u32x4*)(p.y + pix * (size_t)p.ldy + col) = sv;
      if (p.stats_out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0, v1;
          unpack2bf(sv[e], v0, v1);
          ssum[pass][2 * e] += v0, ssq[pass][2 * e] += v0 * v0;
          ssum[pass][2 * e + 1] += v1, ssq[pass][2 * e + 1] += v1 * v1;
        }
      }
    }
  }
  if (p.stats_out) {
    // this lane's pixel rows -> over the 16 x positions (lane bits 2..5, fixed tree) -> LDS [wave][128 columns][2] -> per group over its
    // channels and the waves that share its columns (two of a column half; NT = 128: all four), in a fixed order
#pragma unroll
    for (int pass = 0; pass < 4; ++pass)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
          ssum[pass][e] += __shfl_xor(ssum[pass][e], o, 64);
          ssq[pass][e] += __shfl_xor(ssq[pass][e], o, 64);
        }
      }
    if (rrow == 0) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = (pass >> 1) * 64 + (pass & 1) * 32 + rc * 8 + e;  // column inside the wave's 128
          *(__attribute__((address_space(3))) float*)(lds + C4_RED_LDS + ((wave * 128 + c) * 2 + 0) * 4) = ssum[pass][e];
          *(__attribute__((address_space(3))) float*)(lds + C4_RED_LDS + ((wave * 128 + c) * 2 + 1) * 4) = ssq[pass][e];
        }
    }
    __syncthreads();
    const int cpg = p.O / p.G_out;  // channels per group
    const int gpt = NT / cpg;       // groups of this tile
    if (tid < 2 * gpt) {
      const int g = tid >> 1, stat = tid & 1;
      float a = 0.f;
      for (int c = 0; c < cpg; ++c) {
        const int ch = g * cpg + c;  // channel inside the tile: column half ch >> 7 (wave wn2), column ch & 127 of the wave
        if constexpr (N128) {
          for (int m = 0; m < 4; ++m)
            a += *(const __attribute__((address_space(3))) float*)(lds + C4_RED_LDS + ((m * 128 + ch) * 2 + stat) * 4);
        } else {
          for (int m = 0; m < 2; ++m)
            a += *(const __attribute__((address_space(3))) float*)(lds + C4_RED_LDS + ((((m * 2 + (ch >> 7)) * 128) + (ch & 127)) * 2 + stat) * 4);
        }
      }
      const int tiles_img = tiles_x * tiles_y;
      p.stats_out[(((size_t)b * tiles_img + (pt % tiles_img)) * p.G_out + (n0 / cpg + g)) * 2 + stat] = a;
    }
  }
}

int g_dk_conv_v4 = 1;  // dk_tune_set("conv_v4", v): 0 = conv_halo.hip for every fused conv, 1 = this kernel where a launch fills the CUs, 2 = wherever eligible

// tile width of a launch: 256 output channels per workgroup where O allows it, 128 for the 128-channel stages
static int conv4_nt(const ConvHaloParams& p) { return p.O % 256 == 0 ? 256 : 128; }

bool dk_conv256v4_eligible(const ConvHaloParams& p) {
  if (!dk_conv_halo_eligible(p, false)) return false;
  if (p.img || p.u8 || p.raw || p.x2) return false;
  if (p.O % 128 != 0 || p.C < 128) return false;
  if (p.gn_ss && !p.gn_silu) return false;
  if (p.stats_out && (conv4_nt(p) % (p.O / p.G_out) != 0)) return false;
  return (size_t)p.ldw * 2 * 256 < (1ull << 31);
}

bool dk_conv256v4_wanted(const ConvHaloParams& p) {
  if (g_dk_conv_v4 == 0 || !dk_conv256v4_eligible(p)) return false;
  // one workgroup per CU: a launch of 128 tiles leaves half the chip idle -- conv_halo.hip's 128-column tiles fill it.  The rule looks at ONE
  // image: the two kernels sum their GroupNorm partials in different orders, and a batch must decode to what its images decode to alone
  const long tiles = (long)(p.H >> 4) * (p.W >> 4) * (p.O / conv4_nt(p));
  return g_dk_conv_v4 == 2 || tiles >= 256;
}

int dk_launch_conv256v4(const ConvHaloParams& p, hipStream_t stream) {
  DK_REQUIRE(dk_conv256v4_eligible(p), "conv256v4: shape / alignment not supported (O multiple of 128, C multiple of 64 and >= 128, no shortcut extension)");
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_conv256v4_kernel<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_conv256v4_kernel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_conv256v4_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_conv256v4_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS_BYTES));
    attr_once.mark();
  }
  const int nt = conv4_nt(p);
  const long tiles = (long)p.B * (p.H >> 4) * (p.W >> 4) * (p.O / nt);
  const double flops = 2.0 * p.B * p.H * p.W * 9.0 * p.C * p.O;
  dk_prof_begin(1, flops, stream);
  if (nt == 256) {
    if (p.gn_ss)
      hipLaunchKernelGGL((dk_conv256v4_kernel<true, 256>), dim3((unsigned)tiles), dim3(256), C4_LDS_BYTES, stream, p);
    else
      hipLaunchKernelGGL((dk_conv256v4_kernel<false, 256>), dim3((unsigned)tiles), dim3(256), C4_LDS_BYTES, stream, p);
  } else {
    if (p.gn_ss)
      hipLaunchKernelGGL((dk_conv256v4_kernel<true, 128>), dim3((unsigned)tiles), dim3(256), C4_LDS_BYTES, stream, p);
    else
      hipLaunchKernelGGL((dk_conv256v4_kernel<false, 128>), dim3((unsigned)tiles), dim3(256), C4_LDS_BYTES, stream, p);
  }
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
