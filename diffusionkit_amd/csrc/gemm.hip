// bf16 MFMA GEMM for gfx950: C = epi(alpha * A . W^T + bias), fp32 accumulate.
//
// Replaces every nn.Linear / nn.Conv2d call site of the reference hot path
// (python/src/diffusionkit/mlx/mmdit.py:56,285,358-360,373-375,432,771,777,821-832;
//  python/src/diffusionkit/mlx/vae.py:36-39,73,79,84,134,349,384).
//
// Tile: 128(M) x 128(N) x 64(K) per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// v_mfma_f32_32x32x16_bf16).  Both operands are K-major, staged HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a double-buffered 2 x 32 KiB image.
// The LDS image is lane-linear (a DMA constraint); bank conflicts on the ds_read_b128
// fragment reads are removed by XOR-ing the 16-byte chunk index with (row>>1)&7 on the
// *source* address and again on the read (same involution both sides).
// The MFMA is issued with operands swapped (W-fragment as A, X-fragment as B) so each lane
// ends up owning one output row and 4-element runs of consecutive columns: the epilogue
// (bias, GELU, gate*x+residual) then works on 8-byte vectors.
// AMODE=1 turns the A loader into an im2col gather for 3x3 convolution over NHWC: pad 1 / stride 1
// (optionally reading a nearest-x2-upsampled view), or stride 2 over the input padded by one row and
// column at the bottom / right (the VAE encoder's downsample, vae.py:141-143); padding taps read a zero page.
#include "dk_kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define A_TILE_BYTES (BM * BK * 2)
#define B_TILE_BYTES (BN * BK * 2)
#define STAGE_BYTES (A_TILE_BYTES + B_TILE_BYTES)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ int lds_row_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int AMODE>
__global__ __launch_bounds__(256) void dk_gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- workgroup -> tile: XCD-contiguous chunks, then 8-row groups (A/W panels shared in L2) ----
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int t;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int GROUP = 8;
  const int tpg = GROUP * nbn;
  const int g = t / tpg;
  const int first_m = g * GROUP;
  const int gsz = min(nbm - first_m, GROUP);
  const int tm = first_m + (t % tpg) % gsz;
  const int tn = (t % tpg) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane source pointers for the DMA loads (4 A + 4 W row-instructions per wave) ----
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
  int cy[4], cx[4];       // conv: output pixel coordinates of this lane's rows
  long cimg[4];           // conv: image base (elements)
  const int srow = lane >> 3;  // row within an 8-row instruction
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + srow;  // tile-local row
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    int n = min(n0 + r, p.N - 1);
    w_src[i] = p.W + (size_t)n * p.ldw + chunk * 8;
    int m = min(m0 + r, p.M - 1);
    if (AMODE == 0) {
      const int phys = (m / p.a_seg_len) * p.a_seg_stride + (m % p.a_seg_len);
      a_src[i] = p.A + (size_t)phys * p.lda + chunk * 8;
    } else {
      const int hw = p.cH * p.cW;
      const int b = m / hw, rem = m - b * hw;
      cy[i] = rem / p.cW;
      cx[i] = rem - cy[i] * p.cW;
      // stored input size: ups 1 = nearest-x2 view of a half-size tensor, ups 2 = stride-2 conv over a double-size one
      const int Hs = p.ups == 1 ? (p.cH >> 1) : p.ups == 2 ? (p.cH << 1) : p.cH;
      const int Ws = p.ups == 1 ? (p.cW >> 1) : p.ups == 2 ? (p.cW << 1) : p.cW;
      cimg[i] = (long)b * Hs * Ws * p.cC + chunk * 8;
      a_src[i] = p.zeros + chunk * 8;
    }
  }

  const int nk = p.K / BK;
  const int cpt = (AMODE == 1) ? (p.cC / BK) : 1;  // K-tiles per conv tap

  auto issue = [&](int kt, int buf) {
    char* abase = smem + buf * STAGE_BYTES + (wave * 32) * 128;
    char* bbase = abase + A_TILE_BYTES;
    if (AMODE == 1) {
      const int tap = kt / cpt, cb = (kt - tap * cpt) * BK;
      const int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
      const int Ws = p.ups == 1 ? (p.cW >> 1) : p.ups == 2 ? (p.cW << 1) : p.cW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iy = cy[i] + ky, ix = cx[i] + kx;
        bool ok = (iy >= 0) && (iy < p.cH) && (ix >= 0) && (ix < p.cW);
        int sy = p.ups == 1 ? (iy >> 1) : iy, sx = p.ups == 1 ? (ix >> 1) : ix;
        if (p.ups == 2) {  // stride 2 over the input padded by one row / column at the bottom / right (vae.py:141-143)
          sy = 2 * cy[i] + ky + 1, sx = 2 * cx[i] + kx + 1;
          ok = sy < 2 * p.cH && sx < 2 * p.cW;
        }
        const bf16_t* src = ok ? (p.A + cimg[i] + ((long)sy * Ws + sx) * p.cC + cb) : a_src[i];
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(abase + i * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + kt * BK), (lds_ptr_t)(abase + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w_src[i] + kt * BK), (lds_ptr_t)(bbase + i * 1024), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const char* As = smem + buf * STAGE_BYTES;
    const char* Bs = As + A_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = kk * 2 + hi;
      bf16x8 wf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wf[i] = *(const bf16x8*)(Bs + lds_row_off(wn * 64 + i * 32 + l31, c));
        xf[i] = *(const bf16x8*)(As + lds_row_off(wm * 64 + i * 32 + l31, c));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane owns row m (per mi) and columns nb + {0..3} per (ni, g) ----
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wm * 64 + mi * 32 + l31;
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
        if (nb >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * g4 + e] * p.alpha;
        const bool full = (nb + 3 < p.N);
        if (full) {
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
        } else {
          for (int e = 0; e < 4 && nb + e < p.N; ++e) {
            float x = v[e];
            if (p.bias) x += bf2f(p.bias[nb + e]);
            x = round_bf16(x);
            if (p.epi == DK_EPI_BIAS_GELU) x = gelu_erf_f(x);
            else if (p.epi == DK_EPI_BIAS_SILU) x = silu_f(x);
            else if (p.epi == DK_EPI_GATE_RES) x = bf2f(p.res[rrow + nb + e]) + round_bf16(bf2f(gate[nb + e]) * x);
            else if (p.epi == DK_EPI_RES) x += bf2f(p.res[rrow + nb + e]);
            p.C[crow + nb + e] = f2bf(x);
          }
        }
      }
    }
  }
}

// tuning knob (dk_tune_set("gemm", v)): -1 = automatic choice, 128 = always the 128^2-tile kernel of this file, 9 = the 256^2 kernel
// (gemm256v3.hip) on every shape it accepts, 10 = the one-wave-per-SIMD 256^2 kernel (gemm256v4.hip) on every shape IT accepts (others: 9)
int g_dk_gemm_mode = -1;
thread_local DkGemmPlan* g_dk_gemm_plan = nullptr;

// Which of the two 256^2 kernels takes a launch both accept.  gemm256v4.hip (one wave per SIMD, asm body) runs its K loop 5-7 % faster
// (profiles/r05_gemm_v4_*.log: 8192^3 1587 against 1488 TF with cold weights, the model's linear1 / linear2 / fc1 / fc2 +4-5 %), but has no
// remainder handling: gemm256v3.hip keeps the launches whose last round of the CUs is a small remainder (its 224-row tiles and the
// remainder-wave K split fill those: the q / k / v projections at 2.25-2.4 rounds, the grouped image + text fc1 at 3.2).
int g_dk_pair_split_nk = -1;  // dk_tune_set("gemm_pair_nk", v): K-tile steps from which an image + text pair whose extra round is a small remainder is grouped and cut along K; -1: 24 (32 until the raw-accumulator exchange of round 6: profiles/r06_gemm_pair_nk.log)
int g_dk_v4_auto = -1;  // dk_tune_set("gemm_v4", v): -1 (default) the rule below, 0 never in the automatic choice (A/B runs)
static bool dk_use_v4(const GemmParams& a, const GemmParams* b) {
  if (g_dk_gemm_mode != 10 && (g_dk_gemm_mode != -1 || g_dk_v4_auto == 0)) return false;
  if (!dk_gemm256v4_eligible(a) || (b != nullptr && !dk_gemm256v4_eligible(*b))) return false;
  if (g_dk_gemm_mode == 10) return true;
  // Small launches (round 6): at most half a round of tiles, and gemm256v3.hip would cut every one of them along K -- FLUX at the reference CLI's
  // 512 x 512 default: o_proj / fc2 / linear2 are 60 - 84 tiles on 256 CUs, linear2 229 us on this kernel (profiles/r06_flux_512_kernel_stats_before.md)
  if (g_dk_v4_auto != 2 && dk_gemm256v3_splits_whole_launch(a, b)) return false;
  // launches with tiles that straddle row segments or short reductions with ragged rows stay on gemm256v3.hip: this kernel's per-row tail path
  // is slow (the lab's 1178 x 6144 x 1536 text fc1: 80 us here against 49 there) and its fixed cost per tile ~ 1 us higher (SD3-medium in the
  // model: every eligible launch 22.0 against 21.5 ms per step, whole-tile launches only 21.2; profiles/r05_gemm_v4_in_model.log).
  // ("gemm_v4" 2: lab, no such restriction)
  const int n_cu = dk_device_cu_count();
  const int mf = dk_gemm256v4_pick_mf(a, b, n_cu), bm = 32 * mf;
  const bool uniform = dk_gemm256v4_uniform_tiles(a, bm) && (b == nullptr || dk_gemm256v4_uniform_tiles(*b, bm));
  // (K < 2048: the image stream's fc1 of SD3 alone on this kernel measured +1.1 % per step on one box and -0.9 % on another: short reductions stay
  //  on gemm256v3.hip)
  if ((!uniform || a.K < 2048) && g_dk_v4_auto != 2) return false;
  long tiles = (long)((a.M + bm - 1) / bm) * (a.N / 256);
  if (b) tiles += (long)((b->M + bm - 1) / bm) * (b->N / 256);
  // (the rounds test in units of the device's CUs: the constants were fitted on 256 CUs -- a last round more than 56 % full, or 8 rounds and more)
  const long frac = tiles % n_cu;
  return tiles <= n_cu || tiles >= 8L * n_cu || frac == 0 || frac * 16 > 9L * n_cu;
}

int dk_launch_gemm(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  if (p.ldw <= 0) p.ldw = p.K;
  DK_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
  DK_REQUIRE(p.ldw >= p.K && p.ldw % 8 == 0, "ldw must be >= K and a multiple of 8 elements");
  // the 256^2 kernel (16x16x32 MFMA, LDS-DMA ring, any M and any row-segment map) takes every large-M shape it accepts; small M
  // (modulation tables, embedders, a lone text stream) and N % 256 != 0 stay on the 128^2 tiles
  // (mode 10 forces gemm256v4.hip on what IT accepts and leaves every other launch to the automatic choice: usable around a whole model)
  // (N % 256 == 128 -- the half column tile -- only where the wasted half tile is a small fraction: SD3.5-large's N = 2432 / 7296; a small
  //  N such as the VAE's 128-channel shortcut Linear stays on the 128^2 tiles: ADVICE r4)
  const bool half_ok = p.N % 256 == 0 || p.N >= 1024 || g_dk_gemm_mode == 9;
  const bool big = !p.conv && g_dk_gemm_mode != 128 && half_ok && dk_gemm256v3_eligible(p) &&
                   (p.M >= 1024 || g_dk_gemm_mode == 9 || (g_dk_gemm_mode == 10 && dk_gemm256v4_eligible(p)));
  if (g_dk_gemm_mode == 9 && !p.conv) DK_REQUIRE(big, "gemm256v3 forced but the shape does not allow it");
  if (big && dk_use_v4(p, nullptr)) return dk_launch_gemm256v4(p, nullptr, stream);
  if (big) return dk_launch_gemm256v3(p, nullptr, stream);
  if (p.kn_w != nullptr) {
    // the fused key QKNorm + RoPE lives in the 256^2 kernel's tail: any other route runs the projection plain and the
    // stand-alone pass over its key columns afterwards
    GemmParams plain = p;
    plain.kn_w = nullptr;
    plain.qn_w = nullptr;
    const int rc = dk_launch_gemm(plain, stream);
    if (rc) return rc;
    DK_REQUIRE(p.c_seg_len == p.kn_seg_len || p.c_seg_len >= p.M, "fused key QKNorm: the output's row segments must be the sequences");
    // (with the query side asked for as well -- qn_w -- the pass covers both column ranges: they hold the same number of heads)
    DK_REQUIRE(p.qn_w == nullptr || p.qn_col1 - p.qn_col0 == p.kn_col1 - p.kn_col0, "fused QKNorm: query and key ranges must hold the same heads");
    return dk_launch_qk_norm_rope(p.C, p.ldc, p.qn_w ? p.qn_col0 : 0, p.kn_col0, p.M, (p.kn_col1 - p.kn_col0) / p.kn_D, p.kn_D, p.qn_w ? p.qn_w : p.kn_w, p.kn_w,
                                  p.kn_eps, p.kn_rope, p.kn_seg_len, p.c_seg_len == p.kn_seg_len ? p.c_seg_stride : p.kn_seg_len, p.kn_pos_off, 0,
                                  stream, p.qn_w ? 0 : 1);
  }
  if (p.n_split > 0) {
    // column-split GEMM on the kernel without split support: two GEMMs over the two column ranges
    DK_REQUIRE(p.n_split < p.N && p.C2 != nullptr, "bad column split");
    GemmParams a = p, b = p;
    a.N = p.n_split; a.n_split = 0;
    b.N = p.N - p.n_split; b.n_split = 0; b.W = p.W + (size_t)p.n_split * p.ldw; b.bias = p.bias ? p.bias + p.n_split : nullptr;
    b.C = p.C2; b.ldc = p.ldc2; b.epi = p.epi2;
    const int rc = dk_launch_gemm(a, stream);
    return rc ? rc : dk_launch_gemm(b, stream);
  }
  if (p.conv && g_dk_gemm_mode != 128 && dk_gemm256v3_eligible(p)) {
    // implicit-GEMM convolutions with O % 256 == 0 ride the 256^2 kernel once their tiles fill most of the CUs (the VAE's 256^2-pixel
    // and larger stages); the 128^2-tile kernel below keeps the small stages and O = 128
    const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    // (with the K-split workspace a stage of about half the CUs' worth of tiles goes there too: its tiles are cut in two along K)
    if (g_dk_gemm_mode == 9 || t256 >= 192 || (p.workspace != nullptr && t256 >= 96 && t256 <= 128)) return dk_launch_gemm256v3(p, nullptr, stream);
  }
  DK_REQUIRE(p.K % BK == 0, "K must be a multiple of 64");
  DK_REQUIRE(p.ldc % 4 == 0, "ldc must be a multiple of 4 elements");
  DK_REQUIRE(p.conv || (p.lda % 8 == 0), "lda must be a multiple of 8 elements");
  if (p.conv) {
    DK_REQUIRE(p.cC % BK == 0 && p.K == 9 * p.cC, "conv: C must be a multiple of 64 and K = 9*C");
    DK_REQUIRE(p.M == p.cB * p.cH * p.cW, "conv: M must equal B*H*W");
    DK_REQUIRE(p.zeros != nullptr, "conv: zero page missing");
  }
  if (p.epi == DK_EPI_GATE_RES) DK_REQUIRE(p.gate && p.res, "gate/res missing");
  if (p.epi == DK_EPI_RES) DK_REQUIRE(p.res, "res missing");
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  if (g_dk_gemm_plan != nullptr) {
    DkGemmPlan& pl = *g_dk_gemm_plan;
    pl.kernel = 128; pl.tile_rows = BM; pl.tiles = pl.workgroups = nbm * nbn; pl.split_tiles = 0; pl.k_pieces = 1; pl.ks = p.K / BK;
    pl.n_cu = dk_device_cu_count(); pl.launches += 1;
    return 0;
  }
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_gemm_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
    attr_once.mark();
  }
  dim3 grid(nbm * nbn), block(256);
  dk_prof_begin(p.conv ? 1 : 0, 2.0 * (double)p.M * (double)p.N * (double)p.K, stream);
  if (p.conv)
    hipLaunchKernelGGL(dk_gemm_bf16_kernel<1>, grid, block, 2 * STAGE_BYTES, stream, p);
  else
    hipLaunchKernelGGL(dk_gemm_bf16_kernel<0>, grid, block, 2 * STAGE_BYTES, stream, p);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}

int dk_launch_gemm_pair(const GemmParams& a_in, const GemmParams& b_in, hipStream_t stream) {
  GemmParams a = a_in, b = b_in;
  if (a.ldw <= 0) a.ldw = a.K;
  if (b.ldw <= 0) b.ldw = b.K;
  const bool same = a.N == b.N && a.K == b.K && a.epi == b.epi && a.alpha == b.alpha && a.n_split == 0 && b.n_split == 0;
  if ((g_dk_gemm_mode == -1 || g_dk_gemm_mode == 10) && same && (a.M >= 1024 || b.M >= 1024) && (a.N % 256 == 0 || a.N >= 1024) &&
      dk_gemm256v3_eligible(a) && dk_gemm256v3_eligible(b)) {
    // group only when the extra tiles do not open another wave of the 256 CUs (kernel lab: a partial extra wave costs more
    // than the small separate launch) ...
    const long ta = (long)((a.M + 255) / 256) * ((a.N + 255) / 256), tb = (long)((b.M + 255) / 256) * ((b.N + 255) / 256);
    const long n_cu = dk_device_cu_count();  // (rounds in units of THIS device's CUs; the fractions below were fitted on 256)
    // ... unless the kernel can cut that extra, small wave along K (remainder-wave split, needs the workspace)
    const bool split_ok = g_dk_v3_split != 0 && a.workspace != nullptr && (ta + tb) % n_cu <= n_cu / 4 && a.K / 64 >= (g_dk_pair_split_nk >= 0 ? g_dk_pair_split_nk : 24);
    // (gemm256v4.hip: the same test at ITS tile height -- with 224-row tiles the image + text fc1 of FLUX is 912 + 96 tiles: both 4 rounds)
    if (dk_gemm256v4_eligible(a) && dk_gemm256v4_eligible(b)) {
      const int bm4 = 32 * dk_gemm256v4_pick_mf(a, &b, (int)n_cu);
      const long ta4 = (long)((a.M + bm4 - 1) / bm4) * (a.N / 256), tb4 = (long)((b.M + bm4 - 1) / bm4) * (b.N / 256);
      if ((ta4 + n_cu - 1) / n_cu == (ta4 + tb4 + n_cu - 1) / n_cu && dk_use_v4(a, &b)) return dk_launch_gemm256v4(a, &b, stream);
    }
    if ((ta + n_cu - 1) / n_cu == (ta + tb + n_cu - 1) / n_cu || split_ok) return dk_launch_gemm256v3(a, &b, stream);
  }
  int rc = dk_launch_gemm(a, stream);
  if (rc) return rc;
  return dk_launch_gemm(b, stream);
}
