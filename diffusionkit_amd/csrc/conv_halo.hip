// 3x3 convolution of the VAE's high-resolution stage with LDS halo staging and the GroupNorm-apply + SiLU of the layer in
// front of it as the load prologue (BASELINE north_star: "VAE conv/GroupNorm with coalesced HBM loads and LDS halo staging").
// reference: python/src/diffusionkit/mlx/vae.py:60-101 (ResnetBlock2D: norm -> silu -> conv, conv_shortcut, + x), :72,78 (GroupNorm
// sites), :381,397-399 (conv_norm_out -> silu -> conv_out), python/src/diffusionkit/mlx/__init__.py:581-584,525-526 (clip, uint8).
//
// Why a second conv kernel: the implicit-GEMM forms (gemm.hip / gemm256v3.hip CONV) fetch every input pixel nine times -- once
// per tap, through the L2 -- and need the normalised + activated tensor in memory, i.e. a GroupNorm-apply pass (read + write of
// the whole activation) in front of every conv.  At the 1024 x 1024 x 128 stage those passes and the nine-fold operand traffic
// weigh as much as the MFMA work (128-column tiles: 32 KiB of operands per 128 x 128 x 64 MACs).  Here a workgroup owns a
// 16 x 16 pixel tile: per 64-channel chunk it loads the 18 x 18 halo ONCE (coalesced 16-byte loads, 128 contiguous bytes per
// pixel), applies x * scale[c] + shift[c] -> bf16 -> SiLU -> bf16 in registers (the same arithmetic and rounding points as
// dk_gn_apply_kernel; padding pixels become zeros AFTER the transform, as the conv pads the activated tensor), writes it to LDS,
// and all nine taps read their shifted windows from that tile: 1.27 instead of 9 operand fetches per pixel and channel, and no
// activated tensor in HBM at all.
//
// GEMM view: M-tile = 256 pixels (16 x 16), N-tile = NT output channels, K = (64-channel chunk, tap).  8 waves: WM x WN, a wave
// owns MF pixel rows (16 pixels = one MFMA fragment each) x NF 16-column fragments; v_mfma_f32_16x16x32_bf16 with the weight
// fragment as the A operand (lane holds pixel l15, columns 4q .. 4q+3: row-major stores), fp32 accumulation in chunk-major order.
//   LDS: two halo slots (324 rows x 160 B: 128 B of channels + 32 B pad, conflict-free ds_read_b128 for 16 consecutive rows)
//        + three weight slots (NT rows x 128 B, XOR-swizzled 16-byte chunks as in gemm256v3.hip).
//   Pipeline: weights of K-tile s+4 -> registers, K-tile s+2 registers -> LDS, while K-tile s multiplies; the halo of chunk c+1
//   is loaded at the first tap of chunk c and transformed + stored during its later taps; one barrier per K-tile.  The fragment
//   reads run one half K-tile ahead of the MFMAs ACROSS that barrier: K-tile s+1's weights have been in LDS since the barrier
//   before, its halo window since the chunk began, so the first fragments of s+1 are fetched while the second half of s multiplies
//   (first version, one set of reads per K-tile behind the barrier: 760-915 TFLOP/s; profiles/r03_vae_kernel_stats_halo_v1.md).
// Optional K extension (x2): the 1x1 conv_shortcut of a channel-changing resnet (vae.py:86-89,98-99) as extra K-tiles over the
// raw block input (centre tap only, no transform) -- the shortcut never exists as a tensor.
// Optional statistics of the OUTPUT (stats_out): per workgroup the (sum, sum of squares) of the stored bf16 values per output
// channel group -- the partials dk_gn_finalize_kernel combines for the GroupNorm that reads this tensor next.
// NT = 16 (IMG): conv_out (3 of 16 columns live) with the clip / uint8 / float image tail of dk_image_post_kernel fused.
#include "dk_kernels.h"

typedef __attribute__((address_space(3))) char lds_c;

#ifndef CH_SGB
#define CH_SGB 0  // lab: 1 = a sched_group_barrier pipeline inside a K-tile (one LDS read / memory instruction / a few VALU of the halo
                  // transform behind every MFMA, guide T19) instead of the compiler's own order.  Measured flat (+-2 %,
                  // profiles/r03_conv_halo_ablations.md): the order of a wave's instructions is not what keeps the MFMA time and
                  // everything else from overlapping.
#endif
#ifndef CH_ABL
#define CH_ABL 0  // lab only (scripts/build_halo_abl.sh), bit mask of what the K loop leaves out: 1 the MFMAs, 2 the fragment reads, 4 the
                  // barriers, 8 the GroupNorm / SiLU transform, 16 the weight stream (loads + LDS stores), 32 the halo stream of the next
                  // chunk (loads + LDS stores), 64 the output stores of the tail.  Results are wrong; the timings say what each part costs.
#endif

#define CH_ROWS 324          // 18 x 18 halo pixels
#define CH_ROWB 160          // bytes per halo row in LDS
#define CH_A_SLOT (CH_ROWS * CH_ROWB)
#define CH_ITEMS 6           // 16-byte halo items per thread and chunk: 324 * 8 = 2592 = 5 * 512 + 32

template <int NT, bool IMG>
__global__ __launch_bounds__(512, 2) void dk_conv_halo_kernel(ConvHaloParams p) {
  constexpr int WN = NT >= 128 ? 2 : 1;
  constexpr int WM = 8 / WN;
  constexpr int MF = 16 / WM;            // pixel rows (16-pixel fragments) per wave
  constexpr int NF = NT / (16 * WN);     // 16-column fragments per wave
  constexpr int W_SLOT = NT * 128;
  constexpr int W_OFF = 2 * CH_A_SLOT;
  constexpr int W_ITEMS = (NT * 8 + 511) / 512;  // 16-byte weight items per thread and K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_c*)smem != 0u) __builtin_trap();  // the LDS image is addressed from 0
  lds_c* const lds = (lds_c*)0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, q = lane >> 4;

  // ---- this workgroup's tile ----
  const int tiles_n = (p.O + NT - 1) / NT, tiles_x = p.W >> 4, tiles_y = p.H >> 4;
  const int bid = blockIdx.x;
  const int nt = bid % tiles_n, pt = bid / tiles_n;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, b = pt / (tiles_x * tiles_y);
  const int n0 = nt * NT;
  const int Hs = p.H >> p.ups, Ws = p.W >> p.ups;

  // ---- halo items of this thread: (halo row, 16-byte channel chunk c8 = tid & 7) ----
  const int c8 = tid & 7;
  // (item i = halo row (tid >> 3) + 64 i: the LDS address is one lane-constant + i * 64 rows; the global offset is rebuilt per load
  //  from the source pixel index -- 6 registers instead of 18 for the three address sets)
  unsigned hpix[CH_ITEMS];  // source pixel of the item: (y >> ups) * Ws + (x >> ups) in the stored tensor (== y * W + x for the shortcut input)
  const unsigned lds_w0 = (unsigned)((tid >> 3) * CH_ROWB + c8 * 16);
  unsigned okmask = 0u, inmask = 0u;
#pragma unroll
  for (int i = 0; i < CH_ITEMS; ++i) {
    const int id = tid + 512 * i;
    const int hrow = id >> 3;
    const int hy = hrow / 18, hx = hrow - hy * 18;
    const int y = ty * 16 - 1 + hy, x = tx * 16 - 1 + hx;
    const bool in = id < CH_ROWS * 8;
    const bool ok = in && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    hpix[i] = ok ? (unsigned)((y >> p.ups) * Ws + (x >> p.ups)) : 0u;
    okmask |= (ok ? 1u : 0u) << i;
    inmask |= (in ? 1u : 0u) << i;
  }
  // buffer descriptors from PROVABLY wave-uniform words (readfirstlane): a descriptor the compiler cannot prove uniform is wrapped in
  // a waterfall loop per memory instruction (guide T20)
  auto uptr = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  auto uint_ = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  const bf16_t* const xb = p.x + (size_t)b * Hs * Ws * p.C;
  const bf16_t* const x2b = p.x2 ? p.x2 + (size_t)b * p.H * p.W * p.C2 : p.x;
  const int nrec_x = Hs * Ws * p.C * 2, nrec_x2 = p.x2 ? p.H * p.W * p.C2 * 2 : 0;
  const float* gss = p.gn_ss ? p.gn_ss + (size_t)b * 2 * p.C : nullptr;
  // (scale | shift) of the chunk whose halo is in flight: fetched WITH the halo (unconditionally -- from the weights when there is
  // no table, values unused -- so that the loads are old by the time the transform needs them and no branch hides a load)
  const float* const gsrc = gss ? gss : (const float*)p.w;
  const int gsh_off = gss ? p.C : 0;
  f32x4 gt[4];  // scale[0:4], scale[4:8], shift[0:4], shift[4:8] of this thread's 8 channels

  // ---- weight items of this thread: row = (tid >> 3) + 64 * j, chunk = tid & 7 ----
  unsigned wsrc[W_ITEMS], wdst[W_ITEMS];
#pragma unroll
  for (int j = 0; j < W_ITEMS; ++j) {
    const int row = (tid >> 3) + 64 * j;
    const bool okw = row < NT && n0 + row < p.O;
    wsrc[j] = okw ? ((unsigned)(n0 + row) * (unsigned)p.ldw + c8 * 8u) * 2u : 0x80000000u;
    wdst[j] = (unsigned)(W_OFF + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
  }
  const bool w_thread = NT * 8 >= 512 || (tid >> 3) < NT;  // (NT = 16: only the first 128 threads carry a weight item)
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(uptr(p.w), 0, uint_(p.O * p.ldw * 2), 0x00020000);

  // ---- K-tile sequence: main chunks (9 taps each), then the shortcut chunks (centre tap) ----
  const int n_main = p.C >> 6, n_sc = p.x2 ? (p.C2 >> 6) : 0;
  const int n_chunks = n_main + n_sc;
  const int nkt = 9 * n_main + n_sc;
  auto kt_col = [&](int s) -> int {  // first weight column of K-tile s, in BYTES
    if (s < 9 * n_main) {
      const int cc = s / 9, tap = s - cc * 9;
      return (tap * p.C + cc * 64) * 2;
    }
    return (9 * p.C + (s - 9 * n_main) * 64) * 2;
  };
// (loads are issued UNCONDITIONALLY, with clamped indices where there is nothing left to fetch: a load under a branch leaves the
//  compiler's vmcnt bookkeeping with two different histories at the merge, and it then drains every outstanding load in front of the
//  next LDS read -- measured in the .s: vmcnt(0) at the head of every K-tile)
#define CH_LOAD_W(REG, S)                                                                                           \
  do {                                                                                                              \
    if (w_thread && !(CH_ABL & 16)) { /* (NT = 128: a compile-time true) */                                         \
      const int sc_ = (S) < nkt ? (S) : nkt - 1;                                                                    \
      const int col_ = uint_(kt_col(sc_));                                                                          \
      _Pragma("unroll") for (int j = 0; j < W_ITEMS; ++j) REG[j] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)wsrc[j], col_, 0); \
    }                                                                                                               \
  } while (0)
#define CH_STORE_W(REG, SLOT)                                                                                       \
  do {                                                                                                              \
    if (w_thread && !((CH_ABL & 16) && in_loop)) {                                                                  \
      _Pragma("unroll") for (int j = 0; j < W_ITEMS; ++j)                                                           \
          *(__attribute__((address_space(3))) u32x4*)(lds + wdst[j] + (SLOT) * W_SLOT) = REG[j];                    \
    }                                                                                                               \
  } while (0)
  // halo of chunk c -> registers (raw bf16); chunks >= n_main come from the shortcut input.  (One code path, descriptor rebuilt from
  // uniform scalars: two descriptors selected by a branch end up in VGPRs and every load in a waterfall loop.)
#define CH_LOAD_HALO(REG, CHUNK)                                                                                              \
  do {                                                                                                                        \
    const int c_ = uint_(CHUNK);                                                                                              \
    const bool m_ = c_ < n_main;                                                                                              \
    const __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc(uptr(m_ ? (const void*)xb : (const void*)x2b), 0,     \
                                                                        uint_(m_ ? nrec_x : nrec_x2), 0x00020000);            \
    const int so_ = uint_((m_ ? c_ : c_ - n_main) * 128);                                                                     \
    const unsigned cb_ = (unsigned)uint_((m_ ? p.C : p.C2) * 2); /* bytes per pixel of the source tensor */                   \
    _Pragma("unroll") for (int i = 0; i < CH_ITEMS; ++i)                                                                      \
        REG[i] = __builtin_amdgcn_raw_buffer_load_b128(                                                                        \
            r_, (int)(((okmask >> i) & 1u) ? hpix[i] * cb_ + c8 * 16u : 0x80000000u /* padding: out of range -> zeros */), so_, 0); \
    const float* g_ = gsrc + (m_ ? c_ : 0) * 64 + c8 * 8;                                                                     \
    gt[0] = *(const f32x4*)g_, gt[1] = *(const f32x4*)(g_ + 4);                                                               \
    gt[2] = *(const f32x4*)(g_ + gsh_off), gt[3] = *(const f32x4*)(g_ + gsh_off + 4);                                         \
  } while (0)
  u32x4 hreg[CH_ITEMS];
  // registers (hreg) -> (GroupNorm-apply + SiLU) -> LDS halo slot `slot`; items i0 .. i1-1 of chunk c (the 480 threads without a sixth
  // item store theirs to a dummy LDS zone behind the weight slots: no divergent branch around the store)
  const unsigned dummy_w = (unsigned)(W_OFF + 3 * W_SLOT + tid * 16);
  auto store_halo = [&](int slot, int c, int i0, int i1) {
    const bool xform = gss != nullptr && c < n_main && !((CH_ABL & 8) && c > 0);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) sc[e] = gt[0][e], sc[4 + e] = gt[1][e], sh[e] = gt[2][e], sh[4 + e] = gt[3][e];
#pragma unroll
    for (int i = 0; i < CH_ITEMS; ++i) {
      if (i < i0 || i >= i1) continue;
      u32x4 o = hreg[i];
      if (xform) {
        const unsigned keep = 0u - ((okmask >> i) & 1u);  // padding: zeros of the ACTIVATED tensor
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a0, a1;
          unpack2bf(hreg[i][e], a0, a1);
          float g0 = round_bf16(a0 * sc[2 * e] + sh[2 * e]), g1 = round_bf16(a1 * sc[2 * e + 1] + sh[2 * e + 1]);
          if (p.gn_silu) g0 = silu_f(g0), g1 = silu_f(g1);
          o[e] = pack2bf(g0, g1) & keep;
        }
      }
      const unsigned dst = ((inmask >> i) & 1u) ? lds_w0 + i * (64 * CH_ROWB) + slot * CH_A_SLOT : dummy_w;
      *(__attribute__((address_space(3))) u32x4*)(lds + dst) = o;
    }
  };

  f32x4 acc[NF][MF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane parts of the fragment addresses
  const unsigned a_lane = (unsigned)((wm * MF * 18 + l15) * CH_ROWB + q * 16);
  unsigned w_lane[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
    w_lane[kk] = (unsigned)(W_OFF + (wn * NF * 16 + l15) * 128 + (((kk * 4 + q) ^ (l15 >> 1)) << 4));

  // ---- prologue: halo of chunk 0, weights of K-tiles 0, 1 (-> LDS slots 0, 1) and 2, 3 (-> registers) ----
  // Weight registers: K-tile k travels in register set k % 3 -- loaded in iteration k - 4, stored to LDS slot k % 3 in iteration
  // k - 2 -- so the loops are unrolled by three with compile-time set / slot names (a rotation by copies would make every
  // iteration wait for the loads it has just issued).
  u32x4 wr0[W_ITEMS], wr1[W_ITEMS], wr2[W_ITEMS];
#pragma unroll
  for (int j = 0; j < W_ITEMS; ++j) wr0[j] = wr1[j] = wr2[j] = u32x4{0u, 0u, 0u, 0u};
  constexpr bool in_loop = false;  // (lab ablations leave the prologue alone)
  CH_LOAD_HALO(hreg, 0);
  CH_LOAD_W(wr0, 0);
  CH_LOAD_W(wr1, 1);
  store_halo(0, 0, 0, CH_ITEMS);
  CH_STORE_W(wr0, 0);
  CH_STORE_W(wr1, 1);
  CH_LOAD_W(wr2, 2);
  CH_LOAD_W(wr0, 3);
  __syncthreads();

  // fragment sets: F0 = the K = 0..31 half of a K-tile, F1 = the K = 32..63 half
  bf16x8 wf0[NF], af0[MF], wf1[NF], af1[MF];
#define CH_READ(WF, AF, WSLOT, ABASE, KK)                                                                                       \
  if (!((CH_ABL & 2) && in_loop)) do {                                                                                          \
    _Pragma("unroll") for (int nf = 0; nf < NF; ++nf)                                                                           \
        WF[nf] = *(const __attribute__((address_space(3))) bf16x8*)(lds + w_lane[KK] + (WSLOT) * W_SLOT + nf * 2048);           \
    _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                                                                           \
        AF[mf] = *(const __attribute__((address_space(3))) bf16x8*)(lds + (ABASE) + mf * (18 * CH_ROWB) + (KK) * 64);           \
  } while (0)
#define CH_MMA(WF, AF)                                                                                                          \
  if (!(CH_ABL & 1)) do {                                                                                                       \
    _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) _Pragma("unroll") for (int mf = 0; mf < MF; ++mf)                         \
        acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nf], AF[mf], acc[nf][mf], 0, 0, 0);                            \
  } while (0)
  // halo window of (chunk, tap): byte offset of halo row (dy * 18 + dx) in the chunk's slot
#define CH_ABASE(CHUNK, DY, DX) (a_lane + (unsigned)(((CHUNK) & 1) * CH_A_SLOT + ((DY) * 18 + (DX)) * CH_ROWB))
  CH_READ(wf0, af0, 0, CH_ABASE(0, n_main > 0 ? 0 : 1, n_main > 0 ? 0 : 1), 0);

  // The order the instructions of a K-tile are to be issued in (sched_group_barrier, guide T19): every MFMA is followed by one LDS
  // fragment read while there are any (2 x (NF + MF) per K-tile), the memory instructions (weight loads, weight stores to LDS, at a
  // chunk's first tap the halo loads) go behind MFMAs too -- so that a wave's own LDS / memory issue sits in the shadow of its MFMAs
  // instead of in front of them, and the two waves of a SIMD do not meet in the same phase.
  // masks: 0x8 MFMA, 0x100 DS read, 0x200 DS write, 0x20 VMEM read
#if CH_SGB == 1
#define CH_PIPELINE(HLOAD, NH)                                                                                                  \
  do {                                                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 2 * (NF + MF); ++i_) {                                                              \
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                        \
      if ((NH) > 0) __builtin_amdgcn_sched_group_barrier(0x2, 3, 0); /* the halo transform rides along, 3 VALU per MFMA */      \
    }                                                                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < W_ITEMS; ++i_) {                                                                    \
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                          \
      __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                                                         \
      if ((NH) > 0) __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);                                                            \
    }                                                                                                                           \
    if (HLOAD) {                                                                                                                \
      _Pragma("unroll") for (int i_ = 0; i_ < CH_ITEMS + 4; ++i_) {                                                             \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                                                       \
      }                                                                                                                         \
    }                                                                                                                           \
    if ((NH) > 0) {                                                                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < 2 * NF * MF - 2 * (NF + MF) - 2 * W_ITEMS; ++i_) {                                \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);                                                                        \
      }                                                                                                                         \
    }                                                                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < W_ITEMS + (NH); ++i_) {                                                             \
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                                          \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                                        \
    }                                                                                                                           \
  } while (0)
#else
#define CH_PIPELINE(HLOAD, NH) do { } while (0)
#endif
  // One K-tile.  S: K-tile index (weights in LDS slot WS3 = S % 3), CHUNK / DY / DX: its chunk and tap, NEXT_ABASE: halo window of
  // K-tile S + 1 (its weights: slot (WS3 + 1) % 3), WL: the register set that receives K-tile S + 4 (set (WS3 + 1) % 3),
  // WST: the set holding K-tile S + 2 (-> LDS slot (WS3 + 2) % 3), HLOAD: issue the halo loads of chunk CHUNK + 1 (clamped),
  // H0 .. H1: halo items of chunk CHUNK + 1 to transform + store
#define CH_KTILE(S, CHUNK, DY, DX, WS3, NEXT_ABASE, WL, WST, HLOAD, H0, H1)                                                     \
  {                                                                                                                             \
    constexpr bool in_loop = true;                                                                                              \
    const int s_ = uint_(S), c_k = uint_(CHUNK);                                                                                \
    CH_LOAD_W(WL, s_ + 4);                                                                                                      \
    if ((HLOAD) && !(CH_ABL & 32)) {                                                                                            \
      /* the weight loads stay OLDER than the halo loads: vmcnt retires in order, and these weights are stored two K-tiles from \
         here -- behind the halo loads they would wait for the halo's HBM latency */                                            \
      if (!CH_SGB) __builtin_amdgcn_sched_barrier(0);                                                                            \
      CH_LOAD_HALO(hreg, c_k + 1 < n_chunks ? c_k + 1 : n_chunks - 1);                                                          \
    }                                                                                                                           \
    CH_READ(wf1, af1, WS3, CH_ABASE(c_k, DY, DX), 1);                                                                           \
    CH_MMA(wf0, af0);                                                                                                           \
    /* (behind the last K-tile this fetches a window nobody multiplies: unconditional, so that the K-tile stays one scheduling  \
       region) */                                                                                                               \
    CH_READ(wf0, af0, ((WS3) + 1) % 3, (NEXT_ABASE), 0);                                                                        \
    CH_MMA(wf1, af1);                                                                                                           \
    CH_STORE_W(WST, ((WS3) + 2) % 3);                                                                                           \
    if ((H1) > (H0) && c_k + 1 < n_chunks && !(CH_ABL & 32)) store_halo((c_k + 1) & 1, c_k + 1, (H0), (H1));                    \
    CH_PIPELINE(HLOAD, (H1) - (H0));                                                                                            \
    if (!(CH_ABL & 4)) __syncthreads();                                                                                         \
  }
  // main chunks: nine taps, K-tile 9 * cc + tap: LDS slot / register set tap % 3; the next chunk's halo is loaded at tap 0 and stored
  // two items per tap over taps 5 - 7 (visible, behind tap 7's barrier, when tap 8 prefetches the next chunk's first window)
  for (int cc = 0; cc < n_main; ++cc) {
    const int s0 = 9 * cc;
    // the K-tile behind this chunk's last tap: tap (0, 0) of the next main chunk, or the centre tap of the first shortcut chunk
    const unsigned nxt = cc + 1 < n_main ? CH_ABASE(cc + 1, 0, 0) : CH_ABASE(cc + 1, 1, 1);
    CH_KTILE(s0 + 0, cc, 0, 0, 0, CH_ABASE(cc, 0, 1), wr1, wr2, true, 0, 0)
    CH_KTILE(s0 + 1, cc, 0, 1, 1, CH_ABASE(cc, 0, 2), wr2, wr0, false, 0, 0)
    CH_KTILE(s0 + 2, cc, 0, 2, 2, CH_ABASE(cc, 1, 0), wr0, wr1, false, 0, 0)
    CH_KTILE(s0 + 3, cc, 1, 0, 0, CH_ABASE(cc, 1, 1), wr1, wr2, false, 0, 0)
    CH_KTILE(s0 + 4, cc, 1, 1, 1, CH_ABASE(cc, 1, 2), wr2, wr0, false, 0, 0)
    CH_KTILE(s0 + 5, cc, 1, 2, 2, CH_ABASE(cc, 2, 0), wr0, wr1, false, 0, 2)
    CH_KTILE(s0 + 6, cc, 2, 0, 0, CH_ABASE(cc, 2, 1), wr1, wr2, false, 2, 4)
    CH_KTILE(s0 + 7, cc, 2, 1, 1, CH_ABASE(cc, 2, 2), wr2, wr0, false, 4, 6)
    CH_KTILE(s0 + 8, cc, 2, 2, 2, nxt, wr0, wr1, false, 0, 0)
  }
  // shortcut chunks: one K-tile each (centre tap); the next one's halo is loaded and stored inside the K-tile BEFORE the prefetch of
  // its first window would need it -- so the prefetch is dropped here: these K-tiles read both halves behind the barrier (their
  // latency is exposed, at most C2 / 64 times per workgroup)
#define CH_KTILE_SC(S, CHUNK, WS3, WL, WST)                                                                                     \
  {                                                                                                                             \
    constexpr bool in_loop = true;                                                                                              \
    const int s_ = uint_(S), c_k = uint_(CHUNK);                                                                                \
    CH_LOAD_W(WL, s_ + 4);                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
    CH_LOAD_HALO(hreg, c_k + 1 < n_chunks ? c_k + 1 : n_chunks - 1);                                                            \
    CH_READ(wf1, af1, WS3, CH_ABASE(c_k, 1, 1), 1);                                                                             \
    CH_MMA(wf0, af0);                                                                                                           \
    CH_MMA(wf1, af1);                                                                                                           \
    CH_STORE_W(WST, ((WS3) + 2) % 3);                                                                                           \
    if (c_k + 1 < n_chunks) store_halo((c_k + 1) & 1, c_k + 1, 0, CH_ITEMS);                                                    \
    __syncthreads();                                                                                                            \
    if (s_ + 1 < nkt) CH_READ(wf0, af0, ((WS3) + 1) % 3, CH_ABASE(c_k + 1, 1, 1), 0);                                           \
  }
  for (int j = 0; j < n_sc; j += 3) {
    const int s0 = 9 * n_main + j, c0 = n_main + j;
    CH_KTILE_SC(s0, c0, 0, wr1, wr2)
    if (j + 1 < n_sc) CH_KTILE_SC(s0 + 1, c0 + 1, 1, wr2, wr0)
    if (j + 2 < n_sc) CH_KTILE_SC(s0 + 2, c0 + 2, 2, wr0, wr1)
  }
#undef CH_KTILE
#undef CH_KTILE_SC
#undef CH_PIPELINE
#undef CH_READ
#undef CH_MMA
#undef CH_ABASE
#undef CH_LOAD_W
#undef CH_STORE_W
#undef CH_LOAD_HALO

  // ---- tail ----
  const int px0 = tx * 16, py0 = ty * 16;
  if constexpr (IMG) {
    // conv_out: columns 0 .. out_channels-1 of fragment 0 (q == 0 lanes hold columns 0..3): raw bf16 (4 per pixel), clip(x/2+0.5), uint8
    if (q == 0) {
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      for (int e = 0; e < 4; ++e)
        if (e < p.out_channels) b4[e] = bf2f(p.bias[e]);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const size_t pix = ((size_t)b * p.H + py0 + wm * MF + mf) * p.W + px0 + l15;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = e < p.out_channels ? round_bf16(acc[0][mf][e] + b4[e]) : 0.f;
        if (p.raw) *(u32x2*)(p.raw + pix * 4) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        for (int e = 0; e < 3; ++e) {
          if (e >= p.out_channels) break;
          // the arithmetic of dk_image_post_kernel: bf16 products like the reference's (__init__.py:581-584; :525-526 truncation)
          const float im = fminf(fmaxf(round_bf16(v[e] * 0.5f + 0.5f), 0.f), 1.f);
          if (p.img) p.img[pix * 3 + e] = im;
          if (p.u8) p.u8[pix * 3 + e] = (unsigned char)round_bf16(im * 255.0f);
        }
      }
    }
  } else {
    // accumulators -> wave-private LDS image (bf16 of acc + bias, 64 pixels x 128 B, 16-byte chunk c at position c ^ (row & 7))
    // -> row-major read-back, 16 bytes per lane: 8 lanes cover a pixel's 64 columns
    static_assert(IMG || (MF * 16 == 64 && NF * 16 == 64), "tail written for 64 x 64 wave tiles");
    const unsigned img0 = (unsigned)wave * 8192u;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int col = n0 + wn * 64 + nf * 16 + 4 * q;
      u32x2 bq = *(const u32x2*)(p.bias + col);
      float b4[4];
      unpack2bf(bq[0], b4[0], b4[1]);
      unpack2bf(bq[1], b4[2], b4[3]);
      if (p.bias2) {
        const u32x2 b2 = *(const u32x2*)(p.bias2 + col);
        float c4[4];
        unpack2bf(b2[0], c4[0], c4[1]);
        unpack2bf(b2[1], c4[2], c4[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] += c4[e];
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int row = mf * 16 + l15;
        const f32x4 a = acc[nf][mf];
        *(__attribute__((address_space(3))) u32x2*)(lds + img0 + row * 128 + (((nf * 2 + (q >> 1)) ^ (row & 7)) << 4) + (q & 1) * 8) =
            u32x2{pack2bf(a[0] + b4[0], a[1] + b4[1]), pack2bf(a[2] + b4[2], a[3] + b4[3])};
      }
    }
    // (same wave writes and reads its image: program order + the compiler's lgkmcnt suffice)
    const int rr = lane >> 3, rc = lane & 7;
    const int ocol = n0 + wn * 64 + rc * 8;
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
    // the residual rows of all eight passes up front: one memory latency instead of eight (the compiler cannot move a load of
    // `res` above a store to `y` -- the two may alias for all it knows)
    u32x4 resv[8];
    if (p.res) {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 8 + rr;
        const size_t pix = ((size_t)b * p.H + py0 + wm * MF + (row >> 4)) * p.W + px0 + (row & 15);
        resv[pass] = *(const u32x4*)(p.res + pix * (size_t)p.ldr + ocol);
      }
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 8 + rr;  // pixel inside the wave tile: pixel row row >> 4, x = row & 15
      const size_t pix = ((size_t)b * p.H + py0 + wm * MF + (row >> 4)) * p.W + px0 + (row & 15);
      u32x4 sv = *(const __attribute__((address_space(3))) u32x4*)(lds + img0 + row * 128 + ((rc ^ (row & 7)) << 4));
      if (p.res) {
        const u32x4 rv = resv[pass];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0, v1, r0, r1;
          unpack2bf(sv[e], v0, v1);
          unpack2bf(rv[e], r0, r1);
          sv[e] = pack2bf(v0 + r0, v1 + r1);
        }
      }
      if (!(CH_ABL & 64) || p.ldy == -1) *(u32x4*)(p.y + pix * (size_t)p.ldy + ocol) = sv;
      if (p.stats_out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0, v1;
          unpack2bf(sv[e], v0, v1);
          ssum[2 * e] += v0, ssq[2 * e] += v0 * v0;
          ssum[2 * e + 1] += v1, ssq[2 * e + 1] += v1 * v1;
        }
      }
    }
    if (p.stats_out) {
      // per-channel sums of this lane's 8 pixels -> over the 8 row lanes of the wave (lane bits 3..5, fixed tree) -> LDS ->
      // per-group sums over the 4 waves of a column half and the group's channels, in a fixed order (no atomics: the statistics,
      // and with them the decoded image, stay bit-reproducible)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          ssum[e] += __shfl_xor(ssum[e], o, 64);
          ssq[e] += __shfl_xor(ssq[e], o, 64);
        }
      }
      __syncthreads();  // every wave is done with its staging image
      float* red = (float*)smem;  // [wave][64 channels][2]
      if (rr == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          *(__attribute__((address_space(3))) float*)(lds + ((wave * 64 + rc * 8 + e) * 2 + 0) * 4) = ssum[e];
          *(__attribute__((address_space(3))) float*)(lds + ((wave * 64 + rc * 8 + e) * 2 + 1) * 4) = ssq[e];
        }
      }
      __syncthreads();
      (void)red;
      const int cpg = p.O / p.G_out;          // channels per group
      const int gpt = NT / cpg;               // groups of this N-tile
      if (tid < 2 * gpt) {
        const int g = tid >> 1, stat = tid & 1;
        float a = 0.f;
        for (int c = 0; c < cpg; ++c) {
          const int ch = g * cpg + c;         // channel inside the N-tile: column half ch >> 6, channel ch & 63 of the half
          for (int w4 = 0; w4 < WM; ++w4)
            a += *(const __attribute__((address_space(3))) float*)(lds + (((w4 * WN + (ch >> 6)) * 64 + (ch & 63)) * 2 + stat) * 4);
        }
        const int tiles_img = tiles_x * tiles_y;
        p.stats_out[(((size_t)b * tiles_img + (pt % tiles_img)) * p.G_out + (n0 / cpg + g)) * 2 + stat] = a;
      }
    }
  }
}

bool dk_conv_halo_eligible(const ConvHaloParams& p, bool img) {
  if (p.B <= 0 || p.H % 16 != 0 || p.W % 16 != 0 || p.C % 64 != 0 || p.ups < 0 || p.ups > 1) return false;
  if (p.ups == 1 && (p.H % 2 != 0 || p.W % 2 != 0)) return false;
  if ((size_t)(p.H >> p.ups) * (p.W >> p.ups) * p.C * 2 >= (1ull << 31) || (size_t)p.O * p.ldw * 2 >= (1ull << 31)) return false;
  if (p.x2 && (p.C2 % 64 != 0 || (size_t)p.H * p.W * p.C2 * 2 >= (1ull << 31) || p.ups != 0)) return false;
  if (p.ldw % 8 != 0 || p.ldw < 9 * p.C + (p.x2 ? p.C2 : 0)) return false;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!al16(p.x) || !al16(p.x2) || !al16(p.w) || !al16(p.gn_ss)) return false;
  if (img) return p.O >= 1 && p.O <= 4 && p.out_channels == p.O && p.res == nullptr && p.x2 == nullptr && p.stats_out == nullptr;
  if (p.O % 128 != 0 || p.ldy % 8 != 0 || !al16(p.y) || !al16(p.bias) || !al16(p.bias2)) return false;
  if (p.res && (p.ldr % 8 != 0 || !al16(p.res))) return false;
  if (p.stats_out && (p.G_out <= 0 || p.O % p.G_out != 0 || 128 % (p.O / p.G_out) != 0 || (p.O / p.G_out) > 64)) return false;
  return true;
}

int dk_launch_conv_halo(const ConvHaloParams& p, hipStream_t stream) {
  const bool img = p.img != nullptr || p.u8 != nullptr || p.raw != nullptr;
  if (!img && dk_conv256v4_wanted(p)) return dk_launch_conv256v4(p, stream);  // the one-wave-per-SIMD frame (conv256v4.hip): 256-column tiles
  DK_REQUIRE(dk_conv_halo_eligible(p, img), "conv_halo: shape / alignment not supported (H, W multiples of 16; C multiple of 64; O multiple of 128, or <= 4 for the image tail)");
  static DkDeviceOnce attr_once;
  constexpr int LDS128 = 2 * CH_A_SLOT + 3 * 128 * 128 + 8192, LDS16 = 2 * CH_A_SLOT + 3 * 16 * 128 + 8192;  // (+ the dummy store zone)
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_conv_halo_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_conv_halo_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS16));
    attr_once.mark();
  }
  const long tiles = (long)p.B * (p.H >> 4) * (p.W >> 4);
  const double flops = 2.0 * p.B * p.H * p.W * (9.0 * p.C + (p.x2 ? p.C2 : 0)) * p.O;
  dk_prof_begin(1, flops, stream);
  if (img)
    hipLaunchKernelGGL((dk_conv_halo_kernel<16, true>), dim3((unsigned)tiles), dim3(512), LDS16, stream, p);
  else
    hipLaunchKernelGGL((dk_conv_halo_kernel<128, false>), dim3((unsigned)(tiles * (p.O / 128))), dim3(512), LDS128, stream, p);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
