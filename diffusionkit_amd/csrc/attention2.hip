// Joint text/image attention forward, VALU-lean variant (dk_attn2_fwd_kernel).
//
// Same algorithm, layouts and MFMA operand mapping as dk_attn_fwd_kernel (attention.hip; reference
// call sites python/src/diffusionkit/mlx/mmdit.py:562,643,687,736) -- transposed scores S^T = K Q^T,
// lane-local online softmax, O^T += V^T P^T with V read through ds_read_b64_tr_b16 -- but the per-tile
// instruction stream is cut down, because rocprofv3 counters showed the first kernel VALU-bound
// (13 VALU per MFMA, VALU busy 59 %, MFMA busy 33 %; profiles/archive/r01_attention_pmc.md):
//   * K/V tile loads use ONE 32-bit lane offset per 16-byte chunk against a wave-uniform base that
//     advances per tile (global_load saddr form): no per-tile address arithmetic, the key clamp is
//     evaluated only in the tail tile;
//   * every LDS offset (swizzled K fragments, V transpose-read base, staging stores) is computed
//     once per kernel; the tile body only adds compile-time immediates (the loop is unrolled over
//     the two LDS buffers for that);
//   * the running-max rescale of O (64 multiplies) runs only when some query of the wave raised its
//     maximum by more than DK_RESCALE_THR (wave-uniform vote); probabilities are then bounded by
//     e^THR instead of 1, which fp32 accumulation and bf16 P tolerate (guide T13; the safe order is
//     kept: the decision precedes the exponentials of the tile it covers, nothing else is pending);
//   * the two 32-key score chains of a tile are interleaved (two independent accumulators).
#include "dk_kernels.h"

#define DK_RESCALE_THR 4.0f  // natural-log units of the scaled scores

template <int D, int NW>
struct Attn2Cfg {
  static constexpr int KV = 64;
  static constexpr int ROWB = D * 2;
  static constexpr int TILE_BYTES = KV * D * 2;
  static constexpr int NT = NW * 64;
  static constexpr int NCHUNK = KV * D / 8;                 // 16-byte chunks per K (or V) tile
  static constexpr int NCH = (NCHUNK + NT - 1) / NT;        // per thread (the last one may be partial: NW = 7)
  static constexpr bool RAGGED = (NCHUNK % NT) != 0;
  static constexpr int CPR = D / 8;
  static constexpr int QB = NW * 32;
  static constexpr int LDS_BYTES = 4 * TILE_BYTES;  // K[2] V[2]
};

template <int D>
__device__ __forceinline__ int k2_swz(int r) { return D == 128 ? (r & 15) : ((r >> 1) & 7); }

typedef __attribute__((address_space(3))) char lds_char;

template <int D, int NW, bool HAS_BIAS = false, bool QFUSE = false>
__global__ __launch_bounds__(NW * 64, 2) void dk_attn2_fwd_kernel(AttnParams p) {
  using C = Attn2Cfg<D, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // The dynamic LDS region starts at LDS address 0 (this kernel has no static __shared__ objects), so
  // LDS accesses are formed from small integer addresses: hipcc then knows the address bits and
  // folds every compile-time offset into the ds instruction's immediate field instead of emitting a
  // v_add per access (it cannot prove base + offset stays non-negative for a symbol address).
  if ((unsigned)(size_t)(lds_char*)smem != 0u) __builtin_trap();
  lds_char* const lds = (lds_char*)0;
  constexpr int K_OFF = 0, V_OFF = 2 * C::TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = p.S;

  const int nq = (S + C::QB - 1) / C::QB;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int qb = t % nq, head = (t / nq) % p.H, b = t / (nq * p.H);
  const int q0 = qb * C::QB + wave * 32;

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + (size_t)b * S * p.ld + head * D);  // wave-uniform bases
  const char* Vb = (const char*)(p.V + (size_t)b * S * p.ld + head * D);
  const unsigned row_bytes = (unsigned)p.ld * 2u;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + l31][kk*16 + hi*8 .. +7]
  bf16x8 qf[D / 16];
  {
    const int qrow = min(q0 + l31, S - 1);
    const bf16_t* qp = Qb + (size_t)qrow * p.ld + hi * 8;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
    if (QFUSE) {
      // QKNorm + RoPE of this lane's query row on the fly (same fp32 arithmetic and bf16 rounding points as
      // dk_qk_norm_rope_kernel): the lane and its partner (lane ^ 32) hold the two halves of every 16-element group
      float v[D / 16][8];
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[kk][e] = (float)qf[kk][e];
      if (p.qn_a != nullptr) {
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk)
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += v[kk][e] * v[kk][e];
        ss += __shfl_xor(ss, 32, 64);
        const float r = rsqrtf(ss / (float)D + p.qn_eps);
        const bf16_t* w = (qrow < p.qn_split ? p.qn_a : p.qn_b) + hi * 8;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) {
          const bf16x8 wv = *(const bf16x8*)(w + kk * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[kk][e] = round_bf16(v[kk][e] * r * (float)wv[e]);
        }
      }
      if (p.q_rope != nullptr) {
        const float* tab = p.q_rope + ((size_t)qrow * (D / 2) + hi * 4) * 2;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) {
          const f32x4 t0 = *(const f32x4*)(tab + kk * 16), t1 = *(const f32x4*)(tab + kk * 16 + 4);
          const float cs[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float c = cs[2 * i], sn = cs[2 * i + 1], xe = v[kk][2 * i], xo = v[kk][2 * i + 1];
            v[kk][2 * i] = c * xe - sn * xo;
            v[kk][2 * i + 1] = sn * xe + c * xo;
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[kk][e] = (__bf16)v[kk][e];
    }
  }

  // ---- per-thread constants: staging chunk coordinates, global lane offsets, LDS offsets ----
  bool st_on[C::NCH];       // chunk exists (NT does not divide the chunk count for 7-wave workgroups)
  unsigned g_off[C::NCH];   // byte offset of chunk i inside a 64-key tile (key-local row, 16-byte column)
  unsigned ks_off[C::NCH];  // LDS store offset inside a K tile
  unsigned vs_off[C::NCH];  // LDS store offset inside a V tile
  int st_kl[C::NCH];
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int id0 = tid + C::NT * i;
    st_on[i] = !C::RAGGED || id0 < C::NCHUNK;
    const int id = st_on[i] ? id0 : 0;
    const int kl = id / C::CPR, c8 = id % C::CPR;
    st_kl[i] = kl;
    g_off[i] = (unsigned)kl * row_bytes + (unsigned)c8 * 16u;
    ks_off[i] = (unsigned)(kl * C::ROWB + ((c8 ^ k2_swz<D>(kl)) << 4));
    // V image: [d/16][key][16]; d-block b stores key row kl at row kl ^ f(b), f(b) = ((b & 1) << 2) | (b & 3): bit 2 puts the
    // two blocks a tr-read touches on different bank halves, bits 0-1 spread the four blocks one staging store writes
    // (same key, 2 KiB apart = the same banks) over the four 32-byte rows of a 128-byte bank period (it was a 4-way conflict)
    vs_off[i] = (unsigned)((c8 >> 1) * 2048 + (kl ^ ((((c8 >> 1) & 1) << 2) | ((c8 >> 1) & 3))) * 32 + (c8 & 1) * 16);
  }
  unsigned kr_off[D / 16];  // K fragment read: row l31 (+32 per sub-tile as an immediate), swizzled chunk
#pragma unroll
  for (int kk = 0; kk < D / 16; ++kk) kr_off[kk] = (unsigned)(l31 * C::ROWB + (((kk * 2 + hi) ^ k2_swz<D>(l31)) << 4));
  const int x16 = (lane >> 4) & 1, p16 = lane & 15;
  unsigned vr_off[2];  // by parity of the 32-wide d-tile dt: the lane reads block b = 2*dt + x16, f(b) & 3 = 2*(dt & 1) + x16
#pragma unroll
  for (int par = 0; par < 2; ++par)
    vr_off[par] = (unsigned)(x16 * 2048 + ((4 * (hi ^ x16) + (p16 >> 2)) ^ (2 * par + x16)) * 32 + (p16 & 3) * 8);

  u32x4 kreg[C::NCH], vreg[C::NCH];
  const int ntiles = (S + 63) / 64;
  auto load_tile = [&](int j) {
    const char* kb = Kb + (size_t)j * 64 * row_bytes;
    const char* vb = Vb + (size_t)j * 64 * row_bytes;
    if (j * 64 + 64 <= S) {
#pragma unroll
      for (int i = 0; i < C::NCH; ++i) {
        if (C::RAGGED && !st_on[i]) continue;
        kreg[i] = *(const u32x4*)(kb + g_off[i]);
        vreg[i] = *(const u32x4*)(vb + g_off[i]);
      }
    } else {  // tail tile: rows beyond S - 1 re-read the last key (their scores are masked below)
#pragma unroll
      for (int i = 0; i < C::NCH; ++i) {
        if (C::RAGGED && !st_on[i]) continue;
        const int kl = min(st_kl[i], S - 1 - j * 64);
        const unsigned off = (unsigned)kl * row_bytes + (g_off[i] - (unsigned)st_kl[i] * row_bytes);
        kreg[i] = *(const u32x4*)(kb + off);
        vreg[i] = *(const u32x4*)(vb + off);
      }
    }
  };
#define DK2_STORE_TILE(BUF)                                                            \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) {                                 \
    if (C::RAGGED && !st_on[i]) continue;                                              \
    *(__attribute__((address_space(3))) u32x4*)(lds + K_OFF + (BUF) * C::TILE_BYTES + ks_off[i]) = kreg[i]; \
    *(__attribute__((address_space(3))) u32x4*)(lds + V_OFF + (BUF) * C::TILE_BYTES + vs_off[i]) = vreg[i]; \
  }

  f32x16 o[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;  // p = 2^(s*c - m*c)
  const float thr = DK_RESCALE_THR / p.scale;          // threshold on the raw scores
  const float inv_scale = 1.0f / p.scale;
  const bf16_t* bias_row = HAS_BIAS ? p.bias + (size_t)head * p.bias_head_stride + (size_t)min(q0 + l31, S - 1) * p.ldb : nullptr;

  load_tile(0);
  DK2_STORE_TILE(0)
  __syncthreads();

#define DK2_TILE(BUF, J)                                                                                   \
  do {                                                                                                     \
    const int j_ = (J);                                                                                    \
    if (j_ + 1 < ntiles) load_tile(j_ + 1);                                                                \
    f32x16 s0, s1;                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }                           \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int kk = 0; kk < D / 16; ++kk) {                                                \
      const bf16x8 k0 = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (BUF) * C::TILE_BYTES + kr_off[kk]); \
      const bf16x8 k1 = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (BUF) * C::TILE_BYTES + 32 * C::ROWB + kr_off[kk]); \
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[kk], s0, 0, 0, 0);                               \
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[kk], s1, 0, 0, 0);                               \
    }                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    if (j_ * 64 + 64 > S) {                                                                                \
      asm volatile("; tail tile" ::: "memory"); /* keeps hipcc from if-converting the mask into every tile */ \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                     \
        const int key = j_ * 64 + (e & 3) + 8 * (e >> 2) + 4 * hi;                                         \
        if (key >= S) s0[e] = -1e30f;                                                                      \
        if (key + 32 >= S) s1[e] = -1e30f;                                                                 \
      }                                                                                                    \
    }                                                                                                      \
    if (HAS_BIAS) { /* scores += bias / scale (the exponent below multiplies by scale * log2 e) */            \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                   \
        const uint2 b0 = *(const uint2*)(bias_row + j_ * 64 + 8 * g4 + 4 * hi);                             \
        const uint2 b1 = *(const uint2*)(bias_row + j_ * 64 + 32 + 8 * g4 + 4 * hi);                        \
        float f0[4], f1[4];                                                                                \
        unpack2bf(b0.x, f0[0], f0[1]); unpack2bf(b0.y, f0[2], f0[3]);                                      \
        unpack2bf(b1.x, f1[0], f1[1]); unpack2bf(b1.y, f1[2], f1[3]);                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
          s0[4 * g4 + e] += f0[e] * inv_scale;                                                             \
          s1[4 * g4 + e] += f1[e] * inv_scale;                                                             \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    float mloc = fmaxf(s0[0], s1[0]);                                                                      \
    _Pragma("unroll") for (int e = 1; e < 16; ++e) mloc = fmaxf(mloc, fmaxf(s0[e], s1[e]));                \
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                          \
    if (!__all(mloc - m_run <= thr)) {                                                                     \
      const float m_new = fmaxf(m_run, mloc);                                                              \
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);                                     \
      m_run = m_new;                                                                                       \
      l_run *= alpha;                                                                                      \
      _Pragma("unroll") for (int i = 0; i < D / 32; ++i) _Pragma("unroll") for (int e = 0; e < 16; ++e) o[i][e] *= alpha; \
    }                                                                                                      \
    const float mc = m_run * c;                                                                            \
    float psum = 0.f;                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                       \
      s0[e] = __builtin_amdgcn_exp2f(s0[e] * c - mc);                                                      \
      s1[e] = __builtin_amdgcn_exp2f(s1[e] * c - mc);                                                      \
      psum += s0[e] + s1[e];                                                                               \
    }                                                                                                      \
    l_run += psum;                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                        \
      _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                                   \
        bf16x8 pf;                                                                                         \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) pf[e] = (__bf16)(u == 0 ? s0[8 * tt + e] : s1[8 * tt + e]); \
        _Pragma("unroll") for (int dt = 0; dt < D / 32; ++dt) {                                            \
          const int imm = V_OFF + (BUF) * C::TILE_BYTES + dt * 4096 + (32 * u + 16 * tt) * 32;             \
          const s16x4 vh0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm + vr_off[dt & 1])); \
          const s16x4 vh1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm + 256 + vr_off[dt & 1])); \
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7)); \
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);                         \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    if (j_ + 1 < ntiles) { DK2_STORE_TILE((BUF) ^ 1) }                                                     \
    __syncthreads();                                                                                       \
  } while (0)

  int j = 0;
  for (; j + 1 < ntiles; j += 2) {
    DK2_TILE(0, j);
    DK2_TILE(1, j + 1);
  }
  if (j < ntiles) DK2_TILE(0, j);
#undef DK2_TILE
#undef DK2_STORE_TILE

  // ---- normalise and store: lane owns query q0+l31, d = dt*32 + 8g + 4hi + {0..3} ----
  const float lsum = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / lsum;
  const int q = q0 + l31;
  if (q < S) {
    bf16_t* op = p.O + ((size_t)b * S + q) * p.ldo + head * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 w;
        w.x = pack2bf(o[dt][4 * g4 + 0] * inv, o[dt][4 * g4 + 1] * inv);
        w.y = pack2bf(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
        *(uint2*)(op + dt * 32 + 8 * g4 + 4 * hi) = w;
      }
  }
}

template <int D, int NW, bool HAS_BIAS = false, bool QFUSE = false>
static int launch_attn2(const AttnParams& p, hipStream_t stream) {
  using C = Attn2Cfg<D, NW>;
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn2_fwd_kernel<D, NW, HAS_BIAS, QFUSE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     C::LDS_BYTES));
    attr_once.mark();
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  hipLaunchKernelGGL((dk_attn2_fwd_kernel<D, NW, HAS_BIAS, QFUSE>), dim3(nq * p.H * p.B), dim3(C::NT), C::LDS_BYTES, stream, p);
  return 0;
}

#ifdef DK_ATTN2_DEFAULT_SCHED_TU
// attention2p.hip: this file compiled a second time, under LLVM's default machine scheduler, for the D = 128 instantiations -- the
// iterative-ilp strategy the D = 64 kernels are built under (Makefile) drives them into spills (5 registers with the fused query norm).
// (Round 4 lab, profiles/lab_kernels/attention2_pipelined.patch: a D = 64 form with the next tile's score MFMAs and this tile's P.V
// MFMAs interleaved with the exponentials by a sched_group_barrier pipeline -- parity-green, 200 registers = 2 waves per SIMD,
// 6 % SLOWER than the lean form on the SD3 shapes, profiles/r04_attention_d64_pipelined.log; not built.)
int dk_launch_attention2_d128(const AttnParams& p, hipStream_t stream) {  // (arguments checked by dk_launch_attention2)
  if (p.bias != nullptr) return launch_attn2<128, 4, true>(p, stream);
  if (p.qn_a != nullptr || p.q_rope != nullptr) return launch_attn2<128, 4, false, true>(p, stream);
  return launch_attn2<128, 4>(p, stream);
}
#else
int dk_launch_attention2_d128(const AttnParams& p, hipStream_t stream);  // attention2p.hip
int dk_launch_attention2(const AttnParams& p, int waves, hipStream_t stream) {
  DK_REQUIRE((size_t)p.S * p.ld * 2 < (1ull << 32), "attention2: one batch row of QKV must span < 4 GiB");
  if (p.bias != nullptr) {  // text encoders: D = 64, short sequences
    DK_REQUIRE(p.ldb % 64 == 0 && p.ldb >= p.S && ((uintptr_t)p.bias & 7) == 0 && p.bias_head_stride % 4 == 0,
               "attention bias: row stride must be a multiple of 64 >= S, 8-byte aligned");
    return p.D == 128 ? dk_launch_attention2_d128(p, stream) : launch_attn2<64, 4, true>(p, stream);
  }
  DK_REQUIRE(waves == 4, "attention2: 4 waves per workgroup (the 8- and 7-wave forms were pruned in round 3)");
  if (p.qn_a != nullptr || p.q_rope != nullptr) {  // query-side QKNorm / RoPE fused into the Q load (MMDiT call sites)
    DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
    return p.D == 128 ? dk_launch_attention2_d128(p, stream) : launch_attn2<64, 4, false, true>(p, stream);
  }
  return p.D == 128 ? dk_launch_attention2_d128(p, stream) : launch_attn2<64, 4>(p, stream);
}
#endif
