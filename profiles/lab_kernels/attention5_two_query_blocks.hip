// LAB KERNEL, not part of libdk_hip.so (round 4).  Built, parity-tested on the GPU (oracle: rel-L2 < 6e-3 on nine shapes up to
// B 2 x S 4685; against the lean kernel: max 2.4e-4 = rare bf16 flips of a probability) and measured: +1.3 % on the SD3-medium
// shape, +3 % on SD3.5-large, +17 % at B 1 (grid rounds), -13 % at S = 1613 in isolation (profiles/r04_attention_d64_two_blocks.log);
// inside the model +0.4 % (SD3-medium) / -0.9 % (SD3.5-large) (profiles/r04_attention_d64_in_model.log).  The hypothesis it tested --
// the D = 64 kernel is bound by LDS bytes per MFMA -- is refuted: halving them changes nothing; DESIGN.md section 7.
// To rebuild: copy to diffusionkit_amd/csrc/attention5.hip, add it to the Makefile (flags of attention4.o) and a case to attention.hip.
// Joint text/image attention forward for head_dim 64 with TWO 32-query blocks per wave (dk_attn5_fwd_kernel): the SD3 family
// (mmdit.py:608-625,643; 24 / 38 heads of 64 -- config.py:19-74).
//
// Same algorithm, layouts and MFMA operand mapping as dk_attn2_fwd_kernel (attention2.hip): transposed scores S^T = K Q^T on
// v_mfma_f32_32x32x16_bf16, lane-local online softmax with the deferred rescale (threshold 4), O^T += V^T P^T with V through
// ds_read_b64_tr_b16, K / V staged through two LDS slots each.
//
// Why (round 4, profiles/r04_sd3_pmc.md): at D = 64 a 64-key tile is 16 MFMAs per 32-query block -- half of D = 128's -- for the
// same softmax arithmetic and the same K / V tile through the LDS.  Counted per workgroup and tile of the 4-wave lean kernel:
// 16 KiB of staging stores (~ 79 B / clk) + 4 x 16 KiB of fragment reads (256 B / clk) = ~ 460 LDS cycles against 512 matrix-pipe
// cycles per SIMD; the counters show the matrix pipe 41 % busy and the VALU ~ 43 % busy with the two barely overlapping.  The
// lever at D = 64 is LDS bytes per MFMA, not VALU instructions: here every K and V fragment a wave reads feeds TWO MFMAs (one per
// query block), and a workgroup of NW waves stages a tile for 64 NW queries instead of 32 NW -- half the fragment reads and half
// the staging stores per FLOP; per lane the two blocks are two independent softmax streams (more ILP for the in-order issue).
// Registers: 64 (O) + 64 (S) + 32 (Q) + staging + fragments: two waves per SIMD.
#include "dk_kernels.h"

#define DK5_RESCALE_THR 4.0f  // natural-log units of the scaled scores

template <int NW>
struct Attn5Cfg {
  static constexpr int D = 64, KV = 64, QPW = 2;
  static constexpr int ROWB = D * 2;
  static constexpr int TILE_BYTES = KV * D * 2;
  static constexpr int NT = NW * 64;
  static constexpr int NCHUNK = KV * D / 8;  // 16-byte chunks per K (or V) tile
  static constexpr int NCH = NCHUNK / NT;    // per thread
  static constexpr int CPR = D / 8;
  static constexpr int QB = NW * 32 * QPW;
  static constexpr int LDS_BYTES = 4 * TILE_BYTES;  // K[2] V[2]
};

typedef __attribute__((address_space(3))) char lds_char5;

__device__ __forceinline__ float dk5_max_halves(float x) {  // max over a lane and the lane 32 away (one VALU instruction)
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int NW, bool QFUSE>
__global__ __launch_bounds__(NW * 64, 2) void dk_attn5_fwd_kernel(AttnParams p) {
  using C = Attn5Cfg<NW>;
  constexpr int D = C::D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char5*)smem != 0u) __builtin_trap();  // LDS addressed from 0: offsets fold into instruction immediates
  lds_char5* const lds = (lds_char5*)0;
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
  const int q0 = qb * C::QB + wave * 64;  // block blk covers queries q0 + 32 blk + (0..31)

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + (size_t)b * S * p.ld + head * D);  // wave-uniform bases
  const char* Vb = (const char*)(p.V + (size_t)b * S * p.ld + head * D);
  const unsigned row_bytes = (unsigned)p.ld * 2u;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + 32 blk + l31][kk*16 + hi*8 .. +7]
  bf16x8 qf[2][D / 16];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int qrow = min(q0 + 32 * blk + l31, S - 1);
    const bf16_t* qp = Qb + (size_t)qrow * p.ld + hi * 8;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) qf[blk][kk] = *(const bf16x8*)(qp + kk * 16);
    if (QFUSE) {
      // QKNorm + RoPE of this lane's query row on the fly (same fp32 arithmetic and bf16 rounding points as
      // dk_qk_norm_rope_kernel): the lane and its partner (lane ^ 32) hold the two halves of every 16-element group
      float v[D / 16][8];
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[kk][e] = (float)qf[blk][kk][e];
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
        for (int e = 0; e < 8; ++e) qf[blk][kk][e] = (__bf16)v[kk][e];
    }
  }

  // ---- per-thread constants: staging chunk coordinates, global lane offsets, LDS offsets (as dk_attn2_fwd_kernel<64>) ----
  unsigned g_off[C::NCH], ks_off[C::NCH], vs_off[C::NCH];
  int st_kl[C::NCH];
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int id = tid + C::NT * i;
    const int kl = id / C::CPR, c8 = id % C::CPR;
    st_kl[i] = kl;
    g_off[i] = (unsigned)kl * row_bytes + (unsigned)c8 * 16u;
    ks_off[i] = (unsigned)(kl * C::ROWB + ((c8 ^ ((kl >> 1) & 7)) << 4));
    // V image [d/16][key][16], key row kl of d-block b at row kl ^ f(b) (bank spreading, attention2.hip)
    vs_off[i] = (unsigned)((c8 >> 1) * 2048 + (kl ^ ((((c8 >> 1) & 1) << 2) | ((c8 >> 1) & 3))) * 32 + (c8 & 1) * 16);
  }
  unsigned kr_off[D / 16];  // K fragment read: row l31 (+32 per sub-tile as an immediate), swizzled chunk
#pragma unroll
  for (int kk = 0; kk < D / 16; ++kk) kr_off[kk] = (unsigned)(l31 * C::ROWB + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4));
  const int x16 = (lane >> 4) & 1, p16 = lane & 15;
  unsigned vr_off[2];
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
        kreg[i] = *(const u32x4*)(kb + g_off[i]);
        vreg[i] = *(const u32x4*)(vb + g_off[i]);
      }
    } else {  // tail tile: rows beyond S - 1 re-read the last key (their scores are masked below)
#pragma unroll
      for (int i = 0; i < C::NCH; ++i) {
        const int kl = min(st_kl[i], S - 1 - j * 64);
        const unsigned off = (unsigned)kl * row_bytes + (g_off[i] - (unsigned)st_kl[i] * row_bytes);
        kreg[i] = *(const u32x4*)(kb + off);
        vreg[i] = *(const u32x4*)(vb + off);
      }
    }
  };
#define DK5_STORE_TILE(BUF)                                                            \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) {                                 \
    *(__attribute__((address_space(3))) u32x4*)(lds + K_OFF + (BUF) * C::TILE_BYTES + ks_off[i]) = kreg[i]; \
    *(__attribute__((address_space(3))) u32x4*)(lds + V_OFF + (BUF) * C::TILE_BYTES + vs_off[i]) = vreg[i]; \
  }

  f32x16 o[2][D / 32];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[blk][i][e] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  const float c = p.scale * 1.44269504088896340736f;  // p = 2^(s*c - m*c)
  const float thr = DK5_RESCALE_THR / p.scale;         // threshold on the raw scores
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  DK5_STORE_TILE(0)
  __syncthreads();

  // the softmax of one block's 64 x 32 score tile, in place: s0 / s1 become the probabilities
#define DK5_SOFTMAX(BLK, S0, S1)                                                                                 \
  {                                                                                                              \
    float mx_ = fmaxf(S0[0], S1[0]);                                                                             \
    _Pragma("unroll") for (int e = 1; e < 16; ++e) mx_ = __builtin_fmaxf(__builtin_fmaxf(mx_, S0[e]), S1[e]);    \
    mx_ = dk5_max_halves(mx_);                                                                                   \
    if (!__all(mx_ - m_run[BLK] <= thr)) {                                                                       \
      const float m_new_ = fmaxf(m_run[BLK], mx_);                                                               \
      const float alpha_ = __builtin_amdgcn_exp2f((m_run[BLK] - m_new_) * c);                                    \
      m_run[BLK] = m_new_;                                                                                       \
      l_run[BLK] *= alpha_;                                                                                      \
      _Pragma("unroll") for (int i = 0; i < D / 32; ++i) _Pragma("unroll") for (int e = 0; e < 16; ++e) o[BLK][i][e] *= alpha_; \
    }                                                                                                            \
    const float mc_ = m_run[BLK] * c;                                                                            \
    float psum_ = 0.f;                                                                                           \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                             \
      S0[e] = __builtin_amdgcn_exp2f(S0[e] * c - mc_);                                                           \
      S1[e] = __builtin_amdgcn_exp2f(S1[e] * c - mc_);                                                           \
      psum_ += S0[e] + S1[e];                                                                                    \
    }                                                                                                            \
    l_run[BLK] += psum_;                                                                                         \
  }

#define DK5_TILE(BUF, J)                                                                                   \
  do {                                                                                                     \
    const int j_ = (J);                                                                                    \
    if (j_ + 1 < ntiles) load_tile(j_ + 1);                                                                \
    f32x16 sa0, sa1, sb0, sb1; /* block a / b, keys 0-31 / 32-63 */                                        \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int kk = 0; kk < D / 16; ++kk) {                                                \
      const bf16x8 k0 = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (BUF) * C::TILE_BYTES + kr_off[kk]); \
      const bf16x8 k1 = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (BUF) * C::TILE_BYTES + 32 * C::ROWB + kr_off[kk]); \
      sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[0][kk], kk == 0 ? zero16 : sa0, 0, 0, 0);       \
      sb0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[1][kk], kk == 0 ? zero16 : sb0, 0, 0, 0);       \
      sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[0][kk], kk == 0 ? zero16 : sa1, 0, 0, 0);       \
      sb1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[1][kk], kk == 0 ? zero16 : sb1, 0, 0, 0);       \
    }                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    if (j_ * 64 + 64 > S) {                                                                                \
      asm volatile("; tail tile" ::: "memory"); /* keeps hipcc from if-converting the mask into every tile */ \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                     \
        const int key = j_ * 64 + (e & 3) + 8 * (e >> 2) + 4 * hi;                                         \
        if (key >= S) { sa0[e] = -1e30f; sb0[e] = -1e30f; }                                                \
        if (key + 32 >= S) { sa1[e] = -1e30f; sb1[e] = -1e30f; }                                           \
      }                                                                                                    \
    }                                                                                                      \
    DK5_SOFTMAX(0, sa0, sa1)                                                                               \
    DK5_SOFTMAX(1, sb0, sb1)                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                        \
      _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                                   \
        bf16x8 pa, pb;                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                    \
          pa[e] = (__bf16)(u == 0 ? sa0[8 * tt + e] : sa1[8 * tt + e]);                                    \
          pb[e] = (__bf16)(u == 0 ? sb0[8 * tt + e] : sb1[8 * tt + e]);                                    \
        }                                                                                                  \
        _Pragma("unroll") for (int dt = 0; dt < D / 32; ++dt) {                                            \
          const int imm = V_OFF + (BUF) * C::TILE_BYTES + dt * 4096 + (32 * u + 16 * tt) * 32;             \
          const s16x4 vh0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm + vr_off[dt & 1])); \
          const s16x4 vh1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm + 256 + vr_off[dt & 1])); \
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7)); \
          o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pa, o[0][dt], 0, 0, 0);                   \
          o[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[1][dt], 0, 0, 0);                   \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    if (j_ + 1 < ntiles) { DK5_STORE_TILE((BUF) ^ 1) }                                                     \
    __syncthreads();                                                                                       \
  } while (0)

  int j = 0;
  for (; j + 1 < ntiles; j += 2) {
    DK5_TILE(0, j);
    DK5_TILE(1, j + 1);
  }
  if (j < ntiles) DK5_TILE(0, j);
#undef DK5_TILE
#undef DK5_SOFTMAX
#undef DK5_STORE_TILE

  // ---- normalise and store: lane owns query q0 + 32 blk + l31, d = dt*32 + 8g + 4hi + {0..3} ----
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const float inv = 1.0f / (l_run[blk] + __shfl_xor(l_run[blk], 32, 64));
    const int q = q0 + 32 * blk + l31;
    if (q < S) {
      bf16_t* op = p.O + ((size_t)b * S + q) * p.ldo + head * D;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          uint2 w;
          w.x = pack2bf(o[blk][dt][4 * g4 + 0] * inv, o[blk][dt][4 * g4 + 1] * inv);
          w.y = pack2bf(o[blk][dt][4 * g4 + 2] * inv, o[blk][dt][4 * g4 + 3] * inv);
          *(uint2*)(op + dt * 32 + 8 * g4 + 4 * hi) = w;
        }
    }
  }
}

template <int NW, bool QFUSE>
static int launch_attn5(const AttnParams& p, hipStream_t stream) {
  using C = Attn5Cfg<NW>;
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn5_fwd_kernel<NW, QFUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_once.mark();
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  hipLaunchKernelGGL((dk_attn5_fwd_kernel<NW, QFUSE>), dim3(nq * p.H * p.B), dim3(C::NT), C::LDS_BYTES, stream, p);
  return 0;
}

// D = 64, no score bias; waves = 4 (256 queries per workgroup) or 8 (512)
int dk_launch_attention5(const AttnParams& p, int waves, hipStream_t stream) {
  DK_REQUIRE(p.bias == nullptr, "attention5: no score-bias variant");
  DK_REQUIRE(p.D == 64, "attention5: head_dim 64");
  DK_REQUIRE(waves == 4 || waves == 8, "attention5: 4 or 8 waves per workgroup");
  DK_REQUIRE((size_t)p.S * p.ld * 2 < (1ull << 32), "attention5: one batch row of QKV must span < 4 GiB");
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  if (waves == 4) return qfuse ? launch_attn5<4, true>(p, stream) : launch_attn5<4, false>(p, stream);
  return qfuse ? launch_attn5<8, true>(p, stream) : launch_attn5<8, false>(p, stream);
}
