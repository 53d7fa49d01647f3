// LAB (not built): joint text/image attention forward on v_mfma_f32_16x16x32_bf16 (dk_attn4_fwd_kernel), head dim 128.
// Round 2 result: correct (it passed every attention parity test as tune mode 9, incl. the fused QKNorm + RoPE query load at model
// level), 256 registers without a spill -- and 4 % slower than dk_attn3_fwd_kernel in isolation (878 against 913 TF on the FLUX
// shape), 7 % slower inside the model (750 against 804 TF): twice as many MFMA issues, four cross-lane reductions per tile instead
// of one, more s_nop / s_waitcnt between the two instruction streams under hipcc's scheduler.  Kept for the operand layouts, which
// a hand-scheduled version would reuse.  To build: add it to csrc/Makefile with attention3's flags and a dispatch case.
//
// Same algorithm, software pipeline and LDS images as dk_attn3_fwd_kernel (attention3.hip; reference call sites
// python/src/diffusionkit/mlx/mmdit.py:562,643,687,736): a wave owns 32 queries, carries two 64-key tiles in flight
// (exponentials of tile j beside the score MFMAs of tile j+1, P.V of tile j beside the row maxima of tile j+1), online softmax
// with the deferred rescale (threshold 4).  What changes is the matrix instruction: the GEMM's K loop gained 5-12 % from the
// 16x16x32 form at equal FLOPs (less power per FLOP on a power-bound chip), so the tiles here are 16 wide:
//   S^T (16 keys x 16 queries) = K (16 keys x 32 d) . Q^T (32 d x 16 queries)        4 key blocks x 2 query blocks x 4 d steps
//   O^T (16 d x 16 queries)   += V^T (16 d x 32 keys) . P^T (32 keys x 16 queries)   8 d blocks x 2 query blocks x 2 key steps
// Operand layouts (lane l: r = l & 15, g = l >> 4): A row r, k = 8g..8g+7; B column r, k = 8g..8g+7; C column r, rows 4g..4g+3.
// So a lane owns query 16 qb + r of both query blocks (two softmax states per lane); of a score tile it holds keys 4g..4g+3 of
// every 16-key block, row maxima / sums finish with two cross-lane steps (lanes r, r+16, r+32, r+48: v_permlane16/32_swap).
// The probabilities a lane holds of two adjacent key blocks -- keys {4g..4g+3} and {16+4g..16+4g+3} of a 32-key step -- ARE
// its B-operand elements k = 8g..8g+7 if the contraction index is read as that permutation of the keys; the V^T operand uses
// the same permutation: two ds_read_b64_tr_b16 of 4 keys x 16 d each (16 lanes of one g), from attention3's V image unchanged.
#include "dk_kernels.h"

#define DK4_RESCALE_THR 4.0f  // natural-log units of the scaled scores

template <int NW>
struct Attn4Cfg {
  static constexpr int D = 128;
  static constexpr int KV = 64;
  static constexpr int ROWB = D * 2;
  static constexpr int TILE_BYTES = KV * D * 2;
  static constexpr int NT = NW * 64;
  static constexpr int NCHUNK = KV * D / 8;  // 16-byte chunks per K (or V) tile
  static constexpr int NCH = NCHUNK / NT;    // per thread
  static constexpr int CPR = D / 8;
  static constexpr int QB = NW * 32;
  static constexpr int Q_OFF = 4 * TILE_BYTES;                      // Q fragments: wave-private lane-linear image, 8 x 1 KiB
  static constexpr int LDS_BYTES = 4 * TILE_BYTES + NW * 32 * ROWB;  // K[2] V[2] Q
  static_assert(NCHUNK % NT == 0, "tile chunks must divide over the workgroup");
};

typedef __attribute__((address_space(3))) char lds_char4;

__device__ __forceinline__ float a4_max_g(float v) {  // maximum over the four lanes r, r+16, r+32, r+48
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float a4_sum_g(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int NW, bool QFUSE>
__global__ __launch_bounds__(NW * 64, 2) void dk_attn4_fwd_kernel(AttnParams p) {
  using C = Attn4Cfg<NW>;
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char4*)smem != 0u) __builtin_trap();  // LDS addressed from 0: offsets fold into instruction immediates
  lds_char4* const lds = (lds_char4*)0;
  constexpr int K_OFF = 0, V_OFF = 2 * C::TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int S = p.S;

  const int nq = (S + C::QB - 1) / C::QB;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int qb_ = t % nq, head = (t / nq) % p.H, b = t / (nq * p.H);
  const int q0 = qb_ * C::QB + wave * 32;

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + (size_t)b * S * p.ld + head * D);  // wave-uniform bases
  const char* Vb = (const char*)(p.V + (size_t)b * S * p.ld + head * D);
  const unsigned row_bytes = (unsigned)p.ld * 2u;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + 16 qb + r][32 ks + 8 g .. +7], optionally normalised / rotated here
  // (same fp32 arithmetic and bf16 rounding points as dk_qk_norm_rope_kernel); they live in a lane-linear LDS image
  const unsigned q_lds = C::Q_OFF + wave * (32 * C::ROWB) + lane * 16;  // + (qb * 4 + ks) * 1024
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = min(q0 + 16 * qb + r, S - 1);
    const bf16_t* qp = Qb + (size_t)qrow * p.ld + g * 8;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
    if (QFUSE) {
      float v[4][8];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[ks][e] = (float)qf[ks][e];
      if (p.qn_a != nullptr) {
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += v[ks][e] * v[ks][e];
        ss = a4_sum_g(ss);
        const float rs = rsqrtf(ss / (float)D + p.qn_eps);
        const bf16_t* w = (qrow < p.qn_split ? p.qn_a : p.qn_b) + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 wv = *(const bf16x8*)(w + ks * 32);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[ks][e] = round_bf16(v[ks][e] * rs * (float)wv[e]);
        }
      }
      if (p.q_rope != nullptr) {
        const float* tab = p.q_rope + ((size_t)qrow * (D / 2) + g * 4) * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f32x4 t0 = *(const f32x4*)(tab + ks * 32), t1 = *(const f32x4*)(tab + ks * 32 + 4);
          const float cs[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float c = cs[2 * i], sn = cs[2 * i + 1], xe = v[ks][2 * i], xo = v[ks][2 * i + 1];
            v[ks][2 * i] = c * xe - sn * xo;
            v[ks][2 * i + 1] = sn * xe + c * xo;
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)v[ks][e];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) *(__attribute__((address_space(3))) bf16x8*)(lds + q_lds + (qb * 4 + ks) * 1024) = qf[ks];
  }

  // ---- per-thread constants: staging chunk coordinates, global lane offsets, LDS offsets (K / V images as dk_attn3_fwd_kernel) ----
  unsigned g_off[C::NCH], ks_off[C::NCH], vs_off[C::NCH];
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int id = tid + C::NT * i;
    const int kl = id / C::CPR, c8 = id % C::CPR;
    g_off[i] = (unsigned)kl * row_bytes + (unsigned)c8 * 16u;
    ks_off[i] = (unsigned)(kl * C::ROWB + ((c8 ^ (kl & 15)) << 4));
    vs_off[i] = (unsigned)((c8 >> 1) * 2048 + (kl ^ ((((c8 >> 1) & 1) << 2) | ((c8 >> 1) & 3))) * 32 + (c8 & 1) * 16);
  }
  // K fragment (key block kb, d step ks): row 16 kb + r, chunk (4 ks + g) ^ r  =  kr_base ^ (ks << 6), + kb * 4096
  const unsigned kr_base = (unsigned)(r * C::ROWB + ((g ^ r) << 4));
  // V^T fragment (d block db, key step ks): 4 keys x 16 d per 16 lanes of one g: key row (32 ks + 4 g + (r >> 2)) ^ s(db) of sub-tile db,
  // d quad r & 3; the second read 16 keys further
  const unsigned vr_base = (unsigned)((4 * g + (r >> 2)) * 32 + (r & 3) * 8);

  u32x4 kreg[C::NCH], vreg[C::NCH];
  const int ntiles = (S + 63) / 64;
  const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, -1, 0x00020000);
  auto load_op = [&](const __amdgpu_buffer_rsrc_t rs, u32x4* reg, int jt, bool full) {
    const int soff = jt * 64 * (int)row_bytes;
    if (full) {
#pragma unroll
      for (int i = 0; i < C::NCH; ++i) reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)g_off[i], soff, 0);
    } else {  // tail tile: rows beyond S - 1 re-read the last key (their scores are masked)
#pragma unroll
      for (int i = 0; i < C::NCH; ++i) {
        const int id = tid + C::NT * i;
        const int kl0 = id / C::CPR, kl = min(kl0, S - 1 - jt * 64);
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((unsigned)kl * row_bytes + (unsigned)(id % C::CPR) * 16u), soff, 0);
      }
    }
  };
#define DK4_STORE_K(SLOT) \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) *(__attribute__((address_space(3))) u32x4*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + ks_off[i]) = kreg[i];
#define DK4_STORE_V(SLOT) \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) *(__attribute__((address_space(3))) u32x4*)(lds + V_OFF + (SLOT) * C::TILE_BYTES + vs_off[i]) = vreg[i];
// S^T of one tile from K slot SLOT: SX[kb][qb] += K(kb, ks) . Q(qb, ks)
#define DK4_QK(SLOT, SX)                                                                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                                          \
    const bf16x8 qa_ = *(const __attribute__((address_space(3))) bf16x8*)(lds + q_lds + ks * 1024);                                            \
    const bf16x8 qb2_ = *(const __attribute__((address_space(3))) bf16x8*)(lds + q_lds + (4 + ks) * 1024);                                     \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                                                                        \
      const bf16x8 kf_ = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + kb * 4096 + (kr_base ^ (unsigned)(ks << 6))); \
      SX[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf_, qa_, SX[kb][0], 0, 0, 0);                                                      \
      SX[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf_, qb2_, SX[kb][1], 0, 0, 0);                                                     \
    }                                                                                                                                         \
  }
// scores of keys beyond the sequence end (tail tile JT) -> -1e30
#define DK4_MASK(JT, SX)                                                                  \
  _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
    if ((JT) * 64 + 16 * kb + 4 * g + e >= S) { SX[kb][0][e] = -1e30f; SX[kb][1][e] = -1e30f; }    \
  }
#define DK4_ROWMAX(SX, OUT)                                                                                     \
  _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                            \
    float m_ = fmaxf(fmaxf(SX[0][qb][0], SX[0][qb][1]), fmaxf(SX[0][qb][2], SX[0][qb][3]));                      \
    _Pragma("unroll") for (int kb = 1; kb < 4; ++kb) _Pragma("unroll") for (int e = 0; e < 4; ++e) m_ = fmaxf(m_, SX[kb][qb][e]); \
    OUT[qb] = a4_max_g(m_);                                                                                     \
  }
// the rare rescale: every accumulator still at the old maxima (O, l) exactly once; nothing else is pending
#define DK4_RESCALE(MLOC)                                                                                                  \
  if (!__all((MLOC)[0] - m_run[0] <= thr && (MLOC)[1] - m_run[1] <= thr)) {                                                \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                                     \
      const float m_new_ = fmaxf(m_run[qb], (MLOC)[qb]);                                                                   \
      const float alpha_ = __builtin_amdgcn_exp2f((m_run[qb] - m_new_) * c);                                               \
      m_run[qb] = m_new_;                                                                                                  \
      l_run[qb] *= alpha_;                                                                                                 \
      _Pragma("unroll") for (int db = 0; db < 8; ++db) _Pragma("unroll") for (int e = 0; e < 4; ++e) o[db][qb][e] *= alpha_; \
    }                                                                                                                      \
  }
// region A's VALU half: S(j) -> P(j) as four bf16 B-operand fragments pf[qb][key step], row sums
#define DK4_SOFTMAX(SX)                                                                                        \
  _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                           \
    const float mc_ = m_run[qb] * c;                                                                           \
    float psum_ = 0.f;                                                                                         \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) _Pragma("unroll") for (int e = 0; e < 4; ++e) {            \
      SX[kb][qb][e] = __builtin_amdgcn_exp2f(SX[kb][qb][e] * c - mc_);                                         \
      psum_ += SX[kb][qb][e];                                                                                  \
    }                                                                                                          \
    l_run[qb] += psum_;                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int e = 0; e < 4; ++e) {           \
      pf[qb][ks][e] = (__bf16)SX[2 * ks][qb][e];                                                               \
      pf[qb][ks][4 + e] = (__bf16)SX[2 * ks + 1][qb][e];                                                       \
    }                                                                                                          \
  }
// region B's MFMA half: O += V(slot) P
#define DK4_PV(SLOT)                                                                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int db = 0; db < 8; ++db) {                                        \
    const unsigned va_ = (unsigned)(V_OFF + (SLOT) * C::TILE_BYTES + db * 2048 + ks * 1024) + (vr_base ^ (unsigned)(((((db & 1) << 2) | (db & 3))) << 5)); \
    const s16x4 vh0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + va_));                       \
    const s16x4 vh1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + va_ + 512));                 \
    const bf16x8 vf_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0_, vh1_, 0, 1, 2, 3, 4, 5, 6, 7));                               \
    o[db][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf_, pf[0][ks], o[db][0], 0, 0, 0);                                                   \
    o[db][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf_, pf[1][ks], o[db][1], 0, 0, 0);                                                   \
  }
#define DK4_ZERO(SX) _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) _Pragma("unroll") for (int e = 0; e < 4; ++e) { SX[kb][0][e] = 0.f; SX[kb][1][e] = 0.f; }

  f32x4 o[8][2];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[db][0][e] = 0.f, o[db][1][e] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  const float c = p.scale * 1.44269504088896340736f;  // p = 2^(s*c - m*c)
  const float thr = DK4_RESCALE_THR / p.scale;         // threshold on the raw scores
  bf16x8 pf[2][2];
  f32x4 sa[4][2], sb[4][2];  // scores of the tile being exponentiated / of the tile after it (the roles alternate per tile)

  // ---- prologue: K(0), V(0), K(1) staged; S(0) and its row maxima ----
  load_op(rK, kreg, 0, 64 <= S);
  load_op(rV, vreg, 0, 64 <= S);
  DK4_STORE_K(0)
  DK4_STORE_V(0)
  if (ntiles > 1) {
    load_op(rK, kreg, 1, 128 <= S);
    DK4_STORE_K(1)
  }
  __syncthreads();
  DK4_ZERO(sa)
  DK4_QK(0, sa)
  if (64 > S) { DK4_MASK(0, sa) }
  {
    float mloc[2];
    DK4_ROWMAX(sa, mloc)
    DK4_RESCALE(mloc)
  }

  // One tile.  CUR / NXT: score registers of tile j / j+1; slots: V(j) in j & 1, K(j+1) in (j+1) & 1; the loads fetch K(j+2) and
  // V(j+1) and store them into K slot j & 1 and V slot (j+1) & 1 (both last read in the previous iteration).
  // HAVE_N: tile j+1 exists; LOADS: 0 none, 1 full tiles (steady state: no row clamp, no branches), 2 generic (existence and tail checks).
#define DK4_TILE(J, PAR, CUR, NXT, HAVE_N, LOADS)                                                              \
  {                                                                                                            \
    const int j_ = (J);                                                                                        \
    bool have_k2_ = false, have_v1_ = false;                                                                   \
    if ((LOADS) == 1) {                                                                                        \
      load_op(rK, kreg, j_ + 2, true);                                                                         \
      load_op(rV, vreg, j_ + 1, true);                                                                         \
      have_k2_ = have_v1_ = true;                                                                              \
    } else if ((LOADS) == 2) {                                                                                 \
      have_k2_ = j_ + 2 < ntiles;                                                                              \
      have_v1_ = j_ + 1 < ntiles;                                                                              \
      if (have_k2_) load_op(rK, kreg, j_ + 2, (j_ + 3) * 64 <= S);                                             \
      if (have_v1_) load_op(rV, vreg, j_ + 1, (j_ + 2) * 64 <= S);                                             \
    }                                                                                                          \
    if (HAVE_N) { DK4_ZERO(NXT) }                                                                              \
    /* region A: exponentials of tile j || scores of tile j+1 (independent streams, one basic block) */        \
    DK4_SOFTMAX(CUR)                                                                                           \
    if (HAVE_N) { DK4_QK((PAR) ^ 1, NXT) }                                                                     \
    if ((HAVE_N) && (LOADS) != 1) {                                                                            \
      if ((j_ + 2) * 64 > S) { DK4_MASK(j_ + 1, NXT) }                                                         \
    }                                                                                                          \
    /* region B: P.V of tile j || row maxima of tile j+1 */                                                    \
    float mloc_[2] = {-1e30f, -1e30f};                                                                         \
    DK4_PV(PAR)                                                                                                \
    if (HAVE_N) { DK4_ROWMAX(NXT, mloc_) }                                                                     \
    if (have_k2_) { DK4_STORE_K(PAR) }                                                                         \
    if (have_v1_) { DK4_STORE_V((PAR) ^ 1) }                                                                   \
    if (HAVE_N) { DK4_RESCALE(mloc_) }                                                                         \
    __syncthreads();                                                                                           \
  }

  int j = 0;
  const int n_full = S / 64;  // tiles 0 .. n_full - 1 are complete
  for (; j + 3 < n_full; j += 2) {  // needs K(j+3), V(j+2) full for the second body: j + 3 <= n_full - 1
    DK4_TILE(j, 0, sa, sb, true, 1)
    DK4_TILE(j + 1, 1, sb, sa, true, 1)
  }
  for (; j < ntiles; j += 2) {  // remaining tiles (at most 4 + the tail): generic bodies; j is even here
    if (j + 1 < ntiles) {
      DK4_TILE(j, 0, sa, sb, true, 2)
      if (j + 2 < ntiles) {
        DK4_TILE(j + 1, 1, sb, sa, true, 2)
      } else {
        DK4_TILE(j + 1, 1, sb, sa, false, 0)
      }
    } else {
      DK4_TILE(j, 0, sa, sb, false, 0)
    }
  }
#undef DK4_TILE
#undef DK4_STORE_K
#undef DK4_STORE_V
#undef DK4_QK
#undef DK4_MASK
#undef DK4_ROWMAX
#undef DK4_RESCALE
#undef DK4_SOFTMAX
#undef DK4_PV
#undef DK4_ZERO

  // ---- normalise and store: lane owns queries q0 + 16 qb + r, d = 16 db + 4 g + {0..3} ----
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float inv = 1.0f / a4_sum_g(l_run[qb]);
    const int q = q0 + 16 * qb + r;
    if (q < S) {
      bf16_t* op = p.O + ((size_t)b * S + q) * p.ldo + head * D + 4 * g;
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        uint2 w;
        w.x = pack2bf(o[db][qb][0] * inv, o[db][qb][1] * inv);
        w.y = pack2bf(o[db][qb][2] * inv, o[db][qb][3] * inv);
        *(uint2*)(op + db * 16) = w;
      }
    }
  }
}

template <int NW, bool QFUSE>
static int launch_attn4(const AttnParams& p, hipStream_t stream) {
  using C = Attn4Cfg<NW>;
  static bool attr_set = false;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn4_fwd_kernel<NW, QFUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_set = true;
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  hipLaunchKernelGGL((dk_attn4_fwd_kernel<NW, QFUSE>), dim3(nq * p.H * p.B), dim3(C::NT), C::LDS_BYTES, stream, p);
  return 0;
}

// head dim 128, 8 waves per workgroup, no score bias, no MX-fp8 output (dk_launch_attention quantises behind it)
int dk_launch_attention4(const AttnParams& p, hipStream_t stream) {
  DK_REQUIRE(p.bias == nullptr && p.D == 128, "attention4: head dim 128, no score bias");
  DK_REQUIRE((size_t)p.S * p.ld * 2 < (1ull << 32), "attention4: one batch row of QKV must span < 4 GiB");
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  return qfuse ? launch_attn4<8, true>(p, stream) : launch_attn4<8, false>(p, stream);
}
