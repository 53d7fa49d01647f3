// Joint text/image attention forward, software-pipelined variant (dk_attn3_fwd_kernel).
//
// Same algorithm, layouts and MFMA operand mapping as dk_attn2_fwd_kernel (attention2.hip; reference call sites
// python/src/diffusionkit/mlx/mmdit.py:562,643,687,736): transposed scores S^T = K Q^T on v_mfma_f32_32x32x16_bf16,
// lane-local online softmax with the deferred rescale (threshold 4), O^T += V^T P^T with V through ds_read_b64_tr_b16.
//
// Why: in the second-generation kernel a wave's tile is one dependent chain -- 16 score MFMAs, then ~190 VALU instructions of
// softmax, then 16 P.V MFMAs -- so the matrix pipe only works when the OTHER wave of the SIMD happens to be in a different part of
// its chain, and the per-tile barrier keeps pulling the two into step (rocprofv3 PMC, profiles/r01_pmc_bench_v13.md: MFMA busy
// 34 %, 6.0 VALU per MFMA).  Here every wave carries two tiles in flight (guide T15):
//   region A   P(j) = exp2(S(j) c - m c), row sums, bf16 packing     (VALU, ~115 instructions)
//              || S(j+1) = K(j+1) Q^T                                (16 MFMAs + their 16 LDS fragment reads)
//   region B   O += V(j) P(j)                                        (16 MFMAs + 32 transpose reads)
//              || row maximum of S(j+1), vote on the rescale         (VALU, ~40 instructions)
// Both regions are straight-line code in ONE basic block (global loads of the tiles after next at its top, their LDS stores at
// its end), so hipcc's scheduler can interleave the independent MFMA and VALU streams; the rare rescale of O sits behind the
// block, after P.V(j) has completed -- the order T13 requires (decision and rescale after the pending tile's P.V, before the
// exponentials of the tile the new maximum covers).  K runs one tile ahead of V through the same two LDS slots each.
#include "dk_kernels.h"

#define DK3_RESCALE_THR 4.0f  // natural-log units of the scaled scores

template <int D, int NW>
struct Attn3Cfg {
  static constexpr int KV = 64;
  static constexpr int ROWB = D * 2;
  static constexpr int TILE_BYTES = KV * D * 2;
  static constexpr int NT = NW * 64;
  static constexpr int NCHUNK = KV * D / 8;  // 16-byte chunks per K (or V) tile
  static constexpr int NCH = NCHUNK / NT;    // per thread
  static constexpr int CPR = D / 8;
  static constexpr int QB = NW * 32;
  // D = 128: the Q fragments (32 registers) live in a wave-private LDS image instead of registers -- with two score tiles in
  // flight the register file (256 per wave at 2 waves per SIMD) does not hold them without spilling.  Every lane re-reads the 16
  // bytes it wrote, so the image is simply lane-linear per fragment ([kk][lane][16 B]: one address register + immediates,
  // conflict-free; 8 more ds_read_b128 per tile)
  static constexpr bool QLDS = D == 128;
  static constexpr int Q_OFF = 4 * TILE_BYTES;
  static constexpr int LDS_BYTES = 4 * TILE_BYTES + (QLDS ? NW * 32 * ROWB : 0);  // K[2] V[2] (+ Q per wave)
  static_assert(NCHUNK % NT == 0, "tile chunks must divide over the workgroup");
};

template <int D>
__device__ __forceinline__ int k3_swz(int r) { return D == 128 ? (r & 15) : ((r >> 1) & 7); }

typedef __attribute__((address_space(3))) char lds_char3;

template <int D, int NW, bool QFUSE>
__global__ __launch_bounds__(NW * 64, 2) void dk_attn3_fwd_kernel(AttnParams p) {
  using C = Attn3Cfg<D, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char3*)smem != 0u) __builtin_trap();  // LDS addressed from 0: offsets fold into instruction immediates
  lds_char3* const lds = (lds_char3*)0;
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

  // ---- per-thread constants: staging chunk coordinates, global lane offsets, LDS offsets (as dk_attn2_fwd_kernel) ----
  unsigned g_off[C::NCH];   // byte offset of chunk i inside a 64-key tile (key-local row, 16-byte column)
  unsigned ks_off[C::NCH];  // LDS store offset inside a K tile
  unsigned vs_off[C::NCH];  // LDS store offset inside a V tile
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int id = tid + C::NT * i;
    const int kl = id / C::CPR, c8 = id % C::CPR;
    g_off[i] = (unsigned)kl * row_bytes + (unsigned)c8 * 16u;
    ks_off[i] = (unsigned)(kl * C::ROWB + ((c8 ^ k3_swz<D>(kl)) << 4));
    vs_off[i] = (unsigned)((c8 >> 1) * 2048 + (kl ^ ((((c8 >> 1) & 1) << 2) | ((c8 >> 1) & 3))) * 32 + (c8 & 1) * 16);
  }
  // K fragment read: row l31 (+32 per sub-tile as an immediate), swizzled chunk (kk * 2 + hi) ^ swz(l31).  The chunk index of
  // fragment kk differs from fragment 0's by an XOR with 2 kk, so ONE lane-constant register serves all fragments (one v_xor with
  // a literal per fragment instead of D / 16 registers held across the loop)
  const unsigned kr_base = (unsigned)(l31 * C::ROWB + ((hi ^ k3_swz<D>(l31)) << 4));
  const int x16 = (lane >> 4) & 1, p16 = lane & 15;
  unsigned vr_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    vr_off[par] = (unsigned)(x16 * 2048 + ((4 * (hi ^ x16) + (p16 >> 2)) ^ (2 * par + x16)) * 32 + (p16 & 3) * 8);

  if (C::QLDS) {
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk)
      *(__attribute__((address_space(3))) bf16x8*)(lds + C::Q_OFF + wave * (32 * C::ROWB) + lane * 16 + kk * 1024) = qf[kk];
  }
  const unsigned q_lds = C::Q_OFF + wave * (32 * C::ROWB) + lane * 16;  // (same wave writes and reads: program order + lgkmcnt suffice)

  u32x4 kreg[C::NCH], vreg[C::NCH];
  const int ntiles = (S + 63) / 64;
  // one operand's 64-key tile jt -> registers through a buffer descriptor: one 32-bit lane offset per chunk (shared by K and V)
  // plus a scalar tile offset -- no 64-bit per-lane addresses.  full: the tile lies inside the sequence (no row clamp)
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
#define DK3_STORE_K(SLOT) \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) *(__attribute__((address_space(3))) u32x4*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + ks_off[i]) = kreg[i];
#define DK3_STORE_V(SLOT) \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) *(__attribute__((address_space(3))) u32x4*)(lds + V_OFF + (SLOT) * C::TILE_BYTES + vs_off[i]) = vreg[i];
// S^T of one tile from K slot SLOT into two independent 32-key accumulators
#define DK3_QK(SLOT, A0, A1)                                                                                                                   \
  _Pragma("unroll") for (int kk = 0; kk < D / 16; ++kk) {                                                                                      \
    const bf16x8 k0_ = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + (kr_base ^ (unsigned)(kk << 5)));    \
    const bf16x8 k1_ = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + 32 * C::ROWB + (kr_base ^ (unsigned)(kk << 5)));  \
    const bf16x8 q_ = C::QLDS ? *(const __attribute__((address_space(3))) bf16x8*)(lds + q_lds + kk * 1024) : qf[kk];                           \
    A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0_, q_, A0, 0, 0, 0);                                                                        \
    A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1_, q_, A1, 0, 0, 0);                                                                        \
  }
// scores of keys beyond the sequence end (tail tile JT) -> -1e30
#define DK3_MASK(JT, A0, A1)                                              \
  _Pragma("unroll") for (int e = 0; e < 16; ++e) {                        \
    const int key_ = (JT) * 64 + (e & 3) + 8 * (e >> 2) + 4 * hi;         \
    if (key_ >= S) A0[e] = -1e30f;                                        \
    if (key_ + 32 >= S) A1[e] = -1e30f;                                   \
  }
#define DK3_ROWMAX(A0, A1, OUT)                                                                          \
  do {                                                                                                   \
    float m_ = fmaxf(A0[0], A1[0]);                                                                      \
    _Pragma("unroll") for (int e = 1; e < 16; ++e) m_ = fmaxf(m_, fmaxf(A0[e], A1[e]));                  \
    OUT = fmaxf(m_, __shfl_xor(m_, 32, 64));                                                             \
  } while (0)
// the rare rescale: every accumulator still at the old maximum (O, l) exactly once; nothing else is pending
#define DK3_RESCALE(MLOC)                                                                                                  \
  if (!__all((MLOC) - m_run <= thr)) {                                                                                     \
    const float m_new_ = fmaxf(m_run, (MLOC));                                                                             \
    const float alpha_ = __builtin_amdgcn_exp2f((m_run - m_new_) * c);                                                     \
    m_run = m_new_;                                                                                                        \
    l_run *= alpha_;                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < D / 32; ++i) _Pragma("unroll") for (int e = 0; e < 16; ++e) o[i][e] *= alpha_;   \
  }
// region A's VALU half: S(j) -> P(j) as four bf16 B-operand fragments, row sums
#define DK3_SOFTMAX(A0, A1)                                                                                   \
  {                                                                                                           \
    const float mc_ = m_run * c;                                                                              \
    float psum_ = 0.f;                                                                                        \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                          \
      A0[e] = __builtin_amdgcn_exp2f(A0[e] * c - mc_);                                                        \
      A1[e] = __builtin_amdgcn_exp2f(A1[e] * c - mc_);                                                        \
      psum_ += A0[e] + A1[e];                                                                                 \
    }                                                                                                         \
    l_run += psum_;                                                                                           \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                           \
      pf[0][e] = (__bf16)A0[e]; pf[1][e] = (__bf16)A0[8 + e]; pf[2][e] = (__bf16)A1[e]; pf[3][e] = (__bf16)A1[8 + e]; \
    }                                                                                                         \
  }
// region B's MFMA half: O += V(slot) P
#define DK3_PV(SLOT)                                                                                                                        \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) _Pragma("unroll") for (int dt = 0; dt < D / 32; ++dt) { \
    const int imm_ = V_OFF + (SLOT) * C::TILE_BYTES + dt * 4096 + (32 * u + 16 * tt) * 32;                                                  \
    const s16x4 vh0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm_ + vr_off[dt & 1]));    \
    const s16x4 vh1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm_ + 256 + vr_off[dt & 1])); \
    const bf16x8 vf_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0_, vh1_, 0, 1, 2, 3, 4, 5, 6, 7));                             \
    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_, pf[2 * u + tt], o[dt], 0, 0, 0);                                                   \
  }
#define DK3_ZERO(A0, A1) _Pragma("unroll") for (int e = 0; e < 16; ++e) { A0[e] = 0.f; A1[e] = 0.f; }

  f32x16 o[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;  // p = 2^(s*c - m*c)
  const float thr = DK3_RESCALE_THR / p.scale;         // threshold on the raw scores
  bf16x8 pf[4];
  f32x16 sa0, sa1, sb0, sb1;  // scores of the tile being exponentiated / of the tile after it (the roles alternate per tile)

  // ---- prologue: K(0), V(0), K(1) staged; S(0) and its row maximum ----
  load_op(rK, kreg, 0, 64 <= S);
  load_op(rV, vreg, 0, 64 <= S);
  DK3_STORE_K(0)
  DK3_STORE_V(0)
  if (ntiles > 1) {
    load_op(rK, kreg, 1, 128 <= S);
    DK3_STORE_K(1)
  }
  __syncthreads();
  DK3_ZERO(sa0, sa1)
  DK3_QK(0, sa0, sa1)
  if (64 > S) { DK3_MASK(0, sa0, sa1) }
  {
    float mloc;
    DK3_ROWMAX(sa0, sa1, mloc);
    DK3_RESCALE(mloc)
  }

  // One tile.  CUR / NXT: score registers of tile j / j+1; slots: V(j) in j & 1, K(j+1) in (j+1) & 1; the loads fetch K(j+2) and
  // V(j+1) and store them into K slot j & 1 and V slot (j+1) & 1 (both last read in the previous iteration).
  // HAVE_N: tile j+1 exists; LOADS: 0 none, 1 full tiles (steady state: no row clamp, no branches), 2 generic (existence and tail checks).
#define DK3_TILE(J, PAR, C0, C1, N0, N1, HAVE_N, LOADS)                                                        \
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
    if (HAVE_N) { DK3_ZERO(N0, N1) }                                                                           \
    /* region A: exponentials of tile j || scores of tile j+1 (independent streams, one basic block) */        \
    DK3_SOFTMAX(C0, C1)                                                                                        \
    if (HAVE_N) { DK3_QK((PAR) ^ 1, N0, N1) }                                                                  \
    if ((HAVE_N) && (LOADS) != 1) {                                                                            \
      if ((j_ + 2) * 64 > S) { DK3_MASK(j_ + 1, N0, N1) }                                                      \
    }                                                                                                          \
    /* region B: P.V of tile j || row maximum of tile j+1 */                                                   \
    float mloc_ = -1e30f;                                                                                      \
    DK3_PV(PAR)                                                                                                \
    if (HAVE_N) DK3_ROWMAX(N0, N1, mloc_);                                                                     \
    if (have_k2_) { DK3_STORE_K(PAR) }                                                                         \
    if (have_v1_) { DK3_STORE_V((PAR) ^ 1) }                                                                   \
    if (HAVE_N) { DK3_RESCALE(mloc_) }                                                                         \
    __syncthreads();                                                                                           \
  }

  // steady state: tiles j with j + 2 full tiles behind them (K(j+2) and V(j+1) complete tiles); two tiles per trip so that the
  // score registers keep compile-time names
  int j = 0;
  const int n_full = S / 64;  // tiles 0 .. n_full - 1 are complete
  for (; j + 3 < n_full; j += 2) {  // needs K(j+3), V(j+2) full for the second body: j + 3 <= n_full - 1
    DK3_TILE(j, 0, sa0, sa1, sb0, sb1, true, 1)
    DK3_TILE(j + 1, 1, sb0, sb1, sa0, sa1, true, 1)
  }
  // remaining tiles (at most 4 + the tail): generic bodies; j is even here
  for (; j < ntiles; j += 2) {
    if (j + 1 < ntiles) {
      DK3_TILE(j, 0, sa0, sa1, sb0, sb1, true, 2)
      if (j + 2 < ntiles) {
        DK3_TILE(j + 1, 1, sb0, sb1, sa0, sa1, true, 2)
      } else {
        DK3_TILE(j + 1, 1, sb0, sb1, sa0, sa1, false, 0)
      }
    } else {
      DK3_TILE(j, 0, sa0, sa1, sb0, sb1, false, 0)
    }
  }
#undef DK3_TILE
#undef DK3_STORE_K
#undef DK3_STORE_V
#undef DK3_QK
#undef DK3_MASK
#undef DK3_ROWMAX
#undef DK3_RESCALE
#undef DK3_SOFTMAX
#undef DK3_PV
#undef DK3_ZERO

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

template <int D, int NW, bool QFUSE>
static int launch_attn3(const AttnParams& p, hipStream_t stream) {
  using C = Attn3Cfg<D, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn3_fwd_kernel<D, NW, QFUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_set = true;
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  hipLaunchKernelGGL((dk_attn3_fwd_kernel<D, NW, QFUSE>), dim3(nq * p.H * p.B), dim3(C::NT), C::LDS_BYTES, stream, p);
  return 0;
}

// waves: 8 or 4 per workgroup; no score bias (the text encoders keep dk_attn2_fwd_kernel)
int dk_launch_attention3(const AttnParams& p, int waves, hipStream_t stream) {
  DK_REQUIRE(p.bias == nullptr, "attention3: no score-bias variant");
  DK_REQUIRE((size_t)p.S * p.ld * 2 < (1ull << 32), "attention3: one batch row of QKV must span < 4 GiB");
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  if (p.D == 128) {
    if (waves == 8) return qfuse ? launch_attn3<128, 8, true>(p, stream) : launch_attn3<128, 8, false>(p, stream);
    return qfuse ? launch_attn3<128, 4, true>(p, stream) : launch_attn3<128, 4, false>(p, stream);
  }
  if (waves == 8) return qfuse ? launch_attn3<64, 8, true>(p, stream) : launch_attn3<64, 8, false>(p, stream);
  return qfuse ? launch_attn3<64, 4, true>(p, stream) : launch_attn3<64, 4, false>(p, stream);
}
