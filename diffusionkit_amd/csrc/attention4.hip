// Joint text/image attention forward, PHASE-ALTERNATING variant (dk_attn4_fwd_kernel), D = 128, 8 waves of 32 queries.
//
// Same algorithm, layouts and MFMA operand mapping as dk_attn3_fwd_kernel (attention3.hip; reference call sites
// python/src/diffusionkit/mlx/mmdit.py:562,643,687,736): transposed scores S^T = K Q^T on v_mfma_f32_32x32x16_bf16, lane-local
// online softmax with the deferred rescale (threshold 4), O^T += V^T P^T with V through ds_read_b64_tr_b16.
//
// Why (round 3 measurements; profiles/r03_attention_phase_alternating.md): the pipelined kernel's ablation builds say the matrix pipe
// alone needs 52 % of the launch and everything else alone 64 %: the two barely overlap, because both waves of a SIMD run the same
// part of the same tile at the same time (the per-tile barrier re-aligns them), and a wave that carries two tiles in flight has no
// registers left to read its fragments ahead (256 VGPRs, 14 spilled).
//
// Here the two waves of a SIMD are kept in OPPOSITE phases by construction.  A tile is two phases per wave:
//   M phase (matrix pipe):  O += V(j-1) P(j-1), then S(j) = K(j) Q^T: 32 MFMAs and
//                           48 fragment reads through a ring of 8 register slots, each read issued 7 MFMAs before its use
//   V phase (vector ALU):   row maximum, rescale vote, P(j) = exp2(...), row sums, bf16 packing; LDS stores of the staged K / V
// and every phase ends at the workgroup barrier.  Waves 0-3 (group A) start with M(0); waves 4-7 (group B: wave w + 4 shares its
// SIMD with wave w) pass one extra barrier first, so that B runs M(j) while A runs V(j), and A runs M(j+1) while B runs V(j).  One
// score tile in flight per wave: 64 (O) + 32 (S) + 16 (P) + 32 (Q) + 16 (staging) + 32 (ring) registers, no spills.
//
// What is left (same file of measurements; FLUX shape, 228-230 us): the MFMAs alone take 139 us; with the fragment reads and the
// two barriers per tile but no softmax arithmetic and no K / V staging 201 us; staging +10, softmax +18 (half of it the
// exponentials).  The skeleton -- one wave per SIMD feeding the matrix pipe at a time, an LDS latency at the start of every M phase,
// two workgroup barriers per tile -- is the larger part of the gap; the V phase of a lone wave (33 exponentials at ~12 cycles, ~115
// other VALU instructions at ~5) is about as long as its M phase.  Tried on top and measured flat or worse: three LDS slots with
// the first fragments of an M phase read before its barrier, one barrier per tile, the barrier a few MFMAs before the end of the M
// phase, the scores ahead of P.V, the row sums through the matrix pipe, pre-scaled queries (no multiply-subtract per score), one
// loop body instead of two, wave priorities.
//
// K / V staging through two LDS slots each; with the groups one period apart the rule "a slot is rewritten after its last reader
// and before its next" gives: in V(j) group A stores K(j+1) and V(j), group B stores K(j+2) and V(j+1) (B's threads hold the tile
// one further ahead); a V phase starts with these stores and the global loads of what the NEXT V phase stores.
#include "dk_kernels.h"

#define DK4_RESCALE_THR 4.0f  // natural-log units of the scaled scores
// lab only (scripts/build_attn_abl.sh, K=4): parts of a tile taken out to see what each costs; results are garbage then.
// 1 softmax VALU, 4 global loads + LDS stores, 32 MFMAs (the fragment reads stay), 64 V fragment reads, 128 K fragment reads
#ifndef DK4_ABL
#define DK4_ABL 0
#endif
// lab only (scripts/attn_trace.py): the eight waves of workgroup 0 stamp s_memtime at the phase boundaries of tiles 20..27 into p.bal_ws
#ifndef DK4_TRACE
#define DK4_TRACE 0
#endif

struct Attn4Cfg {
  static constexpr int D = 128, NW = 8;
  static constexpr int KV = 64;
  static constexpr int ROWB = D * 2;
  static constexpr int TILE_BYTES = KV * D * 2;
  static constexpr int NT = NW * 64;
  static constexpr int NCHUNK = KV * D / 8;  // 16-byte chunks per K (or V) tile
  static constexpr int NCH = NCHUNK / NT;    // per thread
  static constexpr int CPR = D / 8;
  static constexpr int QB = NW * 32;
  static constexpr int LDS_BYTES = 4 * TILE_BYTES;  // K[2] V[2]
};

typedef __attribute__((address_space(3))) char lds_char4;

// max of a value over a lane and the lane 32 away (v_permlane32_swap: one VALU instruction; __shfl_xor(x, 32) is a ds_bpermute
// round trip through the LDS pipe)
__device__ __forceinline__ float dk4_max_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0]: the lower half's value in both halves, r[1]: the upper half's
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <bool QFUSE>
__global__ __launch_bounds__(512, 2) void dk_attn4_fwd_kernel(AttnParams p) {
  using C = Attn4Cfg;
  constexpr int D = C::D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_char4*)smem != 0u) __builtin_trap();  // LDS addressed from 0: offsets fold into instruction immediates
  lds_char4* const lds = (lds_char4*)0;
  constexpr int K_OFF = 0, V_OFF = 2 * C::TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // 0: group A, 1: group B (one period behind)
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = p.S;

  const int nq = (S + C::QB - 1) / C::QB;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const unsigned row_bytes = (unsigned)p.ld * 2u;
  const int nt = (S + 63) / 64;  // key tiles
  const int qb = t % nq, head = (t / nq) % p.H, b = t / (nq * p.H);
  const int q0 = qb * C::QB + wave * 32;

  // ---- per-thread constants: staging chunk coordinates, global lane offsets, LDS offsets (as dk_attn3_fwd_kernel) ----
  unsigned g_off[C::NCH];
  unsigned ks_off[2][C::NCH];  // LDS store offsets of this thread's K chunks in the V phase of an even / odd tile (slot folded in)
  unsigned vs_off[2][C::NCH];
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int id = tid + C::NT * i;
    const int kl = id / C::CPR, c8 = id % C::CPR;
    g_off[i] = (unsigned)kl * row_bytes + (unsigned)c8 * 16u;
    const unsigned ks = (unsigned)(kl * C::ROWB + ((c8 ^ (kl & 15)) << 4));
    const unsigned vs = (unsigned)((c8 >> 1) * 2048 + (kl ^ ((((c8 >> 1) & 1) << 2) | ((c8 >> 1) & 3))) * 32 + (c8 & 1) * 16);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      ks_off[par][i] = K_OFF + (unsigned)(((par + 1 + grp) & 1) * C::TILE_BYTES) + ks;  // V(j) stores K(j + 1 + grp)
      vs_off[par][i] = V_OFF + (unsigned)(((par + grp) & 1) * C::TILE_BYTES) + vs;      // ... and V(j + grp)
    }
  }
  const unsigned kr_base = (unsigned)(l31 * C::ROWB + ((hi ^ (l31 & 15)) << 4));
  const int x16 = (lane >> 4) & 1, p16 = lane & 15;
  unsigned vr_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    vr_off[par] = (unsigned)(x16 * 2048 + ((4 * (hi ^ x16) + (p16 >> 2)) ^ (2 * par + x16)) * 32 + (p16 & 3) * 8);

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + (size_t)b * S * p.ld + head * D);  // wave-uniform bases
  const char* Vb = (const char*)(p.V + (size_t)b * S * p.ld + head * D);

  u32x4 kreg[C::NCH], vreg[C::NCH];
  const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, -1, 0x00020000);
  // one operand's 64-key tile jt -> registers: one 32-bit lane offset per chunk plus a scalar tile offset.  full: the tile lies
  // inside the sequence (no row clamp)
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
#define DK4_STORE_K(PAR) \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) *(__attribute__((address_space(3))) u32x4*)(lds + ks_off[PAR][i]) = kreg[i];
#define DK4_STORE_V(PAR) \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) *(__attribute__((address_space(3))) u32x4*)(lds + vs_off[PAR][i]) = vreg[i];
// The M phase as a numbered sequence of 32 MFMAs with one 4-register fragment each: i = 0..15: O[i & 3] += V(SLOT ^ 1) P over the
// 16-key step i >> 2 ((u, tt) = (0,0) (0,1) (1,0) (1,1)); i = 16..31: S[half i & 1] += K(SLOT) Q^T over the 16-element step
// (i - 16) >> 1 of the head dimension.  The fragments go through a ring of 8 register slots: the read for MFMA i + 8 is issued
// right behind MFMA i, 7 MFMAs (224 matrix-pipe cycles) before it is needed, and sched_barriers pin that order -- left to itself
// hipcc reads each fragment directly in front of its MFMA and waits out the LDS latency 32 times per tile.
#define DK4_R1(I, SLOT)                                                                                                                        \
  if (((I) < 16 && (DK4_ABL & 64)) || ((I) >= 16 && (DK4_ABL & 128))) {                                                                        \
    asm volatile("" : "+v"(fr[(I) & 7]));                                                                                                      \
  } else if ((I) < 16) {                                                                                                                              \
    const int dt_ = (I) & 3, n_ = (I) >> 2;                                                                                                    \
    const int imm_ = V_OFF + ((SLOT) ^ 1) * C::TILE_BYTES + dt_ * 4096 + (32 * (n_ >> 1) + 16 * (n_ & 1)) * 32;                                \
    const s16x4 vh0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm_ + vr_off[dt_ & 1]));       \
    const s16x4 vh1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm_ + 256 + vr_off[dt_ & 1])); \
    fr[(I) & 7] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0_, vh1_, 0, 1, 2, 3, 4, 5, 6, 7));                                      \
  } else {                                                                                                                                     \
    const int kk_ = ((I) - 16) >> 1, half_ = (I) & 1;                                                                                          \
    fr[(I) & 7] = *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + half_ * 32 * C::ROWB + (kr_base ^ (unsigned)(kk_ << 5))); \
  }
#define DK4_M1(I)                                                                                                     \
  if (DK4_ABL & 32) {                                                                                                 \
    asm volatile("" ::"v"(fr[(I) & 7]));                                                                              \
  } else if ((I) < 16) {                                                                                              \
    o[(I) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(I) & 7], pf[((I) >> 2) & 3], o[(I) & 3], 0, 0, 0);       \
  } else if ((I) & 1) { /* (the first step of a score chain accumulates onto the inline constant 0: no 32 v_mov per tile) */ \
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(I) & 7], qf[(((I) - 16) >> 1) & 7], (I) == 17 ? zero16 : s1, 0, 0, 0); \
  } else {                                                                                                            \
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(I) & 7], qf[(((I) - 16) >> 1) & 7], (I) == 16 ? zero16 : s0, 0, 0, 0); \
  }
// a phase ends at the workgroup barrier, and nothing may move across it (the compiler would otherwise sink the exponentials whose
// P fragments are needed late into the next M phase: s_barrier orders memory, not VALU work)
#define DK4_PHASE_END                     \
  __builtin_amdgcn_sched_barrier(0);      \
  __syncthreads();                        \
  __builtin_amdgcn_sched_barrier(0);
// MFMAs FIRST .. LAST - 1 of the sequence against slot parity SLOT
#define DK4_MSEQ(SLOT, FIRST, LAST)                                                         \
  _Pragma("unroll") for (int i_ = (FIRST); i_ < (FIRST) + 8; ++i_) { DK4_R1(i_, SLOT) }     \
  __builtin_amdgcn_sched_barrier(0);                                                        \
  _Pragma("unroll") for (int i_ = (FIRST); i_ < (LAST); ++i_) {                             \
    DK4_M1(i_)                                                                              \
    if (i_ + 8 < (LAST)) { DK4_R1(i_ + 8, SLOT) }                                           \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  }
// scores of keys beyond the sequence end (tail tile JT) -> -1e30
#define DK4_MASK(JT, A0, A1)                                              \
  _Pragma("unroll") for (int e = 0; e < 16; ++e) {                        \
    const int key_ = (JT) * 64 + (e & 3) + 8 * (e >> 2) + 4 * hi;         \
    if (key_ >= S) A0[e] = -1e30f;                                        \
    if (key_ + 32 >= S) A1[e] = -1e30f;                                   \
  }

  f32x16 o[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;  // p = 2^(s*c - m*c)
  const float thr = DK4_RESCALE_THR / p.scale;         // threshold on the raw scores
  const bool trace_on = DK4_TRACE && blockIdx.x == 0 && p.bal_ws != nullptr;
  unsigned long long* const trace_buf = (unsigned long long*)p.bal_ws + wave * 64;  // 8 tiles x 6 stamps per wave
  bf16x8 pf[4];
  f32x16 s0, s1;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bf16x8 fr[8];  // fragment ring of the M phase

  // ---- prologue: K(0) by everybody; group B also what its (non-existent) V phase of tile -1 would store: K(1), V(0) ----
  load_op(rK, kreg, 0, 64 <= S);  // (in flight while the query rows are fetched and prepared below: one memory round trip for the whole prologue)

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + l31][kk*16 + hi*8 .. +7]
  bf16x8 qf[D / 16];
  {
    const int qrow = min(q0 + l31, S - 1);
    const bf16_t* qp = Qb + (size_t)qrow * p.ld + hi * 8;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
    if (QFUSE) {
      // QKNorm + RoPE of this lane's query row on the fly (same fp32 arithmetic and bf16 rounding points as
      // dk_qk_norm_rope_kernel): the lane and its partner (lane ^ 32) hold the two halves of every 16-element group.
      // Round 4: the norm weights and the cos / sin table rows are FETCHED UP FRONT, next to the query row -- as written before
      // (loads at their uses) the prologue was four dependent memory round trips (row, weights two at a time behind the reduction,
      // table, then the first key tile): 12 us per FLUX launch over two rounds of workgroups (profiles/r04_attention_qfuse_in_model.log).
      bf16x8 wv[D / 16];
      f32x4 t0[D / 16], t1[D / 16];
      if (p.qn_a != nullptr) {
        const bf16_t* w = (qrow < p.qn_split ? p.qn_a : p.qn_b) + hi * 8;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) wv[kk] = *(const bf16x8*)(w + kk * 16);
      }
      if (p.q_rope != nullptr) {
        const float* tab = p.q_rope + ((size_t)qrow * (D / 2) + hi * 4) * 2;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) t0[kk] = *(const f32x4*)(tab + kk * 16), t1[kk] = *(const f32x4*)(tab + kk * 16 + 4);
      }
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
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[kk][e] = round_bf16(v[kk][e] * r * (float)wv[kk][e]);
      }
      if (p.q_rope != nullptr) {
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) {
          const float cs[8] = {t0[kk][0], t0[kk][1], t0[kk][2], t0[kk][3], t1[kk][0], t1[kk][1], t1[kk][2], t1[kk][3]};
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

#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    // K(0) -> slot 0 for both groups (ks_off carries the steady-state slot, which differs per group: take the slot bits out)
    const unsigned off = (ks_off[0][i] - K_OFF) & (unsigned)(C::TILE_BYTES - 1);
    *(__attribute__((address_space(3))) u32x4*)(lds + K_OFF + off) = kreg[i];
  }
  if (grp == 1) {
    if (nt > 1) {
      load_op(rK, kreg, 1, 128 <= S);
      DK4_STORE_K(1)  // parity of tile -1: stores K(-1 + 1 + 1) = K(1) into slot 1
    }
    load_op(rV, vreg, 0, 64 <= S);
    DK4_STORE_V(1)  // ... and V(-1 + 1) = V(0) into slot 0
  }
  __syncthreads();
  if (grp == 1) __syncthreads();  // B idles through A's M(0)
  // the tiles V(0) stores: K(1 + grp), V(grp)
  if (!(DK4_ABL & 4)) {
    if (1 + grp < nt) load_op(rK, kreg, 1 + grp, (2 + grp) * 64 <= S);
    if (grp < nt) load_op(rV, vreg, grp, (1 + grp) * 64 <= S);
  }

  // One tile: M phase, barrier, V phase, barrier.  PAR = J & 1 (compile-time: LDS slots as immediates).
  // LOADS: 1 = this group's K(J + 1 + grp) and V(J + grp) are complete tiles (steady state: no checks), 2 = generic
#define DK4_STAMP(J, K)                                                                   \
  if (DK4_TRACE && trace_on && (J) >= 20 && (J) < 28) {                                   \
    unsigned long long t_;                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    if (lane == 0) trace_buf[((J) - 20) * 6 + (K)] = t_;                                  \
  }
#define DK4_STEP(J, PAR, LOADS, FIRST)                                                                              \
  {                                                                                                            \
    const int j_ = (J);                                                                                        \
    const int kt_ = j_ + 1 + grp, vt_ = j_ + grp; /* the tiles this V phase stores (in registers since the V phase before) */ \
    DK4_STAMP(j_, 0)                                                                                           \
    /* ---- M phase ---- */                                                                                    \
    DK4_STAMP(j_, 1)                                                                                           \
    if (DK4_TRACE && (FIRST) == 0) {                                                                           \
      DK4_MSEQ(PAR, 0, 16)                                                                                     \
      DK4_STAMP(j_, 2)                                                                                         \
      DK4_MSEQ(PAR, 16, 32)                                                                                    \
    } else {                                                                                                   \
      DK4_MSEQ(PAR, FIRST, 32)                                                                                 \
    }                                                                                                          \
    DK4_STAMP(j_, 3)                                                                                           \
    DK4_PHASE_END                                                                                              \
    DK4_STAMP(j_, 4)                                                                                           \
    /* ---- V phase ---- */                                                                                    \
    /* staging first: the tiles loaded a tile ago go to their slots (free since the barrier just passed), the loads of the \
       next ones (one tile further on) are issued: an M phase meets no global memory instruction */            \
    if (!(DK4_ABL & 4)) {                                                                                      \
      if ((LOADS) == 1) {                                                                                      \
        DK4_STORE_K(PAR)                                                                                       \
        DK4_STORE_V(PAR)                                                                                       \
        load_op(rK, kreg, kt_ + 1, true);                                                                      \
        load_op(rV, vreg, vt_ + 1, true);                                                                      \
      } else {                                                                                                 \
        if (kt_ < nt) { DK4_STORE_K(PAR) }                                                                     \
        if (vt_ < nt) { DK4_STORE_V(PAR) }                                                                     \
        if (kt_ + 1 < nt) load_op(rK, kreg, kt_ + 1, (kt_ + 2) * 64 <= S);                                     \
        if (vt_ + 1 < nt) load_op(rV, vreg, vt_ + 1, (vt_ + 2) * 64 <= S);                                     \
      }                                                                                                        \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if ((LOADS) != 1) {                                                                                        \
      if ((j_ + 1) * 64 > S) { DK4_MASK(j_, s0, s1) }                                                          \
    }                                                                                                          \
    {                                                                                                          \
      float mx_ = fmaxf(s0[0], s1[0]);                                                                         \
      if (!(DK4_ABL & 1)) {                                                                                    \
        _Pragma("unroll") for (int e = 1; e < 16; ++e) mx_ = __builtin_fmaxf(__builtin_fmaxf(mx_, s0[e]), s1[e]); /* v_max3 */ \
        mx_ = dk4_max_halves(mx_);                                                                             \
      }                                                                                                        \
      if (!(DK4_ABL & 1) && !__all(mx_ - m_run <= thr)) {                                                      \
        const float m_new_ = fmaxf(m_run, mx_);                                                                \
        const float alpha_ = __builtin_amdgcn_exp2f((m_run - m_new_) * c);                                     \
        m_run = m_new_;                                                                                        \
        l_run *= alpha_;                                                                                       \
        _Pragma("unroll") for (int i = 0; i < D / 32; ++i) _Pragma("unroll") for (int e = 0; e < 16; ++e) o[i][e] *= alpha_; \
      }                                                                                                        \
      const float mc_ = m_run * c;                                                                             \
      float psum_ = 0.f;                                                                                       \
      if (!(DK4_ABL & 1)) _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                     \
        s0[e] = __builtin_amdgcn_exp2f(s0[e] * c - mc_);                                                       \
        s1[e] = __builtin_amdgcn_exp2f(s1[e] * c - mc_);                                                       \
        psum_ += s0[e] + s1[e];                                                                                \
      }                                                                                                        \
      l_run += psum_;                                                                                          \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                          \
        pf[0][e] = (__bf16)s0[e]; pf[1][e] = (__bf16)s0[8 + e]; pf[2][e] = (__bf16)s1[e]; pf[3][e] = (__bf16)s1[8 + e]; \
      }                                                                                                        \
    }                                                                                                          \
    DK4_STAMP(j_, 5)                                                                                           \
    DK4_PHASE_END                                                                                              \
  }

  const int n_full = S / 64;  // tiles 0 .. n_full - 1 are complete
  // tile 0 has no P.V in front of its scores
  if (4 < n_full) {
    DK4_STEP(0, 0, 1, 16)
  } else {
    DK4_STEP(0, 0, 2, 16)
  }
  int j = 1;
  for (; j + 4 < n_full; j += 2) {  // second body: group B loads K(j + 4), which must be a complete tile
    DK4_STEP(j, 1, 1, 0)
    DK4_STEP(j + 1, 0, 1, 0)
  }
  for (; j < nt; j += 2) {  // j is odd here
    DK4_STEP(j, 1, 2, 0)
    if (j + 1 < nt) { DK4_STEP(j + 1, 0, 2, 0) }
  }
  // the last tile's P.V (its own M phase); group A passes the barrier B had in front
  if (nt & 1) {  // V(nt - 1) sits in slot (nt - 1) & 1 = SLOT ^ 1 of the macros
    DK4_MSEQ(1, 0, 16)
  } else {
    DK4_MSEQ(0, 0, 16)
  }
  if (grp == 0) __syncthreads();

  // ---- normalise and store: lane owns query q0+l31, d = dt*32 + 8g + 4hi + {0..3} ----
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
  const int q = q0 + l31;
  if (p.O8 != nullptr) {
    // MX-fp8 output: a 32-column block (one dt) of a query row lives in this lane and lane ^ 32 (16 values each)
    const size_t orow = (size_t)b * S + min(q, S - 1);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
      float v[16], amax = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = round_bf16(o[dt][e] * inv);
        amax = fmaxf(amax, fabsf(v[e]));
      }
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const float ts = amax * (1.0f / 448.0f);
      unsigned e8 = (__float_as_uint(ts) + 0x7FFFFFu) >> 23;  // ceil(log2 t) + 127 (dk_mx8_quantize8)
      e8 = e8 < 1u ? 1u : (e8 > 254u ? 254u : e8);
      const float sc = __uint_as_float((254u - e8) << 23);
      if (q < S) {
        unsigned char* orow8 = p.O8 + orow * (size_t)p.o8_ld + head * D + dt * 32 + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * g4 + 0] * sc, v[4 * g4 + 1] * sc, w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * g4 + 2] * sc, v[4 * g4 + 3] * sc, w, true);
          *(int*)(orow8 + 8 * g4) = w;
        }
        if (hi == 0) p.O8_scales[dk_mx_scale_index((unsigned)orow, (unsigned)(head * (D / 32) + dt), (unsigned)p.o8_nblk)] = (unsigned char)e8;
      }
    }
  } else if (q < S) {
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
#undef DK4_STEP
#undef DK4_STAMP
#undef DK4_STORE_K
#undef DK4_STORE_V
#undef DK4_R1
#undef DK4_M1
#undef DK4_MSEQ
#undef DK4_PHASE_END
#undef DK4_MASK
}

template <bool QFUSE>
static int launch_attn4(const AttnParams& p, hipStream_t stream) {
  using C = Attn4Cfg;
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn4_fwd_kernel<QFUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_once.mark();
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  const long tasks = (long)nq * p.H * p.B;
  hipLaunchKernelGGL((dk_attn4_fwd_kernel<QFUSE>), dim3((unsigned)tasks), dim3(C::NT), C::LDS_BYTES, stream, p);
  return 0;
}

// 8 waves per workgroup, D = 128, no score bias
int dk_launch_attention4(const AttnParams& p, hipStream_t stream) {
  DK_REQUIRE(p.bias == nullptr, "attention4: no score-bias variant");
  DK_REQUIRE(p.D == 128, "attention4: head_dim 128");
  DK_REQUIRE((size_t)p.S * p.ld * 2 < (1ull << 32), "attention4: one batch row of QKV must span < 4 GiB");
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  return qfuse ? launch_attn4<true>(p, stream) : launch_attn4<false>(p, stream);
}
