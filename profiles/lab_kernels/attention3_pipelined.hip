// Joint text/image attention forward, software-pipelined variant (dk_attn3_fwd_kernel).
//
// Same algorithm, layouts and MFMA operand mapping as dk_attn2_fwd_kernel (attention2.hip; reference call sites
// python/src/diffusionkit/mlx/mmdit.py:562,643,687,736): transposed scores S^T = K Q^T on v_mfma_f32_32x32x16_bf16,
// lane-local online softmax with the deferred rescale (threshold 4), O^T += V^T P^T with V through ds_read_b64_tr_b16.
//
// Why: in the second-generation kernel a wave's tile is one dependent chain -- 16 score MFMAs, then ~190 VALU instructions of
// softmax, then 16 P.V MFMAs -- so the matrix pipe only works when the OTHER wave of the SIMD happens to be in a different part of
// its chain, and the per-tile barrier keeps pulling the two into step (rocprofv3 PMC, profiles/archive/r01_pmc_bench_v13.md: MFMA busy
// 34 %, 6.0 VALU per MFMA).  Here every wave carries two tiles in flight (guide T15):
//   region A   P(j) = exp2(S(j) c - m c), row sums, bf16 packing     (VALU, ~115 instructions)
//              || S(j+1) = K(j+1) Q^T                                (16 MFMAs + their 16 LDS fragment reads)
//   region B   O += V(j) P(j)                                        (16 MFMAs + 32 transpose reads)
//              || row maximum of S(j+1), vote on the rescale         (VALU, ~40 instructions)
// Both regions are straight-line code in ONE basic block (global loads of the tiles after next at its top, their LDS stores at
// its end), so hipcc's scheduler can interleave the independent MFMA and VALU streams; the rare rescale of O sits behind the
// block, after P.V(j) has completed -- the order T13 requires (decision and rescale after the pending tile's P.V, before the
// exponentials of the tile the new maximum covers).  K runs one tile ahead of V through the same two LDS slots each.
//
// Balanced form (BAL): with one image the grid is 1.6 rounds of the CUs (FLUX: 408 workgroups of equal length on 256 CUs), so a
// fifth of the chip idles through the second round whatever the workgroup size.  Here the launch has one workgroup per CU and the
// iteration space (task = (batch, head, 256-query block)) x (64-key tiles) is cut into equal contiguous ranges, the way a stream-K
// GEMM cuts (tile, k).  A range is at least one task long, so a task is either whole inside one range or split in two: its HEAD
// (keys from 0) is the last segment of workgroup c, its TAIL the first segment of workgroup c+1.  The tail's owner stores its
// unnormalised (m, l, O) to a workspace slot first thing and raises a flag (guide G16 hand-off: write-through stores, vmcnt(0) in
// every wave, barrier, one relaxed agent-scope flag store); the head's owner still has its own (m, l, O) in registers when it
// gets there, merges the two softmax states and writes the output.  Dependencies only point at higher workgroup indices, which
// are dispatched later and wait for nobody's output but their own successor's: no deadlock whatever the residency.
#include "dk_kernels.h"

#define DK3_RESCALE_THR 4.0f  // natural-log units of the scaled scores
// lab only (scripts/build_attn_abl.sh): parts of the tile body taken out to see what each costs; results are garbage then.
// 1 softmax VALU, 2 barriers, 4 global loads + LDS stores, 8 K / Q fragment reads, 16 V reads, 32 MFMAs
#ifndef DK3_ABL
#define DK3_ABL 0
#endif
extern int g_dk_attn_balance;
#define DK3_SLOT_BYTES (8 * 17 * 1024)  // one partial state: 8 waves x (16 chunks of O + 1 chunk of (m, l)) x 64 lanes x 16 B

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

__device__ __forceinline__ void a3_store_sc1_b128(float* ptr, f32x4 v) {  // 16-byte write-through store (no release fence needed)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

template <int D, int NW, bool QFUSE, bool BAL>
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
  const unsigned row_bytes = (unsigned)p.ld * 2u;
  const int nt_task = (S + 63) / 64;  // key tiles of a task
  // this workgroup's range of the (task, key tile) iteration space: one whole task, or (BAL) an equal share of all of them
  int it = t * nt_task, it_end = it + nt_task;
  if (BAL) {
    const int total = nq * p.H * p.B * nt_task;  // (the launcher checks total * gridDim.x < 2^31)
    it = (int)((long)t * total / (int)gridDim.x);
    it_end = (int)((long)(t + 1) * total / (int)gridDim.x);
  }
  float* const my_slot = BAL ? (float*)((char*)p.bal_ws + (size_t)t * DK3_SLOT_BYTES) + (wave * 17 * 64 + lane) * 4 : nullptr;

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


  do {  // one segment: key tiles [jb, je) of task `task`
  const int task = it / nt_task;
  const int jb = it - task * nt_task;
  const int je = min(nt_task, jb + (it_end - it));
  const int nt = je - jb;
  it += nt;
  const int qb = task % nq, head = (task / nq) % p.H, b = task / (nq * p.H);
  const int q0 = qb * C::QB + wave * 32;

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + (size_t)b * S * p.ld + head * D);  // wave-uniform bases
  const char* Vb = (const char*)(p.V + (size_t)b * S * p.ld + head * D);

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

  if (C::QLDS) {
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk)
      *(__attribute__((address_space(3))) bf16x8*)(lds + C::Q_OFF + wave * (32 * C::ROWB) + lane * 16 + kk * 1024) = qf[kk];
  }
  const unsigned q_lds = C::Q_OFF + wave * (32 * C::ROWB) + lane * 16;  // (same wave writes and reads: program order + lgkmcnt suffice)

  u32x4 kreg[C::NCH], vreg[C::NCH];
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
    const bf16x8 k0_ = (DK3_ABL & 8) ? abl_f[kk & 1] : *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + (kr_base ^ (unsigned)(kk << 5)));    \
    const bf16x8 k1_ = (DK3_ABL & 8) ? abl_f[2 + (kk & 1)] : *(const __attribute__((address_space(3))) bf16x8*)(lds + K_OFF + (SLOT) * C::TILE_BYTES + 32 * C::ROWB + (kr_base ^ (unsigned)(kk << 5)));  \
    const bf16x8 q_ = (DK3_ABL & 8) ? abl_f[kk & 3] : C::QLDS ? *(const __attribute__((address_space(3))) bf16x8*)(lds + q_lds + kk * 1024) : qf[kk];                           \
    if (DK3_ABL & 32) { asm volatile("" ::"v"(k0_), "v"(k1_), "v"(q_)); } else {                                                                \
    A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0_, q_, A0, 0, 0, 0);                                                                        \
    A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1_, q_, A1, 0, 0, 0); }                                                                      \
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
    if (DK3_ABL & 1) { OUT = m_run; break; }                                                             \
    float m_ = fmaxf(A0[0], A1[0]);                                                                      \
    _Pragma("unroll") for (int e = 1; e < 16; ++e) m_ = fmaxf(m_, fmaxf(A0[e], A1[e]));                  \
    OUT = fmaxf(m_, __shfl_xor(m_, 32, 64));                                                             \
  } while (0)
// the rare rescale: every accumulator still at the old maximum (O, l) exactly once; nothing else is pending
#define DK3_RESCALE(MLOC)                                                                                                  \
  if (!(DK3_ABL & 1) && !__all((MLOC) - m_run <= thr)) {                                                                                     \
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
    if (!(DK3_ABL & 1)) _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                      \
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
    bf16x8 vf_;                                                                                                                             \
    if (DK3_ABL & 16) vf_ = abl_f[dt & 3]; else {                                                                                           \
    const s16x4 vh0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm_ + vr_off[dt & 1]));    \
    const s16x4 vh1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + imm_ + 256 + vr_off[dt & 1])); \
    vf_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0_, vh1_, 0, 1, 2, 3, 4, 5, 6, 7)); }                                        \
    if (DK3_ABL & 32) { asm volatile("" ::"v"(vf_), "v"(pf[2 * u + tt])); } else                                                           \
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

  bf16x8 abl_f[4];  // (lab builds: loop-invariant fragments in place of LDS reads)
  if (DK3_ABL & 24) {
#pragma unroll
    for (int i = 0; i < 4; ++i) abl_f[i] = qf[i];
  }
  // ---- prologue: K(0), V(0), K(1) staged; S(0) and its row maximum ----
  load_op(rK, kreg, jb, (jb + 1) * 64 <= S);
  load_op(rV, vreg, jb, (jb + 1) * 64 <= S);
  DK3_STORE_K(0)
  DK3_STORE_V(0)
  if (nt > 1) {
    load_op(rK, kreg, jb + 1, (jb + 2) * 64 <= S);
    DK3_STORE_K(1)
  }
  __syncthreads();
  DK3_ZERO(sa0, sa1)
  DK3_QK(0, sa0, sa1)
  if ((jb + 1) * 64 > S) { DK3_MASK(jb, sa0, sa1) }
  {
    float mloc;
    DK3_ROWMAX(sa0, sa1, mloc);
    DK3_RESCALE(mloc)
  }

  // One tile (J counts from the segment's first tile jb).  CUR / NXT: score registers of tile j / j+1; slots: V(j) in j & 1,
  // K(j+1) in (j+1) & 1; the loads fetch K(j+2) and V(j+1) and store them into K slot j & 1 and V slot (j+1) & 1 (both last read
  // in the previous iteration).
  // HAVE_N: tile j+1 exists; LOADS: 0 none, 1 full tiles (steady state: no row clamp, no branches), 2 generic (existence and tail checks).
#define DK3_TILE(J, PAR, C0, C1, N0, N1, HAVE_N, LOADS)                                                        \
  {                                                                                                            \
    const int j_ = (J);                                                                                        \
    bool have_k2_ = false, have_v1_ = false;                                                                   \
    if ((LOADS) != 0 && (DK3_ABL & 4)) {                                                                       \
    } else if ((LOADS) == 1) {                                                                                 \
      load_op(rK, kreg, jb + j_ + 2, true);                                                                    \
      load_op(rV, vreg, jb + j_ + 1, true);                                                                    \
      have_k2_ = have_v1_ = true;                                                                              \
    } else if ((LOADS) == 2) {                                                                                 \
      have_k2_ = j_ + 2 < nt;                                                                                  \
      have_v1_ = j_ + 1 < nt;                                                                                  \
      if (have_k2_) load_op(rK, kreg, jb + j_ + 2, (jb + j_ + 3) * 64 <= S);                                   \
      if (have_v1_) load_op(rV, vreg, jb + j_ + 1, (jb + j_ + 2) * 64 <= S);                                   \
    }                                                                                                          \
    if (HAVE_N) { DK3_ZERO(N0, N1) }                                                                           \
    /* region A: exponentials of tile j || scores of tile j+1 (independent streams, one basic block) */        \
    DK3_SOFTMAX(C0, C1)                                                                                        \
    if (HAVE_N) { DK3_QK((PAR) ^ 1, N0, N1) }                                                                  \
    if ((HAVE_N) && (LOADS) != 1) {                                                                            \
      if ((jb + j_ + 2) * 64 > S) { DK3_MASK(jb + j_ + 1, N0, N1) }                                            \
    }                                                                                                          \
    /* region B: P.V of tile j || row maximum of tile j+1 */                                                   \
    float mloc_ = -1e30f;                                                                                      \
    DK3_PV(PAR)                                                                                                \
    if (HAVE_N) DK3_ROWMAX(N0, N1, mloc_);                                                                     \
    if (have_k2_) { DK3_STORE_K(PAR) }                                                                         \
    if (have_v1_) { DK3_STORE_V((PAR) ^ 1) }                                                                   \
    if (HAVE_N) { DK3_RESCALE(mloc_) }                                                                         \
    if (!(DK3_ABL & 2)) __syncthreads();                                                                       \
  }

  // steady state: tiles j with j + 2 full tiles behind them (K(j+2) and V(j+1) complete tiles); two tiles per trip so that the
  // score registers keep compile-time names
  int j = 0;
  const int n_full = min(je, S / 64) - jb;  // the segment's tiles 0 .. n_full - 1 are complete
  for (; j + 3 < n_full; j += 2) {  // needs K(j+3), V(j+2) full for the second body: j + 3 <= n_full - 1
    DK3_TILE(j, 0, sa0, sa1, sb0, sb1, true, 1)
    DK3_TILE(j + 1, 1, sb0, sb1, sa0, sa1, true, 1)
  }
  // remaining tiles (at most 4 + the tail): generic bodies; j is even here
  for (; j < nt; j += 2) {
    if (j + 1 < nt) {
      DK3_TILE(j, 0, sa0, sa1, sb0, sb1, true, 2)
      if (j + 2 < nt) {
        DK3_TILE(j + 1, 1, sb0, sb1, sa0, sa1, true, 2)
      } else {
        DK3_TILE(j + 1, 1, sb0, sb1, sa0, sa1, false, 0)
      }
    } else {
      DK3_TILE(j, 0, sa0, sa1, sb0, sb1, false, 0)
    }
  }

  // ---- the segment's end: a tail hands its state over, a head merges its tail's state in, then (whole task or head) normalise
  // and store: lane owns query q0+l31, d = dt*32 + 8g + 4hi + {0..3} ----
  const bool seg_tail = BAL && jb > 0, seg_head = BAL && je < nt_task;
  if (seg_tail) {
    // (always this workgroup's first segment)  16 chunks of O, then (m, l): [chunk][lane] x 16 bytes per wave
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        a3_store_sc1_b128(my_slot + (dt * 4 + g4) * 256, f32x4{o[dt][4 * g4 + 0], o[dt][4 * g4 + 1], o[dt][4 * g4 + 2], o[dt][4 * g4 + 3]});
    a3_store_sc1_b128(my_slot + (D / 8) * 256, f32x4{m_run, l_run, 0.f, 0.f});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have completed
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.bal_flags + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (seg_head) {
      // (always this workgroup's last segment)  the tail belongs to workgroup t + 1, which computed it first thing
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(p.bal_flags + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 24)) {
            __hip_atomic_store(p.bal_flags + 1023, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // error word
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      const float* other = my_slot + DK3_SLOT_BYTES / 4;
      const f32x4 ml = *(const f32x4*)(other + (D / 8) * 256);
      const float m_new = fmaxf(m_run, ml[0]);
      const float a_own = __builtin_amdgcn_exp2f((m_run - m_new) * c), a_oth = __builtin_amdgcn_exp2f((ml[0] - m_new) * c);
      l_run = l_run * a_own + ml[1] * a_oth;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 ov = *(const f32x4*)(other + (dt * 4 + g4) * 256);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[dt][4 * g4 + e] = o[dt][4 * g4 + e] * a_own + ov[e] * a_oth;
        }
      __syncthreads();  // every wave has read the slot
      if (tid == 0) __hip_atomic_store(p.bal_flags + t + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float lsum = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / lsum;
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
        const float t = amax * (1.0f / 448.0f);
        unsigned e8 = (__float_as_uint(t) + 0x7FFFFFu) >> 23;  // ceil(log2 t) + 127 (dk_mx8_quantize8)
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
  }
  } while (BAL && it < it_end);
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
}

template <int D, int NW, bool QFUSE>
static int launch_attn3(const AttnParams& p, hipStream_t stream) {
  using C = Attn3Cfg<D, NW>;
  static DkDeviceOnce attr_once;
  static int n_cu = 0;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn3_fwd_kernel<D, NW, QFUSE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn3_fwd_kernel<D, NW, QFUSE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    int dev = 0;
    DK_CHECK_HIP(hipGetDevice(&dev));
    DK_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_once.mark();
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  const long tasks = (long)nq * p.H * p.B, nt = (p.S + 63) / 64;
  // balanced form: one workgroup per CU over equal ranges of (task, key tile); every range holds at least a whole task (so that a
  // task is split in two at most) and the caller provided the hand-off workspace.  OPT-IN (dk_tune_set("attn_balance", 1)): FLUX,
  // one image = 408 tasks on 256 CUs = 1.59 rounds, so the model promises +20 %; measured +7 % in isolation (270 -> 252 us:
  // with every CU busy to the end a key tile takes 2.3 us instead of 2.0 -- the chip's clock follows its power budget, as for
  // the GEMM's tile heights and remainder split) and nothing inside the model (848 -> 864 TF, step time unchanged;
  // profiles/r02_attention_balanced.log), where the fused QK-norm + RoPE of the query load is paid per segment.
  const bool bal = g_dk_attn_balance > 0 && D == 128 && NW == 8 && p.bal_ws != nullptr && p.bal_flags != nullptr && n_cu <= 1022 && tasks > n_cu &&
                   tasks * nt * n_cu < (1l << 31);
  if (bal)
    hipLaunchKernelGGL((dk_attn3_fwd_kernel<D, NW, QFUSE, true>), dim3(n_cu), dim3(C::NT), C::LDS_BYTES, stream, p);
  else
    hipLaunchKernelGGL((dk_attn3_fwd_kernel<D, NW, QFUSE, false>), dim3((unsigned)tasks), dim3(C::NT), C::LDS_BYTES, stream, p);
  return 0;
}

// dk_tune_set("attn_balance", v): 1 = balanced form whenever possible; -1 (default) / 0 = plain grid (see launch_attn3)
int g_dk_attn_balance = -1;

// workspace of the balanced form: one slot per CU + 4 KiB of flags (zero before the first launch; the kernels leave them zero)
size_t dk_attention_balance_workspace_bytes() {
  static int n_cu = 0;  // (queried once: this runs on every attention launch of an engine call)
  if (n_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cu = n;
  }
  return (size_t)(n_cu + 1) * DK3_SLOT_BYTES + 4096;
}

// 8 waves per workgroup, D = 128; no score bias (the text encoders keep dk_attn2_fwd_kernel)
int dk_launch_attention3(const AttnParams& p, int waves, hipStream_t stream) {
  DK_REQUIRE(p.bias == nullptr, "attention3: no score-bias variant");
  DK_REQUIRE(p.D == 128 && waves == 8, "attention3: head_dim 128, 8 waves (the other forms were pruned in round 3)");
  DK_REQUIRE((size_t)p.S * p.ld * 2 < (1ull << 32), "attention3: one batch row of QKV must span < 4 GiB");
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  return qfuse ? launch_attn3<128, 8, true>(p, stream) : launch_attn3<128, 8, false>(p, stream);
}
