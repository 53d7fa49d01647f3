// Single-head attention over D = 512 for the VAE's mid block (flash-style: the T x T score matrix never exists).
// reference: python/src/diffusionkit/mlx/vae.py:28-57 (Attention: q / k / v Linear 512 -> 512, softmax((q / sqrt 512) k^T) v, one head
// over all H * W tokens); SURVEY.md section 8 row a22.  The reference -- and the engine until round 2 -- materialises the scores:
// 537 MB at T = 16384 (latent 128 x 128), written by a GEMM, read and rewritten by the row softmax, read again by the P.V GEMM.
//
// Why not the MMDiT kernel's shape (attention3.hip: a wave owns 32 queries and the whole head): at D = 512 a wave's O^T tile would
// be 32 x 512 fp32 = 256 accumulator registers.  Here the work of a 64-query workgroup is split by ROLE instead (8 waves):
//   waves 0-3 ("S"):  wave s owns queries 16 s .. 16 s + 15: S^T = K Q^T over all of D on v_mfma_f32_16x16x32_bf16 (Q fragments
//                     resident in 64 registers), lane-local online softmax (the four lanes that share a query: two shuffles), P as
//                     bf16 into LDS, plus the per-query rescale factor of the running maximum (deferred: threshold 4, guide T13);
//   waves 4-7 ("PV"): wave v owns output columns 128 v .. 128 v + 127 for ALL 64 queries: O^T += V^T P^T on
//                     v_mfma_f32_32x32x16_bf16 (128 accumulator registers), V^T fragments straight from a TRANSPOSED copy of V
//                     (dk_transpose_kernel, 21 us: plain ds_read_b128 fragments, no transposing reads), P^T fragments from LDS.
// The S waves run one key tile ahead of the PV waves (P double-buffered), one barrier per 32-key tile.  Waves w and w + 4 share a
// SIMD (MI355X_MICROARCH.md: a workgroup's waves go to the SIMDs cyclically), so every SIMD carries one S wave (MFMA + softmax
// VALU) and one PV wave (MFMA + LDS reads): the two never run the same phase -- the arrangement conv_halo.hip's ablations ask for
// (profiles/r03_conv_halo_ablations.md: waves in the same phase add their MFMA and non-MFMA time up).  Both roles issue the same
// number of MFMA cycles per tile (16 x 32 = 32 x 16).
//
// Rounding points: scores stay fp32 (the reference rounds them to the activation dtype, quirk Q4: here the more accurate form, as in
// the MMDiT kernels), P is rounded to bf16 before P.V (as there), the output once.
#include "dk_kernels.h"

typedef __attribute__((address_space(3))) char lds_a5;

#define A5_QB 64
#define A5_KT 32
#define A5_D 512
#define A5_K_SLOT (A5_KT * 1024)        // 32 keys x 1024 B, 16-byte chunk c of key k at position c ^ (k & 15)
#define A5_V_ROWB 80                    // 32 keys x 2 B + 16 B pad per d row: conflict-free ds_read_b128 over 16 consecutive rows
#define A5_V_SLOT (A5_D * A5_V_ROWB)
#define A5_P_ROWB 80
#define A5_P_SLOT (A5_QB * A5_P_ROWB)
#define A5_K_OFF 0
#define A5_V_OFF (2 * A5_K_SLOT)
#define A5_P_OFF (A5_V_OFF + 2 * A5_V_SLOT)
#define A5_AL_OFF (A5_P_OFF + 2 * A5_P_SLOT)   // per slot: 64 floats alpha, then 4 flag words (+ pad) = 320 B
#define A5_AL_SLOT 320
#define A5_L_OFF (A5_AL_OFF + 2 * A5_AL_SLOT)  // 64 floats: the final row sums
#define A5_LDS_BYTES (A5_L_OFF + 256)
#define A5_THR 4.0f

__global__ __launch_bounds__(512, 2) void dk_attn512_fwd_kernel(Attn512Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(lds_a5*)smem != 0u) __builtin_trap();  // LDS addressed from 0
  lds_a5* const lds = (lds_a5*)0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T;
  const int b = blockIdx.y, q0 = blockIdx.x * A5_QB;
  const bf16_t* Qb = p.Q + (size_t)b * T * p.ld;
  const bf16_t* Kb = p.K + (size_t)b * T * p.ld;
  const bf16_t* Vtb = p.Vt + (size_t)b * A5_D * p.Tp;
  const int ntiles = (T + A5_KT - 1) / A5_KT;
  const float c = p.scale * 1.44269504088896340736f;  // p = 2^(s c - m c)
  const float thr = A5_THR / p.scale;

  // ---- staging of one key tile (all 8 waves): K rows -> swizzled 1 KiB rows; V^T rows (64 B of 32 keys) -> 80-byte rows ----
  u32x4 kreg[4], vreg[4];
  const int k_key = tid >> 6, k_chunk = tid & 63;  // item i: key k_key + 8 i
  const int v_d = tid >> 2, v_c = tid & 3;         // item i: row d = v_d + 128 i
  auto load_k = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = min(t * A5_KT + k_key + 8 * i, T - 1);  // (keys beyond the sequence re-read the last one: their scores are masked)
      kreg[i] = *(const u32x4*)(Kb + (size_t)key * p.ld + k_chunk * 8);
    }
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) vreg[i] = *(const u32x4*)(Vtb + (size_t)(v_d + 128 * i) * p.Tp + t * A5_KT + v_c * 8);  // (Tp: zero-padded)
  };
  auto store_k = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k_key + 8 * i;
      *(__attribute__((address_space(3))) u32x4*)(lds + A5_K_OFF + slot * A5_K_SLOT + key * 1024 + ((k_chunk ^ (key & 15)) << 4)) = kreg[i];
    }
  };
  auto store_v = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *(__attribute__((address_space(3))) u32x4*)(lds + A5_V_OFF + slot * A5_V_SLOT + (v_d + 128 * i) * A5_V_ROWB + v_c * 16) = vreg[i];
  };

  if (wave < 4) {
    // =========================== S role ===========================
    const int l15 = lane & 15, qq = lane >> 4;
    const int qrow = min(q0 + wave * 16 + l15, T - 1);
    bf16x8 qf[16];  // B operand of S^T = K Q^T: lane (query l15, k block qq) holds Q[q][32 kk + 8 qq .. + 7]
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) qf[kk] = *(const bf16x8*)(Qb + (size_t)qrow * p.ld + kk * 32 + qq * 8);
    // K fragment of key block blk, K slice kk: key 16 blk + l15, chunk 4 kk + qq at position (4 kk + qq) ^ l15:
    // = (lane part for kk & 3) + 256 * (kk >> 2) -- the bits the XOR with l15 touches (6, 7 of the byte address) are those of kk & 3
    unsigned kbase[4];
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) kbase[k3] = (unsigned)(l15 * 1024 + (((k3 * 4 + qq) ^ l15) << 4));
    float m_run = -1e30f, l_run = 0.f;

    // one tile: scores of the 32 keys of tile t (K slot t & 1) -> P, alpha, flag into P slot t & 1
    auto s_tile = [&](int t) {
      const unsigned ks = (unsigned)(A5_K_OFF + (t & 1) * A5_K_SLOT);
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const bf16x8 k0 = *(const __attribute__((address_space(3))) bf16x8*)(lds + ks + kbase[kk & 3] + (kk >> 2) * 256);
        const bf16x8 k1 = *(const __attribute__((address_space(3))) bf16x8*)(lds + ks + 16 * 1024 + kbase[kk & 3] + (kk >> 2) * 256);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[kk], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[kk], s1, 0, 0, 0);
      }
      // lane: query l15, keys t * 32 + 4 qq + e (s0) and + 16 (s1)
      if ((t + 1) * A5_KT > T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = t * A5_KT + 4 * qq + e;
          if (key >= T) s0[e] = -1e30f;
          if (key + 16 >= T) s1[e] = -1e30f;
        }
      }
      float mt = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      // deferred rescale (guide T13): the running maximum only moves when some row of this wave outgrew it by more than the
      // threshold; then EVERY row of the wave takes its new maximum, and the factor goes to the PV waves with this tile's P
      float alpha = 1.0f;
      const bool grow = !__all(mt - m_run <= thr);
      if (grow) {
        const float m_new = fmaxf(m_run, mt);
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
      }
      const float mc = m_run * c;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] = __builtin_amdgcn_exp2f(s0[e] * c - mc);
        s1[e] = __builtin_amdgcn_exp2f(s1[e] * c - mc);
        psum += s0[e] + s1[e];
      }
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      l_run += psum;
      const unsigned pw = (unsigned)(A5_P_OFF + (t & 1) * A5_P_SLOT + (wave * 16 + l15) * A5_P_ROWB + qq * 8);
      *(__attribute__((address_space(3))) u32x2*)(lds + pw) = u32x2{pack2bf(s0[0], s0[1]), pack2bf(s0[2], s0[3])};
      *(__attribute__((address_space(3))) u32x2*)(lds + pw + 32) = u32x2{pack2bf(s1[0], s1[1]), pack2bf(s1[2], s1[3])};
      const unsigned aw = (unsigned)(A5_AL_OFF + (t & 1) * A5_AL_SLOT);
      if (qq == 0) *(__attribute__((address_space(3))) float*)(lds + aw + (wave * 16 + l15) * 4) = alpha;
      if (lane == 0) *(__attribute__((address_space(3))) unsigned*)(lds + aw + 256 + wave * 4) = grow ? 1u : 0u;
    };

    // prologue: K(0), K(1), V(0) staged; P(0)
    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    if (ntiles > 1) {
      load_k(1);
      store_k(1);
    }
    __syncthreads();
    s_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      const bool k2 = t + 2 < ntiles, v1 = t + 1 < ntiles;
      if (k2) load_k(t + 2);
      if (v1) load_v(t + 1);
      if (v1) s_tile(t + 1);  // K slot (t + 1) & 1 (staged one iteration ago), P slot (t + 1) & 1 (read by the PV waves in iteration t - 1)
      if (k2) store_k(t & 1);        // (K slot t & 1: last read in iteration t - 1)
      if (v1) store_v((t + 1) & 1);  // (V slot (t + 1) & 1: last read by the PV waves in iteration t - 1)
      __syncthreads();
    }
    if (qq == 0) *(__attribute__((address_space(3))) float*)(lds + A5_L_OFF + (wave * 16 + l15) * 4) = l_run;
    __syncthreads();
  } else {
    // =========================== PV role ===========================
    const int v = wave - 4;
    const int l31 = lane & 31, kh = lane >> 5;
    f32x16 o[4][2];  // [32-column block of this wave's 128][32-query block]
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][qb][e] = 0.f;
    const unsigned va = (unsigned)(A5_V_OFF + (v * 128 + l31) * A5_V_ROWB + kh * 16);  // + slot, + dt * 32 rows, + 32 B per 16-key step
    const unsigned pa = (unsigned)(A5_P_OFF + l31 * A5_P_ROWB + kh * 16);              // + slot, + qb * 32 rows, + 32 B per step

    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    if (ntiles > 1) {
      load_k(1);
      store_k(1);
    }
    __syncthreads();
    __syncthreads();  // (the S waves compute P(0) between these two)
    for (int t = 0; t < ntiles; ++t) {
      const bool k2 = t + 2 < ntiles, v1 = t + 1 < ntiles;
      if (k2) load_k(t + 2);
      if (v1) load_v(t + 1);
      const unsigned aw = (unsigned)(A5_AL_OFF + (t & 1) * A5_AL_SLOT);
      const u32x4 fl = *(const __attribute__((address_space(3))) u32x4*)(lds + aw + 256);
      if (__any((fl[0] | fl[1] | fl[2] | fl[3]) != 0u)) {  // rare after the first tiles: O takes the rescale of the running maxima
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const float al = *(const __attribute__((address_space(3))) float*)(lds + aw + (qb * 32 + l31) * 4);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][qb][e] *= al;
        }
      }
      const unsigned vs = (unsigned)((t & 1) * A5_V_SLOT), ps = (unsigned)((t & 1) * A5_P_SLOT);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf[2], vf[4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) pf[qb] = *(const __attribute__((address_space(3))) bf16x8*)(lds + pa + ps + qb * 32 * A5_P_ROWB + st * 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[dt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + va + vs + dt * 32 * A5_V_ROWB + st * 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) o[dt][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pf[qb], o[dt][qb], 0, 0, 0);
      }
      if (k2) store_k(t & 1);
      if (v1) store_v((t + 1) & 1);
      __syncthreads();
    }
    __syncthreads();  // the final row sums are in LDS
    // O^T accumulator: lane holds query l31 (+ 32 qb), columns 128 v + 32 dt + 8 g + 4 kh + {0..3}
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = q0 + qb * 32 + l31;
      const float inv = 1.0f / *(const __attribute__((address_space(3))) float*)(lds + A5_L_OFF + (qb * 32 + l31) * 4);
      if (q < T) {
        bf16_t* op = p.O + ((size_t)b * T + q) * p.ldo + v * 128 + 4 * kh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            uint2 w;
            w.x = pack2bf(o[dt][qb][4 * g4 + 0] * inv, o[dt][qb][4 * g4 + 1] * inv);
            w.y = pack2bf(o[dt][qb][4 * g4 + 2] * inv, o[dt][qb][4 * g4 + 3] * inv);
            *(uint2*)(op + dt * 32 + 8 * g4) = w;
          }
      }
    }
  }
}

int dk_launch_attention512(const Attn512Params& p, hipStream_t stream) {
  DK_REQUIRE(p.Q && p.K && p.Vt && p.O && p.T > 0 && p.B > 0, "attention512: null / empty argument");
  DK_REQUIRE(p.ld % 8 == 0 && p.ld >= A5_D && p.ldo % 4 == 0 && p.ldo >= A5_D, "attention512: row strides (16-byte aligned rows of >= 512 columns)");
  DK_REQUIRE(p.Tp % 8 == 0 && p.Tp >= (p.T + A5_KT - 1) / A5_KT * A5_KT, "attention512: V^T rows padded with zeros to a multiple of 32 keys");
  DK_REQUIRE((((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.Vt) & 15) == 0 && ((uintptr_t)p.O & 7) == 0, "attention512: alignment");
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn512_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A5_LDS_BYTES));
    attr_once.mark();
  }
  dk_prof_begin(2, 4.0 * (double)p.B * (double)p.T * (double)p.T * A5_D, stream);
  hipLaunchKernelGGL(dk_attn512_fwd_kernel, dim3((unsigned)((p.T + A5_QB - 1) / A5_QB), (unsigned)p.B), dim3(512), A5_LDS_BYTES, stream, p);
  dk_prof_end(stream);
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
