// Joint text/image attention forward for gfx950 (flash-style, scores never leave the CU).
//
// Replaces the four mx.fast.scaled_dot_product_attention call sites of the reference
// (python/src/diffusionkit/mlx/mmdit.py:562,643,687,736): softmax(q k^T * scale) v, no mask,
// non-causal, over the concatenated [text, image] sequence.  The reference materialises the
// [H, S, S] score tensor in the activation dtype (quirk Q4); here scores stay fp32 in registers.
//
// Layout: q/k/v are read in place from the token-major projection output (row stride ld, head
// h at column h*D), the output is written token-major so the o-projection GEMM consumes it
// directly -- no [B,H,S,D] transposes exist anywhere.
//
// Workgroup = 4 waves x 32 query rows = 128 queries of one (batch, head); K/V tiles of 64 keys
// are staged through registers into double-buffered LDS (K: row-major, XOR-swizzled 16-byte
// chunks -> conflict-free ds_read_b128; V: [d/16][key][16] sub-tiles read with
// ds_read_b64_tr_b16 so the PV operand needs no transpose pass).
// Per wave the score tile is computed TRANSPOSED (S^T = K Q^T, mfma(K-frag, Q-frag)) so every
// lane owns one query column: row max / row sum are in-lane reductions plus one lane^32
// exchange, and the probabilities already sit in the B-operand layout of the second MFMA,
// which accumulates O^T = V^T P^T (query again per lane => rescaling by alpha is lane-local).
#include "dk_kernels.h"

template <int D, int NW = 4>
struct AttnCfg {
  static constexpr int KV = 64;                          // keys per tile
  static constexpr int ROWB = D * 2;                     // bytes per K row
  static constexpr int TILE_BYTES = KV * D * 2;          // one K (or V) tile
  static constexpr int NT = NW * 64;                     // threads per workgroup
  static constexpr int NCH = KV * D / 8 / NT;            // 16-byte chunks per thread per tile
  static constexpr int CPR = D / 8;                      // chunks per row
  static constexpr int QB = NW * 32;                     // query rows per workgroup
};

template <int D>
__device__ __forceinline__ int k_swz(int r) { return D == 128 ? (r & 15) : ((r >> 1) & 7); }

// NW = waves per workgroup (4: 128 query rows, two workgroups per CU; 8: 256 query rows, K/V staging
// shared by twice as many waves).  VAR bit 0: the two 32-key score chains of a tile are interleaved
// (two independent accumulator chains instead of 8 dependent MFMAs in a row) and the MFMA clusters run
// at raised priority.
template <int D, int NW, int VAR>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 2) void dk_attn_fwd_kernel(AttnParams p) {
  using C = AttnCfg<D, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                       // [2][TILE_BYTES]
  char* Vs = smem + 2 * C::TILE_BYTES;   // [2][TILE_BYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = p.S;

  // XCD-contiguous block order so that the blocks of one head share an L2.
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
  const bf16_t* Kb = p.K + (size_t)b * S * p.ld + head * D;
  const bf16_t* Vb = p.V + (size_t)b * S * p.ld + head * D;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + l31][kk*16 + hi*8 .. +7]
  bf16x8 qf[D / 16];
  {
    const int qrow = min(q0 + l31, S - 1);
    const bf16_t* qp = Qb + (size_t)qrow * p.ld + hi * 8;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
  }

  u32x4 kreg[C::NCH], vreg[C::NCH];
  // per-thread staging coordinates (constant across tiles)
  int st_kl[C::NCH], st_c8[C::NCH];
#pragma unroll
  for (int i = 0; i < C::NCH; ++i) {
    const int id = tid + C::NT * i;
    st_kl[i] = id / C::CPR;
    st_c8[i] = id % C::CPR;
  }
#define DK_LOAD_TILE(J)                                                                  \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) {                                   \
    const int key = min((J) * 64 + st_kl[i], S - 1);                                     \
    kreg[i] = *(const u32x4*)(Kb + (size_t)key * p.ld + st_c8[i] * 8);                   \
    vreg[i] = *(const u32x4*)(Vb + (size_t)key * p.ld + st_c8[i] * 8);                   \
  }
  // V: odd d-blocks store key rows with bit 2 flipped so the two 16-lane groups of a tr-read
  // hit different halves of the 256-byte bank row.
#define DK_STORE_TILE(BUF)                                                               \
  _Pragma("unroll") for (int i = 0; i < C::NCH; ++i) {                                   \
    const int kl = st_kl[i], c8 = st_c8[i];                                              \
    *(u32x4*)(Ks + (BUF) * C::TILE_BYTES + kl * C::ROWB + ((c8 ^ k_swz<D>(kl)) << 4)) = kreg[i]; \
    *(u32x4*)(Vs + (BUF) * C::TILE_BYTES + (c8 >> 1) * 2048 + (kl ^ (((c8 >> 1) & 1) << 2)) * 32 + (c8 & 1) * 16) = vreg[i]; \
  }

  f32x16 o[D / 32];
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;  // fold log2(e): p = 2^(s*c - m*c)

  const int ntiles = (S + 63) / 64;
  DK_LOAD_TILE(0)
  DK_STORE_TILE(0)
  __syncthreads();

  for (int j = 0; j < ntiles; ++j) {
    const int buf = j & 1;
    if (j + 1 < ntiles) { DK_LOAD_TILE(j + 1) }
    const char* Kt = Ks + buf * C::TILE_BYTES;
    const char* Vt = Vs + buf * C::TILE_BYTES;

    // ---- S^T[key, q] for the 64 keys of this tile (2 sub-tiles of 32 keys) ----
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[u][e] = 0.f;
    if (VAR & 1) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = u * 32 + l31;
          const bf16x8 kf = *(const bf16x8*)(Kt + r * C::ROWB + (((kk * 2 + hi) ^ k_swz<D>(r)) << 4));
          s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[u], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = u * 32 + l31;
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) {
          const bf16x8 kf = *(const bf16x8*)(Kt + r * C::ROWB + (((kk * 2 + hi) ^ k_swz<D>(r)) << 4));
          s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[u], 0, 0, 0);
        }
      }
    }
    // tail tile: keys beyond S do not exist
    if (j * 64 + 64 > S) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = j * 64 + u * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          if (key >= S) s[u][e] = -1e30f;
        }
    }
    // ---- online softmax (each lane: one query, half of the keys; partner = lane^32) ----
    float mloc = s[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, s[u][e]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    float psum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = exp2f(s[u][e] * c - mc);
        s[u][e] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][e] *= alpha;

    // ---- O^T += V^T P^T ----
    if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[u][8 * tt + e];
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          const int db = dt * 2 + ((lane >> 4) & 1);
          const int p16 = lane & 15;
          s16x4 vh0, vh1;
#pragma unroll
          for (int eh = 0; eh < 2; ++eh) {
            const int keybase = 32 * u + 8 * (2 * tt + eh) + 4 * hi;
            const char* addr = Vt + db * 2048 + ((keybase ^ ((db & 1) << 2)) + (p16 >> 2)) * 32 + (p16 & 3) * 8;
            const s16x4 got = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
            if (eh == 0) vh0 = got; else vh1 = got;
          }
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    }

    if (VAR & 1) __builtin_amdgcn_s_setprio(0);
    if (j + 1 < ntiles) { DK_STORE_TILE(buf ^ 1) }
    __syncthreads();
  }

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

extern int g_dk_attn_mode;  // dk_tune_set("attn", v): -1 automatic; 0 = 4 waves; 1 = 4 waves interleaved; 2 / 3 = 8 waves

template <int D, int NW, int VAR>
static int launch_attn(const AttnParams& p, hipStream_t stream) {
  using C = AttnCfg<D, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn_fwd_kernel<D, NW, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * C::TILE_BYTES));
    attr_set = true;
  }
  const int nq = (p.S + C::QB - 1) / C::QB;
  hipLaunchKernelGGL((dk_attn_fwd_kernel<D, NW, VAR>), dim3(nq * p.H * p.B), dim3(C::NT), 4 * C::TILE_BYTES, stream, p);
  return 0;
}

int dk_launch_attention(const AttnParams& p, hipStream_t stream) {
  DK_REQUIRE(p.D == 128 || p.D == 64, "head_dim must be 64 or 128");
  DK_REQUIRE(p.S > 0 && p.B > 0 && p.H > 0, "empty attention");
  DK_REQUIRE(p.ld % 8 == 0 && p.ldo % 4 == 0, "row strides must keep 16-byte alignment");
  // automatic choice (kernel lab, profiles/r01_attention_lab.md, profiles/r02_attn_bench.log): D = 128 on long sequences: the
  // software-pipelined kernel with 8 waves per workgroup (K/V staging shared by 8 waves; +4 % over the lean kernel on the FLUX
  // shapes); otherwise the VALU-lean kernel with 4 waves (D = 64: 842 TF against 773 / 823 for the pipelined forms).
  // A score bias (text encoders) is only implemented by the lean kernel's 4-wave form
  const int mode = p.bias != nullptr ? 4 : g_dk_attn_mode < 0 ? ((p.D == 128 && p.S >= 2048) ? 7 : 4) : g_dk_attn_mode;
  dk_prof_begin(2, 4.0 * (double)p.B * p.H * (double)p.S * (double)p.S * p.D, stream);
  int rc = 0;
#define DK_ATTN_CASE(M, NW, VAR)                                                     \
  case M:                                                                           \
    rc = p.D == 128 ? launch_attn<128, NW, VAR>(p, stream) : launch_attn<64, NW, VAR>(p, stream); \
    break;
  switch (mode) {
    DK_ATTN_CASE(0, 4, 0)
    DK_ATTN_CASE(1, 4, 1)
    DK_ATTN_CASE(2, 8, 0)
    DK_ATTN_CASE(3, 8, 1)
    case 4: rc = dk_launch_attention2(p, 4, stream); break;
    case 5: rc = dk_launch_attention2(p, 8, stream); break;
    case 6: rc = dk_launch_attention2(p, 7, stream); break;
    case 7: rc = dk_launch_attention3(p, 8, stream); break;  // software-pipelined kernel (attention3.hip), 8 / 4 waves
    case 8: rc = dk_launch_attention3(p, 4, stream); break;
    default: DK_REQUIRE(false, "unknown attention variant");
  }
#undef DK_ATTN_CASE
  dk_prof_end(stream);
  if (rc) return rc;
  DK_CHECK_HIP(hipGetLastError());
  return 0;
}
