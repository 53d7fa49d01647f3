// Joint text/image attention forward, ONE wave per SIMD (dk_attn5_fwd_kernel), D = 128: a workgroup = 4 waves = 256 query rows, a wave
// owns 64 of them and the whole 512-entry register file (O^T accumulators, the query fragments and the key fragments in AGPRs; scores,
// probabilities and value fragments in VGPRs), and the tile loop is ONE hand-scheduled inline-asm block (attention5_asm.inc, written and
// CPU-checked by scripts/gen_attn5.py + scripts/attn5_emu.py).  Reference call sites: python/src/diffusionkit/mlx/mmdit.py:562,643,687,736.
//
// Same algorithm, LDS images and MFMA operand mapping as dk_attn4_fwd_kernel (attention4.hip): transposed scores S^T = K Q^T on
// v_mfma_f32_32x32x16_bf16, lane-local online softmax with the deferred rescale (threshold 4), O^T += V^T P^T with V through
// ds_read_b64_tr_b16.  What changes is the frame (cdna_hip_programming.md, the 4-wave one-wave-per-SIMD structure):
//   * a wave multiplies every K / V fragment it reads with TWO 32-query blocks: half the LDS reads per MFMA of the 8-wave kernel
//     (attention4.hip's waves read 32 KiB of fragments per 32 MFMAs -- as much as the LDS delivers in the time the matrix pipe needs);
//   * K / V tiles arrive by LDS-DMA (source-side swizzle: each lane fetches the chunk its LDS position wants) into four-slot rings, 2.5 - 3
//     tiles ahead of their reads, instead of through registers: no staging VALU / ds_write at all;
//   * per tile two phases of 32 MFMAs with the other work in the MFMA gaps, <= 5-6 instructions each: S(j+1) beside the exponentials of
//     tile j and the V(j) reads; P(j) V(j) beside the row maxima of S(j+1), the first exponentials of tile j+1, the K(j+2) reads and the
//     DMA pieces of K(j+3), V(j+2); one barrier per tile;
//   * the rescale decision of a tile is taken while the previous tile's P.V is still in flight: it records the factor and switches the
//     exponent offset; accumulators and row sums take the factor once that P.V is complete (scripts/gen_attn5.py, header).
// Shapes: D = 128, S a multiple of 256 with at least 12 key tiles, no score bias (attention4.hip keeps the rest).
#include "dk_kernels.h"

typedef __attribute__((address_space(3))) char a5_lds_char;
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

#define A5_LDS_BYTES 131072  // four-slot K ring + four-slot V ring

template <bool QFUSE>
__global__ __launch_bounds__(256, 1) void dk_attn5_fwd_kernel(AttnParams p) {
  constexpr int D = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((unsigned)(size_t)(a5_lds_char*)smem != 0u) __builtin_trap();  // LDS addressed from 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int S = p.S;
  const int nq = (S + 255) / 256;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const unsigned row_bytes = (unsigned)p.ld * 2u;
  const int qblock = t % nq, head = (t / nq) % p.H, b = t / (nq * p.H);
  const int q0 = qblock * 256 + wave * 64;

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + (size_t)b * S * p.ld + head * D);
  const char* Vb = (const char*)(p.V + (size_t)b * S * p.ld + head * D);

  // ---- LDS read addresses (attention4.hip) and the LDS-DMA source offsets of this wave's pieces ----
  u32x8 kaddr;
  const unsigned kr_base = (unsigned)(l31 * 256 + ((hi ^ (l31 & 15)) << 4));
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) kaddr[kk] = kr_base ^ (unsigned)(kk << 5);
  const int x16 = (lane >> 4) & 1, p16 = lane & 15;
  u32x2 vaddr;
#pragma unroll
  for (int par = 0; par < 2; ++par) vaddr[par] = (unsigned)(65536 + x16 * 2048 + ((4 * (hi ^ x16) + (p16 >> 2)) ^ (2 * par + x16)) * 32 + (p16 & 3) * 8);
  // K image: row kl at kl * 256, 16-byte chunk c8 at position c8 ^ (kl & 15); piece pi = 4 wave + i covers rows 4 pi .. 4 pi + 3
  // V image: d-group dg = c8 >> 1 at dg * 2048, key kl at row position kl ^ (((dg & 1) << 2) | (dg & 3)), 32 B per key; piece pi = (dg, key half)
  u32x4 dk, dv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pi = 4 * wave + i;
    const int row = 4 * pi + (lane >> 4);
    dk[i] = (unsigned)row * row_bytes + (unsigned)(((lane & 15) ^ (row & 15)) << 4);
    const int dg = pi >> 1, hf = pi & 1;
    const int kl = (hf * 32 + (lane >> 1)) ^ (((dg & 1) << 2) | (dg & 3));
    dv[i] = (unsigned)kl * row_bytes + (unsigned)((2 * dg + (lane & 1)) << 4);
  }
  const u32x4 rK = {(unsigned)(size_t)Kb, (unsigned)((size_t)Kb >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  const u32x4 rV = {(unsigned)(size_t)Vb, (unsigned)((size_t)Vb >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  // the DMA pieces of the first seven K / V tiles leave BEFORE the query rows are fetched: one memory round trip for the whole prologue
  asm volatile(
#include "attention5_dma.inc"
      :
      : [tileb] "s"(64 * (int)row_bytes), [dbase] "s"(wave * 4096), "{v[248:251]}"(dk), "{v[252:255]}"(dv), "{s[40:43]}"(rK), "{s[44:47]}"(rV)
      : "s48", "s49", "s50", "s55", "m0", "scc", "memory");

  // ---- Q fragments of the wave's two 32-query blocks (B operand of S^T = K Q^T): lane holds Q[q][kk*16 + hi*8 .. +7] ----
  u32x16 qv[4];  // word (qb*8 + kk)*4 + r of the 64
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    bf16x8 qf[D / 16];
    const int qrow = min(q0 + qb * 32 + l31, S - 1);
    const bf16_t* qp = Qb + (size_t)qrow * p.ld + hi * 8;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 16);
    if (QFUSE) {  // QKNorm + RoPE of the query row on the fly: attention4.hip's arithmetic and rounding points
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
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
      const u32x4 w4 = __builtin_bit_cast(u32x4, qf[kk]);
#pragma unroll
      for (int r = 0; r < 4; ++r) qv[(qb * 8 + kk) >> 2][((qb * 8 + kk) & 3) * 4 + r] = w4[r];
    }
  }

  const int nt = S / 64;
  const int scale_bits = __float_as_int(p.scale);  // (the block forms c = scale * log2(e): p = 2^(s*c - m*c); rescale threshold 4 as in attention4.hip)

  f32x16 o[2][4];
  u32x2 lsum;
  asm volatile(
#include "attention5_asm.inc"
      : "={v[0:15]}"(o[0][0]), "={v[16:31]}"(o[0][1]), "={v[32:47]}"(o[0][2]), "={v[48:63]}"(o[0][3]), "={v[64:79]}"(o[1][0]), "={v[80:95]}"(o[1][1]),
        "={v[96:111]}"(o[1][2]), "={v[112:127]}"(o[1][3]), "={v[226:227]}"(lsum), "+{v[128:143]}"(qv[0]), "+{v[144:159]}"(qv[1]), "+{v[160:175]}"(qv[2]),
        "+{v[176:191]}"(qv[3])
      : [koff] "s"(4 * 64 * (int)row_bytes), [voff] "s"(3 * 64 * (int)row_bytes), [tileb] "s"(64 * (int)row_bytes), [scale] "s"(scale_bits), [ntrip] "s"((nt - 8) / 4), [dbase] "s"(wave * 4096), "{v[238:245]}"(kaddr),
        "{v[246:247]}"(vaddr), "{v[248:251]}"(dk), "{v[252:255]}"(dv), "{s[40:43]}"(rK), "{s[44:47]}"(rV)
      :
#include "attention5_clobbers.inc"
  );

  // ---- normalise and store: lane owns query q0 + qb*32 + l31, d = dt*32 + 8g + 4hi + {0..3} (attention4.hip's tail, per query block) ----
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_run = __uint_as_float(lsum[qb]);
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    const int q = q0 + qb * 32 + l31;
    if (p.O8 != nullptr) {
      const size_t orow = (size_t)b * S + min(q, S - 1);
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        float v[16], amax = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = round_bf16(o[qb][dt][e] * inv);
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
    }
  }
  if (p.O8 == nullptr) {
    // bf16 output: O^T accumulators -> wave-private LDS image (64 rows x 256 B, 16-byte chunk c of row r at position c ^ (r & 15)) -> whole
    // rows, 16 bytes per lane: a store instruction covers 4 complete rows (the per-lane 8-byte stores of attention4.hip's tail touch 32
    // rows per instruction).  Behind the tile loop's last barrier no wave reads the K / V rings any more.
    a5_lds_char* const lds = (a5_lds_char*)0;
    const unsigned img = (unsigned)wave * 16384u;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float l_run = __uint_as_float(lsum[qb]);
      const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
      const int r = qb * 32 + l31;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          u32x2 w;
          w[0] = pack2bf(o[qb][dt][4 * g4 + 0] * inv, o[qb][dt][4 * g4 + 1] * inv);
          w[1] = pack2bf(o[qb][dt][4 * g4 + 2] * inv, o[qb][dt][4 * g4 + 3] * inv);
          *(__attribute__((address_space(3))) u32x2*)(lds + img + r * 256 + (((dt * 4 + g4) ^ (r & 15)) << 4) + hi * 8) = w;
        }
    }
    // (a wave reads back its own image: program order + the compiler's lgkmcnt suffice)
    bf16_t* const ob = p.O + ((size_t)b * S + q0) * p.ldo + head * D;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 4 * i + (lane >> 4), pos = lane & 15;
      const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)(lds + img + r * 256 + (pos << 4));
      if (q0 + r < S) *(u32x4*)(ob + (size_t)r * p.ldo + ((pos ^ (r & 15)) << 3)) = v;
    }
  }
}

bool dk_attention5_eligible(const AttnParams& p) {
  return p.D == 128 && p.bias == nullptr && p.S % 256 == 0 && p.S >= 12 * 64 && (size_t)p.S * p.ld * 2 < (1ull << 32);
}

template <bool QFUSE>
static int launch_attn5(const AttnParams& p, hipStream_t stream) {
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn5_fwd_kernel<QFUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, A5_LDS_BYTES));
    attr_once.mark();
  }
  const int nq = (p.S + 255) / 256;
  const long tasks = (long)nq * p.H * p.B;
  hipLaunchKernelGGL((dk_attn5_fwd_kernel<QFUSE>), dim3((unsigned)tasks), dim3(256), A5_LDS_BYTES, stream, p);
  return 0;
}

int dk_launch_attention5(const AttnParams& p, hipStream_t stream) {
  DK_REQUIRE(dk_attention5_eligible(p), "attention5: head_dim 128, no score bias, S a multiple of 256 and >= 768");
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  return qfuse ? launch_attn5<true>(p, stream) : launch_attn5<false>(p, stream);
}
