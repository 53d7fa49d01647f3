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
  // Jobs: workgroups 0 .. a5_whole - 1 take whole query blocks; the others key range `part` of a5_split of one of the remaining blocks (the
  // launch's last, partial round of the CUs -- dk_launch_attention5).  Inside each class the blocks follow each other on an XCD.
  auto xcd_contiguous = [](int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  };
  int t, part = 0, job = -1;
  if ((int)blockIdx.x < p.a5_whole) {
    t = xcd_contiguous(blockIdx.x, p.a5_whole);
  } else {
    job = xcd_contiguous(blockIdx.x - p.a5_whole, gridDim.x - p.a5_whole);
    t = p.a5_whole + job / p.a5_split;
    part = job % p.a5_split;
  }
  const unsigned row_bytes = (unsigned)p.ld * 2u;
  const int qblock = t % nq, head = (t / nq) % p.H, b = t / (nq * p.H);
  const int q0 = qblock * 256 + wave * 64;
  // key tiles of this job: the S / 64 tiles in groups of four, dealt to the parts as evenly as possible
  const int ng = S / 256, nparts = job < 0 ? 1 : p.a5_split;
  const int g_lo = (part * ng) / nparts, g_hi = ((part + 1) * ng) / nparts;
  const int tile0 = 4 * g_lo, nt = 4 * (g_hi - g_lo);

  const bf16_t* Qb = p.Q + (size_t)b * S * p.ld + head * D;
  const char* Kb = (const char*)(p.K + ((size_t)b * S + (size_t)tile0 * 64) * p.ld + head * D);
  const char* Vb = (const char*)(p.V + ((size_t)b * S + (size_t)tile0 * 64) * p.ld + head * D);

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
  // the DMA pieces of the first K tile leave BEFORE the query rows are fetched: one memory round trip for what the first scores need
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

  const int scale_bits = __float_as_int(p.scale);  // (the block forms c = scale * log2(e): p = 2^(s*c - m*c); rescale threshold 4 as in attention4.hip)

  f32x16 o[2][4];
  u32x2 lsum, mcv;
  asm volatile(
#include "attention5_asm.inc"
      : "={v[0:15]}"(o[0][0]), "={v[16:31]}"(o[0][1]), "={v[32:47]}"(o[0][2]), "={v[48:63]}"(o[0][3]), "={v[64:79]}"(o[1][0]), "={v[80:95]}"(o[1][1]),
        "={v[96:111]}"(o[1][2]), "={v[112:127]}"(o[1][3]), "={v[226:227]}"(lsum), "={v[224:225]}"(mcv), "+{v[128:143]}"(qv[0]), "+{v[144:159]}"(qv[1]), "+{v[160:175]}"(qv[2]),
        "+{v[176:191]}"(qv[3])
      : [koff] "s"(64 * (int)row_bytes), [voff] "s"(0), [tileb] "s"(64 * (int)row_bytes), [scale] "s"(scale_bits), [ntrip] "s"((nt - 8) / 4), [dbase] "s"(wave * 4096), "{v[238:245]}"(kaddr),
        "{v[246:247]}"(vaddr), "{v[248:251]}"(dk), "{v[252:255]}"(dv), "{s[40:43]}"(rK), "{s[44:47]}"(rV)
      :
#include "attention5_clobbers.inc"
  );

  // ---- normalise and store: lane owns query q0 + qb*32 + l31, d = dt*32 + 8g + 4hi + {0..3} (attention4.hip's tail, per query block) ----
  // (the block above owns every VGPR: whatever lane-dependent value the tail needs is formed again from a FRESH lane id -- values carried
  //  across the block would be spilled to scratch in front of it)
  int lane_t;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
  {
    const int lane = lane_t, hi = lane >> 5, l31 = lane & 31;
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
    // (a key-range job leaves its 256 x 128 block of O / l in the workspace, rows of 256 B, and per row the exponent offset and l)
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
    const int njobs = (int)gridDim.x - p.a5_whole;
    bf16_t* const ob = job < 0 ? p.O + ((size_t)b * S + q0) * p.ldo + head * D : (bf16_t*)p.a5_ws + ((size_t)job * 256 + wave * 64) * D;
    const size_t ldo = job < 0 ? (size_t)p.ldo : (size_t)D;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 4 * i + (lane >> 4), pos = lane & 15;
      const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)(lds + img + r * 256 + (pos << 4));
      if (q0 + r < S) *(u32x4*)(ob + (size_t)r * ldo + ((pos ^ (r & 15)) << 3)) = v;
    }
    if (job >= 0) {
      float* st = (float*)((char*)p.a5_ws + (size_t)njobs * 65536) + ((size_t)job * 256 + wave * 64) * 2;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float l_run = __uint_as_float(lsum[qb]);
        const float l_row = l_run + __shfl_xor(l_run, 32, 64);  // (every lane takes part in the exchange)
        if (hi == 0) *(f32x2*)(st + (qb * 32 + l31) * 2) = f32x2{__uint_as_float(mcv[qb]), l_row};
      }
    }
  }
  }
}

// the key ranges of one query block -> its rows of the output: O = sum_i w_i O_i / sum_i w_i with w_i = l_i 2^(mc_i - max mc); a thread owns
// 8 columns of a row
__global__ __launch_bounds__(256) void dk_attn5_merge_kernel(AttnParams p, int njobs) {
  constexpr int D = 128;
  const int blk = blockIdx.x >> 4, row = (blockIdx.x & 15) * 16 + (threadIdx.x >> 4), c8 = threadIdx.x & 15;
  const int nq = (p.S + 255) / 256;
  const int t = p.a5_whole + blk;
  const int qblock = t % nq, head = (t / nq) % p.H, b = t / (nq * p.H);
  const int q = qblock * 256 + row;
  const float* st = (const float*)((const char*)p.a5_ws + (size_t)njobs * 65536);
  float mc[4], l[4], mx = -3.0e38f;
  for (int i = 0; i < p.a5_split; ++i) {
    const f32x2 s2 = *(const f32x2*)(st + ((size_t)(blk * p.a5_split + i) * 256 + row) * 2);
    mc[i] = s2[0], l[i] = s2[1];
    mx = fmaxf(mx, mc[i]);
  }
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
  for (int i = 0; i < p.a5_split; ++i) {
    const float w = l[i] * __builtin_amdgcn_exp2f(mc[i] - mx);
    const u32x4 v = *(const u32x4*)((const bf16_t*)p.a5_ws + ((size_t)(blk * p.a5_split + i) * 256 + row) * D + c8 * 8);
    wsum += w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo, hi2;
      unpack2bf(v[e], lo, hi2);
      acc[2 * e] += w * lo, acc[2 * e + 1] += w * hi2;
    }
  }
  const float inv = 1.0f / wsum;
  if (q < p.S) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e] * inv, acc[2 * e + 1] * inv);
    *(u32x4*)(p.O + ((size_t)b * p.S + q) * p.ldo + head * D + c8 * 8) = o;
  }
}

int g_dk_attn5_split = -1;  // dk_tune_set("attn_split", v): -1 automatic (needs the workspace), 0 never, 2 .. 4 that many key ranges for the last round's blocks

bool dk_attention5_eligible(const AttnParams& p) {
  return p.D == 128 && p.bias == nullptr && p.S % 256 == 0 && p.S >= 12 * 64 && (size_t)p.S * p.ld * 2 < (1ull << 32);
}

int dk_launch_attention5(const AttnParams& p_in, hipStream_t stream) {
  DK_REQUIRE(dk_attention5_eligible(p_in), "attention5: head_dim 128, no score bias, S a multiple of 256 and >= 768");
  AttnParams p = p_in;
  const bool qfuse = p.qn_a != nullptr || p.q_rope != nullptr;
  if (qfuse) DK_REQUIRE(p.qn_a == nullptr || p.qn_b != nullptr, "qn_b missing (pass qn_a twice for one weight)");
  const int n_cu = dk_device_cu_count();
  // One workgroup per CU: a launch of nb blocks runs in ceil(nb / n_cu) rounds, the last one with nb % n_cu blocks.  Those blocks are split
  // into s key ranges each (every range a multiple of four tiles, at least twelve) when that shortens the last round: it then takes
  // ceil(tail * s / n_cu) / s of a block's time.  The partial results go through the workspace and dk_attn5_merge_kernel.
  const int nb = ((p.S + 255) / 256) * p.H * p.B;
  const int tail = nb % n_cu;
  int split = 1;
  if (tail > 0 && p.O8 == nullptr && g_dk_attn5_split != 0) {
    // (measured, profiles/r05_attention5_lab.log: a workgroup costs ~14 us + 1.66 us per tile, and a last round on 152 of 256 CUs runs faster
    //  than a full one -- FLUX, one image, gains nothing from three ranges in two sub-rounds; a tail that fits the CUs in ONE sub-round does:
    //  batch 4: 96 blocks x 2)
    for (int s = 2; s <= 4 && split == 1; ++s) {
      if ((p.S / 256) / s < 3) break;  // >= 12 tiles per range
      if (g_dk_attn5_split > 0 ? s == g_dk_attn5_split : (tail * s <= n_cu && tail * s * 10 >= n_cu * 6)) split = s;
    }
    void* ws = dk_get_attention_workspace();
    if (split > 1 && (ws == nullptr || dk_get_attention_workspace_bytes() < (size_t)tail * split * (65536 + 2048))) split = 1;
    p.a5_ws = ws;
  }
  p.a5_split = split;
  p.a5_whole = split > 1 ? nb - tail : nb;
  const int njobs = split > 1 ? tail * split : 0;
  static DkDeviceOnce attr_once;
  if (attr_once.first()) {
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn5_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, A5_LDS_BYTES));
    DK_CHECK_HIP(hipFuncSetAttribute((const void*)dk_attn5_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, A5_LDS_BYTES));
    attr_once.mark();
  }
  if (qfuse)
    hipLaunchKernelGGL((dk_attn5_fwd_kernel<true>), dim3((unsigned)(p.a5_whole + njobs)), dim3(256), A5_LDS_BYTES, stream, p);
  else
    hipLaunchKernelGGL((dk_attn5_fwd_kernel<false>), dim3((unsigned)(p.a5_whole + njobs)), dim3(256), A5_LDS_BYTES, stream, p);
  if (njobs > 0) hipLaunchKernelGGL(dk_attn5_merge_kernel, dim3((unsigned)(tail * 16)), dim3(256), 0, stream, p, njobs);
  return 0;
}
