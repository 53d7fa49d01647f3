// Joint text/image attention forward for gfx950: entry point and kernel choice.
//
// Replaces the four mx.fast.scaled_dot_product_attention call sites of the reference
// (python/src/diffusionkit/mlx/mmdit.py:562,643,687,736): softmax(q k^T * scale) v, no mask, non-causal, over the concatenated
// [text, image] sequence.  The reference materialises the [H, S, S] score tensor in the activation dtype (quirk Q4); here scores
// stay fp32 in registers (flash-style, they never leave the CU).
//
// Layout: q/k/v are read in place from the token-major projection output (row stride ld, head h at column h*D), the output is
// written token-major so the o-projection GEMM consumes it directly -- no [B,H,S,D] transposes exist anywhere.
//
// Kernels: attention2.hip (VALU-lean, deferred rescale, 4 waves of 32 queries; the only one with a score bias), attention3.hip
// (two key tiles in flight per wave, 8 waves, D = 128; also the balanced stream-K-like launch) and attention4.hip (round 3: the two
// waves of a SIMD in opposite matrix / vector phases; D = 128; the default on long sequences: +12 % over attention3 on the FLUX
// shapes in isolation, 815 -> 932 TF inside the model).  Round 3 pruned the variants whose A/B is settled (profiles/archive/r01_attention_lab.md,
// r02_attn_bench.log): the lean kernel at 8 and 7 waves, the pipelined kernel at 4 waves and for D = 64 (842 TF lean against 773 / 823).
// Round 4: a D = 64 form with two query blocks per wave (half the LDS bytes per MFMA) measured +1.3 % isolated and +0.4 % / -0.9 % inside
// SD3-medium / SD3.5-large -- the lean kernel at D = 64 is bound by VALU issue, not by the LDS (profiles/r04_sd3_pmc.md,
// profiles/lab_kernels/attention5_two_query_blocks.hip); not built.  Nor is its pipelined form (next tile's score MFMAs under this tile's
// exponentials, a sched_group_barrier pipeline: parity-green, 6 % slower at 2 waves per SIMD; profiles/lab_kernels/attention2_pipelined.patch).
#include "dk_kernels.h"

extern int g_dk_attn_mode;  // dk_tune_set("attn", v): -1 automatic; 4 = dk_attn2 (4 waves); 7 = dk_attn3, 9 = dk_attn4 (8 waves, D = 128 only)

// Hand-off workspace of the balanced form of dk_attn3_fwd_kernel for the launches this host thread enqueues (an engine call sets
// it to its own engine's region, dk_attention_set_workspace to a caller's buffer; null = plain grids only)
static thread_local void* g_attn_ws = nullptr;
void dk_set_attention_workspace(void* ws) { g_attn_ws = ws; }
void* dk_get_attention_workspace() { return g_attn_ws; }

int dk_launch_attention(const AttnParams& p_in, hipStream_t stream) {
  AttnParams p = p_in;
  if (p.bal_ws == nullptr && g_attn_ws != nullptr) {
    p.bal_ws = g_attn_ws;
    p.bal_flags = (unsigned*)((char*)g_attn_ws + dk_attention_balance_workspace_bytes() - 4096);
  }
  DK_REQUIRE(p.D == 128 || p.D == 64, "head_dim must be 64 or 128");
  DK_REQUIRE(p.S > 0 && p.B > 0 && p.H > 0, "empty attention");
  DK_REQUIRE(p.ld % 8 == 0 && p.ldo % 4 == 0, "row strides must keep 16-byte alignment");
  // automatic choice (kernel lab, profiles/archive/r01_attention_lab.md, r02_attn_bench.log, r03_attention_phase_alternating.md): D = 128 on
  // long sequences: the phase-alternating kernel (8 waves per workgroup; 1018 / 1053 / 1078 TF against 908 / 961 / 978 for the
  // pipelined kernel on FLUX B1 / FLUX-dev B1 / FLUX B4, same box); otherwise the VALU-lean kernel with 4 waves (D = 64: 842 TF
  // against 773 / 823 for the pipelined forms).  A score bias (text encoders) is only implemented by the lean kernel's 4-wave form;
  // the balanced launch (dk_tune_set("attn_balance", 1)) belongs to the pipelined kernel: it needs "attn" = 7 as well
  const int mode = p.bias != nullptr ? 4 : g_dk_attn_mode < 0 ? ((p.D == 128 && p.S >= 2048) ? 9 : 4) : g_dk_attn_mode;
  dk_prof_begin(2, 4.0 * (double)p.B * p.H * (double)p.S * (double)p.S * p.D, stream);
  int rc = 0;
  switch (mode) {
    case 4: rc = dk_launch_attention2(p, 4, stream); break;
    case 7:  // software-pipelined kernel (attention3.hip); D = 64 has no such form: the lean kernel
      rc = p.D == 128 ? dk_launch_attention3(p, 8, stream) : dk_launch_attention2(p, 4, stream);
      break;
    case 9:  // phase-alternating kernel (attention4.hip); D = 128 only
      rc = p.D == 128 ? dk_launch_attention4(p, stream) : dk_launch_attention2(p, 4, stream);
      break;
    default: DK_REQUIRE(false, "unknown attention variant (4: lean kernel, 7: pipelined kernel, 9: phase-alternating kernel)");
  }
  dk_prof_end(stream);
  if (rc) return rc;
  DK_CHECK_HIP(hipGetLastError());
  if (p.O8 != nullptr && !((mode == 7 || mode == 9) && p.D == 128)) {
    // only the pipelined kernel writes the MX-fp8 copy itself: quantise the bf16 output behind the others
    Mx8Out o8{p.O8, p.O8_scales, p.o8_ld, p.o8_nblk, 0, p.B * p.S, 0, 0};
    return dk_launch_quantize_mx8(p.O, p.ldo, p.B * p.S, 0, p.B * p.S, p.H * p.D, o8, stream);
  }
  return 0;
}
