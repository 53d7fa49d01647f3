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
// Kernels: attention2.hip (VALU-lean, deferred rescale, 4 waves of 32 queries; D = 64, short sequences, and the only one with a score
// bias) and attention4.hip (the two waves of a SIMD in opposite matrix / vector phases; D = 128, the default on long sequences).
// Round 5 moved what no default path takes to profiles/lab_kernels/ (README there): the software-pipelined D = 128 kernel with its
// balanced stream-K-like launch (attention3_pipelined.hip: 908 / 961 / 978 TF against attention4's 1018 / 1053 / 1078 on FLUX B1 /
// FLUX-dev B1 / FLUX B4), the D = 64 form with two query blocks per wave and the pipelined D = 64 patch (rounds 2-4: all within +-6 %
// of the lean kernel, which is bound by VALU issue, profiles/r04_sd3_pmc.md).
#include "dk_kernels.h"

extern int g_dk_attn_mode;  // engine.hip; dk_tune_set("attn", v): -1 automatic; 4 = dk_attn2 (4 waves); 9 = dk_attn4 (8 waves, D = 128 only);
                            // 10 = dk_attn5 (one wave per SIMD, asm tile loop; D = 128, S % 256 == 0: other shapes fall back to 9)

// workspace of the launches this host thread enqueues (dk_attention_set_workspace): the partial results of attention5.hip's key-split
// workgroups; lab: the trace buffer of attention4.hip's DK4_TRACE builds (scripts/attn_trace.py)
static thread_local void* g_attn_ws = nullptr;
static thread_local size_t g_attn_ws_bytes = 0;
void dk_set_attention_workspace(void* ws, size_t bytes) { g_attn_ws = ws, g_attn_ws_bytes = ws ? bytes : 0; }
void* dk_get_attention_workspace() { return g_attn_ws; }
size_t dk_get_attention_workspace_bytes() { return g_attn_ws_bytes; }

int dk_launch_attention(const AttnParams& p_in, hipStream_t stream) {
  AttnParams p = p_in;
  if (p.bal_ws == nullptr) p.bal_ws = g_attn_ws;
  DK_REQUIRE(p.D == 128 || p.D == 64, "head_dim must be 64 or 128");
  DK_REQUIRE(p.S > 0 && p.B > 0 && p.H > 0, "empty attention");
  DK_REQUIRE(p.ld % 8 == 0 && p.ldo % 4 == 0, "row strides must keep 16-byte alignment");
  // automatic choice (kernel lab, profiles/archive/r01_attention_lab.md, r02_attn_bench.log, r03_attention_phase_alternating.md): D = 128 on
  // long sequences: the phase-alternating kernel; otherwise the VALU-lean kernel with 4 waves (D = 64: 842 TF against 773 / 823 for the
  // pipelined forms).  A score bias (text encoders) is only implemented by the lean kernel
  // (round 6, profiles/r06_attention_short_sequences.log: at FLUX's 512 x 512 sequence, S = 1280, the one-wave-per-SIMD kernel wins 16 - 26 % on batches -- 240+
  //  workgroups: a round of the CUs -- and ties on one image in the lab, where the model, with the fused query prologue, measured it 1.6 % per step
  //  behind the lean kernel: below 2048 tokens it takes the launches that fill at least three quarters of a round; at S = 768 the two tie)
  const long blocks5 = (long)p.B * p.H * ((p.S + 255) / 256);
  const bool long5 = p.D == 128 && (p.S >= 2048 || (p.S >= 1024 && blocks5 * 4 >= 3L * dk_device_cu_count()));
  int mode = p.bias != nullptr ? 4 : g_dk_attn_mode < 0 ? (long5 ? 10 : 4) : g_dk_attn_mode;
  if (mode == 10 && !dk_attention5_eligible(p)) mode = 9;
  dk_prof_begin(2, 4.0 * (double)p.B * p.H * (double)p.S * (double)p.S * p.D, stream);
  int rc = 0;
  switch (mode) {
    case 4: rc = dk_launch_attention2(p, 4, stream); break;
    case 9:  // phase-alternating kernel (attention4.hip); D = 128 only
      rc = p.D == 128 ? dk_launch_attention4(p, stream) : dk_launch_attention2(p, 4, stream);
      break;
    case 10: rc = dk_launch_attention5(p, stream); break;  // one wave per SIMD (attention5.hip)
    default: DK_REQUIRE(false, "unknown attention variant (4: lean kernel, 9: phase-alternating kernel, 10: one-wave-per-SIMD kernel)");
  }
  dk_prof_end(stream);
  if (rc) return rc;
  DK_CHECK_HIP(hipGetLastError());
  if (p.O8 != nullptr && !((mode == 9 || mode == 10) && p.D == 128)) {
    // only the D = 128 kernels write the MX-fp8 copy themselves: quantise the bf16 output behind the others
    Mx8Out o8{p.O8, p.O8_scales, p.o8_ld, p.o8_nblk, 0, p.B * p.S, 0, 0};
    return dk_launch_quantize_mx8(p.O, p.ldo, p.B * p.S, 0, p.B * p.S, p.H * p.D, o8, stream);
  }
  return 0;
}
