"""Codegen guards (CPU; reads the gfx950 code objects the build left under diffusionkit_amd/csrc/build/): the hot kernels must stay free of
register spills and scratch.  Round 4 found two defects of this kind by reading the ISA -- a parameter-block select that had become a
scratch copy in the fp8 GEMM (105 scratch_load sites, -7..12 % per launch) and D = 128 attention kernels spilling under a scheduler
strategy meant for their D = 64 siblings -- neither of which any numerical test can see."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "diffusionkit_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernels on the denoise / decode path (mangled-name fragments)
HOT = ["dk_gemm256v4_kernel", "dk_gemm256v3_kernel", "dk_gemm256f8_kernel", "dk_attn5_fwd_kernel", "dk_attn4_fwd_kernel", "dk_attn2_fwd_kernel", "dk_attn512_fwd_kernel", "dk_conv_halo_kernel", "dk_conv256v4_kernel",
       "dk_ln_modulate_kernel", "dk_rows_to_mx8_kernel", "dk_euler_step_kernel", "dk_qk_norm_rope_kernel"]


def kernel_metadata(obj):
    """{kernel name: {vgpr, sgpr, spill, scratch, lds}} from the code object's notes."""
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={t}/fb.bin", obj, f"{t}/copy.o"], capture_output=True)
        if r.returncode != 0:
            return {}  # host-only object (no device code)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={t}/fb.bin",
                               f"--output={t}/k.co", "--unbundle"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{t}/k.co"], text=True)
    out, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            cur[k] = v
        if k == "wavefront_size" and "name" in cur:  # last field of a kernel's record
            out[cur["name"]] = {kk: int(vv) for kk, vv in cur.items() if kk != "name"}
            cur = {}
    return out


@pytest.mark.skipif(not (os.path.isdir(BUILD) and glob.glob(os.path.join(BUILD, "*.o")) and shutil.which(f"{LLVM}/llvm-readelf")),
                    reason="needs the built objects (make -C diffusionkit_amd/csrc) and the ROCm LLVM tools")
def test_hot_kernels_have_no_spills_and_no_scratch():
    seen, bad = set(), []
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        for name, md in kernel_metadata(obj).items():
            hot = next((h for h in HOT if h in name), None)
            if hot is None:
                continue
            seen.add(hot)
            if hot == "dk_attn5_fwd_kernel" and "ILb0E" in name:
                # the instantiation without the fused query transform (no model path takes it: lab / ops.attention only) keeps three registers
                # in scratch ACROSS its asm block, which owns every VGPR: one store in front of the tile loop, one load behind it
                assert md.get("private_segment_fixed_size", 0) <= 16, (name, md)
                continue
            if md.get("vgpr_spill_count", 0) or md.get("sgpr_spill_count", 0) or md.get("private_segment_fixed_size", 0):
                bad.append((os.path.basename(obj), name, md))
            if hot in ("dk_gemm256v4_kernel", "dk_conv256v4_kernel", "dk_attn5_fwd_kernel"):  # one wave per SIMD by design: the 256 accumulators in AGPRs beside at most 256 VGPRs
                assert md["vgpr_count"] <= 512, (name, md)
            else:
                assert md["vgpr_count"] <= 256, (name, md)  # two waves per SIMD at least
    assert not bad, "spills / scratch in hot kernels:\n" + "\n".join(f"{o}: {n}: {m}" for o, n, m in bad)
    assert seen == set(HOT), f"hot kernels not found in the build: {sorted(set(HOT) - seen)}"


def test_gemm256v4_asm_body_is_the_generators_and_passes_the_emulator():
    """The hand-scheduled body of gemm256v4.hip is GENERATED (scripts/gen_gemm256v4.py): the committed include file must be what the
    generator writes, and the instruction list must pass the generator's CPU emulator -- 4 waves x 64 lanes, every LDS-DMA piece and LDS
    read completing as late as its s_waitcnt allows or at issue, waves running in both orders between barriers -- on one tile with 1, 2, 3
    and 5 K-tiles (peeled bodies only / one pass of the loop / odd count).  A schedule edit that drops a wait or a barrier fails here,
    without a GPU."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_gemm256v4 as gen
    P = gen.program(*gen.VARIANTS[0])
    text = "\n".join('    "' + ins.text + '\\n"' for ins in P) + "\n"
    committed = open(os.path.join(ROOT, "diffusionkit_amd", "csrc", "gemm256v4_asm.inc")).read()
    assert committed.split("\n", 2)[2] == text, "gemm256v4_asm.inc is stale: run python scripts/gen_gemm256v4.py"
    for nk in (1, 2, 3, 5):
        for late in (True, False):
            for order in (0, 1):
                assert gen.run(P, nk, late, order, seed=nk), (nk, late, order)
    # ... and the body for 224-row tiles (7 activation fragments per wave, 112 MFMAs per K-tile)
    P7 = gen.program(*gen.VARIANTS[0], mf=7)
    text7 = "\n".join('    "' + ins.text + '\\n"' for ins in P7) + "\n"
    committed7 = open(os.path.join(ROOT, "diffusionkit_amd", "csrc", "gemm256v4_asm7.inc")).read()
    assert committed7.split("\n", 2)[2] == text7, "gemm256v4_asm7.inc is stale: run python scripts/gen_gemm256v4.py"
    for nk in (1, 2, 3, 4):
        for late in (True, False):
            assert gen.run(P7, nk, late, nk & 1, seed=nk), (7, nk, late)


def test_conv256v4_asm_bodies_are_the_generators_and_pass_the_emulator():
    """conv256v4.hip's two asm bodies (GroupNorm + SiLU on the way into the halo / plain input) are GENERATED (scripts/gen_conv256v4.py): the
    committed include files must be what the generator writes, and the instruction lists must pass its CPU emulator on a corner tile (every
    border is padding; two 64-channel chunks: one pass of the chunk loop + the peeled last chunk) with every memory instruction landing as
    late as its wait allows, or at issue, and the waves in both orders.  Then the emulator itself is checked: with the wait in front of the
    halo transform weakened, the late run must come out wrong."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_conv256v4 as gen
    for xform, cfg, name in ((True, 0, "x"), (False, 0, "p"), (True, 1, "x128"), (False, 1, "p128")):  # cfg 1: the 128-channel tile form
        P = gen.program(xform, cfg)
        text = "\n".join('    "' + ins.text + '\\n"' for ins in P) + "\n"
        committed = open(os.path.join(ROOT, "diffusionkit_amd", "csrc", f"conv256v4_asm_{name}.inc")).read()
        assert committed.split("\n", 2)[2] == text, f"conv256v4_asm_{name}.inc is stale: run python scripts/gen_conv256v4.py"
        for late in (True, False):
            assert gen.run(P, xform, (0, 0), late, int(late), C=128, HWimg=32, seed=5, cfg=cfg), (xform, cfg, late)
    P = gen.program(True)
    halo_waits = [i for i in P if i.op == "wait" and "need" in i.kw and i.need[0] == "H"]
    assert len(halo_waits) >= 2  # prologue + loop body
    for w in halo_waits:
        w.kw["vm"] = 63
    assert not gen.run(P, True, (0, 0), True, 0, C=128, HWimg=32, seed=5), "the emulator did not notice a halo transform running ahead of its loads"


def test_attention5_asm_body_is_the_generators_and_passes_the_emulator():
    """attention5.hip's tile loop is GENERATED (scripts/gen_attn5.py): the committed include files must be what the generator writes, and
    the instruction list must pass the instruction-level emulator (scripts/attn5_emu.py: 4 waves x 64 lanes, MFMA 32x32x16, the transposing
    LDS read, LDS-DMA pieces landing as late as the waits allow or at issue) against an fp64 softmax(Q K^T) V -- twelve key tiles (one pass of
    the four-tile loop + the eight peeled tiles), with keys that lift a row's maximum in the middle of the sequence (the deferred rescale:
    factor recorded at the decision, applied when the previous tile's P.V is complete).  Then the emulator itself is checked: with the
    landed-wait of the K / V rings weakened, the late run must come out wrong."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import attn5_emu as emu
    import gen_attn5 as gen
    assert gen.ABL == 0 and not gen.OPT
    P = gen.program()
    k = [i for i, x in enumerate(P) if x.op == "split"][0]
    for part, name in ((P[:k], "attention5_dma.inc"), (P[k + 1:], "attention5_asm.inc")):
        text = "\n".join('    "' + ins.text + '\\n"' for ins in part) + "\n"
        committed = open(os.path.join(ROOT, "diffusionkit_amd", "csrc", name)).read()
        assert committed.split("\n", 2)[2] == text, f"{name} is stale: run python scripts/gen_attn5.py"
    for late in (True, False):
        assert emu.run(P, 12, late, int(late), seed=12, spike=True), late
    P = gen.program()
    ring_waits = [i for i in P if i.op == "wait" and "need" in i.kw and isinstance(i.need, tuple)]
    assert len(ring_waits) >= 12
    for w in ring_waits:
        w.kw["vm"] = 63
    assert not emu.run(P, 12, True, 0, seed=12, spike=True), "the emulator did not notice fragment reads running ahead of the DMA pieces"
