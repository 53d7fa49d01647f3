#!/usr/bin/env python3
"""Generator + CPU checker for the hand-scheduled body of the generation-4 bf16 GEMM (diffusionkit_amd/csrc/gemm256v4.hip).

The kernel runs ONE wave per SIMD (256 threads): a wave owns a 128 x 128 block of the 256 x 256 tile, its 256 accumulators live
in AGPRs a[0:255], and the whole main body -- prologue DMA, K loop, the two peeled last K-tiles and the drain of the accumulators
into the bf16 staging image -- is one inline-asm block with the register file addressed by hand.  This script writes that block
(gemm256v4_asm.inc, a C string literal) from an instruction list, and checks the SAME list on the CPU first:

  python scripts/gen_gemm256v4.py            write diffusionkit_amd/csrc/gemm256v4_asm.inc (the shipped schedule, VARIANTS[0])
  python scripts/gen_gemm256v4.py --lab      also write the measured-and-dropped schedule variants to profiles/lab_kernels/gemm256v4_variants/
  python scripts/gen_gemm256v4.py --check    also run the instruction-level emulator (4 waves x 64 lanes, numpy) on one 256 x 256 tile:
                                             every LDS-DMA piece lands at the LATEST moment its vmcnt wait allows (or at issue), every
                                             ds_read delivers at its lgkmcnt wait (or at issue), waves run in both orders between barriers
                                             -- a missing wait, a read of a ring slot before its barrier, or a fragment overwritten too
                                             early all show up as a wrong tile

Schedule (per K-tile of 64: 128 MFMAs 16x16x32, 32 ds_read_b128, 16 buffer_load ... lds, 4 barriers), after the shape of the vendor
library's MT256x256x64 kernel (studied in its disassembly, NOTES_r05): whole K = 32 slices of both operands read one slice ahead, each
operand's ring slot released by its own barrier as soon as its second slice has been read, the DMA pieces of K-tile i + 2 issued into
the slot just released (so a piece has 1.25 - 1.5 K-tiles to land), one non-MFMA instruction between two MFMAs.

Register map (per lane)              LDS (bytes)
  a[0:255]   acc[(nf*8 + mf)*4 + e]    X (activation) slot s: s * 32768            rows 0-255 x 128 B, chunk c of row r at c ^ ((r >> 1) & 7)
  v[0:7]     X piece offsets (in)      W (weight)     slot s: 65536 + s * 32768
  v[8:15]    W piece offsets (in)      drain image (bf16): virtual wave vw = wm*4 + wn: vw * 16384 + ni * 8192 + row * 64 + ...
  v[16:19]   fragment read addresses X kk0, X kk1, W kk0, W kk1 (in, copied to v[20:23])
  v[24:25]   drain addresses (in)
  v[32:63] XF0  v[64:95] XF1  v[96:127] WF0  v[128:159] WF1      (8 fragments x 4 registers each)
  v[160:191] bias of the lane's 32 columns (in)
  s[60:63] X resource  s[64:67] W resource (in);  s68 K byte offset  s69 K-tiles left  s70 / s71 DMA destination bases  s[72:73] alpha
"""
import os
import sys

import numpy as np

NF = 8
MFX = 8  # 16-row X (activation) fragments per wave: 8 = 256-row tiles, 7 = 224-row tiles (set by program(mf=...))
SLOT = 32768
W_BASE = 65536
XF = [32, 64]    # first register of X fragment set kk
WF = [96, 128]
RX = [20, 21]    # read-address registers (copies)
RW = [22, 23]
BIAS0 = 160


class I:
    """one instruction: text for the assembler + fields for the emulator"""

    def __init__(self, op, text, **kw):
        self.op, self.text, self.kw = op, text, kw

    def __getattr__(self, k):
        return self.kw[k]


def mfma(j, k, kk):
    a = (j * 8 + k) * 4
    return I("mfma", f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{WF[kk] + 4 * j}:{WF[kk] + 4 * j + 3}], v[{XF[kk] + 4 * k}:{XF[kk] + 4 * k + 3}], a[{a}:{a + 3}]",
             acc=a, wa=WF[kk] + 4 * j, xb=XF[kk] + 4 * k)


def ds_read(dst, addr, off):
    return I("ds_read", f"ds_read_b128 v[{dst}:{dst + 3}], v{addr} offset:{off}", dst=dst, addr=addr, off=off)


def dma(opnd, g):
    """piece g (0..7) of operand 'X' / 'W' of the K-tile at byte offset s68, into the slot s70 / s71 point at"""
    hh, j, u = g & 1, (g >> 1) & 1, g >> 2
    const = hh * 16384 + (u * 16 + j * 8) * 128
    base = 70 if opnd == "X" else 71
    rs = 60 if opnd == "X" else 64
    vo = (0 if opnd == "X" else 8) + g
    return [I("s_add", f"s_add_u32 m0, s{base}, {const}", dst="m0", a=base, imm=const),
            I("dma", f"buffer_load_dwordx4 v{vo}, s[{rs}:{rs + 3}], s68 offen lds", opnd=opnd, vo=vo)]


def salu(text, **kw):
    return I(kw.pop("op"), text, **kw)


def wait(vm=None, lgkm=None):
    parts = []
    if vm is not None:
        parts.append(f"vmcnt({vm})")
    if lgkm is not None:
        parts.append(f"lgkmcnt({lgkm})")
    return I("wait", "s_waitcnt " + " ".join(parts), vm=vm, lgkm=lgkm)


BARRIER = lambda: I("barrier", "s_barrier")  # noqa: E731


def drain_tile(nf8, mfi, slot):
    """bf16(alpha * acc + bias) of accumulator tile (nf8, mfi) into the staging image; temporaries v[32 + 4 * slot : +4]"""
    vwn, ni, nf = nf8 >> 2, (nf8 >> 1) & 1, nf8 & 1
    a = (nf8 * 8 + mfi) * 4
    r = 32 + (slot % 8) * 4
    out = [I("acc_read", f"v_accvgpr_read_b32 v{r + e}, a{a + e}", dst=r + e, src=a + e) for e in range(4)]
    b = BIAS0 + nf8 * 4
    out.append(I("pk_fma", f"v_pk_fma_f32 v[{r}:{r + 1}], s[72:73], v[{r}:{r + 1}], v[{b}:{b + 1}]", dst=r, x=r, b=b))
    out.append(I("pk_fma", f"v_pk_fma_f32 v[{r + 2}:{r + 3}], s[72:73], v[{r + 2}:{r + 3}], v[{b + 2}:{b + 3}]", dst=r + 2, x=r + 2, b=b + 2))
    out.append(I("cvt_pk", f"v_cvt_pk_bf16_f32 v{r}, v{r}, v{r + 1}", dst=r, lo=r, hi=r + 1))
    out.append(I("cvt_pk", f"v_cvt_pk_bf16_f32 v{r + 1}, v{r + 2}, v{r + 3}", dst=r + 1, lo=r + 2, hi=r + 3))
    off = vwn * 16384 + ni * 8192 + mfi * 1024
    out.append(I("ds_write", f"ds_write_b64 v{24 + nf}, v[{r}:{r + 1}] offset:{off}", addr=24 + nf, src=r, off=off))
    return out


def body(dma_on, next_on, outstanding_next, phase=None, nbar=4, drain=False):
    """one K-tile.  dma_on: issue the pieces of K-tile i + 2; next_on: read the first slice of K-tile i + 1 at the end;
    outstanding_next: own pieces of K-tile i + 1 that may still be in flight at the head of this body (16, X before W);
    phase: None = every wave issues its DMA pieces at the same MFMA slots; w = this is wave w's copy of the body, pieces 4 slots apart
    and shifted by w (the CU's four waves run in lockstep between barriers: pieces issued at the same slot queue up behind each other);
    nbar: 4 = one release and one landed barrier per operand, 2 = one release and one landed barrier for both."""
    S = 8 * MFX   # MFMAs per K = 32 slice
    T = 2 * S     # ... per K-tile
    mf = [mfma(j, k, 0) for j in range(8) for k in range(MFX)] + [mfma(j, k, 1) for j in range(8) for k in range(MFX)]
    slots = [[] for _ in range(T + 1)]  # slots[m] = instructions in front of MFMA m (T = behind the last one)

    def pos(p):
        """slot of the 128-MFMA schedule -> slot of this one: the first 52 slots (second-slice reads, release barriers) keep their place, the
        rest is compressed linearly (224-row tiles: 112 MFMAs per K-tile)"""
        if T == 128 or p <= 52:
            return p
        return 53 + ((p - 53) * (T - 53)) // (128 - 53)

    def put(m, ins):
        slots[pos(m)].extend(ins if isinstance(ins, list) else [ins])

    def put_dma(m, o, g):  # (SALU write of M0 -> LDS-DMA needs one wait state: the M0 write sits one MFMA ahead of its piece)
        a, b = dma(o, g)
        slots[pos(m) - 1].append(a)
        slots[pos(m)].append(b)

    put(0, I("v_xor", f"v_xor_b32 v{RX[1]}, 0x8000, v{RX[1]}", dst=RX[1], imm=0x8000))
    put(0, I("v_xor", f"v_xor_b32 v{RW[1]}, 0x8000, v{RW[1]}", dst=RW[1], imm=0x8000))
    waits = []  # (slot, operand) of the landed waits, filled in below
    if nbar == 4:
        for k in range(MFX):
            put(1 + 2 * k, ds_read(XF[1] + 4 * k, RX[1], k * 2048))
        put(20, wait(lgkm=0))
        put(21, BARRIER())  # every wave has read both slices of X(i): its slot is free
        for j in range(8):
            put(23 + 2 * j, ds_read(WF[1] + 4 * j, RW[1], j * 2048))
        put(51, wait(lgkm=0))
        put(52, BARRIER())  # W(i)'s slot is free
        if dma_on:
            if phase is None:
                xd = [22, 24, 26, 28, 30, 53, 55, 57]
                wd = [59, 61, 86, 88, 90, 98, 102, 124]
            else:
                xd = [23 + 4 * g + phase for g in range(8)]       # 23 .. 54
                wd = [55 + 4 * g + phase for g in range(8)]       # 55 .. 86  (+3)
            for g in range(8):  # (all X pieces first: where the compressed 224-row schedule puts an X and a W item into one slot, X's come first)
                put_dma(xd[g], "X", g)
            for g in range(8):
                put_dma(wd[g], "W", g)
        if next_on:
            put(66, I("v_xor", f"v_xor_b32 v{RX[0]}, 0x8000, v{RX[0]}", dst=RX[0], imm=0x8000))
            put(67, I("v_xor", f"v_xor_b32 v{RW[0]}, 0x8000, v{RW[0]}", dst=RW[0], imm=0x8000))
            waits.append((68, "X"))
            put(69, BARRIER())  # X(i + 1) has landed for every wave
            for k in range(MFX):
                put(70 + 2 * k, ds_read(XF[0] + 4 * k, RX[0], k * 2048))
            waits.append((105, "W"))
            put(106, BARRIER())  # W(i + 1) has landed
            for j in range(8):
                put(107 + j, ds_read(WF[0] + 4 * j, RW[0], j * 2048))
    else:
        assert MFX == 8
        for k in range(8):
            put(1 + 2 * k, ds_read(XF[1] + 4 * k, RX[1], k * 2048))
            put(2 + 2 * k, ds_read(WF[1] + 4 * k, RW[1], k * 2048))
        put(30, wait(lgkm=0))
        put(31, BARRIER())  # both slots of K-tile i are free
        if dma_on:
            ph = 0 if phase is None else phase
            step = 2 if phase is None else 4
            for g in range(8):
                put_dma(33 + 2 * step * g + ph, "X", g)             # X and W pieces alternate
                put_dma(33 + 2 * step * g + step + ph, "W", g)
        if next_on:
            put(94, I("v_xor", f"v_xor_b32 v{RX[0]}, 0x8000, v{RX[0]}", dst=RX[0], imm=0x8000))
            put(95, I("v_xor", f"v_xor_b32 v{RW[0]}, 0x8000, v{RW[0]}", dst=RW[0], imm=0x8000))
            waits.append((101, "W"))  # both operands of K-tile i + 1
            put(102, BARRIER())
            for k in range(8):
                put(103 + k, ds_read(XF[0] + 4 * k, RX[0], k * 2048))
                put(111 + k, ds_read(WF[0] + 4 * k, RW[0], k * 2048))
    if dma_on:  # advance the K offset and flip the destination slot for the next iteration (behind the last piece)
        slots[T].append(I("s_add", "s_add_u32 s68, s68, 128", dst=68, a=68, imm=128))
        slots[T].append(I("s_xor", "s_xor_b32 s70, s70, 0x8000", dst=70, imm=0x8000))
        slots[T].append(I("s_xor", "s_xor_b32 s71, s71, 0x8000", dst=71, imm=0x8000))
    # the landed waits: pieces of K-tile i + 2 issued so far in program order may stay in flight; for the X wait W(i + 1)'s 8 as well
    for m, o in waits:
        m = pos(m)
        assert not any(ins.op == "barrier" for ins in slots[m]), "a landed wait must sit in front of its barrier's slot"
        issued = sum(1 for mm in range(m + 1) for ins in slots[mm] if ins.op == "dma")
        keep = issued + (8 if (o == "X" and outstanding_next == 16) else 0)
        slots[m].append(wait(vm=keep))
    if drain:
        # last K-tile of the tile (no DMA, no next reads): behind its second release barrier every wave has read everything it will ever
        # read of the ring, so the staging image may be written; the second-slice MFMAs of weight fragment j (m = 64 + 8j ..) finish the
        # accumulator tiles (j, 0..7) -- they are drained under the MFMAs of fragment j + 1 (one tile per MFMA gap, >= 8 MFMAs behind the
        # one that wrote it; temporaries = the dead first-slice X registers).  Row 7 is drained behind the loop.
        assert not dma_on and not next_on and nbar == 4
        for j in range(7):
            for g in range(MFX):
                slots[S + MFX * (j + 1) + g].extend(drain_tile(j, g, g))
    out = []
    for m in range(T):
        out.extend(slots[m])
        out.append(mf[m])
    out.extend(slots[T])
    if next_on:
        out.append(wait(lgkm=0))  # the first slice of the next K-tile
    return out


def program(nbar=4, stagger=False, trace=False, drain_overlap=None, mf=8):
    global MFX
    MFX = mf
    assert mf == 8 or (mf == 7 and nbar == 4 and not stagger)
    if drain_overlap is None:
        drain_overlap = nbar == 4
    """trace (lab): s_memtime stamps in s[74:75] (first DMA piece issued), s[76:77] (first K-tile landed, barrier passed), s[78:79] (K loop
    done), s[80:81] (drain done) -- read back by the -DV4_TRACE build of gemm256v4.hip"""
    P = []

    def stamp(lo):
        return [I("nop", f"s_memtime s[{lo}:{lo + 1}]")] if trace else []
    # ---- inputs -> working registers
    for d, s in ((RX[0], 16), (RX[1], 17), (RW[0], 18), (RW[1], 19)):
        P.append(I("v_mov", f"v_mov_b32 v{d}, v{s}", dst=d, src=s))
    P.append(I("s_mov", "s_mov_b32 s68, %[koff]", dst=68, src="koff"))
    P.append(I("s_mov", "s_mov_b32 s69, %[nk]", dst=69, src="nk"))
    P.append(I("s_mov", "s_mov_b32 s70, %[dstx]", dst=70, src="dstx"))
    P.append(I("s_add", f"s_add_u32 s71, s70, {W_BASE}", dst=71, a=70, imm=W_BASE))
    P.append(I("s_mov", "s_mov_b32 s72, %[alpha]", dst=72, src="alpha"))
    P.append(I("s_mov", "s_mov_b32 s73, %[alpha]", dst=73, src="alpha"))

    def pro_dma(o, g):
        a, b = dma(o, g)
        return [a, I("nop", "s_nop 0"), b]

    def advance():
        return [I("s_add", "s_add_u32 s68, s68, 128", dst=68, a=68, imm=128),
                I("s_xor", "s_xor_b32 s70, s70, 0x8000", dst=70, imm=0x8000),
                I("s_xor", "s_xor_b32 s71, s71, 0x8000", dst=71, imm=0x8000)]
    # ---- prologue: K-tile 0 into slot 0, K-tile 1 (if any) into slot 1; the accumulators are zeroed while the pieces fly
    P.extend(stamp(74))
    for g in range(8):
        P.extend(pro_dma("X", g))
    for g in range(8):
        P.extend(pro_dma("W", g))
    P.append(I("s_cmp_eq", "s_cmp_eq_u32 s69, 1", a=69, imm=1))
    P.append(I("cbranch_scc1", "s_cbranch_scc1 10f", target="L10"))
    P.extend(advance())
    for g in range(8):
        P.extend(pro_dma("X", g))
    for g in range(8):
        P.extend(pro_dma("W", g))
    P.extend(advance())
    for a in range(256):
        P.append(I("acc_write", f"v_accvgpr_write_b32 a{a}, 0", dst=a))
    P.append(wait(vm=16))
    P.append(I("branch", "s_branch 11f", target="L11"))
    P.append(I("label", "10:", name="L10"))
    for a in range(256):
        P.append(I("acc_write", f"v_accvgpr_write_b32 a{a}, 0", dst=a))
    P.append(wait(vm=0))
    P.append(I("label", "11:", name="L11"))
    P.append(BARRIER())
    P.extend(stamp(76))
    # first slice of K-tile 0 (slot 0: the address registers point there); the kk1 registers are toggled at the head of every body,
    # so they start on slot 1
    P.append(I("v_xor", f"v_xor_b32 v{RX[1]}, 0x8000, v{RX[1]}", dst=RX[1], imm=0x8000))
    P.append(I("v_xor", f"v_xor_b32 v{RW[1]}, 0x8000, v{RW[1]}", dst=RW[1], imm=0x8000))
    for k in range(MFX):
        P.append(ds_read(XF[0] + 4 * k, RX[0], k * 2048))
    for j in range(8):
        P.append(ds_read(WF[0] + 4 * j, RW[0], j * 2048))
    P.append(wait(lgkm=0))
    # ---- K loop: nk - 2 full bodies, then the K-tile that issues no DMA, then the last one.  stagger: one copy of the loop per wave
    copies = [None] if not stagger else [0, 1, 2, 3]
    if stagger:
        for w in (1, 2, 3):
            P.append(I("s_cmp_eq", f"s_cmp_eq_u32 %[wave], {w}", a="wave", imm=w))
            P.append(I("cbranch_scc1", f"s_cbranch_scc1 {100 + w * 10}f", target=f"W{w}"))
    for w in copies:
        c = 0 if w is None else w
        L = lambda n: 100 + c * 10 + n  # noqa: E731
        if w is not None:
            P.append(I("label", f"{L(0)}:", name=f"W{c}"))
        P.append(I("s_cmp_lt", "s_cmp_lt_u32 s69, 3", a=69, imm=3))
        P.append(I("cbranch_scc1", f"s_cbranch_scc1 {L(2)}f", target=f"L21_{c}"))
        P.append(I("label", f"{L(1)}:", name=f"L20_{c}"))
        P.extend(body(True, True, 16, w, nbar))
        P.append(I("s_sub", "s_sub_u32 s69, s69, 1", dst=69, a=69, imm=1))
        P.append(I("s_cmp_gt", "s_cmp_gt_u32 s69, 2", a=69, imm=2))
        P.append(I("cbranch_scc1", f"s_cbranch_scc1 {L(1)}b", target=f"L20_{c}"))
        P.append(I("label", f"{L(2)}:", name=f"L21_{c}"))
        P.append(I("s_cmp_lt", "s_cmp_lt_u32 s69, 2", a=69, imm=2))
        P.append(I("cbranch_scc1", f"s_cbranch_scc1 {L(3)}f", target=f"L22_{c}"))
        P.extend(body(False, True, 16, w, nbar))
        P.append(I("label", f"{L(3)}:", name=f"L22_{c}"))
        P.extend(body(False, False, 0, w, nbar, drain=drain_overlap))
        if w is not None and w != copies[-1]:
            P.append(I("branch", "s_branch 90f", target="DRAIN"))
    P.append(I("label", "90:", name="DRAIN"))
    P.extend(stamp(78))
    # ---- drain: bf16(alpha * acc + bias) into the staging image.  With drain_overlap seven of the eight accumulator rows left under the
    # last K-tile's MFMAs; otherwise every wave is past the last body's release barriers and one more barrier frees the ring
    if not drain_overlap:
        P.append(BARRIER())
    n = 0
    for nf8 in range(7 if drain_overlap else 0, 8):
        for mfi in range(MFX):
            P.extend(drain_tile(nf8, mfi, n))
            n += 1
    P.append(wait(lgkm=0))
    P.extend(stamp(80))
    if trace:
        P.append(wait(lgkm=0))
    return P


# (barriers per K-tile, per-wave DMA phase).  VARIANTS[0] ships.  Measured (profiles/r05_gemm_v4_lab_variants.log, r05_gemm_v4_blas_cold.log): with
# weights that come from HBM -- every launch of the model -- variant 0 is the fastest on every shape; variant 3 (pieces staggered per wave, two
# barriers) wins by 1-5 % only when the weights are L2 / Infinity-Cache resident, variant 2 shows what un-staggered pieces cost there (-8 %)
VARIANTS = [(4, False), (4, True), (2, False), (2, True)]

CLOBBERS = [f"v{i}" for i in range(20, 24)] + [f"v{i}" for i in range(32, 160)] + [f"a{i}" for i in range(256)] + \
           [f"s{i}" for i in range(68, 74)] + ["m0", "scc", "memory"]


def emit(path, lab_dir=None):
    progs = []
    for v, (nbar, stagger) in enumerate(VARIANTS):
        P = program(nbar, stagger)
        progs.append(P)
        if v > 0 and lab_dir is None:
            continue
        n_mfma = sum(1 for i in P if i.op == "mfma")
        out = path if v == 0 else os.path.join(lab_dir, f"gemm256v4_asm{v}.inc")
        with open(out, "w") as f:
            f.write("// GENERATED by scripts/gen_gemm256v4.py -- do not edit; the CPU emulator in that script checks this instruction list.\n")
            f.write(f"// variant {v}: {nbar} barriers per K-tile, {'one loop copy per wave with its own DMA slots' if stagger else 'one loop for all waves'}; "
                    f"{len(P)} instructions, {n_mfma} MFMAs; explicit registers: see the script's header.\n")
            f.write("\n".join('    "' + ins.text + '\\n"' for ins in P) + "\n")
    if lab_dir is not None:  # the shipped schedule with s_memtime stamps (lab builds with -DV4_TRACE)
        Pt = program(*VARIANTS[0], trace=True)
        with open(os.path.join(lab_dir, "gemm256v4_asm_trace.inc"), "w") as f:
            f.write("// GENERATED by scripts/gen_gemm256v4.py --lab: variant 0 with s_memtime stamps in s[74:81]\n")
            f.write("\n".join('    "' + ins.text + '\\n"' for ins in Pt) + "\n")
    P7 = program(*VARIANTS[0], mf=7)
    progs.append(P7)
    with open(path.replace("_asm.inc", "_asm7.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_gemm256v4.py -- do not edit; the CPU emulator in that script checks this instruction list.\n")
        f.write(f"// the shipped schedule for 224-row tiles (7 activation fragments per wave, 112 MFMAs per K-tile); {len(P7)} instructions\n")
        f.write("\n".join('    "' + ins.text + '\\n"' for ins in P7) + "\n")
    with open(path.replace("_asm.inc", "_clobbers.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_gemm256v4.py\n")
        f.write(", ".join('"' + c + '"' for c in CLOBBERS) + "\n")
    return progs


# ------------------------------------------------------------------------------------------------------------------------------
# emulator
# ------------------------------------------------------------------------------------------------------------------------------
def bf16_round(x):
    """fp32 array -> bf16 bits (RNE) as uint32 (low 16 bits)"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u & 0xFFFF).astype(np.uint32)


def bf16_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


class Wave:
    def __init__(self, wid):
        self.wid = wid
        self.V = np.zeros((256, 64), np.uint32)
        self.A = np.zeros((256, 64), np.float32)
        self.S = {}
        self.m0 = 0
        self.scc = 0
        self.pc = 0
        self.vm = []    # outstanding DMA pieces: closures that land them
        self.lgkm = []  # outstanding LDS reads: closures that deliver


def run(P, nk, late, order, seed=0, verbose=False):
    rng = np.random.default_rng(seed)
    K = nk * 64
    lda, ldw = K + 64, K + 128  # padded pitches
    HR = 16 * MFX  # rows of a wave row (of a 128-row LDS half)
    Xf = bf16_to_f32(bf16_round(rng.standard_normal((2 * HR, lda)).astype(np.float32)))
    Wf = bf16_to_f32(bf16_round(rng.standard_normal((256, ldw)).astype(np.float32) * 0.25))
    bias = bf16_to_f32(bf16_round(rng.standard_normal(256).astype(np.float32)))
    alpha = np.float32(0.5)
    glob = {"X": bf16_round(Xf).astype(np.uint16).tobytes(), "W": bf16_round(Wf).astype(np.uint16).tobytes()}
    gl = {k: np.frombuffer(v, np.uint8) for k, v in glob.items()}
    lds = np.zeros(160 * 1024, np.uint8)
    labels = {ins.name: i for i, ins in enumerate(P) if ins.op == "label"}
    lane = np.arange(64)
    l15, q = lane & 15, lane >> 4
    srow = lane >> 3
    waves = []
    for w in range(4):
        wv = Wave(w)
        wm, wn2 = w >> 1, w & 1
        # piece offsets (host side of the kernel: gemm256v4.hip)
        for g in range(8):
            hh, j, u = g & 1, (g >> 1) & 1, g >> 2
            rih = (w * 2 + u) * 16 + j * 8 + srow  # row inside the 128-row LDS half
            row = hh * 128 + rih
            chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j)
            wv.V[g] = ((hh * HR + np.minimum(rih, HR - 1)) * lda + chunk * 8) * 2  # (224-row tiles: LDS rows 112-127 of a half hold duplicates nobody reads)
            wv.V[8 + g] = (row * ldw + chunk * 8) * 2
        for kk in range(2):
            offk = l15 * 128 + (((kk * 4 + q) ^ (l15 >> 1)) << 4)
            wv.V[16 + kk] = wm * 16384 + offk
            wv.V[18 + kk] = W_BASE + wn2 * 16384 + offk
        c = (l15 >> 2) & 3
        d0 = (wm * 4 + 2 * wn2) * 16384 + l15 * 64 + (q & 1) * 8 + ((((q >> 1)) ^ c) << 4)
        wv.V[24] = d0
        wv.V[25] = d0 ^ 32
        for nf8 in range(8):
            for e in range(4):
                wv.V[BIAS0 + nf8 * 4 + e] = bias[wn2 * 128 + nf8 * 16 + 4 * q + e].view(np.uint32)
        wv.S = {"koff": 0, "nk": nk, "dstx": w * 4096, "alpha": int(np.float32(alpha).view(np.uint32)), "wave": w}
        waves.append(wv)

    def land_all(wv, keep):
        while len(wv.vm) > keep:
            wv.vm.pop(0)()

    def deliver_all(wv, keep):
        while len(wv.lgkm) > keep:
            wv.lgkm.pop(0)()

    def step(wv):
        """run wave wv to its next barrier (returns True) or to the end (False)"""
        while wv.pc < len(P):
            ins = P[wv.pc]
            wv.pc += 1
            op = ins.op
            if op == "label" or op == "nop":
                continue
            if op == "barrier":
                return True
            if op == "mfma":
                Am = np.zeros((16, 32), np.float32)
                Bm = np.zeros((32, 16), np.float32)
                for r in range(4):
                    wa = wv.V[ins.wa + r]
                    xb = wv.V[ins.xb + r]
                    for half in range(2):
                        ka = 8 * q + 2 * r + half
                        Am[l15, ka] = bf16_to_f32((wa >> (16 * half)) & 0xFFFF)
                        Bm[ka, l15] = bf16_to_f32((xb >> (16 * half)) & 0xFFFF)
                D = Am @ Bm
                for e in range(4):
                    wv.A[ins.acc + e] += D[4 * q + e, l15]
            elif op == "ds_read":
                addr = wv.V[ins.addr].astype(np.int64) + ins.off
                idx = addr[:, None] + np.arange(16)[None, :]

                def sample(idx=idx):
                    return lds[idx].copy().view(np.uint32).reshape(64, 4)

                def deliver(data, dst=ins.dst):
                    for r in range(4):
                        wv.V[dst + r] = data[:, r]
                if late:  # sample at issue, deliver at the wait
                    data = sample()
                    wv.lgkm.append(lambda data=data, deliver=deliver: deliver(data))
                else:     # sample and deliver at the wait as well?  no: early = everything at issue
                    deliver(sample())
                    wv.lgkm.append(lambda: None)
            elif op == "ds_write":
                addr = wv.V[ins.addr].astype(np.int64) + ins.off
                data = np.stack([wv.V[ins.src], wv.V[ins.src + 1]], 1).copy().view(np.uint8).reshape(64, 8)
                lds[addr[:, None] + np.arange(8)[None, :]] = data
                wv.lgkm.append(lambda: None)
            elif op == "dma":
                g = gl[ins.opnd]
                src = wv.V[ins.vo].astype(np.int64) + wv.S[68]
                dst = wv.m0 + lane * 16
                sidx = src[:, None] + np.arange(16)[None, :]
                didx = dst[:, None] + np.arange(16)[None, :]

                def land(sidx=sidx, didx=didx, g=g):
                    lds[didx] = g[sidx]
                if late:
                    wv.vm.append(land)
                else:
                    land()
                    wv.vm.append(lambda: None)
            elif op == "wait":
                if ins.vm is not None:
                    land_all(wv, ins.vm)
                if ins.lgkm is not None:
                    deliver_all(wv, ins.lgkm)
            elif op == "s_mov":
                wv.S[ins.dst] = wv.S[ins.src]
            elif op == "s_add":
                v = (wv.S[ins.a] + ins.imm) & 0xFFFFFFFF
                if ins.dst == "m0":
                    wv.m0 = v
                else:
                    wv.S[ins.dst] = v
            elif op == "s_sub":
                wv.S[ins.dst] = (wv.S[ins.a] - ins.imm) & 0xFFFFFFFF
            elif op == "s_xor":
                wv.S[ins.dst] ^= ins.imm
            elif op == "s_cmp_eq":
                wv.scc = int(wv.S[ins.a] == ins.imm)
            elif op == "s_cmp_lt":
                wv.scc = int(wv.S[ins.a] < ins.imm)
            elif op == "s_cmp_gt":
                wv.scc = int(wv.S[ins.a] > ins.imm)
            elif op == "cbranch_scc1":
                if wv.scc:
                    wv.pc = labels[ins.target]
            elif op == "branch":
                wv.pc = labels[ins.target]
            elif op == "v_mov":
                wv.V[ins.dst] = wv.V[ins.src]
            elif op == "v_xor":
                wv.V[ins.dst] ^= np.uint32(ins.imm)
            elif op == "acc_write":
                wv.A[ins.dst] = 0
            elif op == "acc_read":
                wv.V[ins.dst] = wv.A[ins.src].view(np.uint32)
            elif op == "pk_fma":
                al = np.uint32(wv.S[72]).view(np.float32)
                for h in range(2):
                    x = wv.V[ins.x + h].view(np.float32)
                    b = wv.V[ins.b + h].view(np.float32)
                    wv.V[ins.dst + h] = (x.astype(np.float64) * np.float64(al) + b.astype(np.float64)).astype(np.float32).view(np.uint32)
            elif op == "cvt_pk":
                lo = bf16_round(wv.V[ins.lo].view(np.float32))
                hi = bf16_round(wv.V[ins.hi].view(np.float32))
                wv.V[ins.dst] = lo | (hi << 16)
            else:
                raise ValueError(op)
        return False

    n_bar = 0
    while True:
        alive = [step(wv) for wv in (waves if order == 0 else waves[::-1])]
        if not any(alive):
            break
        assert all(alive), "waves disagree on the barrier count"
        n_bar += 1
    for wv in waves:
        assert not wv.vm or all(True for _ in wv.vm), "DMA pieces outstanding at the end"
        land_all(wv, 0)
    # ---- read the staging image back the way the kernel's tail does and compare
    ref = (Xf[:, :K].astype(np.float64) @ Wf[:, :K].astype(np.float64).T) * float(alpha) + bias[None, :].astype(np.float64)
    got = np.zeros((2 * HR, 256), np.float32)
    for vw in range(8):
        wm, wn = vw >> 2, vw & 3
        for ni in range(2):
            reg0 = vw * 16384 + ni * 8192
            for row in range(HR):
                for ch in range(4):  # 16-byte chunk = 8 columns
                    a = reg0 + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)
                    vals = lds[a:a + 16].view(np.uint16).astype(np.uint32)
                    got[wm * HR + row, wn * 64 + ni * 32 + ch * 8: wn * 64 + ni * 32 + ch * 8 + 8] = bf16_to_f32(vals)
    err = np.abs(got - ref) / (np.abs(ref) + 1.0)
    ok = float(err.max()) < 1.2e-2
    if verbose or not ok:
        print(f"nk {nk} late {late} order {order}: max rel err {err.max():.3e}, barriers {n_bar}, {'ok' if ok else 'WRONG'}")
        if not ok:
            bad = np.argwhere(err > 1.2e-2)
            print("  first bad entries (row, col):", bad[:8].tolist(), " count", len(bad))
    return ok


if __name__ == "__main__":
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "diffusionkit_amd", "csrc", "gemm256v4_asm.inc")
    lab = os.path.join(root, "profiles", "lab_kernels", "gemm256v4_variants") if "--lab" in sys.argv else None
    if lab:
        os.makedirs(lab, exist_ok=True)
    progs = emit(path, lab)
    print(f"wrote {path} ({len(progs[0])} instructions)" + (f" and {len(progs) - 1} lab variants under {lab}" if lab else ""))
    if "--check" in sys.argv:
        allok = True
        for v, P in enumerate(progs):
            MFX = 7 if v == len(progs) - 1 else 8  # (the last one is the 224-row body)
            for nk in (1, 2, 3, 4, 6):
                for late in (True, False):
                    for order in (0, 1):
                        allok &= run(P, nk, late, order, seed=nk, verbose="-v" in sys.argv)
            print(f"variant {v}: {'ok' if allok else 'FAILED'}", flush=True)
        print("ALL OK" if allok else "FAILED")
        sys.exit(0 if allok else 1)
