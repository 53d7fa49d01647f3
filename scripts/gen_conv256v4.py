#!/usr/bin/env python3
"""Generator + CPU checker for the hand-scheduled body of the one-wave-per-SIMD 3x3 convolution (diffusionkit_amd/csrc/conv256v4.hip).

The frame is gemm256v4's (scripts/gen_gemm256v4.py: 256 threads, a wave owns 128 pixels x 128 output channels of a 16 x 16 pixel x 256
channel tile, 256 accumulators in AGPRs, whole K = 32 slices of both operands read one slice ahead, the weight K-tiles through a two-slot
LDS-DMA ring).  What is new is the activation operand: it is not a K-tile stream but the 18 x 18 pixel HALO of a 64-channel chunk, loaded
once per chunk through registers (GroupNorm-apply + SiLU on the way, the arithmetic of conv_halo.hip), kept in one of two LDS slots, and
read by all nine taps of the chunk as shifted windows (an immediate offset per tap and pixel row).  The halo of chunk c + 1 is fetched at
the first tap of chunk c and transformed + stored one instruction per MFMA gap over taps 1 .. 8.

  python scripts/gen_conv256v4.py            write diffusionkit_amd/csrc/conv256v4_asm_x.inc (GroupNorm + SiLU prologue), _asm_p.inc (plain input),
                                             conv256v4_clobbers.inc
  python scripts/gen_conv256v4.py --check    also run the instruction-level emulator (4 waves x 64 lanes, numpy) on a corner tile and an inner
                                             tile of a small image: every memory instruction completes as LATE as its waits allow (or at issue),
                                             waves run in both orders between barriers

Per K-tile (tap t of chunk c): 128 MFMAs 16x16x32, 32 ds_read_b128, 8 buffer_load ... lds (the weight K-tile two ahead), 2 barriers (the weight
slot released / the next weight K-tile landed).  Per chunk: 11 + 4 buffer_load (halo items, scale / shift of the lane's 8 channels), ~70 VALU
per item, 11 ds_write_b128.

Register map (per lane)                                           LDS (bytes)
  a[0:255]   acc[(nf*8 + mf)*4 + e]                                 halo slot s: s * 46656: 324 rows (18 x 18 pixels) x 144 B (128 B of channels + 16 B pad)
  v[0:10]    halo item offsets (in)   v11 item-valid bits (in)      bias table (fp32, 256 channels): 93312
  v[12:19]   weight piece offsets (in)                              weight slot s: 98304 / 131072 (toggle: xor 0x38000), rows 0-255 x 128 B, swizzled
  v20 / v21  halo window base for the kk = 0 / kk = 1 reads         drain image (bf16): gemm256v4's, 0 .. 131071
  v22 / v23  weight fragment read addresses kk0 / kk1               v[24:25] drain addresses   v28 mask temp   v30 halo write base
  v[240:248] (in): halo window base, weight read addresses kk0 / kk1, halo write base | drain addresses, scale / shift table offsets | bias read base
  v[32:63] XF0  v[64:95] XF1  v[96:127] WF0  v[128:159] WF1
  v[160:203] raw halo items (11 x 4); v[160:191] = bias of the lane's 32 columns during the drain
  v[204:211] scale  v[212:219] shift of the lane's 8 channels   v[220:235] transform temporaries
  s[56:59] scale/shift table  s[60:63] input  s[64:67] weights (in);  s68 weight K byte offset  s69 chunks left  s70 halo chunk offset  s71 DMA base
  s[72:73] 1.0  s74 table chunk offset  s75 bytes per tap (2 C)  s76 128 - 16 C  s77 +-46656  s[78:79] lanes that hold an eleventh halo item
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm256v4 import I, BARRIER, Wave, bf16_round, bf16_to_f32, drain_tile, ds_read, mfma, wait  # noqa: E402

H_ROWB = 144
H_SLOT = 324 * H_ROWB  # 46656
BIAS_LDS = 2 * H_SLOT  # 93312
W_BASE = 98304
W_TOG = 0x38000
HO, HMASK, WP = 0, 11, 12
RXA, RXB = 20, 21
RW = [22, 23]
GV = [246, 247]
MT, HW, BA = 28, 30, 248
IN_RX, IN_RW, IN_HW, IN_DR = 240, 241, 243, 244  # inputs the block copies into working registers (an asm input may not be modified)
XF = [32, 64]
WF = [96, 128]
R0, SC, SH, TA = 160, 204, 212, 220
NITEM = 11
VM_OPS = ("dma", "vload")
# Two tile forms.  CFG 0: 16 x 16 pixels x 256 channels, waves 2 x 2 of 128 x 128 (8 pixel-row fragments), weight K-tiles of 256 rows in a two-slot
# ring (xor toggle).  CFG 1 (128-channel stages): 16 x 16 pixels x 128 channels, waves 4 x 1 of 64 pixels x 128 channels (4 pixel-row fragments,
# 64 MFMAs per K-tile), weight K-tiles of 128 rows in a THREE-slot ring: nine taps per chunk = slot tap % 3 as an immediate, no toggles, and the
# pieces of K-tile t + 3 leave right behind the barrier that releases K-tile t's slot (two K-tiles to land: a K-tile is half as long here).
CFG = 0
MFX = 8


def xoff(tap, k, kk):
    dy, dx = divmod(tap, 3)
    return ((k + dy) * 18 + dx) * H_ROWB + kk * 64


def wdma(g, tag, slot=0):
    if CFG == 1:  # piece g (0..3) of a wave: 8 rows of the 128-row K-tile, ring slot `slot`
        const = slot * 16384 + g * 1024
    else:
        hh, j, u = g & 1, (g >> 1) & 1, g >> 2
        const = hh * 16384 + (u * 16 + j * 8) * 128
    return [I("s_add", f"s_add_u32 m0, s71, {const}", dst="m0", a=71, imm=const),
            I("dma", f"buffer_load_dwordx4 v{WP + g}, s[64:67], s68 offen lds", opnd="W", vo=WP + g, tag=tag)]


def vload(dst, voff, rs, soff, imm, src, tag):
    o = f" offset:{imm}" if imm else ""
    return I("vload", f"buffer_load_dwordx4 v[{dst}:{dst + 3}], v{voff}, s[{rs}:{rs + 3}], s{soff} offen{o}", dst=dst, voff=voff, src=src, soff=soff,
             imm=imm, tag=tag)


def need(tag):
    """wait until every memory instruction up to the last one tagged `tag` has landed; the count is filled in by resolve()"""
    return I("wait", None, vm=None, lgkm=None, need=tag)


def halo_loads(xform, tag):
    out = []
    if xform:
        out.append(vload(SC, GV[0], 56, 74, 0, "G", tag))
        out.append(vload(SC + 4, GV[0], 56, 74, 16, "G", tag))
        out.append(vload(SH, GV[1], 56, 74, 0, "G", tag))
        out.append(vload(SH + 4, GV[1], 56, 74, 16, "G", tag))
    for i in range(NITEM):
        out.append(vload(R0 + 4 * i, HO + i, 60, 70, 0, "X", tag))
    return out


def halo_advance():
    return [I("s_add", "s_add_u32 s70, s70, 128", dst=70, a=70, imm=128), I("s_add", "s_add_u32 s74, s74, 256", dst=74, a=74, imm=256)]


def item_stream(i, xform):
    """raw item i -> (GroupNorm-apply -> bf16 -> SiLU -> bf16, zero where the pixel is padding) -> LDS.  Units: lists of instructions that stay
    together in one MFMA gap.  The four channel pairs of the item advance in lockstep: >= 3 instructions between dependent ones (a
    transcendental's result needs one wait state before its first use)."""
    r = R0 + 4 * i
    units = []
    if xform:
        units.append([I("v_bfe_i32", f"v_bfe_i32 v{MT}, v{HMASK}, {i}, 1", dst=MT, src=HMASK, bit=i)])
        st = []
        for p in range(4):
            A, T, rp = TA + 4 * p, TA + 4 * p + 2, r + p
            unpack = lambda: [I("v_lshl16", f"v_lshlrev_b32 v{A}, 16, v{rp}", dst=A, src=rp),  # noqa: E731
                              I("v_andhi", f"v_and_b32 v{A + 1}, 0xffff0000, v{rp}", dst=A + 1, src=rp)]
            pack = lambda: I("cvt_pk", f"v_cvt_pk_bf16_f32 v{rp}, v{A}, v{A + 1}", dst=rp, lo=A, hi=A + 1)  # noqa: E731
            seq = unpack()
            # (scalar forms: a packed f32 VALU beside MFMAs costs ~22 cycles more than the two scalar instructions it replaces -- MI355X_MICROARCH.md)
            for h in range(2):
                seq.append(I("v_fma", f"v_fma_f32 v{A + h}, v{SC + 2 * p + h}, v{A + h}, v{SH + 2 * p + h}", dst=A + h, a=SC + 2 * p + h, b=A + h, c=SH + 2 * p + h))
            seq.append(pack())
            seq.extend(unpack())
            for h in range(2):
                seq.append(I("v_mul_lit", f"v_mul_f32 v{T + h}, 0xbfb8aa3b, v{A + h}", dst=T + h, src=A + h))
            for h in range(2):
                seq.append(I("v_exp", f"v_exp_f32 v{T + h}, v{T + h}", dst=T + h))
            for h in range(2):
                seq.append(I("v_add1", f"v_add_f32 v{T + h}, 1.0, v{T + h}", dst=T + h))
            for h in range(2):
                seq.append(I("v_rcp", f"v_rcp_f32 v{T + h}, v{T + h}", dst=T + h))
            for h in range(2):
                seq.append(I("v_mul", f"v_mul_f32 v{A + h}, v{T + h}, v{A + h}", dst=A + h, a=T + h, b=A + h))
            seq.append(pack())
            seq.append(I("v_and", f"v_and_b32 v{rp}, v{rp}, v{MT}", dst=rp, a=rp, b=MT))
            st.append(seq)
        for s in range(len(st[0])):
            for p in range(4):
                units.append([st[p][s]])
    wr = I("ds_write128", f"ds_write_b128 v{HW}, v[{r}:{r + 3}] offset:{i * 32 * H_ROWB}", addr=HW, src=r, off=i * 32 * H_ROWB)
    if i == NITEM - 1:  # 2592 items = 10 x 256 + 32: only lanes 0-31 of wave 0 hold an eleventh
        units.append([I("s_mov_exec", "s_mov_b64 exec, s[78:79]", src=78), wr, I("s_mov_exec", "s_mov_b64 exec, -1", src=None)])
    else:
        units.append([wr])
    return units


def halo_stream(xform):
    out = []
    for i in range(NITEM):
        out.extend(item_stream(i, xform))
    return out


def w_advance(tap_issued):
    """behind the pieces of the weight K-tile of tap `tap_issued`: the K offset moves on to the next tap (or to tap 0 of the next chunk), the ring flips"""
    adv = [I("s_add_s", f"s_add_u32 s68, s68, s{76 if tap_issued == 8 else 75}", dst=68, a=68, b=76 if tap_issued == 8 else 75)]
    if CFG == 0:
        adv.append(I("s_xor", f"s_xor_b32 s71, s71, {W_TOG:#x}", dst=71, imm=W_TOG))
    return adv


def ktile1(tap, wtile, dma_on, next_on, extra=None, drain=False):
    """CFG 1: 64 MFMAs (8 weight fragments x 4 pixel-row fragments x 2 slices); weight ring slot = tap % 3; the pieces of K-tile wtile + 3 go
    into the slot this K-tile releases"""
    S, T = 32, 64
    mf = [mfma(j, k, 0) for j in range(8) for k in range(4)] + [mfma(j, k, 1) for j in range(8) for k in range(4)]
    slots = [[] for _ in range(T + 1)]

    def put(m, ins):
        slots[m].extend(ins if isinstance(ins, list) else [ins])

    ws = (tap % 3) * 16384
    for k in range(4):
        put(1 + k, ds_read(XF[1] + 4 * k, RXB, xoff(tap, k, 1)))
    for j in range(8):
        put(5 + j, ds_read(WF[1] + 4 * j, RW[1], ws + j * 2048))
    if drain:
        for n in range(8):
            put(13 + n, ds_read(R0 + 4 * n, BA, n * 64))
    put(21, wait(lgkm=0))
    put(22, BARRIER())  # every wave has read both slices of W(wtile): its slot is free (tap 8: and every wave's halo stores of the next chunk are done)
    if dma_on:
        for g in range(4):
            a, b = wdma(g, ("W", wtile + 3), tap % 3)
            slots[24 + 4 * g - 1].append(a)
            slots[24 + 4 * g].append(b)
    for m, ins in sorted((extra or {}).items()):
        put(m, ins)
    if next_on:
        ntap = (tap + 1) % 9
        nws = (ntap % 3) * 16384
        for k in range(4):
            put(32 + 2 * k, ds_read(XF[0] + 4 * k, RXA, xoff(ntap, k, 0)))
        put(42, need(("W", wtile + 1)))
        put(43, BARRIER())  # W(wtile + 1) has landed for every wave
        for j in range(8):
            put(44 + j, ds_read(WF[0] + 4 * j, RW[0], nws + j * 2048))
    if dma_on:
        slots[T].extend(w_advance((tap + 3) % 9))
    if drain:
        for j in range(7):
            for g in range(4):
                slots[S + 4 * (j + 1) + g].extend(drain_tile(j, g, g))
    out = []
    for m in range(T):
        out.extend(slots[m])
        out.append(mf[m])
    out.extend(slots[T])
    if next_on:
        out.append(wait(lgkm=0))
    return out


def ktile(tap, wtile, dma_on, next_on, extra=None, drain=False):
    """K-tile `wtile` = tap `tap` of the current chunk; issues the pieces of K-tile wtile + 2, pre-reads the first slice of wtile + 1.
    extra: {slot: [instructions]} of the chunk-level work (halo loads, transform stream, slot toggles)"""
    if CFG == 1:
        return ktile1(tap, wtile, dma_on, next_on, extra, drain)
    S, T = 64, 128
    mf = [mfma(j, k, 0) for j in range(8) for k in range(8)] + [mfma(j, k, 1) for j in range(8) for k in range(8)]
    slots = [[] for _ in range(T + 1)]

    def put(m, ins):
        slots[m].extend(ins if isinstance(ins, list) else [ins])

    put(0, I("v_xor", f"v_xor_b32 v{RW[1]}, {W_TOG:#x}, v{RW[1]}", dst=RW[1], imm=W_TOG))
    for k in range(8):
        put(1 + 2 * k, ds_read(XF[1] + 4 * k, RXB, xoff(tap, k, 1)))
    for j in range(8):
        put(17 + 2 * j, ds_read(WF[1] + 4 * j, RW[1], j * 2048))
    if drain:  # the bias of the lane's 32 columns, before the barrier behind which the staging image may overwrite the table
        for n in range(8):
            put(33 + n, ds_read(R0 + 4 * n, BA, n * 64))
    put(43, wait(lgkm=0))
    put(44, BARRIER())  # every wave has read both slices of W(wtile): its slot is free (tap 8: and every wave's halo stores of the next chunk are done)
    if dma_on:
        for g in range(8):
            a, b = wdma(g, ("W", wtile + 2))
            slots[46 + 7 * g - 1].append(a)
            slots[46 + 7 * g].append(b)
    for m, ins in sorted((extra or {}).items()):
        put(m, ins)
    if next_on:
        ntap = (tap + 1) % 9
        for k in range(8):
            put(64 + 2 * k, ds_read(XF[0] + 4 * k, RXA, xoff(ntap, k, 0)))
        put(100, I("v_xor", f"v_xor_b32 v{RW[0]}, {W_TOG:#x}, v{RW[0]}", dst=RW[0], imm=W_TOG))
        put(104, need(("W", wtile + 1)))
        put(105, BARRIER())  # W(wtile + 1) has landed for every wave
        for j in range(8):
            put(106 + j, ds_read(WF[0] + 4 * j, RW[0], j * 2048))
    if dma_on:
        slots[T].extend(w_advance((tap + 2) % 9))
    if drain:
        for j in range(7):
            for g in range(8):
                slots[S + 8 * (j + 1) + g].extend(drain_tile(j, g, g))
    out = []
    for m in range(T):
        out.extend(slots[m])
        out.append(mf[m])
    out.extend(slots[T])
    if next_on:
        out.append(wait(lgkm=0))
    return out


def chunk_body(n, xform, last):
    """chunk n of the loop (last: the peeled last chunk: no halo work, the last two K-tiles issue no pieces, the last one drains)"""
    out = []
    stream = [] if last else halo_stream(xform)
    # positions of the stream: (tap 1, slot 47) .. (tap 8, slot 30), one unit per MFMA gap, evenly spread
    if CFG == 1:  # (two units per MFMA gap on average: 576 gaps per chunk for the same halo)
        pos = [(1, m) for m in range(24, 64)] + [(t, m) for t in range(2, 8) for m in range(64)] + [(8, m) for m in range(15)]
        if not xform:
            pos = [(1, m) for m in range(24, 64, 3)][:len(stream)]
    else:
        pos = [(1, m) for m in range(47, 128)] + [(t, m) for t in range(2, 8) for m in range(128)] + [(8, m) for m in range(31)]
        if not xform:
            pos = [(1, m) for m in range(47, 128, 4)][:len(stream)]
        assert len(stream) <= len(pos)
    where = {}
    for s, u in enumerate(stream):
        t, m = pos[(s * len(pos)) // len(stream)]
        where.setdefault(t, {}).setdefault(m, []).extend(u)
    for tap in range(9):
        extra = where.get(tap, {})
        if not last:
            if tap == 0:
                loads = halo_loads(xform, ("H", n + 1))
                if CFG == 1:
                    at = ([24, 26, 28, 30] if xform else []) + [32 + 2 * i for i in range(NITEM)]  # behind the weight pieces of this K-tile
                else:
                    at = ([2, 4, 6, 8] if xform else []) + [10 + 3 * i for i in range(NITEM)]  # scale / shift first, then the items: 10, 13, ..., 40
                for m, ld in zip(at, loads):
                    extra.setdefault(m, []).append(ld)
                extra.setdefault(56 if CFG == 1 else 42, []).extend(halo_advance())
            if tap == 1:
                extra.setdefault(23 if CFG == 1 else 45, []).insert(0, need(("H", n + 1)))
            if tap == 8:  # the slots change roles: kk1 window base behind its last reads, write base behind the last store, kk0 base in front of the pre-read
                q = (6, 16, 31, 63) if CFG == 1 else (16, 32, 62, 127)
                extra.setdefault(q[0], []).append(I("v_add_s", f"v_add_u32 v{RXB}, s77, v{RXB}", dst=RXB, s=77))
                extra.setdefault(q[1], []).append(I("v_subrev_s", f"v_subrev_u32 v{HW}, s77, v{HW}", dst=HW, s=77))
                extra.setdefault(q[2], []).append(I("v_add_s", f"v_add_u32 v{RXA}, s77, v{RXA}", dst=RXA, s=77))
                extra.setdefault(q[3], []).append(I("s_neg", "s_sub_u32 s77, 0, s77", dst=77))
        dma_on = not (last and tap >= (6 if CFG == 1 else 7))
        next_on = not (last and tap == 8)
        out.extend(ktile(tap, 9 * n + tap, dma_on, next_on, extra, drain=last and tap == 8))
    return out


def prologue(xform):
    P = []
    for d, s in ((RXA, IN_RX), (RXB, IN_RX), (RW[0], IN_RW), (RW[1], IN_RW + 1), (HW, IN_HW), (24, IN_DR), (25, IN_DR + 1)):
        P.append(I("v_mov", f"v_mov_b32 v{d}, v{s}", dst=d, src=s))
    for d, s in ((68, "0"), (69, "%[nloop]"), (70, "0"), (71, "%[dstw]"), (72, "0x3f800000"), (73, "0x3f800000"), (74, "0"), (75, "%[tapb]"),
                 (76, "%[wrapb]"), (77, str(H_SLOT)), (78, "%[m10]"), (79, "0")):
        P.append(I("s_mov", f"s_mov_b32 s{d}, {s}", dst=d, src=s))
    P.extend(halo_loads(xform, ("H", 0)))
    P.extend(halo_advance())
    for t in range(3 if CFG == 1 else 2):
        for g in range(4 if CFG == 1 else 8):
            a, b = wdma(g, ("W", t), t)
            P.extend([a, I("nop", "s_nop 0"), b])
        P.extend(w_advance(t + 1))
    for a in range(256):
        P.append(I("acc_write", f"v_accvgpr_write_b32 a{a}, 0", dst=a))
    P.append(need(("H", 0)))
    for u in halo_stream(xform):
        P.extend(u)
    P.append(I("v_add_s", f"v_add_u32 v{HW}, s77, v{HW}", dst=HW, s=77))
    P.append(need(("W", 0)))
    P.append(wait(lgkm=0))
    P.append(BARRIER())
    if CFG == 0:
        P.append(I("v_xor", f"v_xor_b32 v{RW[1]}, {W_TOG:#x}, v{RW[1]}", dst=RW[1], imm=W_TOG))
    for k in range(MFX):
        P.append(ds_read(XF[0] + 4 * k, RXA, xoff(0, k, 0)))
    for j in range(8):
        P.append(ds_read(WF[0] + 4 * j, RW[0], j * 2048))
    P.append(wait(lgkm=0))
    return P


def resolve(seq, prefix):
    """fill the vmcnt of every need() wait of seq: the memory instructions issued behind the last one of the wanted tag may stay in flight"""
    hist = [i.tag for i in prefix if i.op in VM_OPS]
    counts = []
    for ins in seq:
        if ins.op in VM_OPS:
            hist.append(ins.tag)
        elif ins.op == "wait" and "need" in ins.kw:
            idx = [k for k, t in enumerate(hist) if t == ins.need]
            assert idx, f"nothing tagged {ins.need} in front of its wait"
            counts.append(len(hist) - 1 - idx[-1])
    return counts


def set_waits(seq, counts):
    it = iter(counts)
    for ins in seq:
        if ins.op == "wait" and "need" in ins.kw:
            c = next(it)
            assert 0 <= c <= 63
            ins.kw["vm"] = c
            ins.text = f"s_waitcnt vmcnt({c})"


def program(xform=True, cfg=0):
    global CFG, MFX
    CFG, MFX = cfg, (4 if cfg == 1 else 8)
    pro = prologue(xform)
    set_waits(pro, resolve(pro, []))
    # the loop body serves its first iteration (behind the prologue) and every later one (behind itself): the stricter count of the two
    b0, b1 = chunk_body(0, xform, False), chunk_body(1, xform, False)
    c0, c1 = resolve(b0, pro), resolve(b1, b0)
    set_waits(b0, [min(x, y) for x, y in zip(c0, c1)])
    last = chunk_body(1, xform, True)
    set_waits(last, resolve(last, pro + b0))
    P = list(pro)
    P.append(I("label", "20:", name="LOOP"))
    P.extend(b0)
    P.append(I("s_sub", "s_sub_u32 s69, s69, 1", dst=69, a=69, imm=1))
    P.append(I("s_cmp_gt", "s_cmp_gt_u32 s69, 0", a=69, imm=0))
    P.append(I("cbranch_scc1", "s_cbranch_scc1 20b", target="LOOP"))
    P.extend(last)
    n = 0
    for mfi in range(MFX):  # accumulator row 7 (the other seven left under the last K-tile's MFMAs)
        P.extend(drain_tile(7, mfi, n))
        n += 1
    P.append(wait(lgkm=0))
    return P


CLOBBERS = [f"v{i}" for i in range(20, 26)] + ["v28", "v30"] + [f"v{i}" for i in range(32, 236)] + [f"a{i}" for i in range(256)] + \
           [f"s{i}" for i in range(68, 80)] + ["m0", "scc", "memory"]


def emit(csrc):
    progs = {}
    for xform, cfg, name in ((True, 0, "x"), (False, 0, "p"), (True, 1, "x128"), (False, 1, "p128")):
        P = program(xform, cfg)
        progs[(xform, cfg)] = P
        if cfg == 0:
            progs[xform] = P
        n_mfma = sum(1 for i in P if i.op == "mfma")
        with open(os.path.join(csrc, f"conv256v4_asm_{name}.inc"), "w") as f:
            f.write("// GENERATED by scripts/gen_conv256v4.py -- do not edit; the CPU emulator in that script checks this instruction list.\n")
            f.write(f"// {'256' if cfg == 0 else '128'}-channel tiles, {'GroupNorm-apply + SiLU on the way into the halo' if xform else 'plain input (no transform)'}: {len(P)} instructions, "
                    f"{n_mfma} MFMAs; explicit registers: see the script's header.\n")
            f.write("\n".join('    "' + ins.text + '\\n"' for ins in P) + "\n")
    with open(os.path.join(csrc, "conv256v4_clobbers.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_conv256v4.py\n")
        f.write(", ".join('"' + c + '"' for c in CLOBBERS) + "\n")
    return progs


# ------------------------------------------------------------------------------------------------------------------------------
# emulator
# ------------------------------------------------------------------------------------------------------------------------------
def f32(u):
    return np.asarray(u, np.uint32).view(np.float32)


def u32(f):
    return np.asarray(f, np.float32).view(np.uint32)


def t_affine(a, sc, sh):  # v_fma_f32: one rounding
    return (a.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)


def t_silu(g):
    with np.errstate(over="ignore"):
        t = np.exp2((g * f32(np.uint32(0xbfb8aa3b))).astype(np.float32)).astype(np.float32)
        t = (np.float32(1.0) + t).astype(np.float32)
        t = (np.float32(1.0) / t).astype(np.float32)
    return (t * g).astype(np.float32)


def run(P, xform, tile, late, order, C=192, HWimg=48, ups=0, seed=0, verbose=False, cfg=0):
    """one workgroup: pixel tile `tile` = (ty, tx) of an HWimg x HWimg image, output channels 0 .. 255 (cfg 1: 0 .. 127)"""
    rng = np.random.default_rng(seed)
    nch = C // 64
    ldw = 9 * C + 8
    Hs = HWimg >> ups
    NO = 128 if cfg == 1 else 256
    x = bf16_to_f32(bf16_round(rng.standard_normal((Hs, Hs, C)).astype(np.float32)))
    Wf = bf16_to_f32(bf16_round(rng.standard_normal((NO, ldw)).astype(np.float32) * 0.05))
    bias = bf16_to_f32(bf16_round(rng.standard_normal(NO).astype(np.float32)))
    gss = np.stack([rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.3]).astype(np.float32)
    gl = {"X": np.frombuffer(bf16_round(x).astype(np.uint16).tobytes(), np.uint8),
          "W": np.frombuffer(bf16_round(Wf).astype(np.uint16).tobytes(), np.uint8),
          "G": np.frombuffer(gss.tobytes(), np.uint8)}
    nrec = {"X": gl["X"].size, "G": gl["G"].size}
    lds = np.zeros(160 * 1024, np.uint8)
    lds[BIAS_LDS:BIAS_LDS + 4 * NO] = np.frombuffer(bias.astype(np.float32).tobytes(), np.uint8)
    labels = {ins.name: i for i, ins in enumerate(P) if ins.op == "label"}
    lane = np.arange(64)
    l15, q = lane & 15, lane >> 4
    srow = lane >> 3
    ty, tx = tile
    waves = []
    for w in range(4):
        wv = Wave(w)
        wv.exec = np.ones(64, bool)
        wm, wn2 = w >> 1, w & 1
        tid = w * 64 + lane
        c8 = tid & 7
        okbits = np.zeros(64, np.uint32)
        for i in range(NITEM):
            idn = tid + 256 * i
            hrow = idn >> 3
            hy, hx = hrow // 18, hrow % 18
            y, xx = ty * 16 - 1 + hy, tx * 16 - 1 + hx
            ok = (idn < 2592) & (y >= 0) & (y < HWimg) & (xx >= 0) & (xx < HWimg)
            off = (((y >> ups) * Hs + (xx >> ups)) * C * 2 + c8 * 16).astype(np.int64)
            wv.V[HO + i] = np.where(ok, off, 0x80000000).astype(np.uint32)
            okbits |= ok.astype(np.uint32) << i
        wv.V[HMASK] = okbits
        if cfg == 1:
            wm, wn2 = w, 0
            for g in range(4):
                row = (w * 4 + g) * 8 + srow
                chunk = (lane & 7) ^ ((row >> 1) & 7)
                wv.V[WP + g] = (row * ldw + chunk * 8) * 2
        else:
            for g in range(8):
                hh, j, u = g & 1, (g >> 1) & 1, g >> 2
                row = hh * 128 + (w * 2 + u) * 16 + j * 8 + srow
                chunk = (lane & 7) ^ (srow >> 1) ^ (4 * j)
                wv.V[WP + g] = (row * ldw + chunk * 8) * 2
        lanex = (wm * MFX * 18 + l15) * H_ROWB + q * 16
        wv.V[IN_RX] = lanex
        for kk in range(2):
            wv.V[IN_RW + kk] = W_BASE + wn2 * 16384 + l15 * 128 + (((kk * 4 + q) ^ (l15 >> 1)) << 4)
        c = (l15 >> 2) & 3
        d0 = ((2 * w) if cfg == 1 else (wm * 4 + 2 * wn2)) * 16384 + l15 * 64 + (q & 1) * 8 + ((((q >> 1)) ^ c) << 4)
        wv.V[IN_DR] = d0
        wv.V[IN_DR + 1] = d0 ^ 32
        wv.V[GV[0]] = c8 * 32
        wv.V[GV[1]] = c8 * 32 + C * 4
        wv.V[IN_HW] = (tid >> 3) * H_ROWB + c8 * 16
        wv.V[BA] = BIAS_LDS + (wn2 * 128 + 4 * q) * 4
        wv.S = {"%[nloop]": nch - 1, "%[dstw]": W_BASE + w * 4096, "%[tapb]": 2 * C, "%[wrapb]": (128 - 16 * C) & 0xFFFFFFFF,
                "%[m10]": 0xFFFFFFFF if w == 0 else 0}
        waves.append(wv)

    def land_all(wv, keep):
        while len(wv.vm) > keep:
            wv.vm.pop(0)()

    def deliver_all(wv, keep):
        while len(wv.lgkm) > keep:
            wv.lgkm.pop(0)()

    def sval(wv, s):
        if isinstance(s, str):
            return wv.S[s] if s.startswith("%") else int(s, 0)
        return wv.S[s]

    def step(wv):
        while wv.pc < len(P):
            ins = P[wv.pc]
            wv.pc += 1
            op = ins.op
            V = wv.V
            if op in ("label", "nop"):
                continue
            if op == "barrier":
                return True
            if op == "mfma":
                Am = np.zeros((16, 32), np.float32)
                Bm = np.zeros((32, 16), np.float32)
                for r in range(4):
                    wa, xb = V[ins.wa + r], V[ins.xb + r]
                    for half in range(2):
                        ka = 8 * q + 2 * r + half
                        Am[l15, ka] = bf16_to_f32((wa >> (16 * half)) & 0xFFFF)
                        Bm[ka, l15] = bf16_to_f32((xb >> (16 * half)) & 0xFFFF)
                D = Am @ Bm
                for e in range(4):
                    wv.A[ins.acc + e] += D[4 * q + e, l15]
            elif op == "ds_read":
                addr = V[ins.addr].astype(np.int64) + ins.off
                data = lds[addr[:, None] + np.arange(16)[None, :]].copy().view(np.uint32).reshape(64, 4)

                def deliver(data=data, dst=ins.dst):
                    for r in range(4):
                        wv.V[dst + r] = data[:, r]
                if late:
                    wv.lgkm.append(deliver)
                else:
                    deliver()
                    wv.lgkm.append(lambda: None)
            elif op in ("ds_write", "ds_write128"):
                nreg = 2 if op == "ds_write" else 4
                addr = V[ins.addr].astype(np.int64) + ins.off
                data = np.stack([V[ins.src + r] for r in range(nreg)], 1).copy().view(np.uint8).reshape(64, 4 * nreg)
                m = wv.exec
                lds[(addr[:, None] + np.arange(4 * nreg)[None, :])[m]] = data[m]
                wv.lgkm.append(lambda: None)
            elif op == "dma":
                src = V[ins.vo].astype(np.int64) + wv.S[68]
                dst = wv.m0 + lane * 16
                sidx = src[:, None] + np.arange(16)[None, :]
                didx = dst[:, None] + np.arange(16)[None, :]

                def land(sidx=sidx, didx=didx):
                    lds[didx] = gl["W"][sidx]
                if late:
                    wv.vm.append(land)
                else:
                    land()
                    wv.vm.append(lambda: None)
            elif op == "vload":
                voff = V[ins.voff].astype(np.int64) + ins.imm
                inb = voff + 16 <= nrec[ins.src]
                a = np.where(inb, voff + wv.S[ins.soff], 0)
                data = gl[ins.src][a[:, None] + np.arange(16)[None, :]].copy().view(np.uint32).reshape(64, 4)
                data[~inb] = 0

                def landv(data=data, dst=ins.dst):
                    for r in range(4):
                        wv.V[dst + r] = data[:, r]
                if late:
                    wv.vm.append(landv)
                else:
                    landv()
                    wv.vm.append(lambda: None)
            elif op == "wait":
                if ins.vm is not None:
                    land_all(wv, ins.vm)
                if ins.lgkm is not None:
                    deliver_all(wv, ins.lgkm)
            elif op == "s_mov":
                wv.S[ins.dst] = sval(wv, ins.src) & 0xFFFFFFFF
            elif op == "s_add":
                v = (wv.S[ins.a] + ins.imm) & 0xFFFFFFFF
                if ins.dst == "m0":
                    wv.m0 = v
                else:
                    wv.S[ins.dst] = v
            elif op == "s_add_s":
                wv.S[ins.dst] = (wv.S[ins.a] + wv.S[ins.b]) & 0xFFFFFFFF
            elif op == "s_sub":
                wv.S[ins.dst] = (wv.S[ins.a] - ins.imm) & 0xFFFFFFFF
            elif op == "s_neg":
                wv.S[ins.dst] = (-wv.S[ins.dst]) & 0xFFFFFFFF
            elif op == "s_xor":
                wv.S[ins.dst] ^= ins.imm
            elif op == "s_cmp_gt":
                wv.scc = int(wv.S[ins.a] > ins.imm)
            elif op == "cbranch_scc1":
                if wv.scc:
                    wv.pc = labels[ins.target]
            elif op == "s_mov_exec":
                if ins.src is None:
                    wv.exec = np.ones(64, bool)
                else:
                    m = wv.S[ins.src] | (wv.S[ins.src + 1] << 32)
                    wv.exec = np.array([(m >> int(b)) & 1 for b in lane], bool)
            elif op == "v_xor":
                V[ins.dst] ^= np.uint32(ins.imm)
            elif op == "v_mov":
                V[ins.dst] = V[ins.src]
            elif op == "v_add_s":
                V[ins.dst] = ((V[ins.dst].astype(np.int64) + wv.S[ins.s]) & 0xFFFFFFFF).astype(np.uint32)
            elif op == "v_subrev_s":
                V[ins.dst] = ((V[ins.dst].astype(np.int64) - wv.S[ins.s]) & 0xFFFFFFFF).astype(np.uint32)
            elif op == "v_bfe_i32":
                V[ins.dst] = np.where((V[ins.src] >> ins.bit) & 1, 0xFFFFFFFF, 0).astype(np.uint32)
            elif op == "v_lshl16":
                V[ins.dst] = (V[ins.src] << 16).astype(np.uint32)
            elif op == "v_andhi":
                V[ins.dst] = V[ins.src] & np.uint32(0xFFFF0000)
            elif op == "v_and":
                V[ins.dst] = V[ins.a] & V[ins.b]
            elif op == "v_fma":
                V[ins.dst] = u32(t_affine(f32(V[ins.b]), f32(V[ins.a]), f32(V[ins.c])))
            elif op == "v_mul":
                V[ins.dst] = u32((f32(V[ins.a]) * f32(V[ins.b])).astype(np.float32))
            elif op == "v_mul_lit":
                V[ins.dst] = u32((f32(V[ins.src]) * f32(np.uint32(0xbfb8aa3b))).astype(np.float32))
            elif op == "v_exp":
                with np.errstate(over="ignore"):
                    V[ins.dst] = u32(np.exp2(f32(V[ins.dst])).astype(np.float32))
            elif op == "v_add1":
                V[ins.dst] = u32((np.float32(1.0) + f32(V[ins.dst])).astype(np.float32))
            elif op == "v_rcp":
                V[ins.dst] = u32((np.float32(1.0) / f32(V[ins.dst])).astype(np.float32))
            elif op == "acc_write":
                wv.A[ins.dst] = 0
            elif op == "acc_read":
                V[ins.dst] = wv.A[ins.src].view(np.uint32)
            elif op == "pk_fma":  # the drain: alpha (s[72:73] = 1.0) * acc + bias
                al = np.uint32(wv.S[72]).view(np.float32)
                for h in range(2):
                    xa, b = f32(V[ins.x + h]), f32(V[ins.b + h])
                    V[ins.dst + h] = u32((xa.astype(np.float64) * np.float64(al) + b.astype(np.float64)).astype(np.float32))
            elif op == "cvt_pk":
                V[ins.dst] = bf16_round(f32(V[ins.lo])) | (bf16_round(f32(V[ins.hi])) << 16)
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
        land_all(wv, 0)
    # ---- reference: transform (the same float operations) -> zero padding -> 3x3 conv in float64
    full = x
    if ups:
        full = np.repeat(np.repeat(x, 2, 0), 2, 1)
    if xform:
        g = bf16_to_f32(bf16_round(t_affine(full, gss[0][None, None, :], gss[1][None, None, :])))
        full = bf16_to_f32(bf16_round(t_silu(g)))
    pad = np.zeros((HWimg + 2, HWimg + 2, C), np.float32)
    pad[1:-1, 1:-1] = full
    ref = np.zeros((16, 16, NO), np.float64)
    for tap in range(9):
        dy, dx = divmod(tap, 3)
        win = pad[ty * 16 + dy: ty * 16 + dy + 16, tx * 16 + dx: tx * 16 + dx + 16].astype(np.float64)
        ref += win @ Wf[:, tap * C:(tap + 1) * C].astype(np.float64).T
    ref += bias[None, None, :]
    got = np.zeros((256, NO), np.float32)
    for vw in range(8):
        wm, wn = (vw >> 1, vw & 1) if cfg == 1 else (vw >> 2, vw & 3)  # cfg 1: wave w owns regions 2 w (columns 0-63) and 2 w + 1, 64 rows each
        rows = 64 if cfg == 1 else 128
        for ni in range(2):
            reg0 = vw * 16384 + ni * 8192
            for row in range(rows):
                for ch in range(4):
                    a = reg0 + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)
                    vals = lds[a:a + 16].view(np.uint16).astype(np.uint32)
                    got[wm * rows + row, wn * 64 + ni * 32 + ch * 8: wn * 64 + ni * 32 + ch * 8 + 8] = bf16_to_f32(vals)
    got = got.reshape(16, 16, NO)  # row = pixel row * 16 + x
    err = np.abs(got - ref) / (np.abs(ref) + 1.0)
    ok = float(err.max()) < 1.2e-2
    if verbose or not ok:
        print(f"xform {xform} tile {tile} C {C} ups {ups} late {late} order {order}: max rel err {err.max():.3e}, barriers {n_bar}, {'ok' if ok else 'WRONG'}")
        if not ok:
            bad = np.argwhere(err > 1.2e-2)
            print("  first bad entries (py, px, ch):", bad[:8].tolist(), " count", len(bad))
    return ok


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "diffusionkit_amd", "csrc")
    progs = emit(csrc)
    print(f"wrote conv256v4_asm_x.inc ({len(progs[True])} instructions), conv256v4_asm_p.inc ({len(progs[False])}) under {csrc}")
    if "--check" in sys.argv:
        allok = True
        for cfg in (0, 1):
            for xform in (True, False):
                for tile, C, ups in (((0, 0), 128, 0), ((1, 1), 192, 0), ((2, 1), 192, 1 if not xform else 0)):
                    for late in (True, False):
                        for order in (0, 1):
                            program(xform, cfg)  # (sets the module's tile form for the emulator's geometry)
                            allok &= run(progs[(xform, cfg)], xform, tile, late, order, C=C, ups=ups, seed=C + tile[0], verbose="-v" in sys.argv, cfg=cfg)
                print(f"cfg {cfg} xform {xform}: {'ok' if allok else 'FAILED'}", flush=True)
        print("ALL OK" if allok else "FAILED")
        sys.exit(0 if allok else 1)
