#!/usr/bin/env python3
"""Generator + CPU checker for the hand-scheduled body of the one-wave-per-SIMD attention kernel (diffusionkit_amd/csrc/attention5.hip), D = 128.

Frame (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD" structure; the algorithm, LDS images and MFMA operand mapping are attention4.hip's):
a workgroup = 4 waves = 256 query rows, a wave owns 64 of them (two 32-row blocks) and the whole 512-entry register file:
  a[0:127]    O^T accumulators, block (qb, dt) at (qb*4 + dt)*16        a[128:191] Q fragments (qb, kk) at 128 + (qb*8 + kk)*4
  a[192:255]  K fragments of the tile whose scores come next, (half, kk) at 192 + (half*8 + kk)*4
  v[0:63] / v[64:127]  two score sets (tile parity), block (qb, half) at base + (qb*2 + half)*16     v[128:159] P fragments (qb, n)
  v[160:223]  V^T fragments (dt, n) of the tile being multiplied        v[224:255] softmax state + addresses (see below)
K / V tiles (64 keys) arrive by LDS-DMA into two-slot rings (K one and a half tiles ahead of its reads, V one): a wave issues 4 + 4 pieces per tile.

Iteration j (one barrier, in the middle):
  phase 1, 32 MFMAs: S(j+1)^T = K(j+1) Q^T      beside: V(j) tr-reads, the second half of the exponentials of tile j, P(j) -> bf16
  -- vmcnt(0): K(j+2), V(j+1) landed -- barrier: every wave has read V(j) and (a phase ago) K(j+1)'s successor slot is free --
  phase 2, 32 MFMAs: O^T += V(j)^T P(j)^T        beside: K(j+2) -> AGPRs, row maxima of S(j+1) + the rescale decision, the first half of the
                                                  exponentials of tile j+1, the DMA pieces of K(j+3) and V(j+2)
The running maximum moves only when a score exceeds it by the threshold (attention4.hip's deferred rescale).  The decision for tile t is taken
while P(t-1) V(t-1) is still in flight, so it only RECORDS the factor (pend) and switches the exponent offset; the accumulators and the row
sums -- everything still at the old scale, P(t-1)'s products and sums included -- are multiplied once, at the head of the next phase 1, when
that P.V is complete (a rare, out-of-line block).

  python scripts/gen_attn5.py            write diffusionkit_amd/csrc/attention5_asm.inc, attention5_clobbers.inc
  python scripts/gen_attn5.py --check    also run the instruction-level emulator (4 waves x 64 lanes) against an fp64 softmax(Q K^T) V

v[224:225] mc = running max * c   v[226:227] l   v[228:231] psum[parity][qb]   v[232:233] pend   v[234:237] temporaries
v[238:245] K read addresses per kk   v[246:247] V read offsets (dt parity)   v[248:251] K piece offsets   v[252:255] V piece offsets
s[40:43] K resource  s[44:47] V resource  s48 K tile offset  s49 V tile offset  s50 tile bytes  s51 c = scale*log2(e)  s52 threshold*c
s53 loop counter  s54 rescale flag  s55 DMA base of this wave (wave*4096)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_gemm256v4 import I, BARRIER, bf16_round, bf16_to_f32, wait  # noqa: E402
from gen_conv256v4 import need, resolve, set_waits  # noqa: E402

SSET = [0, 64]
PF, VFR = 128, 160
MC, L, PEND = [224, 225], [226, 227], [232, 233]
PS = [[228, 229], [230, 231]]
TMP = [234, 235, 236, 237]
KADDR, VADDR, DK, DV = 238, 246, 248, 252
A_O, A_Q, A_K = 0, 128, 192
K_LDS, V_LDS, TILE = 0, 32768, 16384
VM_OPS = ("dma",)


def sblk(t, qb, half):
    return SSET[t & 1] + (qb * 2 + half) * 16


def mfma_qk(t, qb, half, kk):
    d = sblk(t, qb, half)
    a = A_K + (half * 8 + kk) * 4
    b = A_Q + (qb * 8 + kk) * 4
    c = "0" if kk == 0 else f"v[{d}:{d + 15}]"
    return I("mfma32", f"v_mfma_f32_32x32x16_bf16 v[{d}:{d + 15}], a[{a}:{a + 3}], a[{b}:{b + 3}], {c}", d=("v", d), a=("a", a), b=("a", b), zero=kk == 0)


def mfma_pv(qb, dt, n):
    d = A_O + (qb * 4 + dt) * 16
    a = VFR + (dt * 4 + n) * 4
    b = PF + (qb * 4 + n) * 4
    return I("mfma32", f"v_mfma_f32_32x32x16_bf16 a[{d}:{d + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], a[{d}:{d + 15}]", d=("a", d), a=("v", a), b=("v", b), zero=False)


def k_reads(slot):
    """the 16 K fragments of the tile in ring slot `slot` -> a[192:255] (attention4.hip: row l31 of the 32-key half, chunk (2 kk + hi) ^ (l31 & 15))"""
    out = []
    for kk in range(8):
        for half in range(2):
            d = A_K + (half * 8 + kk) * 4
            off = K_LDS + slot * TILE + half * 8192
            out.append(I("ds_read_a", f"ds_read_b128 a[{d}:{d + 3}], v{KADDR + kk} offset:{off}", dst=d, addr=KADDR + kk, off=off))
    return out


def v_reads(slot):
    """the 16 V^T fragments (dt, n), two ds_read_b64_tr_b16 each (attention4.hip: DK4_R1)"""
    out = []
    for n in range(4):
        for dt in range(4):
            d = VFR + (dt * 4 + n) * 4
            imm = V_LDS + slot * TILE + dt * 4096 + (32 * (n >> 1) + 16 * (n & 1)) * 32
            for h in range(2):
                out.append(I("ds_read_tr", f"ds_read_b64_tr_b16 v[{d + 2 * h}:{d + 2 * h + 1}], v{VADDR + (dt & 1)} offset:{imm + 256 * h}",
                             dst=d + 2 * h, addr=VADDR + (dt & 1), off=imm + 256 * h))
    return out


def dma_pieces(opnd, slot, tag):
    """4 pieces of this wave: LDS ring slot + wave*4096 + i*1024 <- the lane's source chunk (offsets v[DK + i] / v[DV + i]) of the tile at s48 / s49"""
    out = []
    base = (K_LDS if opnd == "K" else V_LDS) + slot * TILE
    for i in range(4):
        out.append([I("s_add", f"s_add_u32 m0, s55, {base + i * 1024}", dst="m0", a=55, imm=base + i * 1024), I("nop", "s_nop 0"),
                    I("dma", f"buffer_load_dwordx4 v{(DK if opnd == 'K' else DV) + i}, s[{40 if opnd == 'K' else 44}:{43 if opnd == 'K' else 47}], s{48 if opnd == 'K' else 49} offen lds",
                      opnd=opnd, vo=(DK if opnd == "K" else DV) + i, soff=48 if opnd == "K" else 49, tag=tag)])
    return out


def exp_stream(t, halves):
    """p = exp2(s * c - mc) in place, row sums into psum[t & 1][qb]; per (half, qb) 16 scores: fma, exp, add"""
    out = []
    for half in halves:
        for qb in range(2):
            s0 = sblk(t, qb, half)
            ps = PS[t & 1][qb]
            for e in range(16):
                r = s0 + e
                out.append(I("v_fma_sc", f"v_fma_f32 v{r}, v{r}, s51, -v{MC[qb]}", dst=r, a=r, mc=MC[qb]))
                out.append(I("v_exp", f"v_exp_f32 v{r}, v{r}", dst=r))
                if half == 0 and e == 1:
                    out.append(I("v_add", f"v_add_f32 v{ps}, v{r - 1}, v{r}", dst=ps, a=r - 1, b=r))
                elif not (half == 0 and e == 0):
                    out.append(I("v_add", f"v_add_f32 v{ps}, v{ps}, v{r}", dst=ps, a=ps, b=r))
    # (an exponential's result needs a wait state before its use: the add of score e is moved behind the fma of score e + 1)
    fixed = []
    i = 0
    while i < len(out):
        if out[i].op == "v_add" and i + 1 < len(out) and out[i + 1].op == "v_fma_sc":
            fixed.extend([out[i + 1], out[i]])
            i += 2
        else:
            fixed.append(out[i])
            i += 1
    if len(fixed) >= 2 and fixed[-1].op == "v_add" and fixed[-2].op == "v_exp":
        fixed.insert(-1, I("nop", "s_nop 0"))
    return fixed


def pack_stream(t):
    """P(t) fragments: (qb, n) word r = bf16 pair of scores 8 (n & 1) + 2 r, + 1 of half n >> 1"""
    out = []
    for n in range(4):
        for qb in range(2):
            s0 = sblk(t, qb, n >> 1) + 8 * (n & 1)
            for r in range(4):
                d = PF + (qb * 4 + n) * 4 + r
                out.append(I("cvt_pk", f"v_cvt_pk_bf16_f32 v{d}, v{s0 + 2 * r}, v{s0 + 2 * r + 1}", dst=d, lo=s0 + 2 * r, hi=s0 + 2 * r + 1))
    return out


def max_stream(t, label):
    """row maxima of S(t) (lane-local over its 32 keys, then across the two halves), the deferred-rescale test; the rare path is out of line"""
    out = []
    chains = []
    for qb in range(2):
        a, b = sblk(t, qb, 0), sblk(t, qb, 1)
        m = TMP[2 * qb]
        ch = [I("v_max3", f"v_max3_f32 v{m}, v{a}, v{a + 1}, v{b}", dst=m, a=a, b=a + 1, c=b)]
        for e in range(1, 16):
            x, y = (a + e + 1, b + e) if e < 15 else (b + 15, b + 15)
            ch.append(I("v_max3", f"v_max3_f32 v{m}, v{m}, v{x}, v{y}", dst=m, a=m, b=x, c=y))
        chains.append(ch)
    for x, y in zip(*chains):  # the two blocks' dependent chains alternate
        out.extend([x, y])
    for qb in range(2):  # (units: what must stay together in one MFMA gap -- a branch and its return label above all)
        m, u = TMP[2 * qb], TMP[2 * qb + 1]
        out.append([I("v_mov", f"v_mov_b32 v{u}, v{m}", dst=u, src=m), I("nop", "s_nop 1"), I("permswap", f"v_permlane32_swap_b32 v{m}, v{u}", a=m, b=u),
                    I("v_max", f"v_max_f32 v{m}, v{m}, v{u}", dst=m, a=m, b=u)])
        out.append([I("v_fma_sc", f"v_fma_f32 v{u}, v{m}, s51, -v{MC[qb]}", dst=u, a=m, mc=MC[qb]), I("v_cmp_lt_s", f"v_cmp_lt_f32 vcc, s52, v{u}", s=52, b=u),
                    I("cbranch_vccnz", f"s_cbranch_vccnz {label + qb}f", target=f"RARE{label + qb}"), I("label", f"{label + 2 + qb}:", name=f"BACK{label + qb}")])
    return out


def rare_blocks(label):
    """a row's maximum grew by more than the threshold: new offset, the factor for everything still at the old scale is recorded in pend"""
    out = []
    for qb in range(2):
        m, u = TMP[2 * qb], TMP[2 * qb + 1]
        out.append(I("label", f"{label + qb}:", name=f"RARE{label + qb}"))
        out.append(I("v_mul_s", f"v_mul_f32 v{m}, s51, v{m}", dst=m, s=51, b=m))
        out.append(I("v_max", f"v_max_f32 v{m}, v{m}, v{MC[qb]}", dst=m, a=m, b=MC[qb]))
        out.append(I("v_sub", f"v_sub_f32 v{u}, v{MC[qb]}, v{m}", dst=u, a=MC[qb], b=m))
        out.append(I("v_exp", f"v_exp_f32 v{u}, v{u}", dst=u))
        out.append(I("v_mov", f"v_mov_b32 v{MC[qb]}, v{m}", dst=MC[qb], src=m))
        out.append(I("nop", "s_nop 0"))
        out.append(I("v_mul", f"v_mul_f32 v{PEND[qb]}, v{PEND[qb]}, v{u}", dst=PEND[qb], a=PEND[qb], b=u))
        out.append(I("s_movi", "s_mov_b32 s54, 1", dst=54, imm=1))
        out.append(I("branch", f"s_branch {label + 2 + qb}b", target=f"BACK{label + qb}"))
    return out


def rescale_block(label):
    """P.V of the previous tile is complete: the accumulators and the row sums of the rows whose maximum moved take their factor"""
    out = [I("label", f"{label}:", name=f"RESC{label}"), I("nop", "s_nop 7"), I("nop", "s_nop 7"), I("nop", "s_nop 7")]
    for qb in range(2):
        out.append(I("v_mul", f"v_mul_f32 v{L[qb]}, v{L[qb]}, v{PEND[qb]}", dst=L[qb], a=L[qb], b=PEND[qb]))
        for r in range(64):
            a = A_O + qb * 64 + r
            t = TMP[r & 3]
            out.append(I("acc_read", f"v_accvgpr_read_b32 v{t}, a{a}", dst=t, src=a))
            out.append(I("nop", "s_nop 0"))
            out.append(I("v_mul", f"v_mul_f32 v{t}, v{t}, v{PEND[qb]}", dst=t, a=t, b=PEND[qb]))
            out.append(I("nop", "s_nop 0"))
            out.append(I("acc_write_v", f"v_accvgpr_write_b32 a{a}, v{t}", dst=a, src=t))
        out.append(I("v_movi", f"v_mov_b32 v{PEND[qb]}, 1.0", dst=PEND[qb], imm=0x3f800000))
    out.append(I("s_movi", "s_mov_b32 s54, 0", dst=54, imm=0))
    out.append(I("branch", f"s_branch {label + 1}b", target=f"RESCBACK{label}"))
    return out


def spread(gaps, items, lo, hi, cap=None):
    """put `items` (instructions or lists that stay together) into gaps lo .. hi - 1, evenly, in order"""
    n = len(items)
    for s, it in enumerate(items):
        g = lo + (s * (hi - lo)) // max(n, 1)
        gaps[g].extend(it if isinstance(it, list) else [it])


def iteration(j, lab, qk=True, pv=True, kread=True, dma_k=True, dma_v=True, mx=True, exp_head=True, landed=True):
    """tile j: phase 1 (S(j+1)), barrier, phase 2 (P(j) V(j)).  Flags switch parts off in the prologue / the peeled last iterations.
    lab: base of this copy's numeric labels (10 per copy)"""
    p = j & 1
    out, tails = [], []
    # ---- phase 1 ----
    g1 = [[] for _ in range(33)]
    head = []
    if pv:  # the row sums of tile j - 1 are final; a recorded rescale is applied now (P(j - 1) V(j - 1) is complete)
        for qb in range(2):
            head.append(I("v_add", f"v_add_f32 v{L[qb]}, v{L[qb]}, v{PS[p ^ 1][qb]}", dst=L[qb], a=L[qb], b=PS[p ^ 1][qb]))
        head.append(I("s_cmp_lg", "s_cmp_lg_u32 s54, 0", a=54, imm=0))
        head.append(I("cbranch_scc1", f"s_cbranch_scc1 {lab + 8}f", target=f"RESC{lab + 8}"))
        head.append(I("label", f"{lab + 9}:", name=f"RESCBACK{lab + 8}"))
        tails.extend(rescale_block(lab + 8))
    mf1 = [mfma_qk(j + 1, qb, half, kk) for kk in range(8) for half in range(2) for qb in range(2)] if qk else []
    if pv:
        spread(g1, v_reads(p), 0, 22)
        # exponentials of the second score half of tile j (the first half's ran a phase ago), the packs of P(j): k-steps 0, 1 beside them,
        # k-steps 2, 3 behind them
        pk = pack_stream(j)
        spread(g1, exp_stream(j, [1]), 0, 26)
        spread(g1, pk[:16], 1, 26)
        spread(g1, pk[16:], 26, 32)
    for m in range(32):
        if m == 0:
            out.extend(head)
        if qk:
            out.append(mf1[m])
        out.extend(g1[m])
    out.extend(g1[32])
    # ---- middle: K(j + 2) and V(j + 1) have landed (every piece issued a phase ago); all fragment reads of phase 1 are done ----
    out.append(need(("T", j + 1)) if landed else wait(vm=0))
    out.append(wait(lgkm=0))
    out.append(BARRIER())
    # ---- phase 2 ----
    g2 = [[] for _ in range(33)]
    mf2 = [mfma_pv(qb, dt, n) for n in range(4) for dt in range(4) for qb in range(2)] if pv else []
    if kread:
        spread(g2, k_reads(p), 0, 16)
    if mx:
        spread(g2, max_stream(j + 1, lab), 3, 13)
        tails.extend(rare_blocks(lab))
    if exp_head:
        spread(g2, exp_stream(j + 1, [0]), 13, 32)
    pieces = []
    if dma_k:
        pieces += dma_pieces("K", (j + 3) & 1, ("T", j + 2))
    if dma_v:
        pieces += dma_pieces("V", (j + 2) & 1, ("T", j + 2))
    spread(g2, pieces, 14, 32)
    for m in range(32):
        if pv:
            out.append(mf2[m])
        out.extend(g2[m])
    out.extend(g2[32])
    if dma_k:
        out.append(I("s_add_s", "s_add_u32 s48, s48, s50", dst=48, a=48, b=50))
    if dma_v:
        out.append(I("s_add_s", "s_add_u32 s49, s49, s50", dst=49, a=49, b=50))
    out.append(wait(lgkm=0))  # K(j + 2) sits in the AGPRs
    return out, tails


def program():
    """nt = S / 64 tiles, even, >= 6.  Prologue (tile -1: scores of tile 0 only), a two-tile loop over j = 0 .. nt - 5, four peeled iterations."""
    P, tails = [], []
    for d, s in ((48, "0"), (49, "0"), (50, "%[tileb]"), (52, "0x40b8aa3b"), (53, "%[ntrip]"), (54, "0"), (55, "%[dbase]")):  # s52 = 4 * log2(e)
        P.append(I("s_mov", f"s_mov_b32 s{d}, {s}", dst=d, src=s))
    # c = scale * log2(e) into s51 (a float product has no scalar instruction: through a VGPR)
    P.append(I("v_mov_s", f"v_mov_b32 v{TMP[0]}, %[scale]", dst=TMP[0], src="%[scale]"))
    P.append(I("v_mul_lit", f"v_mul_f32 v{TMP[0]}, 0x3fb8aa3b, v{TMP[0]}", dst=TMP[0], src=TMP[0], lit=0x3fb8aa3b))
    P.append(I("nop", "s_nop 1"))
    P.append(I("readfirstlane", f"v_readfirstlane_b32 s51, v{TMP[0]}", dst=51, src=TMP[0]))
    # Q fragments arrive in v[128:191] (inputs) -> a[128:191]; accumulators, row sums, offsets
    for r in range(64):
        P.append(I("acc_write_v", f"v_accvgpr_write_b32 a{A_Q + r}, v{128 + r}", dst=A_Q + r, src=128 + r))
    for a in range(128):
        P.append(I("acc_write", f"v_accvgpr_write_b32 a{a}, 0", dst=a))
    for qb in range(2):
        P.append(I("v_movi", f"v_mov_b32 v{MC[qb]}, 0xf149f2ca", dst=MC[qb], imm=0xf149f2ca))  # -1e30
        P.append(I("v_movi", f"v_mov_b32 v{L[qb]}, 0", dst=L[qb], imm=0))
        P.append(I("v_movi", f"v_mov_b32 v{PEND[qb]}, 1.0", dst=PEND[qb], imm=0x3f800000))
        P.append(I("v_movi", f"v_mov_b32 v{PS[1][qb]}, 0", dst=PS[1][qb], imm=0))
    # DMA: K(0) -> slot 0, V(0) -> slot 0, K(1) -> slot 1
    pro = []
    for u in dma_pieces("K", 0, "P0"):
        pro.extend(u)
    pro.append(I("s_add_s", "s_add_u32 s48, s48, s50", dst=48, a=48, b=50))
    for u in dma_pieces("V", 0, ("T", 0)):
        pro.extend(u)
    pro.append(I("s_add_s", "s_add_u32 s49, s49, s50", dst=49, a=49, b=50))
    for u in dma_pieces("K", 1, ("T", 0)):
        pro.extend(u)
    pro.append(I("s_add_s", "s_add_u32 s48, s48, s50", dst=48, a=48, b=50))
    pro.append(need("P0"))
    pro.append(BARRIER())
    pro.extend(k_reads(0))  # K(0) -> AGPRs
    pro.append(wait(lgkm=0))
    P.extend(pro)
    # iteration -1: scores of tile 0; behind its barrier: K(1) -> AGPRs, maxima + decision of tile 0, first exponentials, pieces of K(2) and V(1)
    it, tl = iteration(-1, 100, pv=False)
    P.extend(it)
    tails.extend(tl)
    set_waits(P, resolve(P, []))
    # the loop: two iterations (tile parities 0, 1); the same code serves its first pass (behind the prologue) and every later one
    b0a, t0a = iteration(0, 200)
    b1a, t1a = iteration(1, 300)
    b0b, _ = iteration(2, 200)
    b1b, _ = iteration(3, 300)
    first, again = resolve(b0a + b1a, P), resolve(b0b + b1b, P + b0a + b1a)
    set_waits(b0a + b1a, [min(x, y) for x, y in zip(first, again)])
    loop = [I("label", "20:", name="LOOP")] + b0a + b1a + [I("s_sub", "s_sub_u32 s53, s53, 1", dst=53, a=53, imm=1), I("s_cmp_gt", "s_cmp_gt_u32 s53, 0", a=53, imm=0),
                                                           I("cbranch_scc1", "s_cbranch_scc1 20b", target="LOOP")]
    tails.extend(t0a + t1a)
    # peeled: j = nt - 4 (full), nt - 3 (no K pieces), nt - 2 (no pieces, no K reads... K(nt) does not exist), nt - 1 (no scores)
    pe = []
    for k, (flags, lab) in enumerate(((dict(), 400), (dict(dma_k=False), 500), (dict(dma_k=False, dma_v=False, kread=False), 600),
                                       (dict(dma_k=False, dma_v=False, kread=False, qk=False, mx=False, exp_head=False, landed=False), 700))):
        it, tl = iteration(4 + k, lab, **flags)  # (parities as j = nt - 4 + k with nt even)
        pe.extend(it)
        tails.extend(tl)
    set_waits(pe, resolve(pe, P + b0a + b1a + b0b + b1b))
    end = []
    for qb in range(2):  # the last tile's row sums
        end.append(I("v_add", f"v_add_f32 v{L[qb]}, v{L[qb]}, v{PS[1][qb]}", dst=L[qb], a=L[qb], b=PS[1][qb]))
    end.append(I("nop", "s_nop 7"))
    end.append(I("nop", "s_nop 7"))
    end.append(I("nop", "s_nop 7"))
    for r in range(128):  # O -> v[0:127] (outputs)
        end.append(I("acc_read", f"v_accvgpr_read_b32 v{r}, a{r}", dst=r, src=r))
    end.append(I("branch", "s_branch 99f", target="END"))
    return P + loop + pe + end + tails + [I("label", "99:", name="END")]


CLOBBERS = [f"v{i}" for i in range(192, 226)] + [f"v{i}" for i in range(228, 238)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(48, 56)] + \
           ["m0", "vcc", "scc", "memory"]


def emit(csrc):
    P = program()
    n_mfma = sum(1 for i in P if i.op == "mfma32")
    with open(os.path.join(csrc, "attention5_asm.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_attn5.py -- do not edit; the CPU emulator in that script checks this instruction list.\n")
        f.write(f"// {len(P)} instructions, {n_mfma} MFMAs (prologue, two-tile loop, four peeled tiles, out-of-line rescale blocks); registers: see the script's header.\n")
        f.write("\n".join('    "' + ins.text + '\\n"' for ins in P) + "\n")
    with open(os.path.join(csrc, "attention5_clobbers.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_attn5.py\n")
        f.write(", ".join('"' + c + '"' for c in CLOBBERS) + "\n")
    return P


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "diffusionkit_amd", "csrc")
    P = emit(csrc)
    print(f"wrote attention5_asm.inc ({len(P)} instructions)")
    if "--check" in sys.argv:
        import attn5_emu
        ok = attn5_emu.check_all(P, verbose="-v" in sys.argv)
        print("ALL OK" if ok else "FAILED")
        sys.exit(0 if ok else 1)
