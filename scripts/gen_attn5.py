#!/usr/bin/env python3
"""Generator + CPU checker for the hand-scheduled body of the one-wave-per-SIMD attention kernel (diffusionkit_amd/csrc/attention5.hip), D = 128.

Frame (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD" structure; the algorithm, LDS images and MFMA operand mapping are attention4.hip's):
a workgroup = 4 waves = 256 query rows, a wave owns 64 of them (two 32-row blocks) and the whole 512-entry register file:
  a[0:127]    O^T accumulators, block (qb, dt) at (qb*4 + dt)*16        a[128:191] Q fragments (qb, kk) at 128 + (qb*8 + kk)*4
  a[192:255]  K fragments of the tile whose scores come next, (half, kk) at 192 + (half*8 + kk)*4
  v[0:63] / v[64:127]  two score sets (tile parity), block (qb, half) at base + (qb*2 + half)*16     v[128:159] P fragments (qb, n)
  v[160:223]  V^T fragments (dt, n) of the tile being multiplied        v[224:255] softmax state + addresses (see below)
K / V tiles (64 keys) arrive by LDS-DMA into four-slot rings (K at 0, V at 65536; a piece has 2.5 - 3 tiles to land): a wave issues 4 + 4 pieces
per tile.  Temporaries of the row maxima live in the OTHER score set (dead between the packs of a tile and the scores of the tile after next).

Iteration j (one barrier, in the middle):
  phase 1, 32 MFMAs: S(j+1)^T = K(j+1) Q^T      beside: V(j) tr-reads, the second half of the exponentials of tile j, P(j) -> bf16
  -- vmcnt: K(j+2), V(j+1) landed (the pieces of the last two iterations may fly) -- barrier: every wave has read V(j); K(j+1)'s slot is free --
  phase 2, 32 MFMAs: O^T += V(j)^T P(j)^T        beside: K(j+2) -> AGPRs, row maxima of S(j+1) + the rescale decision, the first half of the
                                                  exponentials of tile j+1, the DMA pieces of K(j+5) and V(j+4)
The exponent offset of a row moves only when a score exceeds it by the threshold (attention4.hip's deferred rescale), per row and without a
branch (v_cmp / v_cndmask; the factor is exactly 1 otherwise).  The decision for tile t is taken while P(t-1) V(t-1) is still in flight, so it
only RECORDS the factor (pend) and switches the offset; the accumulators and the row sums -- everything still at the old scale, P(t-1)'s
products and sums included -- are multiplied once, at the head of the next phase 1, when that P.V is complete (a rare, out-of-line block
behind a scalar test of s[56:57]).  Program: a first asm statement with K(0)'s pieces (in front of the query loads), then prologue (K(1..3),
V(0..2), tile -1 = the scores of tile 0), a four-tile loop (ring slot x score-set parity), eight peeled tiles.

  python scripts/gen_attn5.py            write diffusionkit_amd/csrc/attention5_dma.inc, attention5_asm.inc, attention5_clobbers.inc
  python scripts/gen_attn5.py --check    also run the instruction-level emulator (4 waves x 64 lanes) against an fp64 softmax(Q K^T) V

v[224:225] mc = running max * c   v[226:227] l   v[228:231] psum[parity][qb]   v[232:233] pend   v[234:237] temporaries
v[238:245] K read addresses per kk   v[246:247] V read offsets (dt parity)   v[248:251] K piece offsets   v[252:255] V piece offsets
s[40:43] K resource  s[44:47] V resource  s48 K tile offset  s49 V tile offset  s50 tile bytes  s51 c = scale*log2(e)  s52 threshold*c
s53 loop counter  s55 DMA base of this wave (wave*4096)  s[56:57] rows whose offset moved (rescale pending)
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
K_LDS, V_LDS, TILE, DEPTH = 0, 65536, 16384, 4  # four-slot rings: a piece has 2.5 - 3 tiles to land
VM_OPS = ("dma",)
# lab only (scripts/build_attn5_abl.sh): bit mask of what the tile loop leaves out -- 1 the exponentials (fma, exp, add), 2 the row maxima + the rescale
# test, 4 the bf16 packs, 8 the DMA pieces of the loop, 16 the V reads, 32 the K reads, 64 the MFMAs.  Results are wrong; the timings say what each costs.
ABL = int(os.environ.get("A5_ABL", "0"))
# lab only: schedule options, comma separated -- exp=grouped|pipelined|simple, merge=concat|rr, expop=mul (the exponentials become multiplies: wrong results)
OPT = dict(kv.split("=") for kv in os.environ.get("A5_OPT", "").split(",") if kv)


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
            imm = slot * TILE + dt * 4096 + (32 * (n >> 1) + 16 * (n & 1)) * 32  # (V_LDS sits in the address registers: the offset field has 16 bits)
            for h in range(2):
                out.append(I("ds_read_tr", f"ds_read_b64_tr_b16 v[{d + 2 * h}:{d + 2 * h + 1}], v{VADDR + (dt & 1)} offset:{imm + 256 * h}",
                             dst=d + 2 * h, addr=VADDR + (dt & 1), off=imm + 256 * h))
    return out


def dma_pieces(opnd, slot, tag):
    """4 pieces of this wave: LDS ring slot + wave*4096 + i*1024 <- the lane's source chunk (offsets v[DK + i] / v[DV + i]) of the tile at s48 / s49"""
    out = []
    base = (K_LDS if opnd == "K" else V_LDS) + slot * TILE
    for i in range(4):
        out.append([I("s_add", f"s_add_u32 m0, s55, {base + i * 1024}", dst="m0", a=55, imm=base + i * 1024),
                    I("dma", f"buffer_load_dwordx4 v{(DK if opnd == 'K' else DV) + i}, s[{40 if opnd == 'K' else 44}:{43 if opnd == 'K' else 47}], s{48 if opnd == 'K' else 49} offen lds",
                      opnd=opnd, vo=(DK if opnd == "K" else DV) + i, soff=48 if opnd == "K" else 49, tag=tag)])
    return out


def exp_stream(t, halves):
    """p = exp2(s * c - mc) in place and the row sums: four scores in lockstep -- (e, e + 1) of both query blocks: four fma, four exp, four adds
    into four different sums (psum and a second partial sum per block, TMP[1] / TMP[3], folded in behind the second score half).
    Lab orders (A5_OPT exp=...): pipelined (fma of score i + 4, exp of i + 2, add of i), simple (one score at a time)."""
    out = []
    order = OPT.get("exp", "grouped")
    expi = (lambda r: I("v_mul", f"v_mul_f32 v{r}, v{r}, v{r}", dst=r, a=r, b=r)) if OPT.get("expop") == "mul" else (lambda r: I("v_exp", f"v_exp_f32 v{r}, v{r}", dst=r))

    def fma(qb, r):
        return I("v_fma_sc", f"v_fma_f32 v{r}, v{r}, s51, -v{MC[qb]}", dst=r, a=r, mc=MC[qb])

    def add(half, k, qb, odd, r):
        d = TMP[2 * qb + 1] if odd else PS[t & 1][qb]
        if half == 0 and k < 4:
            return I("v_mov", f"v_mov_b32 v{d}, v{r}", dst=d, src=r)
        return I("v_add", f"v_add_f32 v{d}, v{d}, v{r}", dst=d, a=d, b=r)
    for half in halves:
        el = [(qb, odd, sblk(t, qb, half) + e + odd) for e in range(0, 16, 2) for qb in range(2) for odd in range(2)]
        n = len(el)
        if order == "pipelined":
            for k in range(-4, n):
                if 0 <= k + 4 < n:
                    out.append(fma(el[k + 4][0], el[k + 4][2]))
                if 0 <= k + 2 < n:
                    out.append(expi(el[k + 2][2]))
                if 0 <= k < n:
                    out.append(add(half, k, *el[k]))
        elif order == "simple":
            for k in range(n):
                out.append(fma(el[k][0], el[k][2]))
                out.append(expi(el[k][2]))
                if k > 0:
                    out.append(add(half, k - 1, *el[k - 1]))
            out.append(I("nop", "s_nop 0"))
            out.append(add(half, n - 1, *el[n - 1]))
        else:
            for g0 in range(0, n, 4):
                out.extend(fma(qb, r) for qb, odd, r in el[g0:g0 + 4])
                out.extend(expi(r) for qb, odd, r in el[g0:g0 + 4])
                out.extend(add(half, g0 + i, *el[g0 + i]) for i in range(4))
        if half == 1:
            for qb in range(2):
                out.append(I("v_add", f"v_add_f32 v{PS[t & 1][qb]}, v{PS[t & 1][qb]}, v{TMP[2 * qb + 1]}", dst=PS[t & 1][qb], a=PS[t & 1][qb], b=TMP[2 * qb + 1]))
    return out


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
    """row maxima of S(t) (lane-local over its 32 keys, then across the two halves) and the deferred-rescale test; the rare path is out of line.
    Eight independent chains (four per query block, eight scores each) run in lockstep; their temporaries are registers of the OTHER score set,
    dead between the packs of tile t - 1 and the scores of tile t + 1.  Returns the 36 chain instructions, then four units that stay together."""
    out = []
    chains = []
    for qb in range(2):
        x = [sblk(t, qb, 0) + e for e in range(16)] + [sblk(t, qb, 1) + e for e in range(16)]
        for c in range(4):
            T = SSET[(t & 1) ^ 1] + qb * 4 + c
            v = x[c::4]  # 8 scores
            chains.append([I("v_max3", f"v_max3_f32 v{T}, v{v[0]}, v{v[1]}, v{v[2]}", dst=T, a=v[0], b=v[1], c=v[2]),
                           I("v_max3", f"v_max3_f32 v{T}, v{T}, v{v[3]}, v{v[4]}", dst=T, a=T, b=v[3], c=v[4]),
                           I("v_max3", f"v_max3_f32 v{T}, v{T}, v{v[5]}, v{v[6]}", dst=T, a=T, b=v[5], c=v[6]),
                           I("v_max", f"v_max_f32 v{T}, v{T}, v{v[7]}", dst=T, a=T, b=v[7])])
    for step in zip(*chains):
        out.extend(step)
    for k in range(2):
        for qb in range(2):
            T, m = SSET[(t & 1) ^ 1] + qb * 4, TMP[2 * qb]
            if k == 0:
                out.append(I("v_max3", f"v_max3_f32 v{m}, v{T}, v{T + 1}, v{T + 2}", dst=m, a=T, b=T + 1, c=T + 2))
            else:
                out.append(I("v_max", f"v_max_f32 v{m}, v{m}, v{T + 3}", dst=m, a=m, b=T + 3))
    units = []
    for qb in range(2):  # (units: what stays together in one MFMA gap)
        m, u = TMP[2 * qb], TMP[2 * qb + 1]
        units.append([I("v_mov", f"v_mov_b32 v{u}, v{m}", dst=u, src=m), I("nop", "s_nop 1"), I("permswap", f"v_permlane32_swap_b32 v{m}, v{u}", a=m, b=u),
                      I("v_max", f"v_max_f32 v{m}, v{m}, v{u}", dst=m, a=m, b=u)])
    # The deferred rescale WITHOUT a branch (a branch on VCC costs ~270 cycles here: it waits for the vector and matrix pipes; measured with the
    # lab masks 512 / 256 of this script): per row, if the new maximum exceeds the offset by more than the threshold the offset moves to it and
    # the factor exp2(old - new) is recorded in pend; otherwise new = old and the factor is exactly 1.  s[56:57] collects the rows that moved.
    for qb in range(2):
        m, u, mc = TMP[2 * qb], TMP[2 * qb + 1], MC[qb]
        units.append([I("v_fma_sc", f"v_fma_f32 v{u}, v{m}, s51, -v{mc}", dst=u, a=m, mc=mc), I("v_mul_s", f"v_mul_f32 v{m}, s51, v{m}", dst=m, s=51, b=m),
                      I("v_cmp_lt_s", f"v_cmp_lt_f32 vcc, s52, v{u}", s=52, b=u), I("s_or_vcc", "s_or_b64 s[56:57], s[56:57], vcc", dst=56),
                      I("v_cndmask", f"v_cndmask_b32 v{m}, v{mc}, v{m}, vcc", dst=m, a=mc, b=m)])
        units.append([I("v_sub", f"v_sub_f32 v{u}, v{mc}, v{m}", dst=u, a=mc, b=m), I("v_mov", f"v_mov_b32 v{mc}, v{m}", dst=mc, src=m),
                      I("v_exp", f"v_exp_f32 v{u}, v{u}", dst=u)])
    for qb in range(2):
        units.append([I("v_mul", f"v_mul_f32 v{PEND[qb]}, v{PEND[qb]}, v{TMP[2 * qb + 1]}", dst=PEND[qb], a=PEND[qb], b=TMP[2 * qb + 1])])
    return out + units


def rescale_block(label):
    """P.V of the previous tile is complete: the accumulators and the row sums of the rows whose maximum moved take their factor"""
    out = [I("label", f"{label}:", name=f"RESC{label}"), I("nop", "s_nop 7"), I("nop", "s_nop 7"), I("nop", "s_nop 7")]
    for qb in range(2):
        out.append(I("v_mul", f"v_mul_f32 v{L[qb]}, v{L[qb]}, v{PEND[qb]}", dst=L[qb], a=L[qb], b=PEND[qb]))
        for r in range(64):
            a = A_O + qb * 64 + r
            t = PF + (r & 3)  # (the P fragment registers are dead here; TMP carries partial row sums across the phase boundary)
            out.append(I("acc_read", f"v_accvgpr_read_b32 v{t}, a{a}", dst=t, src=a))
            out.append(I("nop", "s_nop 0"))
            out.append(I("v_mul", f"v_mul_f32 v{t}, v{t}, v{PEND[qb]}", dst=t, a=t, b=PEND[qb]))
            out.append(I("nop", "s_nop 0"))
            out.append(I("acc_write_v", f"v_accvgpr_write_b32 a{a}, v{t}", dst=a, src=t))
        out.append(I("v_movi", f"v_mov_b32 v{PEND[qb]}, 1.0", dst=PEND[qb], imm=0x3f800000))
    out.append(I("s_mov64", "s_mov_b64 s[56:57], 0", dst=56, imm=0))
    out.append(I("branch", f"s_branch {label + 1}b", target=f"RESCBACK{label}"))
    return out


class Gaps:
    """the 32 MFMA gaps of a phase (+ one behind the last MFMA): every stream is spread over its window on its own; inside a gap the streams
    are merged round-robin, so that the exponentials of one stream sit between instructions of the others"""

    def __init__(self):
        self.streams = []  # per stream: 33 lists of units

    def load(self, g):
        return sum(len(u) for st in self.streams for u in st[g])

    def spread(self, items, lo, hi):
        """put `items` (instructions, or lists that stay together) into gaps lo .. hi - 1 in order: each item goes to the less loaded of the gap
        at its even-spread position and the next one (never in front of its predecessor)"""
        st = [[] for _ in range(33)]
        self.streams.append(st)
        n, last = len(items), lo
        for s, it in enumerate(items):
            ideal = lo + (s * (hi - lo)) // max(n, 1)
            cand = [g for g in (ideal, ideal + 1) if last <= g < hi] or [max(last, min(ideal, hi - 1))]
            g = min(cand, key=lambda x: (self.load(x), x))
            st[g].append(it if isinstance(it, list) else [it])
            last = g
        return st

    def put(self, g, unit, front=False):
        st = [[] for _ in range(33)]
        st[g].append(unit if isinstance(unit, list) else [unit])
        if front:
            self.streams.insert(0, st)
        else:
            self.streams.append(st)

    def gap(self, g):
        out = []
        qs = [list(st[g]) for st in self.streams if st[g]]
        if OPT.get("merge", "concat") == "concat":
            return [ins for q in qs for u in q for ins in u]
        while qs:
            for q in qs:
                out.extend(q.pop(0))
            qs = [q for q in qs if q]
        return out


def iteration(j, lab, qk=True, pv=True, kread=True, dma_k=True, dma_v=True, mx=True, exp_head=True, landed=True):
    """tile j: phase 1 (S(j+1)), barrier, phase 2 (P(j) V(j)).  Flags switch parts off in the prologue / the peeled last iterations.
    lab: base of this copy's numeric labels (10 per copy)"""
    p = j & 1
    out, tails = [], []
    if (ABL & 8) and j >= 0:
        dma_k = dma_v = landed = False
    # ---- phase 1 ----
    g1 = Gaps()
    head = []
    if pv:  # the row sums of tile j - 1 are final; a recorded rescale is applied now (P(j - 1) V(j - 1) is complete)
        for qb in range(2):
            head.append(I("v_add", f"v_add_f32 v{L[qb]}, v{L[qb]}, v{PS[p ^ 1][qb]}", dst=L[qb], a=L[qb], b=PS[p ^ 1][qb]))
        head.append(I("s_cmp_lg64", "s_cmp_lg_u64 s[56:57], 0", a=56))
        if not (ABL & 2048):
            head.append(I("cbranch_scc1", f"s_cbranch_scc1 {lab + 8}f", target=f"RESC{lab + 8}"))
        head.append(I("label", f"{lab + 9}:", name=f"RESCBACK{lab + 8}"))
        tails.extend(rescale_block(lab + 8))
    qk = qk and not (ABL & 64) and not (ABL & 4096)
    mf1 = [mfma_qk(j + 1, qb, half, kk) for kk in range(8) for half in range(2) for qb in range(2)] if qk else []
    if pv:
        # V(j) reads; exponentials of the second score half of tile j (the first half's ran a phase ago); the packs of P(j): k-steps 0, 1
        # beside them, k-steps 2, 3 behind them
        pk = pack_stream(j)
        if not (ABL & 1):
            g1.spread(exp_stream(j, [1]), 0, 29)
        if not (ABL & 16):
            g1.spread(v_reads(j % DEPTH), 0, 26)
        if not (ABL & 4):
            g1.spread(pk[:16], 0, 29)
            g1.spread(pk[16:], 29, 32)
    for m in range(32):
        if m == 0:
            out.extend(head)
        if qk:
            out.append(mf1[m])
        out.extend(g1.gap(m))
    # ---- middle: K(j + 2) and V(j + 1) have landed (issued three iterations ago); all fragment reads of phase 1 are done ----
    out.append(need(("T", j - 3)) if landed else wait(vm=0))
    out.append(wait(lgkm=0))
    out.append(BARRIER())
    # ---- phase 2 ----
    g2 = Gaps()
    pv_m = pv and not (ABL & 64) and not (ABL & 8192)
    mf2 = [mfma_pv(qb, dt, n) for n in range(4) for dt in range(4) for qb in range(2)] if pv_m else []
    if exp_head and not (ABL & 1):
        g2.spread(exp_stream(j + 1, [0]), 14, 32)
    if mx and not (ABL & 2):  # the eight maximum chains over gaps 1 .. 7, then exchange / test + select / factor / pend of the two blocks
        ms = max_stream(j + 1, lab)
        if not (ABL & 128):
            g2.spread(ms[:36], 1, 8)
        for k, u in enumerate(ms[36:]):
            if not (ABL & 256):
                g2.put(8 + min(k, 5), u)
    if kread and not (ABL & 32):
        g2.spread(k_reads((j + 2) % DEPTH), 0, 22)
    pieces = []
    if dma_k:
        pieces += dma_pieces("K", (j + 5) % DEPTH, ("T", j))
    if dma_v:
        pieces += dma_pieces("V", (j + 4) % DEPTH, ("T", j))
    for s_, (m0w, piece) in enumerate(pieces):  # the M0 write closes gap g - 1, the piece opens gap g (an SALU write of M0 needs a wait state first)
        g = 2 + (s_ * 28) // max(len(pieces), 1)
        g2.put(g - 1, m0w)
        g2.put(g, piece, front=True)
    if dma_k:  # the tile offsets move on behind the last piece
        g2.put(30, I("s_add_s", "s_add_u32 s48, s48, s50", dst=48, a=48, b=50))
    if dma_v:
        g2.put(31, I("s_add_s", "s_add_u32 s49, s49, s50", dst=49, a=49, b=50))
    for m in range(32):
        if pv_m:
            out.append(mf2[m])
        out.extend(g2.gap(m))
    out.append(wait(lgkm=0))  # K(j + 2) sits in the AGPRs
    return out, tails


def program():
    """nt = S / 64 tiles, a multiple of 4, >= 12.  Prologue (K(0..3), V(0..2); iteration -1: the scores of tile 0), a four-tile loop over
    j = 0 .. nt - 9, eight peeled iterations."""
    P, tails = [], []
    # ---- block 1 (its own asm statement, in front of the query loads): the DMA pieces of K(0); the loads of the query rows behind them share
    # their round trip.  (All seven prologue tiles in front of the query loads made every workgroup wait for 176 KB: the counter retires in order.)
    for d, s in ((48, "0"), (49, "0"), (50, "%[tileb]"), (55, "%[dbase]")):
        P.append(I("s_mov", f"s_mov_b32 s{d}, {s}", dst=d, src=s))
    pro = []

    def issue(opnd, t, tag):
        for m0w, piece in dma_pieces(opnd, t % DEPTH, tag):
            pro.extend([m0w, I("nop", "s_nop 0"), piece])
        r = 48 if opnd == "K" else 49
        pro.append(I("s_add_s", f"s_add_u32 s{r}, s{r}, s50", dst=r, a=r, b=50))
    issue("K", 0, "P0")
    P.extend(pro)
    P.append(I("split", None))
    # ---- block 2
    for d, s in ((48, "%[koff]"), (49, "%[voff]"), (50, "%[tileb]"), (52, "0x40b8aa3b"), (53, "%[ntrip]"), (55, "%[dbase]"), (56, "0"), (57, "0")):  # s52 = 4 * log2(e)
        P.append(I("s_mov", f"s_mov_b32 s{d}, {s}", dst=d, src=s))
    pro = []
    for t in range(3):  # K(1) V(0) | K(2) V(1) | K(3) V(2): the order the waits retire them in
        issue("K", t + 1, ("T", t - 4))
        issue("V", t, ("T", t - 4))
    P.extend(pro)
    # c = scale * log2(e) into s51 (a float product has no scalar instruction: through a VGPR)
    P.append(I("v_mov_s", f"v_mov_b32 v{TMP[0]}, %[scale]", dst=TMP[0], src="%[scale]"))
    P.append(I("v_mul_lit", f"v_mul_f32 v{TMP[0]}, 0x3fb8aa3b, v{TMP[0]}", dst=TMP[0], src=TMP[0], lit=0x3fb8aa3b))
    P.append(I("nop", "s_nop 1"))
    P.append(I("readfirstlane", f"v_readfirstlane_b32 s51, v{TMP[0]}", dst=51, src=TMP[0]))
    # Q fragments arrive in v[128:191] (inputs) -> a[128:191]; accumulators, row sums, offsets -- while the pieces fly
    for r in range(64):
        P.append(I("acc_write_v", f"v_accvgpr_write_b32 a{A_Q + r}, v{128 + r}", dst=A_Q + r, src=128 + r))
    for a in range(128):
        P.append(I("acc_write", f"v_accvgpr_write_b32 a{a}, 0", dst=a))
    for qb in range(2):
        P.append(I("v_movi", f"v_mov_b32 v{MC[qb]}, 0xf149f2ca", dst=MC[qb], imm=0xf149f2ca))  # -1e30
        P.append(I("v_movi", f"v_mov_b32 v{L[qb]}, 0", dst=L[qb], imm=0))
        P.append(I("v_movi", f"v_mov_b32 v{PEND[qb]}, 1.0", dst=PEND[qb], imm=0x3f800000))
        P.append(I("v_movi", f"v_mov_b32 v{PS[1][qb]}, 0", dst=PS[1][qb], imm=0))
    P.append(need("P0"))
    P.append(BARRIER())
    P.extend(k_reads(0))  # K(0) -> AGPRs
    P.append(wait(lgkm=0))
    # iteration -1: scores of tile 0; behind its barrier: K(1) -> AGPRs, maxima + decision of tile 0, first exponentials, pieces of K(4) and V(3)
    it, tl = iteration(-1, 100, pv=False)
    P.extend(it)
    tails.extend(tl)
    set_waits(P, resolve(P, []))
    # the loop: four iterations; the same code serves its first pass (behind the prologue) and every later one
    def group(j0):
        body, tl = [], []
        for k in range(4):
            it, t = iteration(j0 + k, 200 + 100 * k)
            body.extend(it)
            tl.extend(t)
        return body, tl
    ba, ta = group(0)
    bb, _ = group(4)
    first, again = resolve(ba, P), resolve(bb, P + ba)
    set_waits(ba, [min(x, y) for x, y in zip(first, again)])
    loop = [I("label", "20:", name="LOOP")] + ba + [I("s_sub", "s_sub_u32 s53, s53, 1", dst=53, a=53, imm=1), I("s_cmp_gt", "s_cmp_gt_u32 s53, 0", a=53, imm=0),
                                                     I("cbranch_scc1", "s_cbranch_scc1 20b", target="LOOP")]
    tails.extend(ta)
    # peeled: j = nt - 8 + k.  K(j + 5) exists for k <= 2, V(j + 4) for k <= 3, K(j + 2) for k <= 5, tile j + 1 for k <= 6
    pe = []
    for k in range(8):
        it, tl = iteration(8 + k, 600 + 100 * k, dma_k=k <= 2, dma_v=k <= 3, kread=k <= 5, qk=k <= 6, mx=k <= 6, exp_head=k <= 6, landed=k <= 6)
        pe.extend(it)
        tails.extend(tl)
    set_waits(pe, resolve(pe, P + ba + bb))
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


CLOBBERS = [f"v{i}" for i in range(192, 224)] + [f"v{i}" for i in range(228, 238)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(48, 58)] + \
           ["m0", "vcc", "scc", "memory"]


def emit(csrc):
    P = program()
    n_mfma = sum(1 for i in P if i.op == "mfma32")
    k = [i for i, x in enumerate(P) if x.op == "split"][0]
    with open(os.path.join(csrc, "attention5_dma.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_attn5.py -- do not edit.  The DMA pieces of the first K tile (an asm statement of its own, in front of the query loads).\n")
        f.write("// same explicit registers as attention5_asm.inc\n")
        f.write("\n".join('    "' + ins.text + '\\n"' for ins in P[:k]) + "\n")
    P_all, P = P, P[k + 1:]
    with open(os.path.join(csrc, "attention5_asm.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_attn5.py -- do not edit; the CPU emulator in that script checks this instruction list.\n")
        f.write(f"// {len(P)} instructions, {n_mfma} MFMAs (prologue, four-tile loop, eight peeled tiles, out-of-line rescale blocks); registers: see the script's header.\n")
        f.write("\n".join('    "' + ins.text + '\\n"' for ins in P) + "\n")
    with open(os.path.join(csrc, "attention5_clobbers.inc"), "w") as f:
        f.write("// GENERATED by scripts/gen_attn5.py\n")
        f.write(", ".join('"' + c + '"' for c in CLOBBERS) + "\n")
    return P_all


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
