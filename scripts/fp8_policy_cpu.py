"""fp8 precision policy, priced on the CPU oracle: which Linear classes cost how many dB.

The fp8 path quantises every block Linear (e4m3 weights with per-channel scales, MX-fp8 activations; oracle/fp8.py).  This script runs the
fake-quantising fp32 oracle at FLUX width (S_t = 512, latent 128 x 128, depth 4 + 8) with ONE class of Linears left un-quantised at a
time (weights and the activations feeding them) and reports the PSNR / rel-L2 of the model output against the plain fp32 oracle: the class
whose exemption buys the most dB per FLOP kept in bf16 is where a precision policy should start.

    python scripts/fp8_policy_cpu.py [threads]        (about 30 s per variant on 8 cores)
"""
import os
import sys
import time
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_fullsize_fixtures as fx  # noqa: E402
from diffusionkit_amd.config import FLUX_SCHNELL  # noqa: E402
from diffusionkit_amd.weights import synth_mmdit_weights  # noqa: E402
from oracle import fp8 as o8  # noqa: E402
from oracle.mmdit import OracleMMDiT, Prec, embed_dtype  # noqa: E402

DM, DU = (19, 38) if os.environ.get("FP8_POLICY_FULL") else (4, 8)
CFG = replace(FLUX_SCHNELL, depth_multimodal=DM, depth_unified=DU)
C = dict(cfg=CFG, seed_w=1234, B=1, latent=(128, 128), S_t=512, timesteps=[1000.0, 752.0], step=1)


def block_of(prefix):
    parts = prefix.split(".")
    return parts[0], int(parts[1]), (parts[2] if len(parts) > 2 else "")


# class name -> predicate(prefix, site): True = this Linear stays in bf16 / fp32 (not quantised)
def cls_double(stream, site):
    return lambda p, s: block_of(p)[0].startswith("multimodal") and block_of(p)[2].startswith(stream) and s == site


def first_doubles(n):
    return lambda p, s: block_of(p)[0].startswith("multimodal") and block_of(p)[1] < n


CLASSES = {
    "none (all fp8)": lambda p, s: False,
    "first 2 double blocks": first_doubles(2),
    "D.img MLP (fc1 + fc2)": lambda p, s: block_of(p)[0].startswith("multimodal") and block_of(p)[2].startswith("image") and s in ("fc1", "fc2"),
    "first 4 double blocks": first_doubles(4),
    "D.img.qkv": cls_double("image", "qkv"), "D.img.o": cls_double("image", "o"), "D.img.fc1": cls_double("image", "fc1"),
    "D.img.fc2": cls_double("image", "fc2"),
    "D.txt (all four)": lambda p, s: block_of(p)[0].startswith("multimodal") and block_of(p)[2].startswith("text"),
    "S.linear1": lambda p, s: block_of(p)[0].startswith("unified") and s == "qkv",
    "S.linear2": lambda p, s: block_of(p)[0].startswith("unified") and s in ("o", "fc2"),
    "all o_proj inputs (attention output)": lambda p, s: s == "o" and block_of(p)[0].startswith("multimodal"),
    "first double block": lambda p, s: block_of(p)[0].startswith("multimodal") and block_of(p)[1] == 0,
    "last single block": lambda p, s: block_of(p)[0].startswith("unified") and block_of(p)[1] == DU - 1,
    "all doubles": lambda p, s: block_of(p)[0].startswith("multimodal"),
    "all singles": lambda p, s: block_of(p)[0].startswith("unified"),
    # activation-only / weight-only exemptions over everything: which half of the format costs what
    "weights only quantised (activations bf16)": "act_off",
    "activations only quantised (weights bf16)": "w_off",
}

SITE_OF_KEY = {"attn.q_proj": "qkv", "attn.k_proj": "qkv", "attn.v_proj": "qkv", "attn.o_proj": "o", "mlp.fc1": "fc1", "mlp.fc2": "fc2"}


def flops_share(keep):
    """fraction of the block-Linear FLOPs that stays un-quantised under ``keep`` (rows x weight elements)"""
    h, r = CFG.hidden_size, CFG.mlp_ratio
    S_i, S_t = 4096, C["S_t"]
    tot = kept = 0.0
    for i in range(DM):
        for stream, rows in (("image_transformer_block", S_i), ("text_transformer_block", S_t)):
            p = f"multimodal_transformer_blocks.{i}.{stream}"
            for site, w in (("qkv", 3 * h * h), ("o", h * h), ("fc1", r * h * h), ("fc2", r * h * h)):
                tot += rows * w
                kept += rows * w * bool(keep(p, site))
    for i in range(DU):
        p = f"unified_transformer_blocks.{i}.transformer_block"
        rows = S_i + S_t
        for site, w in (("qkv", (3 + r) * h * h), ("o", h * h), ("fc2", r * h * h)):
            tot += rows * w
            kept += rows * w * bool(keep(p, site))
    return kept / tot


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8))
    named = synth_mmdit_weights(CFG, seed=C["seed_w"])
    plain = {k: v.float() for k, v in named.items()}
    fq_all = o8.fake_quant_block_weights(replace(CFG, weight_dtype="fp8_e4m3"), named)
    text, pooled, lat = fx.forward_inputs(C)
    ts = C["timesteps"]

    def run(w, aq):
        m = OracleMMDiT(CFG, w, Prec(), act_quant=aq, embed_prec=Prec(embed_dtype(CFG)))
        m.cache_modulation_params(pooled, torch.tensor(ts))
        taps = {}
        m(lat, text, ts[C["step"]], taps=taps)
        return taps["final"]

    t0 = time.time()
    ref = run(plain, None)
    print(f"fp32 oracle, un-quantised: {time.time() - t0:.0f} s", flush=True)
    rows = []
    only = os.environ.get("FP8_POLICY_ONLY")
    for name, keep in CLASSES.items():
        if only and not any(name.startswith(o) for o in only.split("|")):
            continue
        t0 = time.time()
        if keep == "act_off":
            w, aq, share = fq_all, None, float("nan")
        elif keep == "w_off":
            aq = lambda x, site=None: o8.mx8_fake_quant(x)  # noqa: E731
            aq.takes_site = True
            w, share = plain, float("nan")
        else:
            w = dict(fq_all)
            for key in fq_all:
                if not key.endswith(".weight"):
                    continue
                for frag, site in SITE_OF_KEY.items():
                    if key.endswith(frag + ".weight") and ("transformer_block" in key):
                        prefix = key[: -len("." + frag + ".weight")]
                        # single blocks: linear1 = [q|k|v|fc1] is one Linear of site "qkv"
                        st = "qkv" if (prefix.startswith("unified") and site == "fc1") else site
                        if keep(prefix, st):
                            w[key] = plain[key]

            def aq(x, site=None, keep=keep):
                return x if keep(*site) else o8.mx8_fake_quant(x)
            aq.takes_site = True
            share = flops_share(keep)
        out = run(w, aq)
        p, e = fx.psnr(ref, out), fx.rel_l2(ref, out)
        rows.append((name, share, p, e))
        print(f"{name:45s} bf16 share of Linear FLOPs {share:5.3f}   PSNR {p:6.2f} dB   rel-L2 {e:.3e}   ({time.time() - t0:.0f} s)", flush=True)
    base = rows[0]
    print("\nclass, FLOP share kept in bf16, dB gained over all-fp8, dB per 10 % of the Linear FLOPs")
    for name, share, p, e in rows[1:]:
        gain = p - base[2]
        per = gain / (share * 10) if share == share and share > 0 else float("nan")
        print(f"{name:45s} {share:5.3f}  {gain:+6.2f} dB  {per:6.2f}")


if __name__ == "__main__":
    main()
