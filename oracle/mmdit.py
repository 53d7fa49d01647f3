"""CPU ORACLE (test infrastructure, NOT product code) -- MMDiT denoiser.

PyTorch-CPU restatement of the reference's MLX MMDiT, used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP
path.  Nothing under diffusionkit_amd/ imports this package.

PARITY: WIRING PINNED, MLX ARITHMETIC UNPINNED.  The reference's arithmetic lives in MLX 0.17.3 (setup.py:32), which cannot be
imported in this environment, and the reference ships no local golden vectors for this path (its only numeric gate is an
image-PSNR check against network-hosted PNGs, tests/mlx/test_diffusion_pipeline.py:91-93).  What the restatement is pinned by:
(a) outputs of the reference's OWN MLX model code -- mlx/{config,mmdit,sampler,vae}.py imported unmodified and run on a
PyTorch-backed stand-in for the few dozen MLX operations they call (tests/golden/mlx_standin.py,
tests/golden/make_reference_mlx_fixtures.py; replayed by tests/test_reference_mlx_golden.py): FLUX double + single blocks with
RoPE and QK-norm at batch 1 and 2, SD3 at batch 1 and 2, the SD3.5 shape class, the modulation cache, VAE decoder and encoder,
sampler schedules, and DiffusionPipeline.denoise_latents end to end (CFG, FLUX, img2img) -- the exact-math oracle agrees to
2e-7 .. 4e-7 relative (2e-5 over whole step loops); (b) outputs of the reference's own PyTorch modules
(torch/{mmdit,vae,model_io}.py, tests/golden/make_reference_torch_fixtures.py, tests/test_reference_torch_golden.py: SD3 MMDiT
4e-7, VAE decoder 2e-6, both checkpoint key maps); (c) scalar known-answer values derived from the reference's formulas
(tests/golden/kat_scalars.json); (d) self-consistency goldens.  What stays unpinned is what only real MLX can tell: where its
bf16 / fp16 kernels round (the Prec(act=...) mode below is a model of that).

Each function cites the reference lines it follows
(paths relative to python/src/diffusionkit/mlx/).

Precision model: all tensors are float32 on the host.  ``Prec(act=None)`` is the
exact-math oracle.  ``Prec(act=torch.bfloat16)`` rounds to the activation dtype at every
MLX op boundary of the reference graph (quirks Q2/Q4/Q5 of SURVEY.md §3.4), i.e. it
emulates the reference's low-precision path; weights are expected to hold values that
are already representable in the weight dtype.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class Prec:
    """Rounding policy. act=None: no rounding (fp32 oracle)."""

    def __init__(self, act: Optional[torch.dtype] = None, sdpa: str = "ref"):
        self.act = act
        self.sdpa = sdpa  # "ref": materialised low-precision scores (quirk Q4)

    def r(self, x: Tensor) -> Tensor:
        if self.act is None:
            return x
        return x.to(self.act).to(torch.float32)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], P: Prec) -> Tensor:
    """nn.Linear: x @ W^T + b, fp32 accumulate, one rounding (mmdit.py:56,821-832)."""
    y = x @ w.t()
    if b is not None:
        y = y + b
    return P.r(y)


def silu(x: Tensor, P: Prec) -> Tensor:
    return P.r(x * torch.sigmoid(x))


def gelu_erf(x: Tensor, P: Prec) -> Tensor:
    """nn.GELU() = exact erf GELU (mmdit.py:421, quirk Q3)."""
    return P.r(0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0))))


def gelu_tanh(x: Tensor, P: Prec) -> Tensor:
    """nn.GELU(approximate="tanh"): what the reference's PyTorch MMDiT uses (torch/mmdit.py:242) where its MLX one uses the
    exact form; only selected by tests that replay outputs of that PyTorch module (tests/test_reference_torch_golden.py)."""
    return P.r(torch.nn.functional.gelu(x, approximate="tanh"))


def layer_norm(x: Tensor, eps: float) -> Tensor:
    """mx.fast.layer_norm without affine (mmdit.py:838-849), fp32 statistics."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


def affine_transform(x: Tensor, shift: Tensor, residual_scale: Tensor, eps: float, P: Prec) -> Tensor:
    """Adaptive-LN modulation (mmdit.py:958-972).

    batch 1: fused fast.layer_norm(x, 1+scale, shift): (1+scale) is rounded to the
    activation dtype first, everything else happens in fp32, one output rounding.
    batch>1: LN -> round -> *(1+scale) -> round -> +shift -> round (quirk Q5).
    x: [B, S, h]; shift/scale: [B, 1, h].
    """
    w = P.r(1.0 + residual_scale)
    if x.shape[0] == 1:
        return P.r(layer_norm(x, eps) * w + shift)
    return P.r(P.r(P.r(layer_norm(x, eps)) * w) + shift)


def rms_norm(x: Tensor, w: Tensor, eps: float, P: Prec) -> Tensor:
    """nn.RMSNorm -> mx.fast.rms_norm (mmdit.py:754-764): fp32 accumulate, one rounding."""
    return P.r(x * torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + eps) * w)


def timestep_embedding(t: Tensor, cfg, P_embed: Prec) -> Tensor:
    """Sinusoidal embedding evaluated in config.dtype (mmdit.py:379-389, quirk Q2).

    frequencies = exp(-log(max_period) * arange(half, dtype) / half).astype(dtype):
    evaluated in fp32 (fp32 scalar * low-precision arange promotes), rounded once;
    args = t.astype(dtype) * frequencies (low-precision product); cos/sin outputs rounded.
    """
    half = cfg.frequency_embed_dim // 2
    ar = P_embed.r(torch.arange(half, dtype=torch.float32))
    freqs = P_embed.r(torch.exp(-torch.log(torch.tensor(float(cfg.max_period))) * ar / half))
    args = P_embed.r(P_embed.r(t.to(torch.float32))[:, None] * freqs[None])
    return torch.cat([P_embed.r(torch.cos(args)), P_embed.r(torch.sin(args))], dim=-1)


def embed_dtype(cfg) -> torch.dtype:
    return {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}[cfg.dtype]


def rope_table(cfg, text_len: int, h: int, w: int) -> Tensor:
    """cos/sin table [S, D/2, 2] for the joint [text, image] sequence
    (mmdit.py:865-911, quirk Q14): text positions are all zero; image tokens use
    (0, row, col); axes_dim e.g. (16,56,56); omega = theta^(-2i/dim)."""
    S = text_len + h * w
    pos = torch.zeros(S, 3, dtype=torch.float32)
    rows = torch.arange(h, dtype=torch.float32)[:, None].expand(h, w).reshape(-1)
    cols = torch.arange(w, dtype=torch.float32)[None, :].expand(h, w).reshape(-1)
    pos[text_len:, 1] = rows
    pos[text_len:, 2] = cols
    outs = []
    for i, dim in enumerate(cfg.rope_axes_dim):
        scale = torch.arange(0, dim, 2, dtype=torch.float32) / dim
        omega = 1.0 / (float(cfg.rope_theta) ** scale)
        ang = pos[:, i:i + 1] * omega[None, :]
        outs.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1))
    return torch.cat(outs, dim=1)  # [S, D/2, 2]


def rope_apply(x: Tensor, table: Tensor, P: Prec) -> Tensor:
    """RoPE.apply (mmdit.py:934-942): adjacent pairs, fp32 math, one rounding.
    x: [B, H, S, D]; table: [S, D/2, 2]."""
    xe, xo = x[..., 0::2], x[..., 1::2]
    c, s = table[..., 0], table[..., 1]
    out = torch.stack([c * xe - s * xo, s * xe + c * xo], dim=-1)
    return P.r(out.flatten(-2))


def sdpa(q: Tensor, k: Tensor, v: Tensor, scale: float, P: Prec) -> Tensor:
    """mx.fast.scaled_dot_product_attention, MLX 0.17.3 unfused fallback for L_q>1
    (mmdit.py:562,643,687,736; quirk Q4): (q*scale)@k^T materialised in the activation
    dtype, fp32 softmax, probabilities rounded, @v.  q,k,v: [B,H,S,D]."""
    s = P.r(P.r(q * scale) @ k.transpose(-1, -2))
    p = P.r(torch.softmax(s, dim=-1))
    return P.r(p @ v)


class OracleMMDiT:
    """Functional restatement of MMDiT (mmdit.py:22-266).

    ``weights``: dict keyed by the reference module-tree names, MLX layouts
    (Linear [out,in]; x_embedder.proj.weight [out,kh,kw,in]).
    """

    def __init__(self, cfg, weights: Dict[str, Tensor], prec: Optional[Prec] = None, gelu: str = "erf", act_quant=None,
                 guidance: Optional[float] = None, embed_prec: Optional[Prec] = None):
        self.cfg = cfg
        self.w = weights
        self.P = prec or Prec()
        # fp8 path of the MI355X build (no reference counterpart, oracle/fp8.py): fake-quantisation applied to the INPUT of every
        # Linear of the transformer blocks (q/k/v, o_proj, fc1, fc2); the weights handed in are then the dequantised fp8 ones
        # ``act_quant(x)`` or, for precision-policy studies (scripts/fp8_policy_cpu.py), ``act_quant(x, site)`` with site = (block prefix,
        # "qkv" | "o" | "fc1" | "fc2"): which Linear is about to consume x
        if act_quant is None:
            self.aq = lambda x, site=None: x
        elif getattr(act_quant, "takes_site", False):
            self.aq = act_quant
        else:
            self.aq = lambda x, site=None: act_quant(x)
        # FLUX.1-dev guidance strength (cfg.guidance_embed): see cache_modulation_params
        self.guidance = guidance
        self.gelu = {"erf": gelu_erf, "tanh": gelu_tanh}[gelu]  # "erf" = the MLX path (quirk Q3)
        # timestep embedding is evaluated in config.dtype independently of the activation
        # dtype (quirk Q2); the exact oracle keeps it exact.
        # ``embed_prec`` overrides that: the reference evaluates frequencies / arguments / cos / sin in ``config.dtype`` whatever
        # the activation dtype is (mmdit.py:379-389: bf16 for FLUX, fp16 for SD3, also with a16 = False), so
        # ``OracleMMDiT(cfg, w, Prec(), embed_prec=Prec(embed_dtype(cfg)))`` is the function the reference computes with fp32
        # activations.  It matters: t = 752 times a bf16-rounded frequency moves the phase of the fast sinusoids by radians, and
        # the whole modulation table with it -- a fp32 oracle WITHOUT the quirk sits 11 % (rel-L2 of a FLUX block's image
        # stream) away from any implementation WITH it, bf16 rounding noise is 0.7 % (round 3: what held the FLUX full-size
        # fixtures of round 2 at 27-32 dB).
        self.P_embed = embed_prec if embed_prec is not None else (Prec(embed_dtype(cfg)) if self.P.act is not None else Prec())
        self._mod: Dict[str, Dict[float, Tensor]] = {}
        self._rope = None
        self._rope_key = None

    # ---- small helpers -------------------------------------------------------------
    def _lin(self, x, name, bias=True):
        return linear(x, self.w[name + ".weight"], self.w.get(name + ".bias") if bias else None, self.P)

    def _mlp_embed(self, x, prefix):
        """Linear -> SiLU -> Linear (mmdit.py:357-361, 372-376)."""
        y = self._lin(x, prefix + ".mlp.layers.0")
        y = silu(y, self.P)
        return self._lin(y, prefix + ".mlp.layers.2")

    # ---- modulation cache (mmdit.py:77-180) ----------------------------------------
    def cache_modulation_params(self, pooled: Tensor, timesteps: Tensor) -> None:
        cfg, P = self.cfg, self.P
        B = pooled.shape[0]
        y_embed = self._mlp_embed(P.r(pooled), "y_embedder")  # [B,h]
        if getattr(cfg, "guidance_embed", False):
            # The reference declares guidance_in = MLPEmbedder(frequency_embed_dim -> hidden) (mmdit.py:31-36,945-955) but never
            # reaches its call site (:219-220; model_io.py:109 runs FLUX.1-dev on the schnell preset, quirk Q7).  Restated here
            # with the published FLUX.1-dev semantics the module tree was written for: the sinusoidal embedding of
            # 1000 * guidance through the MLP, added to the modulation vector of every batch row and timestep.
            g = torch.full((1,), 1000.0 * float(self.guidance if self.guidance is not None else 3.5))
            gemb = timestep_embedding(g, cfg, self.P_embed)
            y_embed = P.r(y_embed + self._mlp_embed(P.r(gemb), "guidance_in"))
        self._mod = {}
        for t in timesteps:
            key = float(t)
            temb = timestep_embedding(t.reshape(1).repeat(B), cfg, self.P_embed)
            t_embed = self._mlp_embed(P.r(temb), "t_embedder")
            vec = P.r(y_embed + t_embed)  # [B,h]
            act = silu(vec, P)  # adaLN_modulation = SiLU -> Linear (mmdit.py:430-435)
            for name in self._adaln_names():
                out = self._lin(act, name + ".adaLN_modulation.layers.1")
                self._mod.setdefault(name, {})[key] = out[:, None, :]  # [B,1,n*h]

    def _adaln_names(self):
        cfg = self.cfg
        names = []
        for i in range(cfg.depth_multimodal):
            names.append(f"multimodal_transformer_blocks.{i}.image_transformer_block")
            names.append(f"multimodal_transformer_blocks.{i}.text_transformer_block")
        for i in range(cfg.depth_unified):
            names.append(f"unified_transformer_blocks.{i}.transformer_block")
        names.append("final_layer")
        return names

    # ---- transformer block halves (mmdit.py:440-548) -------------------------------
    def _pre_sdpa(self, x: Tensor, prefix: str, tkey: float, n_mod: int):
        cfg, P = self.cfg, self.P
        mod = self._mod[prefix][tkey].chunk(n_mod, dim=-1)
        m = affine_transform(x, mod[0], mod[1], cfg.layer_norm_eps, P)
        mq = self.aq(m, (prefix, "qkv"))
        q = self._lin(mq, prefix + ".attn.q_proj")
        k = self._lin(mq, prefix + ".attn.k_proj", bias=False)  # quirk Q9
        v = self._lin(mq, prefix + ".attn.v_proj")
        B, S, _ = x.shape
        H, D = cfg.num_heads, cfg.head_dim

        def heads(t):
            return t.reshape(B, S, H, D).transpose(1, 2)

        q, k, v = heads(q), heads(k), heads(v)
        if cfg.use_qk_norm:
            q = rms_norm(q, self.w[prefix + ".qk_norm.q_norm.weight"], 1e-6, P)
            k = rms_norm(k, self.w[prefix + ".qk_norm.k_norm.weight"], 1e-6, P)
        return {"q": q, "k": k, "v": v, "m": mq, "mod": mod}

    def _post_sdpa(self, residual, sdpa_out, inter, prefix, parallel_mlp):
        cfg, P = self.cfg, self.P
        mod = inter["mod"]
        attn_out = self._lin(self.aq(sdpa_out, (prefix, "o")), prefix + ".attn.o_proj")
        if parallel_mlp:
            # fc2 bias is zeroed on every call (mmdit.py:741-742, quirk Q8)
            h1 = self.gelu(self._lin(inter["m"], prefix + ".mlp.fc1"), P)
            mlp_out = linear(self.aq(h1, (prefix, "fc2")), self.w[prefix + ".mlp.fc2.weight"], None, P)
            return P.r(residual + P.r(mod[2] * P.r(attn_out + mlp_out)))
        residual = P.r(residual + P.r(attn_out * mod[2]))
        m2 = affine_transform(residual, mod[3], mod[4], cfg.layer_norm_eps, P)
        h1 = self.gelu(self._lin(self.aq(m2, (prefix, "fc1")), prefix + ".mlp.fc1"), P)
        mlp_out = self._lin(self.aq(h1, (prefix, "fc2")), prefix + ".mlp.fc2")
        return P.r(residual + P.r(mod[5] * mlp_out))

    def _merge(self, t):  # [B,H,S,D] -> [B,S,h]
        B, H, S, D = t.shape
        return t.transpose(1, 2).reshape(B, S, H * D)

    # ---- blocks --------------------------------------------------------------------
    def _double_block(self, i, img, txt, tkey, rope):
        cfg, P = self.cfg, self.P
        pi = f"multimodal_transformer_blocks.{i}.image_transformer_block"
        pt = f"multimodal_transformer_blocks.{i}.text_transformer_block"
        skip_txt = (i == cfg.depth_multimodal - 1) and cfg.depth_unified < 1
        ii = self._pre_sdpa(img, pi, tkey, 6)
        ti = self._pre_sdpa(txt, pt, tkey, 2 if skip_txt else 6)
        S_i, S_t = img.shape[1], txt.shape[1]
        if cfg.depth_unified > 0:  # FLUX order [text, image] (mmdit.py:594-606)
            q, k, v = (torch.cat([ti[n], ii[n]], dim=2) for n in "qkv")
        else:  # SD3 order [image, text] (mmdit.py:607-625)
            q, k, v = (torch.cat([ii[n], ti[n]], dim=2) for n in "qkv")
        if rope is not None:
            q, k = rope_apply(q, rope, P), rope_apply(k, rope, P)
        o = self._merge(sdpa(q, k, v, 1.0 / math.sqrt(cfg.head_dim), P))
        if cfg.depth_unified > 0:
            o_txt, o_img = o[:, :S_t], o[:, S_t:]
        else:
            o_img, o_txt = o[:, :S_i], o[:, S_i:]
        img = self._post_sdpa(img, o_img, ii, pi, False)
        txt = None if skip_txt else self._post_sdpa(txt, o_txt, ti, pt, False)
        return img, txt

    def _single_block(self, i, x, tkey, rope):
        cfg, P = self.cfg, self.P
        p = f"unified_transformer_blocks.{i}.transformer_block"
        par = cfg.parallel_mlp_for_unified_blocks
        it = self._pre_sdpa(x, p, tkey, 3 if par else 6)
        q, k, v = it["q"], it["k"], it["v"]
        if rope is not None:
            q, k = rope_apply(q, rope, P), rope_apply(k, rope, P)
        o = self._merge(sdpa(q, k, v, 1.0 / math.sqrt(cfg.head_dim), P))
        return self._post_sdpa(x, o, it, p, par)

    # ---- embedders -----------------------------------------------------------------
    def _patch_embed(self, x: Tensor) -> Tensor:
        """LatentImageAdapter (mmdit.py:269-302) [+ pos-emb :324-349]. x: [B,Hl,Wl,C]."""
        cfg, P = self.cfg, self.P
        B, Hl, Wl, C = x.shape
        p = cfg.patch_size
        w = self.w["x_embedder.proj.weight"]
        if cfg.patchify_via_reshape:  # features ordered (c, ph, pw)
            t = x.reshape(B, Hl // p, p, Wl // p, p, C).permute(0, 1, 3, 5, 2, 4)
        else:  # strided conv: features ordered (kh, kw, c)
            t = x.reshape(B, Hl // p, p, Wl // p, p, C).permute(0, 1, 3, 2, 4, 5)
        t = t.reshape(B, (Hl // p) * (Wl // p), -1)
        y = linear(t, w.reshape(w.shape[0], -1), self.w["x_embedder.proj.bias"], P)
        if "x_pos_embedder.pos_embed.weight" in self.w:
            mh = cfg.max_latent_resolution
            h, w_ = Hl // p, Wl // p
            y0, x0 = (mh - h) // 2, (mh - w_) // 2
            pe = self.w["x_pos_embedder.pos_embed.weight"].reshape(mh, mh, -1)
            pe = pe[y0:y0 + h, x0:x0 + w_].reshape(1, h * w_, -1)
            y = P.r(y + pe)
        return y

    def _unpatch(self, y: Tensor, Hl: int, Wl: int) -> Tensor:
        cfg = self.cfg
        B = y.shape[0]
        p = cfg.patch_size
        h, w = Hl // p, Wl // p
        if cfg.patchify_via_reshape:  # unpack (mmdit.py:304-321)
            return y.reshape(B, h, w, -1, p, p).permute(0, 1, 4, 2, 5, 3).reshape(B, Hl, Wl, -1)
        # unpatchify (mmdit.py:975-988)
        c = cfg.vae_latent_dim
        return y.reshape(B, h, w, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(B, Hl, Wl, c)

    # ---- forward (mmdit.py:188-266) ------------------------------------------------
    def __call__(self, latent: Tensor, text: Tensor, timestep: float, taps: Optional[dict] = None) -> Tensor:
        cfg, P = self.cfg, self.P
        tkey = float(timestep)
        B, Hl, Wl, _ = latent.shape
        txt = self._lin(P.r(text), "context_embedder")
        img = self._patch_embed(P.r(latent))
        rope = None
        if cfg.rope_axes_dim is not None:
            key = (txt.shape[1], Hl // cfg.patch_size, Wl // cfg.patch_size)
            if self._rope_key != key:
                self._rope, self._rope_key = rope_table(cfg, *key), key
            rope = self._rope
        if taps is not None:
            taps["embed_img"], taps["embed_txt"] = img.clone(), txt.clone()
        for i in range(cfg.depth_multimodal):
            img, txt = self._double_block(i, img, txt, tkey, rope)
            if taps is not None:
                taps[f"double{i}_img"] = img.clone()
                if txt is not None:
                    taps[f"double{i}_txt"] = txt.clone()
        if cfg.depth_unified > 0:
            S_t = txt.shape[1]
            x = torch.cat([txt, img], dim=1)
            for i in range(cfg.depth_unified):
                x = self._single_block(i, x, tkey, rope)
                if taps is not None:
                    taps[f"single{i}"] = x.clone()
            img = x[:, S_t:]
        mod = self._mod["final_layer"][tkey].chunk(2, dim=-1)  # FinalLayer (mmdit.py:767-796)
        y = affine_transform(img, mod[0], mod[1], cfg.layer_norm_eps, P)
        y = self._lin(y, "final_layer.linear")
        if taps is not None:
            taps["final"] = y.clone()
        return self._unpatch(y, Hl, Wl)
