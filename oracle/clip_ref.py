"""fp32 CPU restatement of the CLIP image / text encoder forward pass.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

What it follows
---------------
The reference calls ``model.encode_image`` / ``model.encode_text`` of the
third-party ``clip`` package (openai/CLIP; call sites
``src/clip/eval/evaluator_baseline.py:107,113,119``,
``src/clip/eval/evaluator.py:121,127,133``,
``src/clip/model/fusion_model.py:290,296,302``) after forcing fp32
(``src/clip/model/clip_model.py:43-44``).  That package is not in
``/root/reference``; this file restates its published architecture with the
state-dict names the reference's loader requires
(``src/clip/model/clip_model.py:52-64`` strict load, attribute names at
``:193-216``):

* vision: conv patch-embed (no bias) -> prepend ``class_embedding`` -> add
  ``positional_embedding`` -> ``ln_pre`` -> N x pre-LN residual blocks
  (``ln_1`` -> MHA -> +res -> ``ln_2`` -> ``c_fc`` -> QuickGELU ->
  ``c_proj`` -> +res) -> ``ln_post`` on token 0 -> ``@ proj``.
* text: ``token_embedding`` gather + ``positional_embedding`` -> N x the same
  block with a causal mask -> ``ln_final`` -> row at ``argmax(ids)`` (EOT) ->
  ``@ text_projection``.
* LayerNorm eps 1e-5, heads = width // 64, QuickGELU x * sigmoid(1.702 x).

The same math is stated by ``transformers.models.clip.modeling_clip``
(``:138-218`` vision embed, ``:221-256`` text embed, ``:259-277`` attention,
``:338-350`` MLP, ``:353-384`` block, ``:513-584`` text pooling, ``:613-652``
vision pooling, ``:683-750`` projections), which the reference also calls
(``src/clip/eval/evaluator_hf.py:115,130,144``);
``to_hf_state_dict`` below maps our names onto that class so the tests can
cross-check the two.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

# name -> architecture numbers (openai/CLIP model cards)
ARCHS: Dict[str, Dict[str, int]] = {
    "ViT-L/14": dict(embed_dim=768, image_size=224, patch=14, v_width=1024, v_layers=24,
                     t_width=768, t_layers=12, vocab=49408, ctx=77),
    "ViT-B/16": dict(embed_dim=512, image_size=224, patch=16, v_width=768, v_layers=12,
                     t_width=512, t_layers=12, vocab=49408, ctx=77),
    "ViT-B/32": dict(embed_dim=512, image_size=224, patch=32, v_width=768, v_layers=12,
                     t_width=512, t_layers=12, vocab=49408, ctx=77),
    # small shapes for fast tests (head dim stays 64 like every OpenAI CLIP)
    "tiny": dict(embed_dim=128, image_size=32, patch=8, v_width=256, v_layers=2,
                 t_width=256, t_layers=2, vocab=512, ctx=16),
    "tiny-long": dict(embed_dim=256, image_size=112, patch=8, v_width=256, v_layers=3,
                      t_width=512, t_layers=3, vocab=1024, ctx=77),
}


def random_state_dict(arch: Dict[str, int], seed: int = 0, scale: float = 1.0, outliers: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded random weights with OpenAI-CLIP names/shapes (std follows the
    upstream initialiser so activations stay O(1) through the depth).

    ``outliers=True`` (round 4) post-processes them with what trained CLIP towers show and N(0, sigma) weights
    do not (real weights cannot be fetched here): see :func:`add_outliers`."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * (std * scale)

    sd: Dict[str, torch.Tensor] = {}
    vw, tw, D = arch["v_width"], arch["t_width"], arch["embed_dim"]
    p, grid = arch["patch"], arch["image_size"] // arch["patch"]
    sd["visual.conv1.weight"] = rn(vw, 3, p, p, std=(3 * p * p) ** -0.5)
    sd["visual.class_embedding"] = rn(vw, std=vw ** -0.5)
    sd["visual.positional_embedding"] = rn(grid * grid + 1, vw, std=vw ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"visual.{nm}.weight"] = 1.0 + rn(vw, std=0.1)
        sd[f"visual.{nm}.bias"] = rn(vw, std=0.1)
    sd["visual.proj"] = rn(vw, D, std=vw ** -0.5)

    def blocks(prefix, width, layers):
        attn_std = width ** -0.5
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        fc_std = (2 * width) ** -0.5
        for i in range(layers):
            b = f"{prefix}.resblocks.{i}"
            sd[f"{b}.ln_1.weight"] = 1.0 + rn(width, std=0.1)
            sd[f"{b}.ln_1.bias"] = rn(width, std=0.1)
            sd[f"{b}.attn.in_proj_weight"] = rn(3 * width, width, std=attn_std)
            sd[f"{b}.attn.in_proj_bias"] = rn(3 * width, std=0.02)
            sd[f"{b}.attn.out_proj.weight"] = rn(width, width, std=proj_std)
            sd[f"{b}.attn.out_proj.bias"] = rn(width, std=0.02)
            sd[f"{b}.ln_2.weight"] = 1.0 + rn(width, std=0.1)
            sd[f"{b}.ln_2.bias"] = rn(width, std=0.1)
            sd[f"{b}.mlp.c_fc.weight"] = rn(4 * width, width, std=fc_std)
            sd[f"{b}.mlp.c_fc.bias"] = rn(4 * width, std=0.02)
            sd[f"{b}.mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std)
            sd[f"{b}.mlp.c_proj.bias"] = rn(width, std=0.02)

    blocks("visual.transformer", vw, arch["v_layers"])
    sd["token_embedding.weight"] = rn(arch["vocab"], tw, std=0.02)
    sd["positional_embedding"] = rn(arch["ctx"], tw, std=0.01)
    blocks("transformer", tw, arch["t_layers"])
    sd["ln_final.weight"] = 1.0 + rn(tw, std=0.1)
    sd["ln_final.bias"] = rn(tw, std=0.1)
    sd["text_projection"] = rn(tw, D, std=tw ** -0.5)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    if outliers:
        add_outliers(sd, arch, seed)
    return sd


def add_outliers(sd: Dict[str, torch.Tensor], arch: Dict[str, int], seed: int = 0, gain_lo: float = 3.0, gain_hi: float = 10.0,
                 per_norm: int = 4, massive: float = 60.0, cls_row: bool = True, sharp: float = 4.0,
                 migrate_lo: float = 30.0, migrate_hi: float = 100.0) -> Dict[str, torch.Tensor]:
    """Heavy tails of a trained tower, in place (test infrastructure; real weights cannot be fetched here, the magnitudes follow what
    is published about CLIP / ViT checkpoints -- residual "massive activations" two orders of magnitude above the median in a couple
    of channels, LayerNorm gains spread over two decades, a class / start-of-text row unlike every other row, sharp heads):

    * two MASSIVE residual channels per tower from the first LayerNorm on: ``visual.ln_pre.weight[c] *= massive`` (every token carries
      |x_c| ~ 60 sigma, up to ~200, into all blocks' statistics); text: ``token_embedding.weight[:, c] *= massive``;
    * a class-token-like row: ``visual.class_embedding[c'] = 1.5`` (50 sigma), ``positional_embedding[0, c'] = 0.5`` (50 sigma on the
      start-of-text row);
    * one sharp head per block: the query rows of head 0 scaled by ``sharp`` (logits x 4: near one-hot softmax rows);
    * in EVERY block ``per_norm`` channels of ln_1 and of ln_2 with their gain scaled by U(gain_lo, gain_hi) = 3 .. 10, NOT
      compensated: the function changes, those channels dominate the QKV / fc1 inputs;
    * in EVERY block ``per_norm`` further channels of ln_1 and ln_2 with gain AND bias scaled by g ~ U(migrate_lo, migrate_hi) =
      30 .. 100 and the consuming weight columns (in_proj_weight / c_fc.weight [:, c]) divided by g: the function is unchanged in real
      arithmetic, but the LayerNorm OUTPUT -- the A operand of the QKV / fc1 GEMM -- reaches 4 sigma x 100 = 400 in those channels,
      the neighbourhood of e4m3's +-448.  This is how large gains occur in trained towers (gain and weight column trade places
      freely under training); floating-point bf16 does not care, a unit-scale saturating e4m3 store does.

    Measured with :func:`encode_image` / :func:`encode_text` ``(bf16_operands=True)`` (an emulation of bf16 operands on the CPU, no
    kernel involved): UNcompensated gains of 30 .. 100 put the attention logits at ~1e4 -- the towers become chaotic functions and
    ANY bf16-operand arithmetic leaves the fp32 oracle by 1 - cos 3e-2 .. 1e-1 (ViT-B/32; gains of 10 .. 30: 4e-5 / 4e-4; 3 .. 10:
    4e-6 / 3e-5).  That regime says nothing about an implementation, so the uncompensated gains stop at 10."""
    g = torch.Generator().manual_seed(seed * 7919 + 17)

    def pick(width, k):
        return torch.randperm(width, generator=g)[:k]

    vw, tw = arch["v_width"], arch["t_width"]
    mv, mt = pick(vw, 3), pick(tw, 3)
    if massive:
        sd["visual.ln_pre.weight"][mv[:2]] *= massive
        sd["token_embedding.weight"][:, mt[:2]] *= massive
    if cls_row:
        sd["visual.class_embedding"][mv[2]] = 1.5
        sd["positional_embedding"][0, mt[2]] = 0.5
    for prefix, width, layers in (("visual.transformer", vw, arch["v_layers"]), ("transformer", tw, arch["t_layers"])):
        for i in range(layers):
            b = f"{prefix}.resblocks.{i}"
            for nm, consumer in (("ln_1", f"{b}.attn.in_proj_weight"), ("ln_2", f"{b}.mlp.c_fc.weight")):
                ch = pick(width, 2 * per_norm)
                gain = gain_lo + (gain_hi - gain_lo) * torch.rand(per_norm, generator=g)
                mig = migrate_lo + (migrate_hi - migrate_lo) * torch.rand(per_norm, generator=g)
                if gain_hi > 0:
                    sd[f"{b}.{nm}.weight"][ch[:per_norm]] *= gain
                if migrate_hi > 0:
                    sd[f"{b}.{nm}.weight"][ch[per_norm:]] *= mig
                    sd[f"{b}.{nm}.bias"][ch[per_norm:]] *= mig
                    sd[consumer][:, ch[per_norm:]] /= mig
            if sharp:
                sd[f"{b}.attn.in_proj_weight"][:64] *= sharp
                sd[f"{b}.attn.in_proj_bias"][:64] *= sharp
    return sd


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _block(x: torch.Tensor, sd, prefix: str, heads: int, causal: bool, bf16_operands: bool = False) -> torch.Tensor:
    """One pre-LN residual attention block on x [B, T, W] (fp32).

    ``bf16_operands`` (a DIAGNOSTIC of the precision class, not the oracle): every GEMM / attention operand and every stored
    intermediate is rounded to bf16 where the build's default precision holds it in bf16 (LayerNorm outputs, weights, q | k | v, the
    softmax probabilities, the attention output, the out-proj / fc2 updates, the MLP hidden), fp32 accumulation and an fp32
    residual stream -- what ANY bf16-operand implementation computes, up to summation order.  tools/outlier_stress.py uses it to
    tell the cost of bf16 operands on heavy-tailed weights from a defect of the kernels."""
    r = _bf16 if bf16_operands else (lambda t: t)
    B, T, W = x.shape
    hd = W // heads
    h = r(F.layer_norm(x, (W,), sd[f"{prefix}.ln_1.weight"], sd[f"{prefix}.ln_1.bias"], 1e-5))
    wq, bq = sd[f"{prefix}.attn.in_proj_weight"], sd[f"{prefix}.attn.in_proj_bias"]
    if bf16_operands:                              # the build folds 1 / sqrt(64) into W_q / b_q (exact) and stores q | k | v as bf16
        scale = torch.ones(3 * W)
        scale[:W] = hd ** -0.5
        qkv = r(h @ r(wq * scale[:, None]).T + bq * scale)
        q, k, v = qkv.split(W, dim=-1)
    else:
        qkv = h @ wq.T + bq
        q, k, v = qkv.split(W, dim=-1)
        q = q * (hd ** -0.5)
    q = q.view(B, T, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu_(1)
    if bf16_operands:                              # row sum of the fp32 exponentials, P rounded to bf16 for the PV product
        p = torch.exp(s - s.amax(dim=-1, keepdim=True))
        a = (r(p) @ v) / p.sum(dim=-1, keepdim=True)
    else:
        a = torch.softmax(s, dim=-1) @ v
    a = r(a.transpose(1, 2).reshape(B, T, W))
    x = x + r(a @ r(sd[f"{prefix}.attn.out_proj.weight"]).T + sd[f"{prefix}.attn.out_proj.bias"])
    h = r(F.layer_norm(x, (W,), sd[f"{prefix}.ln_2.weight"], sd[f"{prefix}.ln_2.bias"], 1e-5))
    h = h @ r(sd[f"{prefix}.mlp.c_fc.weight"]).T + sd[f"{prefix}.mlp.c_fc.bias"]
    h = r(h * torch.sigmoid(1.702 * h))
    return x + r(h @ r(sd[f"{prefix}.mlp.c_proj.weight"]).T + sd[f"{prefix}.mlp.c_proj.bias"])


@torch.no_grad()
def encode_image(sd, arch, pixels: torch.Tensor, return_hidden: bool = False, bf16_operands: bool = False) -> torch.Tensor:
    """pixels [B,3,S,S] fp32 (already mean/std normalised) -> [B, embed_dim] fp32 (un-normalised)."""
    sd = {k: v.float() for k, v in sd.items() if k.startswith("visual.")}
    vw, p = arch["v_width"], arch["patch"]
    x = F.conv2d(pixels.float(), sd["visual.conv1.weight"], stride=p)          # [B, W, g, g]
    x = x.flatten(2).transpose(1, 2)                                            # [B, g*g, W]
    cls = sd["visual.class_embedding"].expand(x.shape[0], 1, vw)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (vw,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    for i in range(arch["v_layers"]):
        x = _block(x, sd, f"visual.transformer.resblocks.{i}", vw // 64, causal=False, bf16_operands=bf16_operands)
    if return_hidden:
        return x
    x = F.layer_norm(x[:, 0, :], (vw,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return x @ sd["visual.proj"]


@torch.no_grad()
def encode_text(sd, arch, ids: torch.Tensor, return_hidden: bool = False, bf16_operands: bool = False) -> torch.Tensor:
    """ids [B, ctx] int -> [B, embed_dim] fp32 (un-normalised); pooled at argmax(ids) (EOT = highest id)."""
    sd = {k: v.float() for k, v in sd.items() if not k.startswith("visual.")}
    tw = arch["t_width"]
    ids = ids.long()
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][: ids.shape[1]]
    for i in range(arch["t_layers"]):
        x = _block(x, sd, f"transformer.resblocks.{i}", tw // 64, causal=True, bf16_operands=bf16_operands)
    if return_hidden:
        return x
    x = F.layer_norm(x, (tw,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    x = x[torch.arange(x.shape[0]), ids.argmax(dim=-1)]
    return x @ sd["text_projection"]


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """`x / x.norm(dim=-1, keepdim=True)`, no eps (evaluator_baseline.py:108,114,120)."""
    return x / x.norm(dim=-1, keepdim=True)


def to_hf_state_dict(sd, arch) -> Dict[str, torch.Tensor]:
    """Map OpenAI-CLIP names to ``transformers.CLIPModel`` names."""
    out: Dict[str, torch.Tensor] = {"logit_scale": sd["logit_scale"].clone()}

    def blocks(src, dst, width, layers):
        for i in range(layers):
            s, d = f"{src}.resblocks.{i}", f"{dst}.encoder.layers.{i}"
            w, b = sd[f"{s}.attn.in_proj_weight"], sd[f"{s}.attn.in_proj_bias"]
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                out[f"{d}.self_attn.{nm}.weight"] = w[j * width:(j + 1) * width].clone()
                out[f"{d}.self_attn.{nm}.bias"] = b[j * width:(j + 1) * width].clone()
            out[f"{d}.self_attn.out_proj.weight"] = sd[f"{s}.attn.out_proj.weight"].clone()
            out[f"{d}.self_attn.out_proj.bias"] = sd[f"{s}.attn.out_proj.bias"].clone()
            for a, bname in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2")):
                out[f"{d}.{bname}.weight"] = sd[f"{s}.{a}.weight"].clone()
                out[f"{d}.{bname}.bias"] = sd[f"{s}.{a}.bias"].clone()
            for a, bname in (("c_fc", "fc1"), ("c_proj", "fc2")):
                out[f"{d}.mlp.{bname}.weight"] = sd[f"{s}.mlp.{a}.weight"].clone()
                out[f"{d}.mlp.{bname}.bias"] = sd[f"{s}.mlp.{a}.bias"].clone()

    out["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"].clone()
    out["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"].clone()
    out["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"].clone()
    out["vision_model.pre_layrnorm.weight"] = sd["visual.ln_pre.weight"].clone()
    out["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.bias"].clone()
    out["vision_model.post_layernorm.weight"] = sd["visual.ln_post.weight"].clone()
    out["vision_model.post_layernorm.bias"] = sd["visual.ln_post.bias"].clone()
    out["visual_projection.weight"] = sd["visual.proj"].T.contiguous()
    blocks("visual.transformer", "vision_model", arch["v_width"], arch["v_layers"])
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"].clone()
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"].clone()
    out["text_model.final_layer_norm.weight"] = sd["ln_final.weight"].clone()
    out["text_model.final_layer_norm.bias"] = sd["ln_final.bias"].clone()
    out["text_projection.weight"] = sd["text_projection"].T.contiguous()
    blocks("transformer", "text_model", arch["t_width"], arch["t_layers"])
    return out


def hf_config_kwargs(arch) -> dict:
    """kwargs for ``transformers.CLIPConfig`` describing ``arch`` (local object, nothing is fetched)."""
    eot = arch["vocab"] - 1
    return dict(
        text_config=dict(hidden_size=arch["t_width"], intermediate_size=4 * arch["t_width"],
                         num_attention_heads=arch["t_width"] // 64, num_hidden_layers=arch["t_layers"],
                         projection_dim=arch["embed_dim"], vocab_size=arch["vocab"],
                         max_position_embeddings=arch["ctx"], eos_token_id=eot, bos_token_id=eot - 1,
                         pad_token_id=0, hidden_act="quick_gelu"),
        vision_config=dict(hidden_size=arch["v_width"], intermediate_size=4 * arch["v_width"],
                           num_attention_heads=arch["v_width"] // 64, num_hidden_layers=arch["v_layers"],
                           patch_size=arch["patch"], image_size=arch["image_size"],
                           projection_dim=arch["embed_dim"], hidden_act="quick_gelu"),
        projection_dim=arch["embed_dim"],
    )


def synthetic_ids(arch, n: int, seed: int = 1235) -> torch.Tensor:
    """Synthetic token ids per SURVEY.md section 8(d): SOT, L-1 random ids, one EOT (= row max), zero pad."""
    g = torch.Generator().manual_seed(seed)
    ctx, vocab = arch["ctx"], arch["vocab"]
    sot, eot = vocab - 2, vocab - 1
    ids = torch.zeros(n, ctx, dtype=torch.int32)
    lens = torch.randint(min(8, ctx - 2), ctx, (n,), generator=g)
    body = torch.randint(1, sot, (n, ctx), generator=g, dtype=torch.int32)
    for i in range(n):
        L = int(lens[i])
        ids[i, 0] = sot
        ids[i, 1:L] = body[i, 1:L]
        ids[i, L] = eot
    return ids
