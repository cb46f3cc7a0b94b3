#!/usr/bin/env python3
"""Round 4, CPU only (no kernel involved): what the ALGEBRAICALLY FUSED LayerNorm of DESIGN.md section 7 would cost in accuracy
(VERDICT r3 #4b: "check its error under the outlier weights before anything else, since it rounds raw x to bf16").

Fused form of LN -> GEMM:  y = rstd * (bf16(x) . bf16(gamma * W)^T - mu * colsum(bf16(gamma * W))) + (beta . W^T + b), i.e. the GEMM reads the
RAW residual row rounded to bf16 and the normalisation happens on the fp32 accumulators (statistics from the fp32 row).  Against
it: the build's arithmetic, which normalises in fp32 FIRST and rounds the normalised row to bf16 (clip_ref ... bf16_operands=True).
Both against the fp32 oracle, on plain and on heavy-tailed weights (clip_ref.add_outliers).

    python tools/probe_fused_layernorm.py [ViT-B/32|tiny-long] [n_images] [n_texts]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import clip_ref  # noqa: E402

r = clip_ref._bf16


def fused_ln_gemm(x, gamma, beta, W, b):
    """x [.., K] fp32 residual rows -> LN(x) . W^T + b computed the fused way."""
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    Wg = r(W * gamma[None, :])
    acc = r(x) @ Wg.T
    return rstd * (acc - mu * Wg.sum(-1)[None, :]) + (beta @ W.T + b)


def block(x, sd, prefix, heads, causal, fused):
    B, T, W = x.shape
    hd = W // heads
    wq, bq = sd[f"{prefix}.attn.in_proj_weight"], sd[f"{prefix}.attn.in_proj_bias"]
    scale = torch.ones(3 * W)
    scale[:W] = hd ** -0.5
    if fused:
        qkv = r(fused_ln_gemm(x, sd[f"{prefix}.ln_1.weight"], sd[f"{prefix}.ln_1.bias"], wq * scale[:, None], bq * scale))
    else:
        h = r(F.layer_norm(x, (W,), sd[f"{prefix}.ln_1.weight"], sd[f"{prefix}.ln_1.bias"], 1e-5))
        qkv = r(h @ r(wq * scale[:, None]).T + bq * scale)
    q, k, v = [t.view(B, T, heads, hd).transpose(1, 2) for t in qkv.split(W, dim=-1)]
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu_(1)
    p = torch.exp(s - s.amax(dim=-1, keepdim=True))
    a = r(((r(p) @ v) / p.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, T, W))
    x = x + r(a @ r(sd[f"{prefix}.attn.out_proj.weight"]).T + sd[f"{prefix}.attn.out_proj.bias"])
    if fused:
        h = fused_ln_gemm(x, sd[f"{prefix}.ln_2.weight"], sd[f"{prefix}.ln_2.bias"], sd[f"{prefix}.mlp.c_fc.weight"], sd[f"{prefix}.mlp.c_fc.bias"])
    else:
        h = r(F.layer_norm(x, (W,), sd[f"{prefix}.ln_2.weight"], sd[f"{prefix}.ln_2.bias"], 1e-5))
        h = h @ r(sd[f"{prefix}.mlp.c_fc.weight"]).T + sd[f"{prefix}.mlp.c_fc.bias"]
    h = r(h * torch.sigmoid(1.702 * h))
    return x + r(h @ r(sd[f"{prefix}.mlp.c_proj.weight"]).T + sd[f"{prefix}.mlp.c_proj.bias"])


@torch.no_grad()
def encode_image(sd, arch, px, fused):
    sd = {k: v.float() for k, v in sd.items() if k.startswith("visual.")}
    vw, p = arch["v_width"], arch["patch"]
    x = F.conv2d(px, sd["visual.conv1.weight"], stride=p).flatten(2).transpose(1, 2)
    x = torch.cat([sd["visual.class_embedding"].expand(x.shape[0], 1, vw), x], dim=1) + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (vw,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    for i in range(arch["v_layers"]):
        x = block(x, sd, f"visual.transformer.resblocks.{i}", vw // 64, False, fused)
    return F.layer_norm(x[:, 0, :], (vw,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5) @ sd["visual.proj"]


@torch.no_grad()
def encode_text(sd, arch, ids, fused):
    sd = {k: v.float() for k, v in sd.items() if not k.startswith("visual.")}
    tw, ids = arch["t_width"], ids.long()
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][: ids.shape[1]]
    for i in range(arch["t_layers"]):
        x = block(x, sd, f"transformer.resblocks.{i}", tw // 64, True, fused)
    x = F.layer_norm(x, (tw,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    return x[torch.arange(x.shape[0]), ids.argmax(dim=-1)] @ sd["text_projection"]


name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ntxt = int(sys.argv[3]) if len(sys.argv) > 3 else 4
oa = clip_ref.ARCHS[name]
g = torch.Generator().manual_seed(1234)
px = torch.randn(nimg, 3, oa["image_size"], oa["image_size"], generator=g)
ids = clip_ref.synthetic_ids(oa, ntxt)
cosd = lambda a, b: float((1 - F.cosine_similarity(a.double(), b.double(), dim=-1)).max())
for outliers in (False, True):
    sd = clip_ref.random_state_dict(oa, seed=0, outliers=outliers)
    ri, rt = clip_ref.encode_image(sd, oa, px), clip_ref.encode_text(sd, oa, ids)
    for fused in (False, True):
        ei, et = encode_image(sd, oa, px, fused), encode_text(sd, oa, ids, fused)
        print(f"{name} outliers={outliers} {'FUSED LN (raw x rounded to bf16)' if fused else 'LN in fp32, then bf16 (the build)  '}: "
              f"image 1-cos max {cosd(ei, ri):.2e} | text {cosd(et, rt):.2e}", flush=True)
