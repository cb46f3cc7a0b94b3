#!/usr/bin/env python3
"""Round 4: the encoders on heavy-tailed weights (oracle.clip_ref.add_outliers: massive residual channels,
LayerNorm gains x 30-100, a class-token-like row, a sharp head per block) against the fp32 oracle, per precision.

    python tools/outlier_stress.py [ViT-B/32|ViT-L/14|tiny-long] [n_images] [n_texts]

Prints 1 - cos (max / mean) of every precision for plain and outlier weights, and the largest |LayerNorm output| the oracle sees
(what the fp8 A operand must hold: e4m3 saturates at +-448)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                                        # noqa: E402
import torch.nn.functional as F                                                     # noqa: E402
from knowledge_enhanced_multimodal_retrieval_amd import engine                      # noqa: E402
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS               # noqa: E402
from oracle import clip_ref                                                         # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ntxt = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
arch, oa = ARCHS[name], clip_ref.ARCHS[name]
g = torch.Generator().manual_seed(1234)
px = torch.randn(nimg, 3, arch.image_size, arch.image_size, generator=g)
ids = clip_ref.synthetic_ids(oa, ntxt)


def ln_out_max(sd):
    """largest |LN output| over the first vision block's ln_1 / every block's gains x 5 sigma (cheap bound)"""
    worst = 0.0
    for k, v in sd.items():
        if k.endswith("ln_1.weight") or k.endswith("ln_2.weight"):
            worst = max(worst, float(v.abs().max()))
    return worst


for outliers in (False, True):
    sd = clip_ref.random_state_dict(oa, seed=0, outliers=outliers)
    ref_i, ref_t = clip_ref.encode_image(sd, oa, px), clip_ref.encode_text(sd, oa, ids)
    print(f"{name} outliers={outliers}: largest LayerNorm gain {ln_out_max(sd):.1f}; oracle |image| {float(ref_i.norm(dim=1).mean()):.2f} |text| {float(ref_t.norm(dim=1).mean()):.2f}")
    for prec in ("bf16", "bf16-x24", "bf16-res16", "fp8", "fp8-x24", "fp8-res16", "fp8-mlp"):
        try:
            eng = engine.ClipEngine(arch, dev, precision=prec)
            eng.load_state_dict(sd)
            gi = eng.encode_image(px.to(dev)).cpu()
            gt = eng.encode_text(ids.to(dev)).cpu()
        except Exception as e:                                                      # noqa: BLE001
            print(f"  {prec:11s} FAILED: {e}")
            continue
        ci = 1 - F.cosine_similarity(gi.double(), ref_i.double(), dim=-1)
        ct = 1 - F.cosine_similarity(gt.double(), ref_t.double(), dim=-1)
        fin = bool(torch.isfinite(gi).all() and torch.isfinite(gt).all())
        print(f"  {prec:11s} image 1-cos max {float(ci.max()):.2e} mean {float(ci.mean()):.2e} | text max {float(ct.max()):.2e} mean {float(ct.mean()):.2e} | finite {fin}")
