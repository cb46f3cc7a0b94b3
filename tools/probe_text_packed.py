#!/usr/bin/env python3
"""Where do the packed and the full-context text encoders differ?  (1) the same texts in calls of different sizes, both forms;
(2) one GEMM on the same rows at another row offset / with another M."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import engine  # noqa: E402
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS  # noqa: E402
from oracle import clip_ref  # noqa: E402
dev = torch.device("cuda:0")
name = "ViT-B/32"
arch = ARCHS[name]
sd = clip_ref.random_state_dict(clip_ref.ARCHS[name], seed=0)
eng = engine.ClipEngine(arch, dev)
eng.load_state_dict(sd)
ids = clip_ref.synthetic_ids(clip_ref.ARCHS[name], 300)
def diff(a, b):
    cos = torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)
    return "equal" if torch.equal(a, b) else "max |d| %.2e, rows differing %d of %d, 1 - cos max %.2e" % (float((a - b).abs().max()), int((a != b).any(-1).sum()), a.shape[0], float((1 - cos).max()))
for pack in (False, True):
    eng.pack_text = pack
    big = eng.encode_text(ids, normalize=True)
    half = eng.encode_text(ids[:150], normalize=True)
    print("pack", pack, ": texts 0..149 in a 300-text call vs a 150-text call:", diff(big[:150], half))
eng.pack_text = False
full = eng.encode_text(ids, normalize=True)
eng.pack_text = True
packed = eng.encode_text(ids, normalize=True)
lens = engine.text_lengths(ids)
d = (full != packed).any(-1).cpu()
print("packed vs full:", diff(full, packed), "| lengths of the texts that differ:", sorted(set(lens[d].tolist()))[:20], "of the equal ones:", sorted(set(lens[~d].tolist()))[:20])
ref = clip_ref.l2_normalize(clip_ref.encode_text(sd, clip_ref.ARCHS[name], ids[:64])).to(dev)
print("against the fp32 oracle (64 texts): full", diff(full[:64], ref), "| packed", diff(packed[:64], ref))
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(24 * 256, 512, generator=g, device=dev).to(torch.bfloat16)
w = (torch.randn(1536, 512, generator=g, device=dev) * 512 ** -0.5).to(torch.bfloat16)
bias = torch.randn(1536, generator=g, device=dev)
c1 = engine.op_gemm(a, w, bias, a.shape[0], 0)
a2 = torch.zeros_like(a); a2[:a.shape[0] - 1000] = a[1000:]
c2 = engine.op_gemm(a2, w, bias, a.shape[0] - 1000, 0)
print("GEMM rows 1000.. of a 6144-row call vs the same rows as a 5144-row call:", diff(c1[1000:a.shape[0]].float(), c2[:a.shape[0] - 1000].float()))
c3 = engine.op_gemm(a[:3072].contiguous(), w, bias, 3000, 0)
print("GEMM rows 0..2999 of a 6144-row call vs a 3000-row call:", diff(c1[:3000].float(), c3[:3000].float()))
