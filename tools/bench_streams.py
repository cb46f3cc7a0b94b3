#!/usr/bin/env python3
"""Host-level experiment: does running independent encoder chains on two HIP streams hide the HBM-bound kernels
(LayerNorm, attention) of one chain behind the MFMA-bound GEMMs of the other?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
from oracle import clip_ref
dev = torch.device("cuda:0")
arch = ARCHS["ViT-L/14"]
sd = clip_ref.random_state_dict(clip_ref.ARCHS["ViT-L/14"], seed=0)
e1, e2 = engine.ClipEngine(arch, dev), engine.ClipEngine(arch, dev)
e1.load_state_dict(sd); e2.load_state_dict(sd)
B = 255
g = torch.Generator(device=dev).manual_seed(1)
px = torch.randn(B, 3, 224, 224, generator=g, device=dev)
ids = clip_ref.synthetic_ids(clip_ref.ARCHS["ViT-L/14"], 2 * B).to(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def serial():
    e1.encode_image(px, normalize=True); e1.encode_text(ids, normalize=True)
def img_txt():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev); e1.encode_image(px, normalize=True)
    with torch.cuda.stream(s2):
        s2.wait_event(ev); e2.encode_text(ids, normalize=True)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
h = B // 2
def halves():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev); e1.encode_image(px[:h], normalize=True); e1.encode_text(ids[:2 * h], normalize=True)
    with torch.cuda.stream(s2):
        s2.wait_event(ev); e2.encode_text(ids[2 * h:], normalize=True); e2.encode_image(px[h:], normalize=True)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
res = {}
for rnd in range(3):
    for name, fn in (("serial", serial), ("img||txt", img_txt), ("halves", halves)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(B * 8 / (time.perf_counter() - t0))
print({k: [round(x) for x in v] for k, v in res.items()})
