#!/usr/bin/env python3
"""One-query text tower + search, a few times (for rocprofv3 --kernel-trace --stats): which kernels make up the B = 1 latency."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine, _lib
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
from oracle import clip_ref, metrics_ref
dev = torch.device("cuda:0")
eng = engine.ClipEngine(ARCHS["ViT-L/14"], dev)
eng.load_state_dict(clip_ref.random_state_dict(clip_ref.ARCHS["ViT-L/14"], seed=0))
img, _, tgt = metrics_ref.planted_embeddings(43000, 768, 0)
panel = engine.build_panel([torch.from_numpy(img).to(dev), torch.from_numpy(tgt).to(dev)], _lib.SIDE_GALLERY, 1)
ids = clip_ref.synthetic_ids(clip_ref.ARCHS["ViT-L/14"], 1).to(dev)
for _ in range(20):
    q = eng.encode_text(ids, normalize=True)
    qp = engine.build_panel([q, q], _lib.SIDE_QUERY, 1, part_scale=[0.5, 0.5])
    s, i = engine.sim_topk(qp, panel, 10)
torch.cuda.synchronize()
print("done")
