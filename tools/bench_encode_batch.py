#!/usr/bin/env python3
"""Vision-tower throughput by images per encoder call, inputs resident in HBM: the tile-count cliff behind evaluators.ENCODE_ITEMS
(64 images = 65 row tiles -> 260 tiles of the N = 1024 GEMMs on 256 CUs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
from oracle import clip_ref
dev = torch.device("cuda:0")
eng = engine.ClipEngine(ARCHS["ViT-L/14"], dev)
eng.load_state_dict(clip_ref.random_state_dict(clip_ref.ARCHS["ViT-L/14"], seed=0))
px = torch.randn(255, 3, 224, 224, device=dev)
for b in (32, 63, 64, 127, 128, 255):
    reps = max(2, 1020 // b)
    for _ in range(2):
        eng.encode_image(px[:b], normalize=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.encode_image(px[:b], normalize=True)
    torch.cuda.synchronize()
    print("images per call %3d: %.0f images/s" % (b, b * reps / (time.perf_counter() - t0)), flush=True)
