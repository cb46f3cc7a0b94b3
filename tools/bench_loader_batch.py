#!/usr/bin/env python3
"""encode_dataset throughput for different DataLoader batch sizes (the reference scripts pass 64): the encoders are fed
255 items per call regardless (evaluators.ENCODE_ITEMS); with ENCODE_ITEMS = 0 every loader batch is one encoder call."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import clip
from knowledge_enhanced_multimodal_retrieval_amd import datasets, evaluators
warnings.simplefilter("ignore")
model, _ = clip.load("ViT-L/14", device="cuda")
n = 1020
ds = datasets.SyntheticRetrievalDataset(n, 224, seed=1)
items = [ds[i] for i in range(n)]                      # materialise once: time the encoders, not the synthetic generator
class Mem(torch.utils.data.Dataset):
    def __len__(self): return n
    def __getitem__(self, i): return items[i]
for group in (255, 0):
    evaluators.ENCODE_ITEMS = group if group else 1
    for bs in (64, 63, 255):
        evaluators.encode_dataset(model, Mem(), bs, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evaluators.encode_dataset(model, Mem(), bs, 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("ENCODE_ITEMS", group, "loader batch", bs, "items/s %.0f" % (n / dt), flush=True)
