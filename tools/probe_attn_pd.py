#!/usr/bin/env python3
"""Which (image, head) items of the persistent attention kernel (attn_v = 3) differ from the one-item kernel (attn_v = 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import debug, engine  # noqa: E402
dev = torch.device("cuda:0")
t, batch, width = 257, 100, 512
g = torch.Generator().manual_seed(11)
x = (torch.randn(batch * t, 3 * width, generator=g) * 0.7).to(torch.bfloat16).to(dev)
outs = {}
for v in (2, 3):
    with debug.override(attn_v=v):
        runs = [engine.op_attention(x, batch, t, width, False).float().view(batch, t, width // 64, 64) for _ in range(6)]
    torch.cuda.synchronize()
    for i, r in enumerate(runs[1:]):
        d = (r != runs[0])
        print("attn_v", v, "run", i + 1, "vs run 0: elements differing", int(d.sum()), [tuple(ix) for ix in d.nonzero()[:5].tolist()])
    outs[v] = runs[0]
bad = (outs[2] != outs[3]).any(-1)            # [batch, t, heads]
per_item = bad.sum(1)                          # [batch, heads]
print("items differing:", int((per_item > 0).sum()), "of", batch * (width // 64))
for b in range(batch):
    if per_item[b].sum() > 0:
        rows = bad[b].any(-1).nonzero().flatten().tolist()
        print("image", b, "xcd", b % 8, "i", b // 8, "heads", per_item[b].tolist(), "queries", rows[:6], "...", rows[-3:], "maxdiff", float((outs[2][b] - outs[3][b]).abs().max()))
