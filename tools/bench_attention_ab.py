#!/usr/bin/env python3
"""Interleaved A/B of attention kernels at the vision shape (B = 255 images x 16 heads, T = 257) in ONE process on ONE device
(cdna guide rule 24): rounds of 100 launches per variant, median and min per variant.  Needs a `build.py --ab-variants` library
for every variant but 0.

    python tools/bench_attention_ab.py 0,5 [rounds] [batch]

Between the timed blocks a 512 MiB buffer is rewritten so that every launch reads its q | k | v rows from beyond the caches, as the
launch inside the encoder does (the QKV GEMM in front of it writes 402 MB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import debug, engine  # noqa: E402

variants = [(int(x.split(":")[0]), int(x.split(":")[1]) if ":" in x else 0) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]     # attn_v[:attn_waves]


def select(vw):
    debug.set("attn_v", vw[0])
    debug.set("attn_waves", vw[1])


rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
B = int(sys.argv[3]) if len(sys.argv) > 3 else 255
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
# four q | k | v tensors used in turn: 1.6 GB, far beyond the 256 MiB Infinity Cache -> every launch streams from HBM
qkvs = [(torch.randn(B * 257, 3072, generator=g, device=dev) * 0.5).to(torch.bfloat16) for _ in range(4)]
ref = None
for v in variants:
    select(v)
    out = engine.op_attention(qkvs[0], B, 257, 1024, False)
    if ref is None:
        ref = out
    else:
        d = (out.float() - ref.float()).abs()
        print("attn_v", v, "vs", variants[0], ": max |diff|", float(d.max()), "elements differing", int((d > 0).sum()), "of", d.numel())
times = {v: [] for v in variants}
n = 100
for r in range(rounds + 1):
    for v in variants:
        select(v)
        for i in range(20):
            engine.op_attention(qkvs[i % 4], B, 257, 1024, False)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for i in range(n):
            engine.op_attention(qkvs[i % 4], B, 257, 1024, False)
        t1.record(); torch.cuda.synchronize()
        if r:
            times[v].append(t0.elapsed_time(t1) / n * 1e3)
select((0, 0))
for v in variants:
    ts = sorted(times[v])
    print("attn_v %d:%d: median %.1f us, min %.1f, max %.1f over %d rounds of %d launches (B = %d)" % (v[0], v[1], ts[len(ts) // 2], ts[0], ts[-1], len(ts), n, B))
