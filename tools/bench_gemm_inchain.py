#!/usr/bin/env python3
"""Which in-chain condition costs the GEMM its isolated speed?  Sustained loops of the fc1 / qkv shape with
(a) fixed operands, (b) 24 rotating weight matrices, (c) A rewritten by an elementwise kernel before every GEMM,
(d) a 400 MB streaming write (LayerNorm-like) before every GEMM, (e) all of it.  GEMM time from hipEvents around the GEMM only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
g = torch.Generator(device=dev).manual_seed(0)
for name, m, n, k, epi in [("v.qkv", B * 257, 3072, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1)]:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    a_src = a.clone()
    ws = [(torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16) for _ in range(24)]
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    junk = torch.empty(100_000_000, dtype=torch.float32, device=dev)     # 400 MB
    for variant in (4, 7):
        engine.set_gemm_variant(variant)
        res = {}
        for mode in ("fixed", "rot_w", "fresh_a", "dirty400", "all"):
            evs = []
            def one(i, timed):
                w = ws[i % 24] if mode in ("rot_w", "all") else ws[0]
                if mode in ("fresh_a", "all"):
                    torch.add(a_src, 0, out=a)
                if mode in ("dirty400", "all"):
                    junk.fill_(1.0)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                engine.op_gemm(a, w, bias, m, epi, c=c)
                if timed:
                    e1.record()
                    evs.append((e0, e1))
            for i in range(400):
                one(i, False)
            for i in range(200):
                one(i, True)
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
            res[mode] = round(ts[len(ts) // 2], 1)
        print(name, "variant", variant, res, flush=True)
engine.set_gemm_variant(0)
