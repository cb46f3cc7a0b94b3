#!/usr/bin/env python3
"""What the C stores cost the persistent GEMM (A/B library: the DBG instantiation, gemm_flags bit 0 drops the epilogue's global stores, bit 2 makes
them plain instead of non-temporal; bit 7 = whole-kernel clock stamps only, the calibration arm).  Interleaved, sustained, B = 255 vision shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)


def variant(order, dbg=0, conc=0):
    return 7 | (dbg << 8) | ((order + 1) << 16) | ((conc + 1) << 20)


for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    cands = [("product", variant(3, conc=2)), ("dbg build, stores as in the product", variant(3, dbg=128, conc=2)), ("no stores", variant(3, dbg=129, conc=2)), ("plain stores", variant(3, dbg=132, conc=2))]
    out = {}
    for rnd in range(3):
        for label, v in cands:
            engine.set_gemm_variant(v)
            fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
            for _ in range(300):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(label, []).append(e0.elapsed_time(e1) / 300 * 1e3)
    print(name, {l: "%.1f us (%s)" % (sorted(t)[len(t) // 2], " ".join("%.0f" % x for x in t)) for l, t in out.items()}, flush=True)
engine.set_gemm_variant(7 | (4 << 16) | (3 << 20))
engine.set_gemm_variant(0)
