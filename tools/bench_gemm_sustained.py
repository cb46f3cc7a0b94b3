#!/usr/bin/env python3
"""Sustained-clock A/B: each candidate runs ~1.5 s back to back before and while it is timed (MI355X lowers its clock
under sustained MFMA load, so 10-launch bursts after an idle gap flatter every kernel; rule 24 / DVFS give-back)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)
cands = [4, 7, 7 | (4 << 8), 6, "torch"]
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    bias16 = bias.to(torch.bfloat16)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    out = {}
    for v in cands:
        if v == "torch":
            fn = lambda: torch.nn.functional.linear(a[:m], w, bias16)
        else:
            engine.set_gemm_variant(v)
            fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
        n_warm = int(1.0 / 400e-6)
        for _ in range(n_warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(1000):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 1000 * 1e3
        out[v] = "%.1f us %.0f TF" % (t, 2.0 * m * n * k / t / 1e6)
    print(name, {(v if v == "torch" else "v%d_dbg%d" % (v & 255, v >> 8)): s for v, s in out.items()}, flush=True)
engine.set_gemm_variant(0)
