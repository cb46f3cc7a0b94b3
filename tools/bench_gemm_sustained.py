#!/usr/bin/env python3
"""Sustained-clock A/B: each candidate runs ~1.5 s back to back before and while it is timed (MI355X lowers its clock
under sustained MFMA load, so 10-launch bursts after an idle gap flatter every kernel; rule 24 / DVFS give-back)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0), ("sq4096", 4096, 4096, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)
cands = [("v7", 7), ("v9_regstaged", 9), ("torch", "torch")]
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    bias16 = bias.to(torch.bfloat16)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    out = {}
    for rnd in range(4):                      # interleaved rounds, every candidate warmed ~0.3 s before its timed 300 launches
        for label, v in cands:
            if v == "torch":
                fn = lambda: torch.nn.functional.linear(a[:m], w, bias16)
            else:
                engine.set_gemm_variant(v)
                fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
            for _ in range(700):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(label, []).append(e0.elapsed_time(e1) / 300 * 1e3)
    fl = 2.0 * m * n * k
    print(name, {l: "%.1f us %.0f TF (%s)" % (sorted(t)[len(t) // 2], fl / sorted(t)[len(t) // 2] / 1e6, " ".join("%.0f" % x for x in t)) for l, t in out.items()}, flush=True)
engine.set_gemm_variant(0)
