#!/usr/bin/env python3
"""Tile order of the persistent GEMM by shape (debug switch gemm_order: 0 = N fastest, n = column groups of 2^(n-1) tiles kept on one XCD),
product library, interleaved sustained rounds on the towers' shapes.    python tools/bench_gemm_order.py [orders, e.g. 2,3,4,5] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0),
          ("t851.qkv", 851 * 77, 2304, 768, 0), ("t851.out", 851 * 77, 768, 768, 0), ("t851.fc1", 851 * 77, 3072, 768, 1), ("t851.fc2", 851 * 77, 768, 3072, 0)]
orders = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,5").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device=dev).manual_seed(0)


def variant(order, conc=2):
    return 7 | ((order + 1) << 16) | ((conc + 1) << 20)


for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    out = {}
    for rnd in range(rounds):
        for o in orders:
            engine.set_gemm_variant(variant(o))
            fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
            for _ in range(300):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(o, []).append(e0.elapsed_time(e1) / 300 * 1e3)
    print(name, f"N/256 = {n // 256}", {o: "%.1f us" % sorted(t)[len(t) // 2] for o, t in out.items()}, flush=True)
engine.set_gemm_variant(7 | (4 << 16) | (3 << 20))
engine.set_gemm_variant(0)
