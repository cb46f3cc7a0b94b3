#!/usr/bin/env python3
"""Timing experiment: 256x256 GEMM variants with / without epilogue stores (dbg flag) on the encoder shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1),
          ("v.fc2", B * 257, 1024, 4096, 0), ("sq4096", 4096, 4096, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
    out = {}
    for rnd in range(4):
        for v in (4, 4 | (8 << 8), 4 | (1 << 8)):
            engine.set_gemm_variant(v)
            for _ in range(2):
                engine.op_gemm(a, w, bias, m, epi, c=c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                engine.op_gemm(a, w, bias, m, epi, c=c)
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(v, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    print(name, {("v%d_dbg%d" % (v & 255, v >> 8)): round(sorted(t)[len(t) // 2], 1) for v, t in out.items()}, flush=True)
engine.set_gemm_variant(0)
