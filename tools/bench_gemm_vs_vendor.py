#!/usr/bin/env python3
"""The product's persistent GEMM against hipBLASLt (through torch.nn.functional.linear: the outside yardstick, not used by the build) on the
towers' shapes, same device, one process, interleaved rounds of sustained launches (cdna guide rule 24).  Works with the product library.

    python tools/bench_gemm_vs_vendor.py [rounds]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1 (+QuickGELU in ours only)", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0),
          ("t851.qkv", 851 * 77, 2304, 768, 0), ("t851.fc2", 851 * 77, 768, 3072, 0), ("sq4096", 4096, 4096, 4096, 0)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device=dev).manual_seed(0)
torch.backends.cuda.matmul.allow_tf32 = False
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    bias16 = bias.to(torch.bfloat16)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    got = engine.op_gemm(a, w, bias, m, 0, c=c)[:m].float()
    want = torch.nn.functional.linear(a[:m], w, bias16).float()
    err = float((got - want).abs().max())
    cands = {"ours": lambda: engine.op_gemm(a, w, bias, m, epi, c=c), "hipblaslt": lambda: torch.nn.functional.linear(a[:m], w, bias16)}
    out = {kk: [] for kk in cands}
    for r in range(rounds):
        for kk, fn in cands.items():
            for _ in range(300):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[kk].append(e0.elapsed_time(e1) / 300 * 1e3)
    fl = 2.0 * m * n * k
    med = {kk: sorted(t)[len(t) // 2] for kk, t in out.items()}
    print(f"{name}: M={m} N={n} K={k}  ours {med['ours']:.1f} us ({fl / med['ours'] / 1e6:.0f} TFLOP/s)  hipBLASLt {med['hipblaslt']:.1f} us ({fl / med['hipblaslt'] / 1e6:.0f} TFLOP/s)"
          f"  ratio {med['hipblaslt'] / med['ours']:.3f}  max |ours - vendor| {err:.3g}", flush=True)
