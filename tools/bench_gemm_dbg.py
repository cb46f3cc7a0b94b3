#!/usr/bin/env python3
"""Timing experiment: GEMM variants (low byte) with debug flags (second byte: 1 = no C stores, gemm256u: 4 = plain instead of
non-temporal stores, 32 = half the stores) on the encoder shapes, 10-launch bursts, next to torch's F.linear (hipBLASLt) as
an outside yardstick.  For sustained-clock numbers use bench_gemm_sustained.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1),
          ("v.fc2", B * 257, 1024, 4096, 0), ("t.qkv", 2 * B * 77, 2304, 768, 0), ("t.out", 2 * B * 77, 768, 768, 0),
          ("t.fc1", 2 * B * 77, 3072, 768, 1), ("t.fc2", 2 * B * 77, 768, 3072, 0), ("sq4096", 4096, 4096, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
    out = {}
    bias16 = bias.to(torch.bfloat16)
    for rnd in range(4):
        for v in (4, 7, 7 | (4 << 8), 7 | (1 << 8), 6, "torch"):
            if v == "torch":
                fn = lambda: torch.nn.functional.linear(a[:m], w, bias16)
            else:
                engine.set_gemm_variant(v)
                fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(v, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * m * n * k
    print(name, {(v if v == "torch" else "v%d_dbg%d" % (v & 255, v >> 8)): "%.1f us %.0f TF" % (sorted(t)[len(t) // 2], fl / sorted(t)[len(t) // 2] / 1e6)
                 for v, t in out.items()}, flush=True)
engine.set_gemm_variant(0)
