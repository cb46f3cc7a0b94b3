#!/usr/bin/env python3
"""fp8 (block-scaled MFMA, K = 128) against bf16 on the encoder shapes that run in fp8 (QKV, fc1); sustained, interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import _lib, engine
dev = torch.device("cuda:0")
B = 255
g = torch.Generator(device=dev).manual_seed(0)
for name, m, n, k, epi in [("v.qkv", B * 257, 3072, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("t.qkv", B * 77, 2304, 768, 0), ("t.fc1", B * 77, 3072, 768, 1),
                           ("sq4096", 4096, 4096, 4096, 0)]:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev)
    w = torch.randn(n, k, generator=g, device=dev) * k ** -0.5
    bias = torch.randn(n, generator=g, device=dev)
    a16, w16 = a.to(torch.bfloat16), w.to(torch.bfloat16)
    a8, w8 = a.to(torch.float8_e4m3fn), (w * 16).to(torch.float8_e4m3fn)
    ws = torch.full((n,), 1 / 16, device=dev)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    out = {}
    for rnd in range(3):
        for label in ("bf16", "fp8"):
            fn = (lambda: engine.op_gemm(a16, w16, bias, m, epi, c=c)) if label == "bf16" else (lambda: engine.op_gemm_fp8(a8, w8, ws, bias, m, epi))
            for _ in range(500):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(label, []).append(e0.elapsed_time(e1) / 300 * 1e3)
    fl = 2.0 * m * n * k
    print(name, {l: "%.1f us %.0f TF" % (sorted(t)[1], fl / sorted(t)[1] / 1e6) for l, t in out.items()}, flush=True)
