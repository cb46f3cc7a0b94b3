#!/usr/bin/env python3
"""For rocprofv3 --pmc: the same two vision shapes through gemm256u and through torch's F.linear (hipBLASLt), a few launches
each, so that SQ counters of both kernels can be read side by side (yardstick only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
g = torch.Generator(device=dev).manual_seed(0)
for name, m, n, k in [("v.qkv", B * 257, 3072, 1024), ("v.fc2", B * 257, 1024, 4096)]:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    b16 = bias.to(torch.bfloat16)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    for _ in range(4):
        engine.op_gemm(a, w, bias, m, 0, c=c)
    for _ in range(4):
        torch.nn.functional.linear(a[:m], w, b16)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
