#!/usr/bin/env python3
"""Run each encoder GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace). argv: variant [batch]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 2), ("v.fc1", B * 257, 4096, 1024, 1),
          ("v.fc2", B * 257, 1024, 4096, 2), ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0)]
g = torch.Generator(device=dev).manual_seed(0)
engine.set_gemm_variant(variant)
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
    for _ in range(3):
        engine.op_gemm(a, w, bias, m, epi, c=c)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
