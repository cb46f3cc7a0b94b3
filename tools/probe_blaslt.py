#!/usr/bin/env python3
"""Yardstick probe: which hipBLASLt kernels torch picks for the encoder GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
dev = torch.device("cuda:0")
B = 255
for name, m, n, k in [("v.qkv", B * 257, 3072, 1024), ("v.out", B * 257, 1024, 1024), ("v.fc1", B * 257, 4096, 1024),
                      ("v.fc2", B * 257, 1024, 4096)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = torch.randn(n, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, device=dev).to(torch.bfloat16)
    for _ in range(3):
        c = torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
