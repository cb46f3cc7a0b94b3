#!/usr/bin/env python3
"""Run the attention and LayerNorm kernels at the bench shapes a few times (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
B = 255
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * 257, 3072, generator=g, device=dev) * 0.5).to(torch.bfloat16)
qkv_t = (torch.randn(2 * B * 77, 2304, generator=g, device=dev) * 0.5).to(torch.bfloat16)
x = torch.randn(B * 257, 1024, generator=g, device=dev)
d = torch.randn(B * 257, 1024, generator=g, device=dev).to(torch.bfloat16)
gam = torch.ones(1024, device=dev); bet = torch.zeros(1024, device=dev)
for _ in range(4):
    engine.op_attention(qkv, B, 257, 1024, False)
    engine.op_attention(qkv_t, 2 * B, 77, 768, True)
    engine.op_layernorm_resid(x, d, gam, bet)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10): engine.op_attention(qkv, B, 257, 1024, False)
t1.record(); torch.cuda.synchronize()
print("attention T=257 B=255: %.1f us" % (t0.elapsed_time(t1) * 100))
