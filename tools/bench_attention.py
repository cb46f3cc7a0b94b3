"""Attention kernel alone on the bench shapes (B = 255): vision T = 257 x 16 heads, text T = 77 x 12 heads (causal), checked
against a torch fp32 softmax(QK^T)V of the same bf16 inputs; `waves` = waves per workgroup for the 257-token kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def ref(qkv, b, t, w, causal):
    h = w // 64
    x = qkv.float().view(b, t, 3, h, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2)                    # the 1/8 scale is folded into W_q by the engine: none here
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(t, t, dtype=torch.bool, device=dev), 1), float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b * t, w)


for name, b, t, w, causal in (("vision", 255, 257, 1024, False), ("text", 255, 77, 768, True), ("text x2", 510, 77, 768, True)):
    qkv = (torch.randn(b * t, 3 * w, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    want = ref(qkv[: 8 * t], 8, t, w, causal)
    for waves in ((0, 6) if not causal else (0,)):
        engine.set_gemm_variant((waves + 1) << 24)
        got = engine.op_attention(qkv, b, t, w, causal)
        err = (got[: 8 * t].float() - want).abs().max().item()
        assert err < 2e-2, (name, waves, err)
        for _ in range(50):
            engine.op_attention(qkv, b, t, w, causal)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rnd in range(3):
            e0.record()
            for _ in range(100):
                engine.op_attention(qkv, b, t, w, causal)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        fl = 4.0 * t * t * 64 * b * (w // 64) * (0.5 if causal else 1.0)
        print(f"{name} T={t} waves/wg={'default' if waves == 0 else waves}: {sorted(ts)[1]:.1f} us  ({fl / sorted(ts)[1] / 1e6:.0f} TF/s useful)  max err {err:.1e}", flush=True)
engine.set_gemm_variant(1 << 24)
