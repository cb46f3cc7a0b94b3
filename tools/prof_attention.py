#!/usr/bin/env python3
"""The vision attention launch (B = 255, T = 257, 16 heads) a few times, for rocprofv3 --pmc / --kernel-trace.
argv: attn_v (0 = the 16-query-tile kernel, 1 = 32-query tiles on the 32x32x16 MFMA, 2 = eight waves, keys in two halves) [debug]: with `debug`, the one-hot exact test
of tests/test_ops_gpu.py and the list of (batch, head, query) rows that differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import debug, engine  # noqa: E402

dev = torch.device("cuda:0")
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
debug.set("attn_v", v % 10)
debug.set("attn_xcd", (v // 10) % 10)
debug.set("attn_waves", v // 100)
_=(0)                # 10, 12: the same kernels with the images dealt to the XCDs
if len(sys.argv) > 2 and sys.argv[2] == "debug":
    t, batch, width = 257, 2, 256
    g = torch.Generator().manual_seed(3)
    heads = width // 64
    perm = torch.randperm(t, generator=g)
    x = torch.zeros(batch * t, 3 * width)
    vals = torch.randint(-64, 65, (batch * t, width), generator=g).float()
    x[:, 2 * width:] = vals
    code = torch.arange(t)
    d0, d1 = code % 17, code // 17
    kk = torch.zeros(t, 64); kk[torch.arange(t), d0] = 1.0; kk[torch.arange(t), 17 + d1] = 1.0
    qq = torch.zeros(t, 64); qq[torch.arange(t), d0[perm]] = 40.0; qq[torch.arange(t), 17 + d1[perm]] = 40.0
    for hd in range(heads):
        for b in range(batch):
            x[b * t:(b + 1) * t, hd * 64:(hd + 1) * 64] = qq
            x[b * t:(b + 1) * t, width + hd * 64:width + (hd + 1) * 64] = kk
    got = engine.op_attention(x.to(torch.bfloat16).to(dev), batch, t, width, False).float().cpu()
    want = torch.cat([vals[b * t:(b + 1) * t][perm] for b in range(batch)])
    bad = (got != want).view(batch, t, heads, 64).any(-1)
    rows = [(b, h, q, int(perm[q])) for b in range(batch) for h in range(heads) for q in range(t) if bad[b, q, h]]
    print("attn_v", v, "wrong (batch, head, query, its target key):", rows[:40], "of", len(rows))
    for b, h, q, tk in rows[:3]:
        gq = got.view(batch, t, heads, 64)[b, q, h]
        # which key's V row did it return, if any?
        vv = vals.view(batch, t, heads, 64)[b, :, h]
        match = [(j, float((vv[j] - gq).abs().max())) for j in range(t) if float((vv[j] - gq).abs().max()) < 0.51]
        print("  row", (b, h, q), "target", tk, "closest V rows:", match[:4], "got[:6]", gq[:6].tolist(), "want[:6]", want.view(batch, t, heads, 64)[b, q, h][:6].tolist())
    sys.exit(0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 255
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * 257, 3072, generator=g, device=dev) * 0.5).to(torch.bfloat16)
for _ in range(300):
    engine.op_attention(qkv, B, 257, 1024, False)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(300):
    engine.op_attention(qkv, B, 257, 1024, False)
t1.record(); torch.cuda.synchronize()
print("attention T=257 B=%d attn_v=%d: %.1f us" % (B, v, t0.elapsed_time(t1) / 300 * 1e3))
