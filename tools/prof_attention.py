#!/usr/bin/env python3
"""The vision attention launch (B = 255, T = 257, 16 heads) a few times, for rocprofv3 --pmc / --kernel-trace.
argv[1]: a number whose units digit is attn_v (0 = the 16-query-tile kernel, 1 = 32-query tiles on the 32x32x16 MFMA, 2 = eight
waves / keys in two halves, 3 = the persistent LDS-DMA kernel), tens digit attn_xcd (1 = the images dealt to the XCDs, the library's
default; 0 = grid order) and hundreds attn_waves -- "10" is the product kernel.  argv[2]: a batch size (default 255), or `stamps`
(wave-cycles per phase of the 16-query kernel from its instrumented build), or `debug` (the one-hot exact test of
tests/test_ops_gpu.py with the list of (batch, head, query) rows that differ)."""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "stamps" else 255
if len(sys.argv) > 2 and sys.argv[2] == "stamps":
    # the 16-query kernel with s_memtime stamps between its phases (attn_waves = 2 selects the instrumented build): wave-cycles per phase
    import ctypes as C
    from knowledge_enhanced_multimodal_retrieval_amd import _lib
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = (torch.randn(B * 257, 3072, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.zeros(B * 257 * 1024 + B * 16 * 4 * 8 * 4, dtype=torch.bfloat16, device=dev)
    debug.set("attn_v", 0); debug.set("attn_xcd", 1); debug.set("attn_waves", 2)
    L = _lib.lib()
    n = 20
    for _ in range(n):
        _lib.check(L.kemr_op_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), B, 257, 1024, 0, C.c_void_p(engine._stream_ptr(dev))), "op_attention")
    torch.cuda.synchronize()
    st = out[B * 257 * 1024:].view(torch.int64).view(-1, 8).sum(0).cpu().tolist()
    waves = st[6]
    names = ["staging", "S^T + K reads", "mask + max", "exp + sum", "PV + V reads", "normalise + store"]
    tot = sum(st[:6])
    print("waves", waves, "memtime ticks per wave", tot / waves)
    for nme, v in zip(names, st[:6]):
        print("  %-20s %8.0f ticks per wave  %5.1f %%" % (nme, v / waves, 100.0 * v / tot))
    sys.exit(0)
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
