#!/usr/bin/env python3
"""A/B the GEMM tile variants on the encoder shapes, interleaved rounds in ONE process (cdna guide rule 24)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import _lib, engine

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1),
          ("v.fc2", B * 257, 1024, 4096, 0), ("t.qkv", 2 * B * 77, 2304, 768, 0), ("t.out", 2 * B * 77, 768, 768, 0),
          ("t.fc1", 2 * B * 77, 3072, 768, 1), ("t.fc2", 2 * B * 77, 768, 3072, 0), ("sq4096", 4096, 4096, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
    times = {2: [], 4: [], 5: []}
    for rnd in range(6):
        for v in (2, 4, 5):
            engine.set_gemm_variant(v)
            for _ in range(2):
                engine.op_gemm(a, w, bias, m, epi, c=c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                engine.op_gemm(a, w, bias, m, epi, c=c)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * m * n * k
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    res[name] = {f"v{v}_us": round(med[v] * 1e3, 1) for v in med} | {f"v{v}_tflops": round(fl / med[v] / 1e9, 1) for v in med}
    print(name, m, n, k, res[name], flush=True)
engine.set_gemm_variant(0)
print(json.dumps(res))
