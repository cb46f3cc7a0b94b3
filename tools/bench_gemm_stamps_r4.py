#!/usr/bin/env python3
"""Round 4 (A/B library): in-kernel stamps of the persistent GEMM by wave half and for the first K-tile of a tile alone -- where a tile's
C stores cost time.  gemm_flags: 64 = stamps, + 8 = wave 4 (the trailing half) instead of wave 0, + 16 = K-loop slots of the FIRST K-tile only,
+ 1 = stores dropped.  Slots 0..7: the K-tile's barrier intervals as the stamping wave sees them (L1 M1 L2 M2 L3 M3 L4 M4), 8: behind the K
loop up to the epilogue (CONC: the extra barrier), 9: the epilogue.  Cycles per tile (first-K-tile mode) or per K-tile, median over workgroups."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine, _lib
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0)]
g = torch.Generator(device=dev).manual_seed(0)


def variant(order, dbg=0, conc=0):
    return 7 | (dbg << 8) | ((order + 1) << 16) | ((conc + 1) << 20)


for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    for label, flags in (("wave 0, all K-tiles", 64), ("wave 4, all K-tiles", 64 | 8), ("wave 0, first K-tile", 64 | 16), ("wave 4, first K-tile", 64 | 16 | 8),
                         ("wave 0, first K-tile, no stores", 64 | 16 | 1), ("wave 4, first K-tile, no stores", 64 | 16 | 8 | 1)):
        engine.set_gemm_variant(variant(3, dbg=flags, conc=2))
        for _ in range(200):
            engine.op_gemm(a, w, bias, m, epi, c=c)
        torch.cuda.synchronize()
        buf = (C.c_uint * (256 * 16))()
        _lib.check(_lib.lib().kemr_debug_gemm_stamps(buf, 256 * 16), "stamps")
        st = np.frombuffer(buf, dtype=np.uint32).reshape(256, 16).astype(np.float64)
        tiles, nt = st[:, 14], st[:, 15]
        div = tiles if flags & 16 else tiles * nt
        per = st[:, :8] / div[:, None]
        print(f"{name} {label}: " + " ".join("%.0f" % x for x in np.median(per, 0)) + f" | sum {np.median(per.sum(1)):.0f} | per tile: tail {np.median(st[:, 8] / tiles):.0f}, epilogue {np.median(st[:, 9] / tiles):.0f}", flush=True)
engine.set_gemm_variant(7 | (4 << 16) | (3 << 20))
engine.set_gemm_variant(0)
