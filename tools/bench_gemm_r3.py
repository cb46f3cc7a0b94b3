#!/usr/bin/env python3
"""Round-3 GEMM experiments on the towers' shapes (B = 255 images, 851 texts): the long-interval K loop (debug switch gemm_kl = 1:
four 512-cycle barrier intervals per K-tile) against round 2's (gemm_kl = 0: eight of 256), for every epilogue of the persistent
kernel -- store bf16, QuickGELU, bf16 residual add, fp32 residual add in the accumulator domain -- first checked against an fp32
torch statement of the op, then timed sustained and interleaved (rule 24: one process, alternating rounds, >= 0.3 s of warm-up
per candidate); hipBLASLt through torch as the outside yardstick (not used by the build).

    python tools/bench_gemm_r3.py [check|ab|stamps|all]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import _lib, debug, engine  # noqa: E402

dev = torch.device("cuda:0")
B = 255
EPI_NAMES = {0: "store", 1: "qgelu", 2: "resid_f32", 4: "resadd_bf16"}
shapes = [("v.qkv", B * 257, 3072, 1024, (0,)), ("v.out", B * 257, 1024, 1024, (0, 4, 2)), ("v.fc1", B * 257, 4096, 1024, (1,)),
          ("v.fc2", B * 257, 1024, 4096, (0, 4, 2)), ("t851.qkv", 851 * 77, 2304, 768, (0,)), ("t851.fc2", 851 * 77, 768, 3072, (0, 2)),
          ("sq4096", 4096, 4096, 4096, (0,))]
what = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator(device=dev).manual_seed(0)
torch.backends.cuda.matmul.allow_tf32 = False


def ref(a, w, bias, m, epi, x0):
    y = a[:m].float() @ w.float().t() + bias
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    if epi in (2, 4):
        y = y + x0[:m].float()
    return y


def timed(fn, warm, n):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, m, n, k, epis in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    x32 = torch.randn(ma, n, generator=g, device=dev) * 3
    x16 = x32.to(torch.bfloat16)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    cbuf = {0: c, 1: c, 2: x32.clone(), 4: x16.clone()}
    if what in ("all", "check"):
        for epi in epis:
            want = ref(a, w, bias, m, epi, x32 if epi == 2 else x16)
            outs = []
            for kl in (1, 0):
                with debug.override(gemm_variant=7, gemm_kl=kl):
                    buf = {0: c, 1: c, 2: x32.clone(), 4: x16.clone()}[epi]
                    if epi < 2:
                        buf.zero_()
                    engine.op_gemm(a, w, bias, m, epi, c=buf)
                    got = buf[:m].float()
                err = (got - want).abs().max().item()
                tol = (3e-5 * k ** 0.5) if epi == 2 else 0.06 * max(1.0, want.abs().max().item() / 4)
                print(f"check {name} {EPI_NAMES[epi]} kl={kl}: max abs err {err:.2e} (|ref| max {want.abs().max().item():.2f})", flush=True)
                assert err < tol, (name, epi, kl, err, tol)
                outs.append(got)
            assert torch.equal(outs[0], outs[1]), (name, epi, "the two K loops sum in the same order: bit-identical results expected")
    if what in ("all", "ab"):
        cands = [(f"{EPI_NAMES[e]} kl={kl}", e, kl) for e in epis for kl in (1, 0)] + [("hipblaslt", -1, 1)]
        bias16 = bias.to(torch.bfloat16)
        out = {}
        for rnd in range(3):
            for label, e, kl in cands:
                if e < 0:
                    fn = lambda: torch.nn.functional.linear(a[:m], w, bias16)
                    out.setdefault(label, []).append(timed(fn, 500, 300))
                    continue
                with debug.override(gemm_variant=7, gemm_kl=kl):
                    fn = lambda: engine.op_gemm(a, w, bias, m, e, c=cbuf[e])
                    out.setdefault(label, []).append(timed(fn, 500, 300))
        fl = 2.0 * m * n * k
        print("ab", name, {l: "%.1f us %.0f TF (%s)" % (sorted(t)[1], fl / sorted(t)[1] / 1e6, " ".join("%.0f" % x for x in t)) for l, t in out.items()}, flush=True)
    if what in ("all", "stamps") and name.startswith("v."):
        for e, kl in [(epis[0], 1), (epis[0], 0)]:
            with debug.override(gemm_variant=7, gemm_flags=64, gemm_kl=kl):
                for _ in range(200):
                    engine.op_gemm(a, w, bias, m, e, c=cbuf[e])
                torch.cuda.synchronize()
                st = np.frombuffer(debug.gemm_stamps(256 * 16), dtype=np.uint32).reshape(256, 16).astype(np.float64)
            tiles, nt = st[:, 14], st[:, 15]
            per_ktile = st[:, :(4 if kl else 8)] / (tiles * nt)[:, None]
            print(f"stamps {name} {EPI_NAMES[e]} kl={kl}: cycles per K-tile interval (median over workgroups) "
                  + " ".join("%.0f" % x for x in np.median(per_ktile, 0)) + f" | sum {np.median(per_ktile.sum(1)):.0f}"
                  + f" | per tile: K-loop tail {np.median(st[:, 8] / tiles):.0f}, epilogue(H0) {np.median(st[:, 9] / tiles):.0f}", flush=True)
