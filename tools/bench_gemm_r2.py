#!/usr/bin/env python3
"""Round-2 GEMM experiments on the vision shapes (B = 255): correctness of every tile order against torch, sustained
interleaved A/B of the tile orders (p.order = 0: N fastest; 1 + log2(column-group width) otherwise), and the in-kernel
stamp profile of the diagnostic instantiation (cycles per barrier interval of the K loop, epilogue)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine, _lib
dev = torch.device("cuda:0")
B = 255
shapes = [("v.qkv", B * 257, 3072, 1024, 0), ("v.out", B * 257, 1024, 1024, 0), ("v.fc1", B * 257, 4096, 1024, 1), ("v.fc2", B * 257, 1024, 4096, 0),
          ("t.qkv", B * 77, 2304, 768, 0), ("t.fc1", B * 77, 3072, 768, 1), ("t851.out", 851 * 77, 768, 768, 0), ("t851.fc2", 851 * 77, 768, 3072, 0)]
g = torch.Generator(device=dev).manual_seed(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
orders = [0, 3]          # 0: N fastest; 3: column groups of 4 tiles
R1 = "v2"                 # gemm256 (variant 2, non-persistent lockstep kernel): a fixed point of comparison in the product library


def variant(order, dbg=0, conc=0):
    return 7 | (dbg << 8) | ((order + 1) << 16) | ((conc + 1) << 20)


def ref(a, w, bias, m, epi):
    y = a[:m].float() @ w.float().t() + bias
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    return y


for name, m, n, k, epi in shapes:
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=dev)
    c = torch.zeros(ma, n, dtype=torch.bfloat16, device=dev)
    if what in ("all", "check"):
        want = ref(a, w, bias, m, epi)
        for o in orders:
            engine.set_gemm_variant(variant(o))
            c.zero_()
            engine.op_gemm(a, w, bias, m, epi, c=c)
            err = (c[:m].float() - want).abs().max().item()
            nz = (c[:m] == 0).all(dim=1).sum().item()
            print(f"check {name} order {o}: max abs err {err:.4f} (|ref| max {want.abs().max().item():.2f}), all-zero rows {nz}", flush=True)
            assert err < 0.06 * max(1.0, want.abs().max().item() / 4), (name, o, err)
    if what in ("all", "ab") and name.startswith("v."):
        out = {}
        for rnd in range(4):
            for o in orders + [R1]:
                engine.set_gemm_variant(2 if o == R1 else variant(o))
                fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
                for _ in range(600):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(300):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out.setdefault(o, []).append(e0.elapsed_time(e1) / 300 * 1e3)
        fl = 2.0 * m * n * k
        print("ab", name, {o: "%.1f us %.0f TF (%s)" % (sorted(t)[len(t) // 2], fl / sorted(t)[len(t) // 2] / 1e6, " ".join("%.0f" % x for x in t)) for o, t in out.items()}, flush=True)
    if what in ("res",) and epi == 0 and n <= 1024:
        # residual-add epilogue (reads the bf16 x tile it overwrites) against the store-only epilogue, sustained, interleaved
        x = (torch.randn(ma, n, generator=g, device=dev) * 3).to(torch.bfloat16)
        out = {}
        for rnd in range(3):
            for label, e, cc in (("store", 0, c), ("resadd", 4, x)):
                engine.set_gemm_variant(variant(3, conc=2))
                fn = lambda: engine.op_gemm(a, w, bias, m, e, c=cc)
                for _ in range(400):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(300):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out.setdefault(label, []).append(e0.elapsed_time(e1) / 300 * 1e3)
        print("res", name, {o: "%.1f us (%s)" % (sorted(t)[1], " ".join("%.0f" % v for v in t)) for o, t in out.items()}, flush=True)
    if what in ("conc",):
        # both halves' epilogues in the same barrier interval (conc = 1) against one after the other (0): parity, then
        # sustained interleaved timing, then the stamp profile of the stamped instantiation
        want = ref(a, w, bias, m, epi)
        for cc in (0, 1):
            engine.set_gemm_variant(variant(3, conc=cc))
            c.zero_()
            got = engine.op_gemm(a, w, bias, m, epi, c=c)[:m].float()
            err = (got - want).abs().max().item()
            assert err < 0.08 * max(1.0, want.abs().max().item() / 8), (name, cc, err)
            if cc == 0:
                base = got.clone()
            else:
                assert torch.equal(got, base), (name, "conc differs from serial epilogues")
        out = {}
        for rnd in range(3):
            for cc in (0, 1):
                engine.set_gemm_variant(variant(3, conc=cc))
                fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
                for _ in range(400):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(300):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out.setdefault(cc, []).append(e0.elapsed_time(e1) / 300 * 1e3)
        fl = 2.0 * m * n * k
        print("conc", name, {o: "%.1f us %.0f TF (%s)" % (sorted(t)[1], fl / sorted(t)[1] / 1e6, " ".join("%.0f" % x for x in t)) for o, t in out.items()}, flush=True)
        for cc in (0, 1):
            engine.set_gemm_variant(variant(3, dbg=64, conc=cc))
            for _ in range(200):
                engine.op_gemm(a, w, bias, m, epi, c=c)
            torch.cuda.synchronize()
            buf = (C.c_uint * (256 * 16))()
            _lib.check(_lib.lib().kemr_debug_gemm_stamps(buf, 256 * 16), "stamps")
            st = np.frombuffer(buf, dtype=np.uint32).reshape(256, 16).astype(np.float64)
            tiles, nt = st[:, 14], st[:, 15]
            per_ktile = st[:, :8] / (tiles * nt)[:, None]
            print(f"  stamps conc={cc}: " + " ".join("%.0f" % x for x in np.median(per_ktile, 0)) + f" | sum {np.median(per_ktile.sum(1)):.0f}"
                  + f" | per tile: tail {np.median(st[:, 8] / tiles):.0f}, epilogue(H0) {np.median(st[:, 9] / tiles):.0f}", flush=True)
        engine.set_gemm_variant(variant(3, conc=2))
    if what in ("exp",) and name.startswith("v."):
        # timing experiments of the DBG instantiation: 128 = nothing extra (calibrates the instantiation), 8 = no waits for the
        # staged pieces (garbage results), 16 = L2 prefetch PD K-tiles ahead
        cands = [("order 0", variant(0)), ("order 3", variant(3)), ("order 4", variant(4)), ("dbg", variant(3, dbg=128)), ("v2 (gemm256 lockstep)", 2)]
        out = {}
        for rnd in range(3):
            for label, v in cands:
                engine.set_gemm_variant(v)
                fn = lambda: engine.op_gemm(a, w, bias, m, epi, c=c)
                for _ in range(500):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(300):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out.setdefault(label, []).append(e0.elapsed_time(e1) / 300 * 1e3)
        print("exp", name, {l: "%.1f us (%s)" % (sorted(t)[len(t) // 2], " ".join("%.0f" % x for x in t)) for l, t in out.items()}, flush=True)
    if what in ("all", "clock") and name.startswith("v."):
        # clock the chip holds under the kernel: in-kernel shader cycles / 100 MHz ticks over the whole launch (wave 0 of every
        # workgroup), after ~0.2 s of back-to-back launches
        engine.set_gemm_variant(variant(3, dbg=128))
        for _ in range(500):
            engine.op_gemm(a, w, bias, m, epi, c=c)
        torch.cuda.synchronize()
        buf = (C.c_uint * (256 * 16))()
        _lib.check(_lib.lib().kemr_debug_gemm_stamps(buf, 256 * 16), "stamps")
        st = np.frombuffer(buf, dtype=np.uint32).reshape(256, 16).astype(np.float64)
        cyc, ticks = st[:, 12], st[:, 13]
        print(f"clock {name}: kernel cycles {np.median(cyc):.0f}, {np.median(ticks) / 100:.1f} us in-kernel, clock {np.median(cyc / ticks) * 0.1:.3f} GHz", flush=True)
    if what in ("all", "stamps") and name.startswith("v."):
        for o, fine in ((0, 0), (0, 32)):
            engine.set_gemm_variant(variant(o, dbg=64 | fine))
            for _ in range(200):
                engine.op_gemm(a, w, bias, m, epi, c=c)
            torch.cuda.synchronize()
            buf = (C.c_uint * (256 * 16))()
            _lib.check(_lib.lib().kemr_debug_gemm_stamps(buf, 256 * 16), "stamps")
            st = np.frombuffer(buf, dtype=np.uint32).reshape(256, 16).astype(np.float64)
            tiles, nt = st[:, 14], st[:, 15]
            per_ktile = st[:, :8] / (tiles * nt)[:, None]
            line = (f"stamps {name} order {o} fine {fine}: cycles per K-tile interval (median over workgroups) "
                    + " ".join("%.0f" % x for x in np.median(per_ktile, 0)) + f" | sum {np.median(per_ktile.sum(1)):.0f}"
                    + f" | per tile: K-loop tail {np.median(st[:, 8] / tiles):.0f}, epilogue(H0) {np.median(st[:, 9] / tiles):.0f}")
            if fine & 32:
                pre = st[:, 10:14] / (tiles * nt)[:, None]
                line += " | own work before barrier: L1 %.0f, M1 (incl. LDS wait) %.0f, M4 %.0f, M4 + piece wait %.0f" % tuple(np.median(pre, 0))
            print(line, flush=True)
engine.set_gemm_variant(7 | (4 << 16) | (3 << 20))      # back to the default tile order (3) and epilogue choice (2)
engine.set_gemm_variant(0)
