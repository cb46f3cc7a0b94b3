"""Similarity + top-k timings by route (tools only).

    python tools/bench_sim.py [--nq 43000] [--ng 43000] [--d 768] [--terms 1] [--k 10] [--iters 5]

Prints one JSON line per route: 0 = sim_kernel, 1 = candidate lists (default route), rank-only.  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of the list route.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import _lib, debug, engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=43000)
    ap.add_argument("--ng", type=int, default=43000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--terms", type=int, default=1)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--routes", default="0,1")
    ap.add_argument("--kl", default="0", help="K loops of the list pass to time, e.g. 0,1 (debug switch gemm_kl; 1 needs a build.py --ab-variants library)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stamps", action="store_true", help="in-kernel cycle split of the list pass (stamped instantiation)")
    ap.add_argument("--device-rng", action="store_true", help="the data bench.py uses (device generator, seed 7)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    src = torch.arange(args.nq) % args.ng
    if args.device_rng:
        gg = torch.Generator(device=dev).manual_seed(7 + args.seed)
        gal = torch.nn.functional.normalize(torch.randn(args.ng, args.d, generator=gg, device=dev), dim=-1)
        qry = torch.nn.functional.normalize(gal[src.to(dev)] + 0.04 * torch.randn(args.nq, args.d, generator=gg, device=dev), dim=-1)
    else:
        g = torch.Generator().manual_seed(args.seed)
        gal = torch.nn.functional.normalize(torch.randn(args.ng, args.d, generator=g), dim=-1).to(dev)
        qry = torch.nn.functional.normalize(gal[src.to(dev)] + 0.04 * torch.randn(args.nq, args.d, generator=g).to(dev), dim=-1)
    qp = engine.build_panel([qry], _lib.SIDE_QUERY, args.terms)
    gp = engine.build_panel([gal], _lib.SIDE_GALLERY, args.terms)
    gt = src.int().to(dev)
    sgt = engine.pair_scores(qp, gp, torch.arange(args.nq, device=dev).int(), gt)
    flops = 2.0 * args.nq * args.ng * qp.kdim
    ref = None
    for route, kl in [(int(r), int(q)) for r in args.routes.split(",") for q in args.kl.split(",")]:
        debug.set("sim_lists", route)
        debug.set("gemm_kl", kl)
        for with_rank in (False, True):
            def run():
                ahead = torch.zeros(args.nq, dtype=torch.int32, device=dev) if with_rank else None
                return engine.sim_topk(qp, gp, args.k, 0, gt if with_rank else None, sgt if with_rank else None, ahead, return_workspace=True), ahead
            (s, i, ws), ahead = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.iters * 1e3
            if ref is None:
                ref = (s, i)
            same = bool(torch.equal(s, ref[0]) and torch.equal(i, ref[1]))
            st = debug.sim_lists(ws, args.nq, args.ng, qp.kdim, args.k)
            print(json.dumps({"route": route, "kl": kl, "with_rank": with_rank, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1),
                              "same_as_first": same, "nq": args.nq, "ng": args.ng, "kdim": qp.kdim, "k": args.k,
                              "lists": dict(zip(("flag", "longest", "cap", "chunks", "sampled", "records_per_query"), list(st)))}))
    debug.set("sim_lists", 1)
    debug.set("gemm_kl", 0)
    if args.stamps:
        import ctypes as C
        import numpy as np
        for label, k, rank in (("rank only (SIM 1)", 0, True), ("lists + rank (SIM 2)", args.k, True), ("lists (SIM 2)", args.k, False)):
            debug.set("gemm_flags", 64)
            ahead = torch.zeros(args.nq, dtype=torch.int32, device=dev)
            for _ in range(3):
                if rank:
                    engine.sim_topk(qp, gp, k, 0, gt, sgt, ahead)
                else:
                    engine.sim_topk(qp, gp, k, 0)
            torch.cuda.synchronize()
            buf = (C.c_uint * (1024 * 16))()
            _lib.check(_lib.lib().kemr_debug_gemm_stamps(buf, 1024 * 16), "stamps")
            debug.set("gemm_flags", 0)
            st = np.frombuffer(buf, dtype=np.uint32).reshape(1024, 16).astype(np.float64)[:256]
            tiles, nt = st[:, 14], st[:, 15]
            per_ktile = st[:, :8] / (tiles * nt)[:, None]
            print(f"stamps {label}: tiles/wg {np.median(tiles):.0f}, K-tiles {np.median(nt):.0f}; cycles per K-tile interval "
                  + " ".join("%.0f" % x for x in np.median(per_ktile, 0)) + f" | sum {np.median(per_ktile.sum(1)):.0f}"
                  + f" | per tile: K-loop tail {np.median(st[:, 8] / tiles):.0f}, scan {np.median(st[:, 9] / tiles):.0f}", flush=True)


if __name__ == "__main__":
    main()
