#!/usr/bin/env python3
"""Round-3 similarity routes, each against round 2's route for the same call (debug switch sim_lists = 0: sim_kernel):
  (a) an 8-way shard: Q = 1 024 queries x 5 375 gallery rows, top-10 (BASELINE configs[3]) -- candidate lists from 2 048 rows up;
  (b) the alpha sweep of the SPARQL score fusion: rank-only passes with a bonus list at 43 000 x 43 000, fused T2I + T2T panels
      (kdim 1 536), nine alphas -- rank-count pass on the raw scores + per-query fix-up of the candidates that carry a bonus;
  (c) the headline Q = 43 000 top-10 call for reference.  Results must be identical between the routes."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import _lib, debug, engine  # noqa: E402

dev = torch.device("cuda:0")
gg = torch.Generator(device=dev).manual_seed(7)
N, D = 43000, 768
gal = torch.nn.functional.normalize(torch.randn(N, D, generator=gg, device=dev), dim=-1)
tgt = torch.nn.functional.normalize(gal + 0.5 * torch.randn(N, D, generator=gg, device=dev), dim=-1)
qry = torch.nn.functional.normalize(gal + 0.04 * torch.randn(N, D, generator=gg, device=dev), dim=-1)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


# (a) + (c)
for label, nq, lo, hi, reps in (("shard_q1024_x_5375", 1024, 7 * 5375, 8 * 5375, 20), ("q1024_x_43000", 1024, 0, N, 20), ("q43000_x_43000", N, 0, N, 3)):
    gp = engine.build_panel([gal[lo:hi]], _lib.SIDE_GALLERY, 1)
    res = {}
    for route in (0, 1):
        debug.set("sim_lists", route)
        ms, out = timed(lambda: engine.sim_topk(engine.build_panel([qry[:nq]], _lib.SIDE_QUERY, 1), gp, 10, lo), reps)
        res[route] = (ms, out)
    same = torch.equal(res[0][1][0], res[1][1][0]) and torch.equal(res[0][1][1], res[1][1][1])
    print(json.dumps({"case": label, "sim_kernel_ms": round(res[0][0], 4), "lists_ms": round(res[1][0], 4),
                      "speedup": round(res[0][0] / res[1][0], 2), "identical": bool(same)}), flush=True)
debug.set("sim_lists", 1)

# (b) nine alphas, weighted fusion: score = alpha * (0.5 T2I + 0.5 T2T) + (1 - alpha) * hit; a third of the queries have ~20 hits
rng = np.random.default_rng(0)
ptr, cols = [0], []
for q in range(N):
    if q % 3 == 0:
        c = np.unique(rng.integers(0, N, size=20))
        cols += list(c)
    ptr.append(len(cols))
ptr_t = torch.tensor(ptr, dtype=torch.int32, device=dev)
col_t = torch.tensor(cols, dtype=torch.int32, device=dev)
gp2 = engine.build_panel([gal, tgt], _lib.SIDE_GALLERY, 1)
gt = torch.arange(N, dtype=torch.int32, device=dev)
alphas = [0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1]


def sweep():
    outs = []
    for a in alphas:
        qp = engine.build_panel([qry, qry], _lib.SIDE_QUERY, 1, part_scale=[0.5 * a, 0.5 * a])
        val_t = torch.full((len(cols),), 1.0 - a, dtype=torch.float32, device=dev)
        sgt = engine.pair_scores(qp, gp2, gt, gt)                      # (no query's own item is among its hits here)
        ahead = torch.zeros(N, dtype=torch.int32, device=dev)
        engine.sim_topk(qp, gp2, 0, 0, gt, sgt, ahead, bonus=(ptr_t, col_t, val_t))
        outs.append(ahead)
    return outs


res = {}
for route in (0, 1):
    debug.set("sim_lists", route)
    res[route] = timed(sweep, 2)
debug.set("sim_lists", 1)
same = all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
print(json.dumps({"case": "alpha_sweep_9_rank_only_bonus_43000x43000_kdim1536", "sim_kernel_ms": round(res[0][0], 3),
                  "fast_pass_plus_fixup_ms": round(res[1][0], 3), "speedup": round(res[0][0] / res[1][0], 2), "identical": bool(same),
                  "hits": len(cols)}), flush=True)
