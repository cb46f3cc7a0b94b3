#!/usr/bin/env python3
"""Round 4 experiment: does the chip overlap the HBM-bound kernels of one encoder call (LayerNorm, attention staging: 25 % of the
step, zero MFMA) with the MFMA-bound GEMMs of ANOTHER call when the two are queued on two HIP streams?  The persistent GEMM takes
one workgroup and 128 KiB of LDS per CU, so two GEMMs serialise; LayerNorm workgroups (no LDS) fit beside it.

Legs (same weights, one process, interleaved rounds; images/s or items/s):
  one       255 images per call, one stream (the product's arrangement)
  halves    127 + 127 images, two calls on ONE stream (what the smaller M costs by itself: 128 row tiles x 12 = 6 whole rounds)
  streams   127 images on stream A, 127 on stream B, queued alternately
  it-seq    255 images + 1 501 packed texts (one step's worth), one stream
  it-par    the images on stream A, the texts on stream B
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (random_weights, synthetic_ids)
from knowledge_enhanced_multimodal_retrieval_amd import debug, engine  # noqa: E402
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS  # noqa: E402

dev = torch.device("cuda:0")
arch = ARCHS["ViT-L/14"]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sd = bench.random_weights(arch, 0)
A = engine.ClipEngine(arch, dev, precision=prec); A.load_state_dict(sd)
Bn = engine.ClipEngine(arch, dev, precision=prec); Bn.load_state_dict(sd)
g = torch.Generator().manual_seed(1)
px = torch.randn(255, 3, 224, 224, generator=g).to(dev)
ids = torch.cat([bench.synthetic_ids(arch, 255, 5), bench.synthetic_ids(arch, 255, 6)] * 3)[:1501]
lens = engine.text_lengths(ids)
ids_d = ids.to(dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
sHi, sLo = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)      # the texts beside the images at a higher / the same priority


def leg_one(n):
    for _ in range(n):
        A.encode_image(px, normalize=True)
    return 255 * n


def leg_halves(n):
    for _ in range(n):
        A.encode_image(px[:127], normalize=True)
        A.encode_image(px[127:254], normalize=True)
    return 254 * n


def leg_streams(n):
    for _ in range(n):
        with torch.cuda.stream(sA):
            A.encode_image(px[:127], normalize=True)
        with torch.cuda.stream(sB):
            Bn.encode_image(px[127:254], normalize=True)
    return 254 * n


def leg_it_seq(n):
    for _ in range(n):
        A.encode_image(px, normalize=True)
        A.encode_text(ids_d, normalize=True, lens=lens)
    return (255 + 1501) * n


def leg_it_par(n):
    for _ in range(n):
        with torch.cuda.stream(sA):
            A.encode_image(px, normalize=True)
        with torch.cuda.stream(sB):
            Bn.encode_text(ids_d, normalize=True, lens=lens)
    return (255 + 1501) * n


def leg_it_par_main(n):          # the product's arrangement: images on the caller's (default) stream, texts on a side stream
    for _ in range(n):
        A.encode_image(px, normalize=True)
        with torch.cuda.stream(sB):
            Bn.encode_text(ids_d, normalize=True, lens=lens)
    return (255 + 1501) * n


def leg_it_par_hi(n):            # the text stream at high priority
    for _ in range(n):
        A.encode_image(px, normalize=True)
        with torch.cuda.stream(sHi):
            Bn.encode_text(ids_d, normalize=True, lens=lens)
    return (255 + 1501) * n


def leg_alt255(n):               # consecutive 255-image calls alternate between two streams (two engines = two workspaces)
    for i in range(n):
        if i % 2 == 0:
            with torch.cuda.stream(sA):
                A.encode_image(px, normalize=True)
        else:
            with torch.cuda.stream(sB):
                Bn.encode_image(px, normalize=True)
    return 255 * n


def leg_alt255_text(n):          # the same with the texts on a third stream (another engine would be needed for a clean split: here the
    for i in range(n):           # texts share engine B's text workspace only, which the image calls do not touch)
        if i % 2 == 0:
            with torch.cuda.stream(sA):
                A.encode_image(px, normalize=True)
        else:
            with torch.cuda.stream(sB):
                Bn.encode_image(px, normalize=True)
        with torch.cuda.stream(sHi):
            Bn.encode_text(ids_d, normalize=True, lens=lens)
    return (255 + 1501) * n


def with_grid(fn, cap):         # the same leg with the persistent GEMM's grid capped (debug switch gemm_grid): two half-chip GEMMs side by side
    def run(n):
        debug.set("gemm_grid", cap)
        try:
            return fn(n)
        finally:
            debug.set("gemm_grid", 0)
    return run


legs = {"alt-255": leg_alt255, "alt-255+text": leg_alt255_text, "it-par-main": leg_it_par_main, "it-par-hi": leg_it_par_hi, "one": leg_one, "halves": leg_halves, "streams": leg_streams, "it-seq": leg_it_seq, "it-par": leg_it_par}
legs["alt-255 grid128"] = with_grid(leg_alt255, 128)
legs["streams grid128"] = with_grid(leg_streams, 128)
legs["alt-255+text grid128"] = with_grid(leg_alt255_text, 128)
legs["alt-255 grid192"] = with_grid(leg_alt255, 192)
if len(sys.argv) > 3:
    legs = {k: v for k, v in legs.items() if k in sys.argv[3].split(",")}
res = {k: [] for k in legs}
# results must not depend on the arrangement
torch.cuda.synchronize()
ref = A.encode_image(px[:254], normalize=True)
torch.cuda.synchronize()        # an engine's workspace belongs to ONE call at a time: the next call on another stream must not start
                                # while this one runs (the first version of this script did not wait here: two calls scribbled over
                                # each other's pool_idx / row_start rows in the shared workspace and a gather faulted)
with torch.cuda.stream(sA):
    a = A.encode_image(px[:127], normalize=True)
with torch.cuda.stream(sB):
    b = Bn.encode_image(px[127:254], normalize=True)
torch.cuda.synchronize()
cos = torch.nn.functional.cosine_similarity(torch.cat([a, b]).double(), ref.double()).min().item()
print("two-stream halves vs one call: min cosine", cos)
for r in range(rounds + 1):
    for name, fn in legs.items():
        fn(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        items = fn(12)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r:
            res[name].append(items / dt)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:8s} median {v[len(v) // 2]:9.0f} /s   min {v[0]:9.0f}   max {v[-1]:9.0f}")
