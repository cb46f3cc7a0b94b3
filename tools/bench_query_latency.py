#!/usr/bin/env python3
"""Single-query online path: tokenised query -> text tower (B = 1) -> fused T2I+T2T sim + top-10 over a
43 000-item store.  Wall time per query with the result on the host, and the GPU time of the two halves."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knowledge_enhanced_multimodal_retrieval_amd import engine, _lib, ranking
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
from oracle import clip_ref, metrics_ref
dev = torch.device("cuda:0")
arch = ARCHS["ViT-L/14"]
eng = engine.ClipEngine(arch, dev)
eng.load_state_dict(clip_ref.random_state_dict(clip_ref.ARCHS["ViT-L/14"], seed=0))
n = 43000
img, _, tgt = metrics_ref.planted_embeddings(n, 768, 0)
img, tgt = torch.from_numpy(img).to(dev), torch.from_numpy(tgt).to(dev)
for terms, label in ((1, "bf16"), (3, "fp32x3")):
    panel = engine.build_panel([img, tgt], _lib.SIDE_GALLERY, terms)
    ids = clip_ref.synthetic_ids(clip_ref.ARCHS["ViT-L/14"], 1).to(dev)
    def query():
        q = eng.encode_text(ids, normalize=True)
        qp = engine.build_panel([q, q], _lib.SIDE_QUERY, terms, part_scale=[0.5, 0.5])
        s, i = engine.sim_topk(qp, panel, 10)
        return s[0].cpu().tolist(), i[0].cpu().tolist()
    for _ in range(20):
        query()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        query()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    te = ts = 0.0
    for _ in range(50):
        e0.record(); q = eng.encode_text(ids, normalize=True); e1.record()
        qp = engine.build_panel([q, q], _lib.SIDE_QUERY, terms, part_scale=[0.5, 0.5]); s, i = engine.sim_topk(qp, panel, 10); e2.record()
        torch.cuda.synchronize()
        te += e0.elapsed_time(e1); ts += e1.elapsed_time(e2)
    print(label, "wall per query %.3f ms; gpu: text tower %.3f ms, panel+sim+top10 %.3f ms" % (wall, te / 50, ts / 50), flush=True)

# ---- the same query path captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed: the path is capture-safe
# (no allocation, no synchronisation inside the library), but at B = 1 the ~90 kernels are GPU-bound, so nothing is gained
terms = 1
panel = engine.build_panel([img, tgt], _lib.SIDE_GALLERY, terms)
static_ids = clip_ref.synthetic_ids(clip_ref.ARCHS["ViT-L/14"], 1).to(dev)
def body():
    q = eng.encode_text(static_ids, normalize=True)
    qp = engine.build_panel([q, q], _lib.SIDE_QUERY, terms, part_scale=[0.5, 0.5])
    return engine.sim_topk(qp, panel, 10)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s_out, i_out = body()
torch.cuda.synchronize()
ref_s, ref_i = body()
g.replay(); torch.cuda.synchronize()
print("graph result equals eager:", torch.equal(s_out, ref_s) and torch.equal(i_out, ref_i), flush=True)
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay(); r = (s_out[0].cpu().tolist(), i_out[0].cpu().tolist())
print("graph replay: wall per query %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3), flush=True)
