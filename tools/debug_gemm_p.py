import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowledge_enhanced_multimodal_retrieval_amd import engine
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
m, n, k = 256, 256, 128
a = torch.randn(m, k, generator=g).to(torch.bfloat16); w = (torch.randn(n, k, generator=g) * k**-0.5).to(torch.bfloat16)
ref = a.float() @ w.float().T
engine.set_gemm_variant(4)
out = engine.op_gemm(a.to(dev), w.to(dev), None, m, 0).float().cpu()
engine.set_gemm_variant(0)
torch.set_printoptions(precision=3, linewidth=200)
print("out[112, 0:8]", out[112, 0:8]); print("ref[112, 0:8]", ref[112, 0:8])
print("out[113, 0:8]", out[113, 0:8]); print("ref[113, 0:8]", ref[113, 0:8])
# search: does out[112,2] equal some ref element in the same row / column?
for (r, c) in [(112, 2), (112, 3), (113, 2), (120, 6)]:
    v = out[r, c]
    hits = (ref - v).abs() < 4e-3 * (1 + abs(float(v)))
    print((r, c), "out", float(v), "ref", float(ref[r, c]), "matches ref at", hits.nonzero().tolist()[:8])
# per-k-tile partials
for kt in range(k // 64):
    part = a.float()[:, kt*64:(kt+1)*64] @ w.float()[:, kt*64:(kt+1)*64].T
    print("kt", kt, "out[112,2]-part", float(out[112, 2] - part[112, 2]), "part", float(part[112, 2]))
