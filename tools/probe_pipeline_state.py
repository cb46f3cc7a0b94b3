"""Which piece of process state makes evaluators.encode_dataset crawl (round 3: 0.7 k instead of 8 k items/s inside bench.py)?
Runs the same 4 080-item pipeline in fresh child processes, each with one more piece of bench.py's preamble in front."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, warnings, json
sys.path.insert(0, %(root)r)
os.environ.setdefault("KEMR_ALLOW_RANDOM_WEIGHTS", "1"); os.environ.setdefault("KEMR_ALLOW_HASH_TOKENIZER", "1")
import torch
what = sys.argv[1].split(",")
dev = torch.device("cuda", 0)
if "setdev" in what:
    torch.cuda.set_device(dev)
if "threads16" in what:
    torch.set_num_threads(16)
if "cpu_randn" in what:
    x = [torch.randn(4096, 4096) * 0.5 for _ in range(20)]
if "engine" in what:
    import bench
    from knowledge_enhanced_multimodal_retrieval_amd import engine
    from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
    arch = ARCHS["ViT-L/14"]
    eng = engine.ClipEngine(arch, dev)
    eng.load_state_dict(bench.random_weights(arch, seed=0))
    px = torch.randn(255, 3, 224, 224).to(dev)
    for _ in range(3):
        eng.encode_image(px, normalize=True)
    torch.cuda.synchronize()
from knowledge_enhanced_multimodal_retrieval_amd import clip_api, datasets as kds, evaluators, tokenizer
clip_api.allow_random_weights(True); tokenizer.allow_hash_tokenizer(True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pm, ppre = clip_api.load("ViT-L/14", device=str(dev))
    evaluators.encode_dataset(pm, kds.CLIPEvalDatasetHF(kds.SyntheticHFSplit(510, 12), ppre), 64, 1, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = evaluators.encode_dataset(pm, kds.CLIPEvalDatasetHF(kds.SyntheticHFSplit(4080, 11), ppre), 64, 1, 12)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"preamble": what, "items_per_s": round(3 * 4080 / dt, 1), "threads": torch.get_num_threads()}), flush=True)
''' % {"root": ROOT}

for what, ctx in (("engine", "forkserver"), ("engine", "fork"), ("none", "forkserver"), ("none", "fork")):
    env = dict(os.environ, KEMR_LOADER_CONTEXT=ctx)
    r = subprocess.run([sys.executable, "-c", CHILD, what], capture_output=True, text=True, timeout=280, env=env)
    print(ctx, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + r.stderr[-400:]), flush=True)
