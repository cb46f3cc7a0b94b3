#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container: it imports the reference's own importable modules from
/root/reference (``src.clip.eval.metrics``, ``src.clip.eval.fusion``, ``src.clip.model.fusion_model``)
and a from-config ``transformers.CLIPModel`` (the class the reference calls in
``src/clip/eval/evaluator_hf.py``), feeds them seeded synthetic inputs and stores inputs + outputs.
Nothing here is needed (or present) on the GPU box; the tests read only the files it wrote.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def ref_modules():
    """Import the reference packages without shadowing this repo's own ``src`` package."""
    saved = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        import importlib
        metrics = importlib.import_module("src.clip.eval.metrics")
        fusion = importlib.import_module("src.clip.eval.fusion")
        fusion_model = importlib.import_module("src.clip.model.fusion_model")
        assert metrics.__file__.startswith(REF), metrics.__file__
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return metrics, fusion, fusion_model


def quiet(fn, *a, **kw):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def main():
    from oracle import clip_ref, metrics_ref
    metrics, fusion, fusion_model = ref_modules()

    # ------------------------------------------------------------------ 1. metrics (SURVEY 8(c) recipe)
    for n, d, tag in ((256, 768, "n256_d768"), (192, 128, "n192_d128")):
        img, q, t = metrics_ref.planted_embeddings(n, d, seed=0)
        S = q @ img.T
        out = {
            "image": img, "query": q, "target": t,
            "all_keys": None, "all_vals": None,
        }
        allm = metrics.compute_all_retrieval_metrics(q, t, img)
        final_55 = quiet(metrics.compute_retrieval_metrics_final, q, t, img, t2i_weight=0.5, t2t_weight=0.5)
        final_19 = quiet(metrics.compute_retrieval_metrics_final, q, t, img, prefix="F", t2i_weight=0.1, t2t_weight=0.9)
        train = metrics.compute_training_metrics(q, t, img)
        fus = metrics.compute_retrieval_metrics_fusion(0.5 * (q @ img.T) + 0.5 * (q @ t.T), prefix="X")
        ev = quiet(fusion.evaluate_retrieval, S)
        order = np.argsort(-S, axis=1)
        ranks = np.argmax(order == np.arange(n)[:, None], axis=1) + 1
        del out["all_keys"], out["all_vals"]
        np.savez_compressed(
            os.path.join(HERE, f"metrics_{tag}.npz"), **out,
            t2i_top10=order[:, :10].astype(np.int32), t2i_ranks=ranks.astype(np.int32),
            metrics_json=np.frombuffer(json.dumps({
                "all": allm, "final_0.5_0.5": final_55, "final_0.1_0.9_prefixF": final_19,
                "training": train, "fusion_prefixX": fus, "evaluate_retrieval_t2i": ev,
            }, sort_keys=True).encode(), dtype=np.uint8))
        print(tag, {k: round(v, 4) for k, v in allm.items() if k.startswith("T2I")})

    # ------------------------------------------------------------------ 2. SPARQL score fusion
    rng = np.random.default_rng(7)
    n = 48
    S = rng.standard_normal((n, n)).astype(np.float32) * 0.1
    uuids = [f"uuid-{i:04d}" for i in range(n)]
    results = {}
    for i in range(0, n, 2):
        k = int(rng.integers(0, 9)) if i % 6 else 60
        picks = rng.integers(0, n + 6, size=k)       # some ids fall outside the gallery
        results[uuids[i]] = [
            (f"http://example.org/artefact/uuid-{j:04d}" if (j % 3) else f"uuid-{j:04d}") for j in picks]
    results[uuids[0]] = []                             # empty result set
    fused = {
        "weighted_a0.7": quiet(fusion.fuse_clip_and_text2sparql, S, results, uuids, uuids, "weighted",
                               {"alpha": 0.7, "sparql_weight": 0.3}),
        "weighted_a0.6_w0.6": quiet(fusion.weighted_fusion, S, results, uuids, uuids, 0.6, 0.6),
        "additive_d0.5": quiet(fusion.fuse_clip_and_text2sparql, S, results, uuids, uuids, "additive", {"delta": 0.5}),
        "adaptive_d0.5": quiet(fusion.fuse_clip_and_text2sparql, S, results, uuids, uuids, "adaptive", {"delta": 0.5}),
    }
    fmetrics = {k: quiet(fusion.evaluate_retrieval, v) for k, v in fused.items()}
    np.savez_compressed(os.path.join(HERE, "sparql_fusion.npz"), S=S, **fused,
                        meta_json=np.frombuffer(json.dumps({"uuids": uuids, "results": results,
                                                            "metrics": fmetrics}).encode(), dtype=np.uint8))
    print("sparql fusion", {k: round(v["MRR"], 3) for k, v in fmetrics.items()})

    # ------------------------------------------------------------------ 3. learned fusion heads (eval mode)
    D, Nq, M = 64, 12, 20
    g = torch.Generator().manual_seed(11)
    qe = torch.nn.functional.normalize(torch.randn(Nq, D, generator=g), dim=-1)
    ie = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1)
    te = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1)
    heads = {"q": qe.numpy(), "img": ie.numpy(), "tgt": te.numpy()}

    class _NoClip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

    for ft in ("linear", "gated", "simple_gated", "simple_gated_with_bias", "bilinear", "cross_attention"):
        torch.manual_seed(3)
        fm = fusion_model.FusionModel(_NoClip(), fusion_type=ft, embed_dim=D).eval()
        with torch.no_grad():
            for p in fm.fusion_head.parameters():      # zero-initialised params would make the test vacuous
                p.add_(torch.randn(p.shape, generator=g) * 0.2)
            out = fm(qe, ie, te)
        heads[f"{ft}__out"] = out.numpy()
        for k, v in fm.fusion_head.state_dict().items():
            heads[f"{ft}__sd__{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "fusion_heads.npz"), **heads)
    print("heads ok")

    # ------------------------------------------------------------------ 4. encoder cross-check (HF from-config)
    from transformers import CLIPConfig, CLIPModel
    for name in ("tiny", "tiny-long"):
        arch = clip_ref.ARCHS[name]
        sd = clip_ref.random_state_dict(arch, seed=0)
        cfg = CLIPConfig(**clip_ref.hf_config_kwargs(arch))
        cfg._attn_implementation = "eager"
        m = CLIPModel(cfg).eval().float()
        m.load_state_dict(clip_ref.to_hf_state_dict(sd, arch), strict=True)
        gg = torch.Generator().manual_seed(1234)
        px = torch.randn(4, 3, arch["image_size"], arch["image_size"], generator=gg)
        ids = clip_ref.synthetic_ids(arch, 6)
        with torch.no_grad():
            hi = m.get_image_features(pixel_values=px)
            hi = hi if torch.is_tensor(hi) else hi.pooler_output
            ht = m.get_text_features(input_ids=ids.long())
            ht = ht if torch.is_tensor(ht) else ht.pooler_output
        chk = {k: float(v.double().abs().sum()) for k, v in sd.items()}
        np.savez_compressed(os.path.join(HERE, f"clip_hf_{name}.npz"), pixels=px.numpy(), ids=ids.numpy(),
                            image_features=hi.numpy(), text_features=ht.numpy(),
                            weight_abs_sums=np.frombuffer(json.dumps(chk, sort_keys=True).encode(), dtype=np.uint8))
        oi = clip_ref.encode_image(sd, arch, px)
        ot = clip_ref.encode_text(sd, arch, ids)
        print(name, "oracle vs HF:", float((oi - hi).abs().max()), float((ot - ht).abs().max()))

    # ------------------------------------------------------------------ 4b. the same cross-check at the FULL ViT-L/14 and ViT-B/32
    # shapes (width / depth / heads / patch of the models the bench and the reference scripts use).  Inputs and weights come
    # from seeds, so the fixture holds only the HF outputs and checksums of what was fed in (a few KB).
    hf_full_shape()

    # ------------------------------------------------------------------ 5. RetrievalEngine linear fuse (retrieval.py:23-76 is
    # not importable: it needs dotenv + network-bound constructors; the vector below restates its documented arithmetic on a
    # hand-checkable case and is marked "restated", not "reference-generated")
    clip_results = [{"uuid": f"u{i}", "score": s} for i, s in enumerate([0.91234, 0.5, 0.49996, 0.3, 0.12345, -0.2])]
    sparql = ["u3", "u5", "zz"]
    expect = sorted(({"uuid": it["uuid"], "score": round(0.8 * it["score"] + 0.2 * (it["uuid"] in sparql), 4)}
                     for it in clip_results), key=lambda x: x["score"], reverse=True)
    with open(os.path.join(HERE, "engine_fuse.json"), "w") as f:
        json.dump({"origin": "restated from src/retrieval.py:23-76 (module not importable offline)",
                   "clip_results": clip_results, "sparql_results": sparql, "alpha": 0.8, "beta": 0.2,
                   "expected": expect}, f, indent=1)
    print("done")


def hf_full_shape():
    from transformers import CLIPConfig, CLIPModel
    from oracle import clip_ref
    for name in ("ViT-B/32", "ViT-L/14"):
        arch = clip_ref.ARCHS[name]
        sd = clip_ref.random_state_dict(arch, seed=0)
        cfg = CLIPConfig(**clip_ref.hf_config_kwargs(arch))
        cfg._attn_implementation = "eager"
        m = CLIPModel(cfg).eval().float()
        m.load_state_dict(clip_ref.to_hf_state_dict(sd, arch), strict=True)
        gg = torch.Generator().manual_seed(1234)
        px = torch.randn(2, 3, arch["image_size"], arch["image_size"], generator=gg)
        ids = clip_ref.synthetic_ids(arch, 3)
        with torch.no_grad():
            hi = m.get_image_features(pixel_values=px)
            hi = hi if torch.is_tensor(hi) else hi.pooler_output
            ht = m.get_text_features(input_ids=ids.long())
            ht = ht if torch.is_tensor(ht) else ht.pooler_output
        chk = {k: float(v.double().abs().sum()) for k, v in sd.items()}
        meta = {"pixel_seed": 1234, "pixel_abs_sum": float(px.double().abs().sum()), "n_images": 2, "n_texts": 3,
                "weight_abs_sums": chk}
        np.savez_compressed(os.path.join(HERE, "clip_hf_full_%s.npz" % name.replace("/", "-")), ids=ids.numpy(),
                            image_features=hi.numpy(), text_features=ht.numpy(),
                            meta_json=np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8))
        oi = clip_ref.encode_image(sd, arch, px)
        ot = clip_ref.encode_text(sd, arch, ids)
        print(name, "full shape, oracle vs HF:", float((oi - hi).abs().max()), float((ot - ht).abs().max()), flush=True)
        del m


# (architecture, gallery items, noise levels of the image queries, store the oracle's embeddings in full).  Round 4:
# ViT-L/14 at N = 128 with the embeddings stored -- the oracle takes 0.3-0.7 s per ViT-L/14 image, so the GPU test reads them from
# the fixture (and re-runs the oracle on the first rows only, to pin them) instead of spending minutes of box time per run.
E2E_CASES = (("ViT-B/32", 256, (2.0, 3.0, 4.0), False), ("ViT-L/14", 32, (3.0,), False), ("ViT-L/14", 128, (2.0, 3.0, 4.0), True))


def e2e_inputs(arch, n, levels):
    """Seeded inputs of the end-to-end fixture (regenerated, not stored, by tests/test_e2e_gpu.py): gallery images, query and target
    texts as the reference's eval loop encodes them (evaluator.py:121-135), and noisy copies of every gallery image as further
    query sets (I2I): random-weight towers put all embeddings in a narrow cone (score spread 2e-3), so text queries have no margin
    to speak of, while a noisy copy's ground truth has a real one and Recall@K moves with the noise level."""
    from oracle import clip_ref
    g = torch.Generator().manual_seed(20261004)
    px = torch.randn(n, 3, arch["image_size"], arch["image_size"], generator=g)
    nz = torch.randn(n, 3, arch["image_size"], arch["image_size"], generator=g)
    return px, {lvl: px + lvl * nz for lvl in levels}, clip_ref.synthetic_ids(arch, n, seed=777), clip_ref.synthetic_ids(arch, n, seed=778)


def e2e_oracle_embeddings(sd, arch, px, noisy, q_ids, t_ids):
    from oracle import clip_ref
    with torch.no_grad():
        emb = {"image": clip_ref.l2_normalize(clip_ref.encode_image(sd, arch, px)).numpy(),
               "query": clip_ref.l2_normalize(clip_ref.encode_text(sd, arch, q_ids)).numpy(),
               "target": clip_ref.l2_normalize(clip_ref.encode_text(sd, arch, t_ids)).numpy()}
        for lvl, x in noisy.items():
            emb[f"noisy{lvl}"] = clip_ref.l2_normalize(clip_ref.encode_image(sd, arch, x)).numpy()
    return emb


def e2e_tasks(emb, levels):
    """task -> (query embeddings, [(weight, candidate embeddings)]): what the reference's evaluators score (metrics.py:102, 145-148)."""
    tasks = {"T2I": (emb["query"], [(1.0, emb["image"])]), "T2T": (emb["query"], [(1.0, emb["target"])]),
             "FUSED": (emb["query"], [(0.5, emb["image"]), (0.5, emb["target"])])}
    for lvl in levels:
        tasks[f"I2I@{lvl}"] = (emb[f"noisy{lvl}"], [(1.0, emb["image"])])
    return tasks


def e2e():
    """Oracle encoders -> the REFERENCE's metrics.py, end to end (missing #1; evaluator.py:121-156 -> metrics.py:34-41,
    62-68): per task the reference's metric dict, the oracle's top-11 ids and scores of every query (stable order) and the rank of
    the ground truth.  A few hundred KB; inputs regenerate from seeds, the test re-runs the oracle and checks it against the
    checksums stored here before it trusts its margins."""
    from oracle import clip_ref
    metrics, _, _ = ref_modules()
    only = os.environ.get("KEMR_E2E_ONLY", "")            # e.g. "ViT-L/14:128": regenerate one case
    for name, n, levels, store in E2E_CASES:
        if only and only != f"{name}:{n}":
            continue
        arch = clip_ref.ARCHS[name]
        sd = clip_ref.random_state_dict(arch, seed=0)
        px, noisy, q_ids, t_ids = e2e_inputs(arch, n, levels)
        emb = e2e_oracle_embeddings(sd, arch, px, noisy, q_ids, t_ids)
        out = {"first8_" + k: v[:8] for k, v in emb.items()}
        if store:
            out.update({"emb_" + k: v.astype(np.float32) for k, v in emb.items()})
        meta = {"arch": name, "n": n, "levels": list(levels), "weights_seed": 0,
                "input_abs_sums": {"pixels": float(px.double().abs().sum()), "query_ids": int(q_ids.long().sum()), "target_ids": int(t_ids.long().sum()),
                                   **{f"noisy{lvl}": float(x.double().abs().sum()) for lvl, x in noisy.items()}},
                "embedding_abs_sums": {k: float(np.abs(v.astype(np.float64)).sum()) for k, v in emb.items()}, "metrics": {}}
        for task, (q, parts) in e2e_tasks(emb, levels).items():
            prefix = task.split("@")[0]
            if task == "FUSED":
                m = quiet(metrics.compute_retrieval_metrics_final, q, emb["target"], emb["image"], prefix=prefix, t2i_weight=0.5, t2t_weight=0.5)
            else:
                m = metrics.compute_retrieval_metrics(q, parts[0][1], prefix)
            S = sum(w * (q @ c.T) for w, c in parts).astype(np.float32)           # the reference's expression (metrics.py:102, 145-148)
            order = np.argsort(-S, axis=1, kind="stable")
            ranks = np.argmax(order == np.arange(n)[:, None], axis=1) + 1
            assert abs(float(np.mean(1.0 / ranks) * 100.0) - m[f"{prefix}_MRR"]) < 1e-3, (task, m)       # unstable vs stable sort only moves exact ties
            out[f"{task}_top11_ids"] = order[:, :11].astype(np.int32)
            out[f"{task}_top11_scores"] = np.take_along_axis(S, order[:, :11], axis=1)
            out[f"{task}_ranks"] = ranks.astype(np.int32)
            meta["metrics"][task] = {k: float(v) for k, v in m.items()}
            top = out[f"{task}_top11_scores"]
            print(name, task, {k: round(float(v), 2) for k, v in m.items()}, "| median 10/11 margin %.1e, score spread %.1e" % (np.median(top[:, 9] - top[:, 10]), S.std()), flush=True)
        np.savez_compressed(os.path.join(HERE, "e2e_%s_n%d.npz" % (name.replace("/", "-"), n)), **out,
                            meta_json=np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8))


RECALL_ANCHOR = dict(name="ViT-L/14", n_sub=256, levels=(1.5, 2.0, 2.5), seed=424242)


def recall_anchor_inputs(size=224):
    """The first n_sub gallery images of the ViT-L/14 Recall test and their noise, from a CPU generator (the rest of that test's
    16 384 items come from the GPU's generator: only these rows have to be the same pixels in the build container and on the box)."""
    c = RECALL_ANCHOR
    g = torch.Generator().manual_seed(c["seed"])
    base = torch.randn(c["n_sub"], 3, size, size, generator=g)
    noise = torch.randn(c["n_sub"], 3, size, size, generator=g)
    return base, noise


def recall_anchor():
    """Oracle embeddings for the anchor of tests/test_encoder_gpu.py::test_recall_at_10_...[ViT-L/14] (configs[4]
    names ViT-L/14; the oracle needs minutes for 1 024 ViT-L/14 images, so they are computed HERE, once): the fp32 CPU oracle on the
    first 256 gallery images and on their noisy copies at the three levels, L2-normalised, plus input checksums."""
    from oracle import clip_ref
    c = RECALL_ANCHOR
    arch = clip_ref.ARCHS[c["name"]]
    sd = clip_ref.random_state_dict(arch, seed=0)
    base, noise = recall_anchor_inputs(arch["image_size"])
    out, meta = {}, {"arch": c["name"], "n_sub": c["n_sub"], "levels": list(c["levels"]), "seed": c["seed"], "weights_seed": 0,
                     "input_abs_sums": {"base": float(base.double().abs().sum()), "noise": float(noise.double().abs().sum())}}
    with torch.no_grad():
        out["gallery"] = clip_ref.l2_normalize(clip_ref.encode_image(sd, arch, base)).numpy().astype(np.float32)
        print("recall anchor: gallery done", flush=True)
        for lvl in c["levels"]:
            out[f"query_{lvl}"] = clip_ref.l2_normalize(clip_ref.encode_image(sd, arch, base + lvl * noise)).numpy().astype(np.float32)
            S = out[f"query_{lvl}"] @ out["gallery"].T
            ranks = (S > np.diag(S)[:, None]).sum(1) + 1
            meta[f"recall10_oracle_{lvl}"] = float(100.0 * np.mean(ranks <= 10))
            print("recall anchor: level", lvl, "oracle Recall@10 on the subset", meta[f"recall10_oracle_{lvl}"], flush=True)
    np.savez_compressed(os.path.join(HERE, "recall_anchor_ViT-L-14.npz"), **out,
                        meta_json=np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hf-full":
        hf_full_shape()
    elif len(sys.argv) > 1 and sys.argv[1] == "e2e":
        e2e()
    elif len(sys.argv) > 1 and sys.argv[1] == "recall-anchor":
        recall_anchor()
    else:
        main()
        e2e()
