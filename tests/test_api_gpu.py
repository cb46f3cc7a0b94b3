"""GPU: the drop-in Python API (src.clip.*, clip shim, RetrievalEngine) end to end on the HIP engine against the oracle."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from knowledge_enhanced_multimodal_retrieval_amd import _lib
from oracle import clip_ref, fusion_ref, metrics_ref

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)


def test_config0_vit_b32_zeroshot_256(device, tmp_path):
    """BASELINE configs[0]: ViT-B/32 zero-shot on a 256-item subset through the evaluate_zeroshot plumbing
    (`src.clip.eval.evaluator.main`), synthetic data + seeded random weights (nothing can be fetched offline)."""
    import clip
    from knowledge_enhanced_multimodal_retrieval_amd import datasets, evaluators
    from src.clip.eval.evaluator import main
    out = tmp_path / "res" / "b32.json"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = main(["--model_name", "ViT-B/32", "--split", "test", "--splits_file", "splits.json", "--batch_size", "64",
                    "--device", "cuda", "--output_file", str(out), "--seed", "42", "--synthetic", "256"])
        saved = json.loads(out.read_text())
        # the reference's keys (evaluator.py:379-387) + the build's provenance block (what the numbers were computed with)
        assert set(saved) == {"image_transform", "loader_workers", "model_name", "checkpoint", "split", "tasks", "num_samples", "seed", "metrics",
                              "weights_source", "tokenizer", "precision", "data"}
        assert saved["weights_source"] == "random(seed 0)" and saved["data"] == "synthetic" and saved["precision"] == _lib.DEFAULT_PRECISION == "bf16-x24"
        assert saved["num_samples"] == 256 and saved["metrics"] == res["metrics"]
        keys = {f"{t}_{m}" for t in ("T2I", "I2T", "T2T") for m in ("R@1", "R@5", "R@10", "R@20", "MRR", "Mean_Rank")}
        assert set(res["metrics"]) == keys
        assert set(evaluators.evaluate_clip_model.last_analysis) == {"0.5_0.5", "0.1_0.9"}
        assert len(evaluators.evaluate_clip_model.last_analysis["0.5_0.5"]) == 3 + 9

        # the same pipeline piece by piece against the oracle
        model, _ = clip.load("ViT-B/32", device="cuda")
        ds = datasets.SyntheticRetrievalDataset(256, 224, seed=42)
        img, qry, tgt, uuids = evaluators.encode_dataset(model, ds, 64, 42)
        oa = clip_ref.ARCHS["ViT-B/32"]
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        n_or = 48                                                  # oracle encode on a slice (CPU seconds)
        px = torch.stack([ds[i][0] for i in range(n_or)])
        ids_q = clip.tokenize([ds[i][1] for i in range(n_or)], truncate=True)
        ids_t = clip.tokenize([ds[i][2] for i in range(n_or)], truncate=True)
    ref_i = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, px))
    ref_q = clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, ids_q))
    ref_t = clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, ids_t))
    for got, ref in ((img, ref_i), (qry, ref_q), (tgt, ref_t)):
        assert float((1 - _cos(got[:n_or].cpu(), ref)).max()) < 1e-3
        assert float((got.norm(dim=-1) - 1).abs().max()) < 1e-5
    # ranking parity: identical embeddings in -> identical metrics out (fp32 oracle ranking of OUR embeddings)
    gi, gq, gt_ = img.cpu().numpy(), qry.cpu().numpy(), tgt.cpu().numpy()
    want = metrics_ref.all_retrieval_metrics(gq, gt_, gi)
    for k, v in want.items():
        assert res["metrics"][k] == pytest.approx(v, abs=0.5), k       # <= 1 near-tie rank flip out of 256
    exact = metrics_ref.all_retrieval_metrics(gq.astype(np.float64), gt_.astype(np.float64), gi.astype(np.float64))
    assert sum(abs(res["metrics"][k] - exact[k]) > 1e-9 for k in exact) <= 4


@pytest.mark.parametrize("precision", ["bf16-res16", "fp8", "fp8-mlp"])
def test_precision_switch_reaches_the_drop_in_modules(device, tmp_path, monkeypatch, precision):
    """KEMR_PRECISION selects the encoder precision behind the unchanged CLI: the evaluator runs and its
    embeddings differ from the default (bf16 operands, fp32 residual stream) run by what that precision costs, no more."""
    import clip
    from knowledge_enhanced_multimodal_retrieval_amd import datasets, evaluators
    ds = datasets.SyntheticRetrievalDataset(64, 224, seed=7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.delenv("KEMR_PRECISION", raising=False)
        base, _ = clip.load("ViT-B/32", device="cuda")
        assert base.engine().precision == _lib.DEFAULT_PRECISION
        ref_img, ref_qry, _, _ = evaluators.encode_dataset(base, ds, 32, 7)
        monkeypatch.setenv("KEMR_PRECISION", precision)
        model, _ = clip.load("ViT-B/32", device="cuda")
        assert model.engine().precision == precision
        img, qry, _, _ = evaluators.encode_dataset(model, ds, 32, 7)
    tol = {"bf16-res16": 1e-3, "fp8": 5e-3, "fp8-mlp": 2e-2}[precision]
    di, dq = float((1 - _cos(img, ref_img)).max()), float((1 - _cos(qry, ref_qry)).max())
    assert 0 < di < tol and dq < tol
    if not precision.startswith("fp8"):
        assert dq > 0                                  # (fp8 operands are confined to the vision tower: the text embeddings are the default's)


def test_cli_runs_the_batched_device_pipeline_behind_the_reference_dataset_call(device, tmp_path, monkeypatch):
    """`python -m src.clip.eval.evaluator` on camera-sized uint8 sources.  The CLI builds its dataset exactly as the
    reference's main does -- CLIPEvalDatasetHF(split, preprocess) with the object load_clip_model returned (evaluator.py:330-333) --
    and by default that object defers the image transform to the GPU (one launch pair per loader batch, loader workers on):
    the metrics, and the embeddings behind them, are IDENTICAL to the run with the host transform (KEMR_GPU_PREPROCESS=0)."""
    from src.clip.eval.evaluator import main
    from knowledge_enhanced_multimodal_retrieval_amd import clip_api, datasets, evaluators, tokenizer
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("KEMR_GPU_PREPROCESS", mode)
        out = tmp_path / f"uint8_{mode}.json"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = main(["--model_name", "ViT-B/32", "--batch_size", "24", "--device", "cuda", "--synthetic", "72", "--synthetic_images", "uint8",
                        "--output_file", str(out)] + (["--num_workers", "0"] if mode == "0" else []))
        saved = json.loads(out.read_text())
        assert saved["image_transform"].startswith("gpu" if mode == "1" else "host"), saved["image_transform"]
        assert saved["num_samples"] == 72 and (saved["loader_workers"] == evaluators.default_loader_workers() if mode == "1" else saved["loader_workers"] == 0)
        runs[mode] = res["metrics"]
    assert runs["1"] == runs["0"], (runs["1"], runs["0"])                              # bit-identical pixels in, same kernels: equal, not close
    # the embeddings themselves, through the function API with the two preprocess objects
    monkeypatch.setenv("KEMR_GPU_PREPROCESS", "1")
    clip_api.allow_random_weights(True)
    tokenizer.allow_hash_tokenizer(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, pre = clip_api.load("ViT-B/32", device="cuda")
    assert pre.defer_to_gpu and tuple(pre(__import__("PIL.Image").Image.new("RGB", (300, 200))).shape) == (3, 224, 224)      # still the host transform when called
    split = datasets.SyntheticHFSplit(40, 3)
    gpu = evaluators.encode_dataset(model, datasets.CLIPEvalDatasetHF(split, pre), 16, 3, 2)
    from knowledge_enhanced_multimodal_retrieval_amd.preprocess import ClipPreprocess
    host = evaluators.encode_dataset(model, datasets.CLIPEvalDatasetHF(split, ClipPreprocess(224)), 16, 3, 0)
    assert torch.equal(gpu[0], host[0]) and torch.equal(gpu[1], host[1]) and gpu[3] == host[3]


def test_config2_fused_scoring_cli(device, tmp_path):
    """BASELINE configs[2] plumbing (scripts/fusion/eval.sh -> evaluator_baseline) on a checkpoint file."""
    from knowledge_enhanced_multimodal_retrieval_amd import clip_api, clip_model
    from src.clip.eval.evaluator_baseline import main
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = clip_api.load("ViT-B/32", device="cpu")
        torch.manual_seed(5)
        with torch.no_grad():
            model.visual.proj.add_(0.01 * torch.randn_like(model.visual.proj))
        ck = tmp_path / "checkpoint_best.pt"
        clip_model.save_checkpoint(model, None, 1, 50.0, 1, str(ck))
        out = tmp_path / "fused.json"
        res = main(["--model_name", "ViT-B/32", "--checkpoint", str(ck), "--split", "test", "--splits_file", "s.json",
                    "--batch_size", "64", "--device", "cuda", "--output_file", str(out), "--t2i_weight", "0.5",
                    "--t2t_weight", "0.5", "--synthetic", "96"])
    assert set(res["metrics"]) == {"R@1", "R@5", "R@10", "R@20", "MRR", "Mean_Rank"}
    assert json.loads(out.read_text())["checkpoint"] == str(ck)


@pytest.mark.parametrize("ft", ["gated", "linear", "cross_attention"])
def test_evaluate_fusion_model_end_to_end(device, tmp_path, ft):
    """SURVEY 8 row a13 (reference eval/evaluator_fusion.py:28-144, main :147-235): encode -> fused scoring under a learned
    head -> Recall@K / MRR, through the drop-in module path and its CLI, on synthetic data.  The metrics must equal what the
    head's own dense score matrix gives under the reference's ranking rule (the reference fills that matrix in 50 x 500
    blocks; here it never exists for the gated family)."""
    from src.clip.eval import evaluator_fusion as EF
    from knowledge_enhanced_multimodal_retrieval_amd import ranking
    from knowledge_enhanced_multimodal_retrieval_amd.datasets import SyntheticRetrievalDataset
    from knowledge_enhanced_multimodal_retrieval_amd.evaluators import encode_dataset
    out = tmp_path / f"fusion_{ft}.json"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = EF.main(["--model_name", "ViT-B/32", "--fusion_type", ft, "--batch_size", "32", "--device", "cuda",
                       "--synthetic", "80", "--output_file", str(out)])
    saved = json.loads(out.read_text())
    assert saved["fusion_type"] == ft and saved["num_samples"] == 80 and saved["weights_source"].startswith("random")
    assert set(res["metrics"]) == {"R@1", "R@5", "R@10", "R@20", "MRR", "Mean_Rank"}
    # the same numbers from the dense matrix of the same head (seeded construction: rebuild model and head identically)
    from knowledge_enhanced_multimodal_retrieval_amd import clip_model
    from src.clip.models import FusionModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cm, _ = clip_model.load_clip_model("ViT-B/32", None, "cuda")
    fm = FusionModel(clip_model=cm, fusion_type=ft, embed_dim=512).to("cuda").eval()
    with torch.no_grad():                    # a freshly initialised head scores every pair almost alike (cross_attention: 78 of 80 queries
        gh = torch.Generator().manual_seed(31)      # with a competitor within 1e-5 of the ground truth): perturb it, seeded, as the golden heads are
        for p_ in fm.fusion_head.parameters():
            p_.add_((torch.randn(p_.shape, generator=gh) * 0.2).to(p_.device))
    ds = SyntheticRetrievalDataset(80, cm.arch.image_size)
    res2 = EF.evaluate_fusion_model(fm, ds, 32, "cuda")
    image, query, target, _ = encode_dataset(cm, ds, 32, 42, 0, None)
    S = fm(query, image, target).double().cpu().numpy()
    want = metrics_ref.retrieval_metrics_from_similarity(S)
    # ALWAYS asserted.  linear / cross_attention: rank() ranks the very matrix forward() returns (dense head,
    # then kemr_rank_dense), so the metrics EQUAL those of that fp32 matrix under the path's order rule (score desc, index asc),
    # near-ties or not -- a cross_attention head scores all 80 candidates of a query within 1e-5 of one another on these inputs.
    # The gated family never materialises the matrix (fused kernel pass, fp32x3 products): there a query is ambiguous only if another
    # candidate scores within 1e-5 of its ground truth; at most four of the 80 may be, each can move one Recall@K by 100 / 80
    # points and Mean_Rank by its number of near-ties / 80 -- with none ambiguous the metrics are EQUAL.
    d = np.abs(S - np.diag(S)[:, None])
    np.fill_diagonal(d, np.inf)
    amb = 0 if ft in ("linear", "cross_attention") else int((d.min(axis=1) <= 1e-5).sum())
    near = 0 if ft in ("linear", "cross_attention") else int((d <= 1e-5).sum())
    assert amb <= 4, amb
    for key in ("R@1", "R@5", "R@10", "R@20"):
        assert abs(res2[key] - want[key]) <= 100.0 * amb / 80 + 1e-9, (key, res2[key], want[key], amb)
    assert abs(res2["Mean_Rank"] - want["Mean_Rank"]) <= near / 80 + 1e-9 and abs(res2["MRR"] - want["MRR"]) <= 100.0 * amb / 80 + 1e-9
    assert 1.0 <= res2["Mean_Rank"] <= 80.0


def test_writes_through_dot_data_need_refresh_or_the_digest(device, monkeypatch):
    """`p.data.copy_()` / `p.data.mul_()` do not bump the parameter's version counter, so the cheap fingerprint cannot
    see them.  Documented behaviour: stale until `refresh()`; with KEMR_WEIGHT_DIGEST=1 the content digest catches them."""
    from knowledge_enhanced_multimodal_retrieval_amd import clip_api
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = clip_api.load("ViT-B/32", device="cuda")
    px = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
    monkeypatch.delenv("KEMR_WEIGHT_DIGEST", raising=False)
    a = model.encode_image(px).clone()
    v0 = model.visual.proj._version
    model.visual.proj.data.mul_(2.0)
    assert model.visual.proj._version == v0                          # the blind spot itself
    assert torch.equal(model.encode_image(px), a)                    # stale packed weights, as documented ...
    model.refresh()
    b = model.encode_image(px)
    assert torch.allclose(b, 2.0 * a, rtol=2e-2, atol=1e-3)          # ... until refresh()
    monkeypatch.setenv("KEMR_WEIGHT_DIGEST", "1")
    model.encode_image(px)                                           # first call with the digest: re-packs once (fingerprint changed shape)
    model.visual.proj.data.copy_(model.visual.proj.data * 0.5)
    c = model.encode_image(px)
    assert torch.allclose(c, a, rtol=2e-2, atol=1e-3) and not torch.allclose(c, b)


def test_module_repacks_after_in_place_weight_edits_and_refuses_nan(device):
    """(a) the packed engine copy follows in-place parameter edits that autograd's version counters see
    (optimizer.step(), p.mul_() under no_grad) without a manual refresh(), and a deepcopy never shares the raw library
    handle; (b) non-finite scores are refused instead of ranking first.  (Writes through p.data: the test above.)"""
    import copy
    from knowledge_enhanced_multimodal_retrieval_amd import clip_api, ranking
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = clip_api.load("ViT-B/32", device="cuda")
    px = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
    a = model.encode_image(px).clone()
    with torch.no_grad():
        model.visual.proj.mul_(2.0)                      # what an optimizer step does: in place, same storage
    b = model.encode_image(px)
    assert torch.allclose(b, 2.0 * a, rtol=2e-2, atol=1e-3) and not torch.allclose(b, a)
    twin = copy.deepcopy(model)
    assert twin._engine is None
    assert torch.equal(twin.encode_image(px), b) and twin._engine is not model._engine
    # NaN guard
    q = torch.nn.functional.normalize(torch.randn(8, 64), dim=-1)
    g = torch.nn.functional.normalize(torch.randn(20, 64), dim=-1)
    ranking.ranks_and_topk([q], [g], k=3)                # finite: fine
    q[2, 5] = float("nan")
    with pytest.raises(ValueError, match="non-finite"):
        ranking.ranks_and_topk([q], [g], k=3)
    S = (q @ g.t()).numpy()
    with pytest.raises(ValueError, match="non-finite"):
        ranking.ranks_of_matrix(S)


@pytest.mark.parametrize("ft", ["linear", "gated", "simple_gated", "simple_gated_with_bias", "bilinear", "cross_attention"])
def test_fusion_heads_match_reference_golden(device, golden_dir, ft):
    from src.clip.models import FusionModel
    z = np.load(os.path.join(golden_dir, "fusion_heads.npz"))

    class NoClip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

    fm = FusionModel(NoClip(), fusion_type=ft, embed_dim=64)
    sd = {k.split("__sd__")[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{ft}__sd__")}
    fm.fusion_head.load_state_dict(sd, strict=True)
    fm = fm.to(device).eval()
    out = fm(z["q"], z["img"], z["tgt"]).cpu().numpy()
    np.testing.assert_allclose(out, z[f"{ft}__out"], rtol=1e-4, atol=5e-6)
    gt = np.arange(12) % 20
    ranks, top_s, top_i = fm.rank(z["q"], z["img"], z["tgt"], k=3, gt_idx=gt)
    want = z[f"{ft}__out"].astype(np.float64)
    d = np.abs(want - want[np.arange(12), gt][:, None])
    d[np.arange(12), gt] = np.inf
    clear = d.min(axis=1) > 1e-5
    assert np.array_equal(ranks.cpu().numpy()[clear], metrics_ref.ranks_by_count(want, gt)[clear])
    assert np.array_equal(top_i.cpu().numpy()[:, 0], want.argmax(axis=1))


def test_cross_attention_head_at_clip_width(device):
    """D = 768 (head dim 96, not a multiple of 64) against the float64 oracle restatement of the reference head."""
    from src.clip.models import FusionModel
    g = torch.Generator().manual_seed(3)
    D, N, M = 768, 37, 53
    fm = FusionModel(torch.nn.Linear(1, 1), fusion_type="cross_attention", embed_dim=D)
    with torch.no_grad():
        for p_ in fm.fusion_head.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.05 if p_.dim() > 1 else 0.1))
    q = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=-1).numpy()
    im = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1).numpy()
    tg = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1).numpy()
    sd = {k: v.numpy() for k, v in fm.fusion_head.state_dict().items()}
    want = fusion_ref.head_scores("cross_attention", sd, q, im, tg)
    got = fm.to(device)(q, im, tg).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-5)
    with pytest.raises(ValueError):
        FusionModel(torch.nn.Linear(1, 1), fusion_type="bogus")


def test_metrics_module_matches_reference_golden(device, golden_dir):
    from src.clip.eval import fusion as F
    from src.clip.eval import metrics as M
    z = np.load(os.path.join(golden_dir, "metrics_n192_d128.npz"))
    ref = json.loads(bytes(z["metrics_json"]).decode())
    img, q, t = z["image"], z["query"], z["target"]
    assert M.compute_all_retrieval_metrics(q, t, img) == pytest.approx(ref["all"], abs=1e-9)
    assert M.compute_retrieval_metrics_final(q, t, img) == pytest.approx(ref["final_0.5_0.5"], abs=1e-9)
    assert M.compute_retrieval_metrics_final(q, t, img, prefix="F", t2i_weight=0.1, t2t_weight=0.9) == \
        pytest.approx(ref["final_0.1_0.9_prefixF"], abs=1e-9)
    assert M.compute_training_metrics(q, t, img) == pytest.approx(ref["training"], abs=1e-9)
    S = q @ img.T
    assert M.compute_retrieval_metrics_fusion(0.5 * (q @ img.T) + 0.5 * (q @ t.T), prefix="X") == \
        pytest.approx(ref["fusion_prefixX"], abs=1e-9)
    assert F.evaluate_retrieval(S) == pytest.approx(ref["evaluate_retrieval_t2i"], abs=1e-9)
    assert {**M.compute_recall_at_k(S), **M.compute_mrr_and_mean_rank(S)} == pytest.approx(ref["evaluate_retrieval_t2i"], abs=1e-9)
    # device tensors are accepted as well as numpy
    tq, ti = torch.from_numpy(q).to(device), torch.from_numpy(img).to(device)
    assert M.compute_retrieval_metrics(tq, ti, "T2I") == pytest.approx({k: v for k, v in ref["all"].items() if k.startswith("T2I")}, abs=1e-9)
    assert M.compute_metrics_multi_mode(img, [t])["T2T_R@1"] == 100.0            # deprecated self-retrieval shim


def test_sparql_fused_metrics_match_dense_reference(device, golden_dir):
    from src.clip.eval import fusion as F
    z = np.load(os.path.join(golden_dir, "sparql_fusion.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    uu, res = meta["uuids"], meta["results"]
    img, q, t = metrics_ref.planted_embeddings(len(uu), 128, seed=2)
    for strategy, params in (("weighted", {"alpha": 0.7, "sparql_weight": 0.3}), ("additive", {"delta": 0.5}),
                             ("adaptive", {"delta": 0.5})):
        got = F.fused_metrics([q, q], [img, t], [0.5, 0.5], res, uu, uu, strategy, params)
        S = 0.5 * (q.astype(np.float64) @ img.astype(np.float64).T) + 0.5 * (q.astype(np.float64) @ t.astype(np.float64).T)
        want = metrics_ref.retrieval_metrics_from_similarity(fusion_ref.fuse(S, res, uu, uu, strategy, params))
        assert got == pytest.approx(want, abs=1e-9), strategy


def test_retriever_store_and_engine(device, tmp_path):
    from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
    from knowledge_enhanced_multimodal_retrieval_amd.clip_module import CLIP
    from knowledge_enhanced_multimodal_retrieval_amd.retriever import CLIPRetriever, EmbeddingStore
    from src.clip.clip_retrieval import CLIPRetrieval
    from src.retrieval import RetrievalEngine
    arch, oa = ARCHS["tiny"], clip_ref.ARCHS["tiny"]
    sd = clip_ref.random_state_dict(oa, seed=0)
    model = CLIP(arch)
    model.load_state_dict(sd)
    model = model.to(device).eval()
    n = 300
    img, _, txt = metrics_ref.planted_embeddings(n, arch.embed_dim, seed=1)
    uuids = [f"u{i:04d}" for i in range(n)]
    store = EmbeddingStore(img, txt, uuids, device)
    store.save(str(tmp_path / "emb"))
    store2 = EmbeddingStore.load(str(tmp_path / "emb"), device)
    assert len(store2) == n and store2.dim == arch.embed_dim and torch.equal(store2.image, store.image)

    words = {}

    def tok(texts):                                       # tiny vocab: a fixed toy tokenizer
        out = torch.zeros(len(texts), arch.ctx, dtype=torch.int32)
        for r, s in enumerate(texts):
            ids = [arch.sot] + [1 + words.setdefault(w, len(words)) % (arch.sot - 1) for w in s.split()][:arch.ctx - 2] + [arch.eot]
            out[r, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out

    ret = CLIPRetriever(model, store2, tokenize_fn=tok)
    query = "bronze statue of a seated king"
    hits = ret.search(query, alpha=0.3, top_k=7)
    qe = clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, tok([query]))).numpy().astype(np.float64)
    S = 0.3 * (qe @ img.astype(np.float64).T) + 0.7 * (qe @ txt.astype(np.float64).T)
    order = np.argsort(-S[0], kind="stable")[:7]
    assert [h["uuid"] for h in hits][:3] == [uuids[i] for i in order[:3]] or abs(S[0, order[2]] - S[0, order[3]]) < 2e-3
    assert all(abs(h["score"] - S[0, uuids.index(h["uuid"])]) < 2e-3 for h in hits)       # bf16 encoder vs fp32 oracle
    assert [h["score"] for h in hits] == sorted((h["score"] for h in hits), reverse=True)

    class T2S:
        def retrieval(self, q):
            return [hits[5]["uuid"], "unknown"]

    eng = RetrievalEngine(clip_retriever=CLIPRetrieval(retriever=ret), t2s_retriever=T2S())
    fused = eng.retrieve_text(query, alpha=0.8, beta=0.2, alpha_clip=0.3, threshold=-1)
    assert fused[0]["uuid"] == hits[5]["uuid"] and fused[0]["score"] == round(0.8 * hits[5]["score"] + 0.2, 4)
    assert len(eng.retrieve_text_noknowledge(query, alpha_clip=0.3, threshold=-1)) == 10
    with pytest.raises(ValueError):
        ret.search(query, top_k=100)


def test_sharded_gallery_single_process(device):
    """dist.ShardedGallery at world_size 1 + hand-simulated shards == single-gallery answer (config 4's data path)."""
    from knowledge_enhanced_multimodal_retrieval_amd import _lib, engine
    from knowledge_enhanced_multimodal_retrieval_amd.dist import ShardedGallery, shard_bounds
    n, d, nq, k = 1000, 128, 96, 10
    img, q, t = metrics_ref.planted_embeddings(n, d, seed=4)
    ti, tq = torch.from_numpy(img).to(device), torch.from_numpy(q[:nq]).to(device)
    gal = ShardedGallery([ti], n)
    assert (gal.lo, gal.hi, gal.world) == (0, n, 1)
    s, i = gal.search([tq], k=k)
    ranks, s2, i2 = gal.ranks([tq], torch.arange(nq), k=k)
    assert torch.equal(i, i2) and torch.equal(s, s2)
    S = q[:nq].astype(np.float64) @ img.astype(np.float64).T
    assert np.array_equal(i.cpu().numpy(), metrics_ref.topk(S, k)[1])
    assert np.array_equal(ranks.cpu().numpy(), metrics_ref.ranks_by_count(S))
    parts_s, parts_i = [], []
    qp = engine.build_panel([tq], _lib.SIDE_QUERY, 3)
    for r in range(8):
        lo, hi = shard_bounds(n, 8, r)
        ps, pi = engine.sim_topk(qp, engine.build_panel([ti[lo:hi]], _lib.SIDE_GALLERY, 3), k, lo)
        parts_s.append(ps)
        parts_i.append(pi)
    ms, mi = engine.topk_merge(torch.stack(parts_s, 1), torch.stack(parts_i, 1), k)
    assert torch.equal(mi, i) and torch.equal(ms, s)


@pytest.mark.parametrize("ft", ["gated", "simple_gated", "simple_gated_with_bias"])
def test_gate_of_the_gated_heads_through_the_library(device, ft):
    """FusionModel._gate (fp32x3 dense kernel for Linear(d, 128) + kemr_gate_rows) against the head's own torch gate() in fp64 on the
    CPU, with non-trivial seeded parameters, d = 768 and a row count that is not a multiple of the kernel's four rows per workgroup."""
    from src.clip.models import FusionModel

    class NoClip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

    torch.manual_seed(5)
    fm = FusionModel(NoClip(), fusion_type=ft, embed_dim=768)
    with torch.no_grad():
        for p in fm.fusion_head.parameters():
            p.copy_(torch.randn_like(p) * (0.2 if p.dim() > 1 else 0.5))
    q = torch.nn.functional.normalize(torch.randn(1001, 768), dim=-1)
    want = fm.fusion_head.double().eval().gate(q.double()).reshape(-1)
    fm.fusion_head.float()
    got = fm.to(device)._gate(q.to(device)).cpu().double()
    assert got.shape == (1001,) and float((got - want).abs().max()) < 2e-6, float((got - want).abs().max())
    assert float(got.min()) > 0 and float(got.max()) < 1 and float(got.std()) > 1e-3       # a gate, and not a constant one
