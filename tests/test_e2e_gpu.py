"""End-to-end parity of the whole hot path against the oracle's encoders composed with the REFERENCE's metrics.py
(VERDICT r2 "missing #1"; north_star: "outputs (embeddings, ranked ids, Recall@K) match the reference CPU path within 1e-3 cosine
and identical top-k sets").

tests/golden/e2e_<arch>_n<N>.npz was written in the build container by tests/golden/make_golden.py: seeded inputs -> oracle
encoders (fp32 CPU) -> the imported reference ``src/clip/eval/metrics.py`` (evaluator.py:121-156 -> metrics.py:34-41, 62-68) ->
metric dicts, the oracle's top-11 ids / scores and the ground truth's rank, per task.  Here:

  CPU (not gpu): the oracle re-run on the first items reproduces the fixture's embeddings (pins the restatement to the fixture).
  GPU: HIP encoders -> HIP fused similarity / top-k / rank (the product path, default precision) on the same inputs against
       (a) the oracle's embeddings, re-computed here and checked against the fixture's checksums: cosine >= 1 - 1e-3, every item;
       (b) the fixture's top-10 sets and ranks under the margin rule of SURVEY section 8(c): a candidate may change sides of the
           top-10 boundary, or of the ground truth, only if the ORACLE scores it within E = 2e-3 of that boundary;
       (c) the reference's own Recall@K / Mean_Rank numbers, within what (b) allows query by query -- and EQUAL where it allows
           nothing.
Random-weight towers put every embedding in a narrow cone (score spread 2e-3 .. 5e-2), so the text tasks have hardly any margin
at E = 2e-3 and (b) binds little there; the I2I tasks (a noisy copy of a gallery image as the query) have real ground-truth
margins and Recall@K between 17 and 100 %, which is where a wrong embedding shows.  The tighter E_T = 2e-4 rows are the
sensitive ones: FIXED bars, set from the first measurement of the default precision with the headroom noted beside them.
"""
import json
import os

import numpy as np
import pytest
import torch

from knowledge_enhanced_multimodal_retrieval_amd import _lib, engine, ranking
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
from oracle import clip_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
E_HARD = 2e-3        # top-k sets equal where the oracle's k / (k+1) margin exceeds this
E_TIGHT = 2e-4       # the bar that binds on these inputs (typical 10 / 11 margins are 2e-5 .. 8e-4)
CASES = [("ViT-B/32", 256), ("ViT-L/14", 64)]


def _load(name, n):
    z = np.load(os.path.join(GOLDEN, "e2e_%s_n%d.npz" % (name.replace("/", "-"), n)))
    return z, json.loads(bytes(z["meta_json"]).decode())


def _inputs(arch, n, levels):
    g = torch.Generator().manual_seed(20261004)
    px = torch.randn(n, 3, arch["image_size"], arch["image_size"], generator=g)
    nz = torch.randn(n, 3, arch["image_size"], arch["image_size"], generator=g)
    return px, {lvl: px + lvl * nz for lvl in levels}, clip_ref.synthetic_ids(arch, n, seed=777), clip_ref.synthetic_ids(arch, n, seed=778)


def _tasks(emb, levels):
    t = {"T2I": (emb["query"], [(1.0, emb["image"])]), "T2T": (emb["query"], [(1.0, emb["target"])]),
         "FUSED": (emb["query"], [(0.5, emb["image"]), (0.5, emb["target"])])}
    for lvl in levels:
        t[f"I2I@{lvl}"] = (emb[f"noisy{lvl}"], [(1.0, emb["image"])])
    return t


def test_fixture_is_what_the_oracle_produces():
    """CPU: oracle/clip_ref on the regenerated inputs reproduces the embeddings stored with the fixture (first 8 items of every
    set, ViT-B/32), and the input checksums match -- the GPU test's margins come from the same oracle."""
    z, meta = _load("ViT-B/32", 256)
    arch = clip_ref.ARCHS["ViT-B/32"]
    sd = clip_ref.random_state_dict(arch, seed=0)
    px, noisy, q_ids, t_ids = _inputs(arch, 256, meta["levels"])
    assert abs(float(px.double().abs().sum()) - meta["input_abs_sums"]["pixels"]) < 1e-6 * meta["input_abs_sums"]["pixels"]
    assert int(q_ids.long().sum()) == meta["input_abs_sums"]["query_ids"] and int(t_ids.long().sum()) == meta["input_abs_sums"]["target_ids"]
    with torch.no_grad():
        got = {"image": clip_ref.l2_normalize(clip_ref.encode_image(sd, arch, px[:8])),
               "query": clip_ref.l2_normalize(clip_ref.encode_text(sd, arch, q_ids[:8])),
               "target": clip_ref.l2_normalize(clip_ref.encode_text(sd, arch, t_ids[:8]))}
        lvl = meta["levels"][0]
        got[f"noisy{lvl}"] = clip_ref.l2_normalize(clip_ref.encode_image(sd, arch, noisy[lvl][:8]))
    for k, v in got.items():
        assert float((v - torch.from_numpy(z["first8_" + k])).abs().max()) < 2e-6, k
    for task, m in meta["metrics"].items():
        p = task.split("@")[0]
        assert sorted(m) == sorted(f"{p}_{s}" for s in ("R@1", "R@5", "R@10", "R@20", "MRR", "Mean_Rank")), task     # the reference's key names
        assert z[f"{task}_top11_ids"].shape == (256, 11) and z[f"{task}_ranks"].min() >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", CASES)
def test_hip_path_end_to_end_against_oracle_and_reference_metrics(device, name, n):
    z, meta = _load(name, n)
    oa, levels = clip_ref.ARCHS[name], meta["levels"]
    sd = clip_ref.random_state_dict(oa, seed=0)
    px, noisy, q_ids, t_ids = _inputs(oa, n, levels)
    # ---- the oracle, re-run on this machine and pinned to the fixture
    with torch.no_grad():
        oe = {"image": clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, px)).numpy(),
              "query": clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, q_ids)).numpy(),
              "target": clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, t_ids)).numpy()}
        for lvl in levels:
            oe[f"noisy{lvl}"] = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, noisy[lvl])).numpy()
    for k, v in oe.items():
        want = meta["embedding_abs_sums"][k]
        assert abs(float(np.abs(v.astype(np.float64)).sum()) - want) < 2e-5 * want, k          # same oracle as the fixture's (thread count moves fp32 sums by 1e-7)
        assert float(np.abs(v[:8] - z["first8_" + k]).max()) < 5e-6, k
    # ---- the product path: HIP encoders (default precision), embeddings stay in HBM
    assert _lib.DEFAULT_PRECISION == "bf16"
    eng = engine.ClipEngine(ARCHS[name], device)
    eng.load_state_dict(sd)
    he = {"image": eng.encode_image(px.to(device), normalize=True), "query": eng.encode_text(q_ids.to(device), normalize=True),
          "target": eng.encode_text(t_ids.to(device), normalize=True)}
    for lvl in levels:
        he[f"noisy{lvl}"] = eng.encode_image(noisy[lvl].to(device), normalize=True)
    worst = 0.0
    for k in oe:
        c = torch.nn.functional.cosine_similarity(he[k].cpu().double(), torch.from_numpy(oe[k]).double())
        worst = max(worst, float((1 - c).max()))
        assert float((1 - c).max()) < 1e-3, (k, float((1 - c).max()))                           # north_star: within 1e-3 cosine, EVERY item
    print(f"{name} N={n}: worst 1 - cos over {len(oe) * n} embeddings {worst:.2e}")
    # ---- HIP ranking of the HIP embeddings against the oracle's scores and the reference's metrics
    ids = np.arange(n)
    for task, (oq, oparts) in _tasks(oe, levels).items():
        prefix = task.split("@")[0]
        S = sum(w * (oq @ c.T) for w, c in oparts).astype(np.float32)                          # oracle scores (the reference's expression)
        order = np.argsort(-S, axis=1, kind="stable")
        o_ranks = np.argmax(order == ids[:, None], axis=1) + 1
        stable = (z[f"{task}_top11_scores"][:, 9] - z[f"{task}_top11_scores"][:, 10]) > 1e-6
        assert np.array_equal(order[stable, :10], z[f"{task}_top11_ids"][stable, :10]), task       # live oracle == fixture
        assert np.array_equal(o_ranks, z[f"{task}_ranks"]) or float(np.abs(o_ranks - z[f"{task}_ranks"]).max()) <= 1, task
        hq, hparts = _tasks(he, levels)[task]
        ranks, top_s, top_i = ranking.ranks_and_topk([hq] * len(hparts), [c for _, c in hparts], weights=[w for w, _ in hparts], k=10)
        h_ranks, h_top = ranks.cpu().numpy(), top_i.cpu().numpy().astype(np.int64)
        s10 = np.take_along_axis(S, order[:, 9:10], axis=1)[:, 0]
        sgt = S[ids, ids]
        report = {}
        for E, hard in ((E_HARD, True), (E_TIGHT, False)):
            # (b) top-10: every id the HIP path returns scores >= s10 - E for the oracle, every id the oracle scores > s10 + E is returned
            got_scores = np.take_along_axis(S, h_top, axis=1)
            bad_in = (got_scores < (s10 - E)[:, None]).any(axis=1)
            must = S > (s10 + E)[:, None]
            have = np.zeros_like(must)
            np.put_along_axis(have, h_top, True, axis=1)
            bad_out = (must & ~have).any(axis=1)
            # ranks: the ground truth moves at most by the candidates the oracle scores within E of it
            near = (np.abs(S - sgt[:, None]) <= E).sum(axis=1) - 1
            bad_rank = np.abs(h_ranks - o_ranks) > near
            report[E] = (int(bad_in.sum() + bad_out.sum()), int(bad_rank.sum()), near)
            if hard:
                assert not bad_in.any() and not bad_out.any() and not bad_rank.any(), (task, report[E][:2])
        same_sets = int(sum(set(h_top[i]) == set(order[i, :10]) for i in range(n)))
        same_ranks = int((h_ranks == o_ranks).sum())
        hm = ranking.metrics_from_ranks(ranks)
        rm = meta["metrics"][task]
        near = report[E_TIGHT][2]
        line = {k: (round(hm[k], 2), round(rm[f"{prefix}_{k}"], 2)) for k in ("R@1", "R@10", "MRR", "Mean_Rank")}
        print(f"{name} {task}: identical top-10 sets {same_sets}/{n}, identical ranks {same_ranks}/{n}, violations at E=2e-4: "
              f"sets {report[E_TIGHT][0]}, ranks {report[E_TIGHT][1]}; (hip, reference) {line}")
        # (c) Recall@K against the REFERENCE's numbers: a query may change sides of K only if the oracle scores a competitor within
        # E_TIGHT of its ground truth; elsewhere the membership -- hence the metric -- is equal
        for K in (1, 5, 10, 20):
            may_cross = ((o_ranks - near <= K) & (o_ranks > K)) | ((o_ranks + near > K) & (o_ranks <= K))
            assert abs(hm[f"R@{K}"] - rm[f"{prefix}_R@{K}"]) <= 100.0 * may_cross.sum() / n + 1e-9, (task, K, hm, rm)
        assert abs(hm["Mean_Rank"] - rm[f"{prefix}_Mean_Rank"]) <= near.sum() / n + 1e-9, (task, hm, rm)
        # FIXED bars at E_TIGHT (round 3, first measurement of the default precision: see the printed line; never widened since)
        assert report[E_TIGHT][0] <= 0.02 * n and report[E_TIGHT][1] <= 0.02 * n, (task, report[E_TIGHT][:2])
        if prefix == "I2I":
            for K in (1, 10):
                assert abs(hm[f"R@{K}"] - rm[f"{prefix}_R@{K}"]) <= 100.0 * 2 / n + 1e-9, (task, K, hm[f"R@{K}"], rm[f"{prefix}_R@{K}"])
