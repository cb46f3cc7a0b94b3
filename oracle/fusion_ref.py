"""numpy restatement of the CLIP-score (+) Text2SPARQL-hit fusion and of the learned fusion heads.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows
* ``/root/reference/src/clip/eval/fusion.py``: ``weighted_fusion`` ``:22-85``
  (``alpha*S + (1-alpha)*1[hit]``, weights renormalised when they do not sum to 1),
  ``additive_bonus_fusion`` ``:88-132`` (``S + delta*1[hit]``),
  ``adaptive_additive_fusion`` ``:135-206`` (``S + delta*omega(|R|)*1[hit]``, omega by result-set
  size thresholds {1:1.0, 5:0.8, 20:0.5, 50:0.3, inf:0.1}), dispatcher ``:209-275``;
  a hit is addressed by the tail of the URI after the last ``/``.
* ``/root/reference/src/retrieval.py:23-76``: online linear fuse
  ``score = round(alpha*clip + beta*[uuid in sparql], 4)``, sorted descending (stable), then
  thresholded at ``:88-95``.
* ``/root/reference/src/clip/model/fusion_model.py``: the heads' forward math in eval mode
  (dropout inactive): linear ``:25-48``, gated ``:136-180``, simple_gated ``:182-196``,
  simple_gated_with_bias ``:9-23``, bilinear ``:198-240``, cross_attention ``:51-133``.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

DEFAULT_SIZE_THRESHOLDS = {1: 1.0, 5: 0.8, 20: 0.5, 50: 0.3, float("inf"): 0.1}


def uri_tail(uri: str) -> str:
    return uri.split("/")[-1] if "/" in uri else uri


def hit_pairs(results: Dict[str, List[str]], query_uuids: Sequence[str], artefact_uuids: Sequence[str]):
    """(rows, cols, per-row result-set size) of every SPARQL hit that names a known artefact."""
    col_of = {u: i for i, u in enumerate(artefact_uuids)}
    rows, cols, sizes = [], [], []
    for r, qu in enumerate(query_uuids):
        hits = results.get(qu, [])
        for uri in hits:
            c = col_of.get(uri_tail(uri))
            if c is not None:
                rows.append(r)
                cols.append(c)
                sizes.append(len(hits))
    return np.asarray(rows, np.int64), np.asarray(cols, np.int64), np.asarray(sizes, np.int64)


def omega_of_size(size: int, thresholds=None) -> float:
    thresholds = DEFAULT_SIZE_THRESHOLDS if thresholds is None else thresholds
    if size == 0:
        return 0.0
    for thr, w in sorted(thresholds.items()):
        if size <= thr:
            return w
    return 0.0


def weighted(S, results, query_uuids, artefact_uuids, alpha=0.7, sparql_weight=0.3):
    assert S.shape == (len(query_uuids), len(artefact_uuids))
    if not np.isclose(alpha + sparql_weight, 1.0):
        tot = alpha + sparql_weight
        alpha, sparql_weight = alpha / tot, sparql_weight / tot
    H = np.zeros_like(S)
    r, c, _ = hit_pairs(results, query_uuids, artefact_uuids)
    H[r, c] = 1.0
    return alpha * S + sparql_weight * H


def additive(S, results, query_uuids, artefact_uuids, delta=0.5):
    assert S.shape == (len(query_uuids), len(artefact_uuids))
    out = S.copy()
    r, c, _ = hit_pairs(results, query_uuids, artefact_uuids)
    # the reference adds delta once per listed URI (duplicates add twice): np.add.at keeps that
    np.add.at(out, (r, c), np.float32(delta) if out.dtype == np.float32 else delta)
    return out


def adaptive(S, results, query_uuids, artefact_uuids, delta=0.5, size_thresholds=None):
    assert S.shape == (len(query_uuids), len(artefact_uuids))
    out = S.copy()
    r, c, sz = hit_pairs(results, query_uuids, artefact_uuids)
    bonus = np.asarray([delta * omega_of_size(int(s), size_thresholds) for s in sz], dtype=out.dtype)
    np.add.at(out, (r, c), bonus)
    return out


def fuse(S, results, query_uuids, artefact_uuids, strategy="weighted", params=None):
    params = params or {}
    if strategy == "weighted":
        return weighted(S, results, query_uuids, artefact_uuids,
                        params.get("alpha", 0.7), params.get("sparql_weight", 0.3))
    if strategy == "additive":
        return additive(S, results, query_uuids, artefact_uuids, params.get("delta", 0.5))
    if strategy == "adaptive":
        return adaptive(S, results, query_uuids, artefact_uuids, params.get("delta", 0.5),
                        params.get("size_thresholds"))
    raise ValueError(f"Unknown fusion strategy: {strategy}")


def engine_linear_fuse(clip_results: List[dict], sparql_results: List[str], alpha=0.8, beta=0.2) -> List[dict]:
    if not clip_results:
        return []
    hits = set(sparql_results)
    fused = [{"uuid": it["uuid"],
              "score": round(alpha * it["score"] + beta * (1.0 if it["uuid"] in hits else 0.0), 4)}
             for it in clip_results]
    fused.sort(key=lambda x: x["score"], reverse=True)
    return fused


# --------------------------------------------------------------------------- fusion heads (eval mode)

def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def head_scores(fusion_type: str, sd: Dict[str, np.ndarray], q, img, tgt) -> np.ndarray:
    """Fused score matrix [Nq, M] of one reference fusion head given its ``fusion_head.*`` state dict
    (keys without the ``fusion_head.`` prefix), all float64 inside for a tight oracle."""
    q = np.asarray(q, np.float64)
    img = np.asarray(img, np.float64)
    tgt = np.asarray(tgt, np.float64)
    sd = {k: np.asarray(v, np.float64) for k, v in sd.items()}
    t2i, t2t = q @ img.T, q @ tgt.T
    if fusion_type == "linear":
        h = np.stack([t2i, t2t], -1) @ sd["fusion.0.weight"].T + sd["fusion.0.bias"]
        h = np.maximum(h, 0.0)
        return (h @ sd["fusion.3.weight"].T + sd["fusion.3.bias"])[..., 0]
    if fusion_type == "gated":
        h = np.maximum(q @ sd["gate_net.0.weight"].T + sd["gate_net.0.bias"], 0.0)
        g = _sigmoid(h @ sd["gate_net.3.weight"].T + sd["gate_net.3.bias"])
        return g * t2i + (1 - g) * t2t
    if fusion_type in ("simple_gated", "simple_gated_with_bias"):
        g = _sigmoid((q * sd["query_weight"]).sum(1, keepdims=True) + sd["bias"])
        return g * t2i + (1 - g) * t2t
    if fusion_type == "bilinear":
        a = _sigmoid(sd["alpha"])
        return a * (q @ (img @ sd["W_image.weight"].T).T) + (1 - a) * (q @ (tgt @ sd["W_target.weight"].T).T)
    if fusion_type == "cross_attention":
        D = q.shape[1]
        H = 8
        hd = D // H
        qp = q @ sd["query_proj.weight"].T + sd["query_proj.bias"]
        ip = img @ sd["image_proj.weight"].T + sd["image_proj.bias"]
        tp = tgt @ sd["target_proj.weight"].T + sd["target_proj.bias"]
        Wi, bi = sd["cross_attn.in_proj_weight"], sd["cross_attn.in_proj_bias"]
        aq = (qp @ Wi[:D].T + bi[:D]).reshape(-1, H, hd)                      # [N,H,hd]
        ki = (ip @ Wi[D:2 * D].T + bi[D:2 * D]).reshape(-1, H, hd)            # [M,H,hd]
        kt = (tp @ Wi[D:2 * D].T + bi[D:2 * D]).reshape(-1, H, hd)
        vi = (ip @ Wi[2 * D:].T + bi[2 * D:]).reshape(-1, H, hd)
        vt = (tp @ Wi[2 * D:].T + bi[2 * D:]).reshape(-1, H, hd)
        si = np.einsum("nhd,mhd->nmh", aq, ki) / np.sqrt(hd)                   # [N,M,H]
        st = np.einsum("nhd,mhd->nmh", aq, kt) / np.sqrt(hd)
        mx = np.maximum(si, st)
        ei, et = np.exp(si - mx), np.exp(st - mx)
        wi, wt = ei / (ei + et), et / (ei + et)
        ctx = wi[..., None] * vi[None] + wt[..., None] * vt[None]              # [N,M,H,hd]
        ctx = ctx.reshape(q.shape[0], img.shape[0], D)
        o = ctx @ sd["cross_attn.out_proj.weight"].T + sd["cross_attn.out_proj.bias"]
        h = np.maximum(o @ sd["score_mlp.0.weight"].T + sd["score_mlp.0.bias"], 0.0)
        h = np.maximum(h @ sd["score_mlp.3.weight"].T + sd["score_mlp.3.bias"], 0.0)
        s = (h @ sd["score_mlp.6.weight"].T + sd["score_mlp.6.bias"])[..., 0]
        return np.tanh(s) * 0.5
    raise ValueError(f"Unknown fusion type: {fusion_type}")
