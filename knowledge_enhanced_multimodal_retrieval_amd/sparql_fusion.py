"""CLIP score (+) Text2SPARQL-hit fusion.

Two layers:

* the reference's dense API (same names as /root/reference/src/clip/eval/fusion.py: ``weighted_fusion`` :22-85,
  ``additive_bonus_fusion`` :88-132, ``adaptive_additive_fusion`` :135-206, ``fuse_clip_and_text2sparql`` :209-275,
  ``evaluate_retrieval`` :6-20).  These take and return a dense (N_queries, N_artefacts) matrix exactly like the
  reference -- callers that already hold such a matrix keep working; the hits are applied as a sparse scatter
  (no python double loop over a dense 0/1 matrix) and ``evaluate_retrieval`` ranks the matrix on the GPU;
* the fused path the build's evaluators use: the hits become a CSR list of additive bonuses
  (``sparql_bonus``) that the similarity kernel applies while it scans each score tile, so neither the CLIP matrix nor
  the 0/1 hit matrix ever exists (``fused_metrics``).

A hit is addressed by the tail of the URI after the last ``/`` (reference fusion.py:75).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import ranking

DEFAULT_SIZE_THRESHOLDS = {1: 1.0, 5: 0.8, 20: 0.5, 50: 0.3, float("inf"): 0.1}


def _uri_tail(uri: str) -> str:
    return uri.split("/")[-1] if "/" in uri else uri


def _omega(size: int, thresholds) -> float:
    if size == 0:
        return 0.0
    for thr, w in sorted(thresholds.items()):
        if size <= thr:
            return w
    return 0.0


def hit_list(text2sparql_results: Dict[str, List[str]], query_uuids: Sequence[str], artefact_uuids: Sequence[str]
             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(rows, cols, result-set size per entry) of every listed URI that names a known artefact, in listing order."""
    col_of = {u: i for i, u in enumerate(artefact_uuids)}
    rows, cols, sizes = [], [], []
    for r, qu in enumerate(query_uuids):
        hits = text2sparql_results.get(qu, [])
        n = len(hits)
        for uri in hits:
            c = col_of.get(_uri_tail(uri))
            if c is not None:
                rows.append(r)
                cols.append(c)
                sizes.append(n)
    return np.asarray(rows, np.int64), np.asarray(cols, np.int64), np.asarray(sizes, np.int64)


def sparql_bonus(text2sparql_results, query_uuids, artefact_uuids, strategy: str = "weighted",
                 params: Optional[dict] = None) -> Tuple[float, Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """-> (clip_scale, (rowptr int32 [Q+1], col int32, val fp32)): final score = clip_scale * S + sum of a pair's values.

    weighted: clip_scale = alpha, every DISTINCT hit adds sparql_weight (the reference SETS the 0/1 matrix entry);
    additive: every LISTED hit adds delta (duplicates add twice, as the reference's ``+=``);
    adaptive: every listed hit adds delta * omega(len(result list)).
    """
    params = params or {}
    rows, cols, sizes = hit_list(text2sparql_results, query_uuids, artefact_uuids)
    scale = 1.0
    if strategy == "weighted":
        alpha, w = params.get("alpha", 0.7), params.get("sparql_weight", 0.3)
        if not np.isclose(alpha + w, 1.0):
            print(f"Warning: alpha ({alpha}) + sparql_weight ({w}) != 1.0, normalizing...")
            alpha, w = alpha / (alpha + w), w / (alpha + w)
        scale = float(alpha)
        if len(rows):
            pairs = np.unique(np.stack([rows, cols], 1), axis=0)
            rows, cols = pairs[:, 0], pairs[:, 1]
        vals = np.full(len(rows), w, np.float32)
    elif strategy == "additive":
        vals = np.full(len(rows), params.get("delta", 0.5), np.float32)
    elif strategy == "adaptive":
        thr = params.get("size_thresholds") or DEFAULT_SIZE_THRESHOLDS
        delta = params.get("delta", 0.5)
        vals = np.asarray([delta * _omega(int(s), thr) for s in sizes], np.float32)
    else:
        raise ValueError(f"Unknown fusion strategy: {strategy}")
    order = np.lexsort((cols, rows)) if len(rows) else np.zeros(0, np.int64)
    rows, cols, vals = rows[order], cols[order], vals[order]
    ptr = np.zeros(len(query_uuids) + 1, np.int64)
    np.add.at(ptr, rows + 1, 1)
    return scale, (np.cumsum(ptr).astype(np.int32), cols.astype(np.int32), vals.astype(np.float32))


def fused_ranks(query_parts, gallery_parts, weights, text2sparql_results, query_uuids, artefact_uuids,
                strategy: str = "weighted", params: Optional[dict] = None, k: int = 10, precision: str = "fp32x3"):
    """Ranks / top-k of  clip_scale * sum_p w_p <q_p, g_p>  + SPARQL bonus, fused in one kernel pass."""
    scale, bonus = sparql_bonus(text2sparql_results, query_uuids, artefact_uuids, strategy, params)
    weights = [scale * w for w in (weights if weights is not None else [1.0] * len(query_parts))]
    return ranking.ranks_and_topk(query_parts, gallery_parts, weights=weights, k=k, precision=precision, bonus=bonus)


def fused_metrics(query_parts, gallery_parts, weights, text2sparql_results, query_uuids, artefact_uuids,
                  strategy: str = "weighted", params: Optional[dict] = None) -> Dict[str, float]:
    ranks, _, _ = fused_ranks(query_parts, gallery_parts, weights, text2sparql_results, query_uuids, artefact_uuids,
                              strategy, params, k=0)
    return ranking.metrics_from_ranks(ranks)


# ------------------------------------------------------------------------------------------------ reference dense API
def evaluate_retrieval(similarity_matrix) -> Dict[str, float]:
    ranks, _, _ = ranking.ranks_of_matrix(similarity_matrix)
    metrics = ranking.metrics_from_ranks(ranks)
    print("evaluate_retrieval:", metrics)
    return metrics


def _check(S, query_uuids, artefact_uuids):
    assert S.shape[0] == len(query_uuids), \
        f"Similarity matrix rows ({S.shape[0]}) != query_uuids length ({len(query_uuids)})"
    assert S.shape[1] == len(artefact_uuids), \
        f"Similarity matrix cols ({S.shape[1]}) != artefact_uuids length ({len(artefact_uuids)})"


def _dense(S: np.ndarray, scale: float, bonus) -> np.ndarray:
    ptr, col, val = bonus
    out = S * S.dtype.type(scale) if scale != 1.0 else S.copy()
    rows = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    np.add.at(out, (rows, col), val.astype(out.dtype))
    return out


def weighted_fusion(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids, alpha: float = 0.7,
                    sparql_weight: float = 0.3) -> np.ndarray:
    S = np.asarray(clip_similarity_matrix)
    _check(S, query_uuids, artefact_uuids)
    scale, bonus = sparql_bonus(text2sparql_results, query_uuids, artefact_uuids, "weighted",
                                {"alpha": alpha, "sparql_weight": sparql_weight})
    return _dense(S, scale, bonus)


def additive_bonus_fusion(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids,
                          delta: float = 0.5) -> np.ndarray:
    S = np.asarray(clip_similarity_matrix)
    _check(S, query_uuids, artefact_uuids)
    return _dense(S, 1.0, sparql_bonus(text2sparql_results, query_uuids, artefact_uuids, "additive", {"delta": delta})[1])


def adaptive_additive_fusion(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids,
                             delta: float = 0.5, size_thresholds: Dict[str, float] = None) -> np.ndarray:
    S = np.asarray(clip_similarity_matrix)
    _check(S, query_uuids, artefact_uuids)
    return _dense(S, 1.0, sparql_bonus(text2sparql_results, query_uuids, artefact_uuids, "adaptive",
                                       {"delta": delta, "size_thresholds": size_thresholds})[1])


def fuse_clip_and_text2sparql(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids,
                              fusion_strategy: str = "weighted", fusion_params: Dict = None) -> np.ndarray:
    fusion_params = fusion_params or {}
    if fusion_strategy == "weighted":
        return weighted_fusion(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids,
                               alpha=fusion_params.get("alpha", 0.7), sparql_weight=fusion_params.get("sparql_weight", 0.3))
    if fusion_strategy == "additive":
        return additive_bonus_fusion(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids,
                                     delta=fusion_params.get("delta", 0.5))
    if fusion_strategy == "adaptive":
        return adaptive_additive_fusion(clip_similarity_matrix, text2sparql_results, query_uuids, artefact_uuids,
                                        delta=fusion_params.get("delta", 0.5),
                                        size_thresholds=fusion_params.get("size_thresholds", None))
    raise ValueError(f"Unknown fusion strategy: {fusion_strategy}")
