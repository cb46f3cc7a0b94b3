"""numpy restatement of the reference's similarity + ranking metrics.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``/root/reference/src/clip/eval/metrics.py``:

* similarity ``S = Q @ C.T`` in fp32 (``metrics.py:102``; fused
  ``w_i * Q@I.T + w_t * Q@T.T`` at ``:145-148``),
* Recall@K from a full descending sort of every row, ground truth = the
  diagonal (``:13-44``),
* MRR / Mean_Rank from the 1-based position of the diagonal element in that
  sort (``:47-76``),
* the task routers and key naming ``"{prefix}_{metric}"`` (``:79-116``,
  ``:119-162``, ``:165-185``, ``:188-252``, ``:256-282``) and
  ``evaluate_retrieval`` of ``eval/fusion.py:6-20``.

The reference sorts with ``np.argsort(-S)`` (unstable quicksort): the order of
exactly tied scores is undefined there.  This oracle fixes the rule the build
uses everywhere: higher score first, then lower candidate index (a stable sort
of ``-S``).  ``ranks_by_count`` is the sort-free statement of the same rank,
``1 + #{j : S_ij > S_ii} + #{j < i : S_ij == S_ii}``, that the device kernels
implement.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

K_VALUES = (1, 5, 10, 20)


def similarity(q: np.ndarray, c: np.ndarray) -> np.ndarray:
    return np.asarray(q, np.float32) @ np.asarray(c, np.float32).T


def fused_similarity(q, img, tgt, t2i_weight: float = 0.5, t2t_weight: float = 0.5) -> np.ndarray:
    return (t2i_weight * similarity(q, img)) + (t2t_weight * similarity(q, tgt))


def sorted_indices(S: np.ndarray) -> np.ndarray:
    """Row-wise candidate order: descending score, ties by ascending index."""
    return np.argsort(-S, axis=1, kind="stable")


def ranks_by_sort(S: np.ndarray, gt: np.ndarray | None = None) -> np.ndarray:
    """1-based rank of the ground-truth column in every row (sort formulation)."""
    n = S.shape[0]
    gt = np.arange(n) if gt is None else np.asarray(gt)
    order = sorted_indices(S)
    return np.argmax(order == gt[:, None], axis=1) + 1


def ranks_by_count(S: np.ndarray, gt: np.ndarray | None = None) -> np.ndarray:
    """Same rank without sorting (what the fused device kernel computes)."""
    n = S.shape[0]
    gt = np.arange(n) if gt is None else np.asarray(gt)
    s_gt = S[np.arange(n), gt][:, None]
    cols = np.arange(S.shape[1])[None, :]
    ahead = (S > s_gt) | ((S == s_gt) & (cols < gt[:, None]))
    return ahead.sum(axis=1) + 1


def metrics_from_ranks(ranks: np.ndarray, k_values: Sequence[int] = K_VALUES,
                       compute_recall: bool = True, compute_mrr: bool = True) -> Dict[str, float]:
    out: Dict[str, float] = {}
    ranks = np.asarray(ranks)
    if compute_recall:
        for k in k_values:
            out[f"R@{k}"] = float(np.mean(ranks <= k) * 100.0)
    if compute_mrr:
        out["MRR"] = float(np.mean(1.0 / ranks) * 100.0)
        out["Mean_Rank"] = float(np.mean(ranks))
    return out


def recall_at_k(S: np.ndarray, k_values: Sequence[int] = K_VALUES) -> Dict[str, float]:
    order = sorted_indices(S)
    gt = np.arange(S.shape[0])[:, None]
    return {f"R@{k}": float(np.mean(np.any(order[:, :k] == gt, axis=1)) * 100.0) for k in k_values}


def mrr_and_mean_rank(S: np.ndarray) -> Dict[str, float]:
    pos = ranks_by_sort(S)
    return {"MRR": float(np.mean(1.0 / pos) * 100.0), "Mean_Rank": float(np.mean(pos))}


def topk(S: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    order = sorted_indices(S)[:, :k]
    return np.take_along_axis(S, order, axis=1), order


def _prefixed(prefix: str, d: Dict[str, float]) -> Dict[str, float]:
    return {(f"{prefix}_{k}" if prefix else k): v for k, v in d.items()}


def retrieval_metrics_from_similarity(S, prefix: str = "", k_values=K_VALUES,
                                      compute_recall: bool = True, compute_mrr: bool = True):
    out: Dict[str, float] = {}
    if compute_recall:
        out.update(_prefixed(prefix, recall_at_k(S, k_values)))
    if compute_mrr:
        out.update(_prefixed(prefix, mrr_and_mean_rank(S)))
    return out


def retrieval_metrics(q, c, prefix: str = "", k_values=K_VALUES, compute_recall=True, compute_mrr=True):
    return retrieval_metrics_from_similarity(similarity(q, c), prefix, k_values, compute_recall, compute_mrr)


def retrieval_metrics_final(q, tgt, img, prefix: str = "", k_values=K_VALUES, compute_recall=True,
                            compute_mrr=True, t2i_weight: float = 0.5, t2t_weight: float = 0.5):
    S = fused_similarity(q, img, tgt, t2i_weight, t2t_weight)
    return retrieval_metrics_from_similarity(S, prefix, k_values, compute_recall, compute_mrr)


def all_retrieval_metrics(q, tgt, img, k_values=K_VALUES, tasks: List[str] = ("T2I", "I2T", "T2T"),
                          compute_recall=True, compute_mrr=True) -> Dict[str, float]:
    """T2I = query->image, I2T = image->target, T2T = query->target (metrics.py:219-250)."""
    pairs = {"T2I": (q, img), "I2T": (img, tgt), "T2T": (q, tgt)}
    out: Dict[str, float] = {}
    for task in ("T2I", "I2T", "T2T"):
        if task in tasks:
            a, b = pairs[task]
            out.update(retrieval_metrics(a, b, task, k_values, compute_recall, compute_mrr))
    return out


def planted_embeddings(n: int, d: int, seed: int = 0, query_noise: float = 1.2, target_noise: float = 0.5):
    """Synthetic unit-norm (image, query, target) embeddings with a planted diagonal signal
    (SURVEY.md section 8(c)/(d)): image = unit gaussian direction, query = norm(image + 1.2 * N(0, I)),
    target = norm(image + 0.5 * N(0, I)); one ``default_rng(seed)`` stream drawn in the order
    image -> query noise -> target noise.  N=256, D=768, seed 0 gives T2I R@1 2.73 %, Mean_Rank 70.89."""
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((n, d)).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    q = img + np.float32(query_noise) * rng.standard_normal((n, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = img + np.float32(target_noise) * rng.standard_normal((n, d)).astype(np.float32)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    return img.astype(np.float32), q.astype(np.float32), t.astype(np.float32)
