"""Retrieval metrics with the reference's function names, computed by the fused HIP kernels.

Drop-in for /root/reference/src/clip/eval/metrics.py (same names, arguments, key naming ``"{prefix}_{metric}"``,
percent scaling, k in {1,5,10,20}, diagonal ground truth), but:

* functions that receive EMBEDDINGS never build the N x N matrix: scores, the rank of the diagonal element and the
  top-k come out of one fused kernel pass (``ranking.ranks_and_topk``); T2I+T2T fusion is one contraction over the
  concatenated, weighted embeddings instead of two GEMMs and an N^2 axpy (reference ``metrics.py:145-148``);
* functions that receive a MATRIX (the caller already materialised it) stream it once on the GPU
  (``ranking.ranks_of_matrix``) instead of two full ``np.argsort`` passes (reference ``metrics.py:34,62``);
* Recall@K is derived from the rank (``rank <= K``), which equals the reference's ``i in argsort[:, :K]``.

Inputs may be numpy arrays (as in the reference) or torch tensors already resident on the GPU.
Ties: higher score first, then lower index (the reference's unstable argsort leaves exact ties undefined).
"""
from __future__ import annotations

import logging
from typing import Dict, List, Sequence

from . import ranking

logger = logging.getLogger(__name__)

DEFAULT_K = [1, 5, 10, 20]


def _with_prefix(prefix: str, d: Dict[str, float]) -> Dict[str, float]:
    return {(f"{prefix}_{k}" if prefix else k): v for k, v in d.items()}


def compute_recall_at_k(similarity_matrix, k_values: List[int] = DEFAULT_K) -> Dict[str, float]:
    """Recall@K (percent) of a (N_queries, N_candidates) score matrix; ground truth = diagonal."""
    ranks, _, _ = ranking.ranks_of_matrix(similarity_matrix)
    return ranking.metrics_from_ranks(ranks, k_values, compute_recall=True, compute_mrr=False)


def compute_mrr_and_mean_rank(similarity_matrix) -> Dict[str, float]:
    """MRR (percent) and Mean_Rank of a score matrix; ground truth = diagonal."""
    ranks, _, _ = ranking.ranks_of_matrix(similarity_matrix)
    return ranking.metrics_from_ranks(ranks, compute_recall=False, compute_mrr=True)


def _metrics_of_ranks(ranks, prefix, k_values, compute_recall, compute_mrr) -> Dict[str, float]:
    return _with_prefix(prefix, ranking.metrics_from_ranks(ranks, k_values, compute_recall, compute_mrr))


def compute_retrieval_metrics(query_embeddings, candidate_embeddings, prefix: str = "",
                              k_values: List[int] = DEFAULT_K, compute_recall: bool = True,
                              compute_mrr: bool = True, precision: str = "fp32x3") -> Dict[str, float]:
    """Metrics of `query @ candidate.T` without forming it (reference metrics.py:79-116)."""
    ranks, _, _ = ranking.ranks_and_topk([query_embeddings], [candidate_embeddings], k=0, precision=precision)
    return _metrics_of_ranks(ranks, prefix, k_values, compute_recall, compute_mrr)


def compute_retrieval_metrics_final(query_embeddings, target_embeddings, image_embeddings, prefix: str = "",
                                    k_values: List[int] = DEFAULT_K, compute_recall: bool = True,
                                    compute_mrr: bool = True, t2i_weight: float = 0.5, t2t_weight: float = 0.5,
                                    precision: str = "fp32x3") -> Dict[str, float]:
    """Metrics of ``t2i_weight * Q@I.T + t2t_weight * Q@T.T`` (reference metrics.py:119-162) as ONE fused contraction."""
    ranks, _, _ = ranking.ranks_and_topk([query_embeddings, query_embeddings], [image_embeddings, target_embeddings],
                                         weights=[t2i_weight, t2t_weight], k=0, precision=precision)
    return _metrics_of_ranks(ranks, prefix, k_values, compute_recall, compute_mrr)


def compute_retrieval_metrics_fusion(similarity_matrix, prefix: str = "", k_values: List[int] = DEFAULT_K,
                                     compute_recall: bool = True, compute_mrr: bool = True) -> Dict[str, float]:
    """Metrics of an already fused score matrix (reference metrics.py:165-185)."""
    ranks, _, _ = ranking.ranks_of_matrix(similarity_matrix)
    return _metrics_of_ranks(ranks, prefix, k_values, compute_recall, compute_mrr)


def compute_all_retrieval_metrics(query_embeddings, target_embeddings, image_embeddings,
                                  k_values: List[int] = DEFAULT_K, tasks: Sequence[str] = ("T2I", "I2T", "T2T"),
                                  compute_recall: bool = True, compute_mrr: bool = True,
                                  precision: str = "fp32x3") -> Dict[str, float]:
    """T2I: query -> image, I2T: image -> target, T2T: query -> target (reference metrics.py:188-252)."""
    pairs = {"T2I": (query_embeddings, image_embeddings), "I2T": (image_embeddings, target_embeddings),
             "T2T": (query_embeddings, target_embeddings)}
    out: Dict[str, float] = {}
    for task in ("T2I", "I2T", "T2T"):
        if task in tasks:
            q, c = pairs[task]
            out.update(compute_retrieval_metrics(q, c, task, k_values, compute_recall, compute_mrr, precision))
    return out


def compute_training_metrics(query_embeddings, target_embeddings, image_embeddings,
                             tasks: Sequence[str] = ("T2I", "I2T", "T2T")) -> Dict[str, float]:
    """MRR / Mean_Rank only (early stopping, reference metrics.py:256-282)."""
    return compute_all_retrieval_metrics(query_embeddings, target_embeddings, image_embeddings, tasks=tasks,
                                         compute_recall=False, compute_mrr=True)


# ---- deprecated shims kept for callers of the reference's old API (metrics.py:285-351): text -> text self retrieval
def _deprecated(name: str):
    logger.warning("%s is DEPRECATED. Use compute_all_retrieval_metrics / compute_training_metrics with separate "
                   "query and target embeddings.", name)


def compute_metrics_multi_mode(image_embeddings, text_embeddings_by_variant) -> Dict[str, float]:
    _deprecated("compute_metrics_multi_mode")
    text = text_embeddings_by_variant[0]
    return compute_all_retrieval_metrics(text, text, image_embeddings, tasks=["T2I", "I2T", "T2T"])


def compute_metrics_single_4train(image_embeddings, text_embeddings_by_variant) -> Dict[str, float]:
    _deprecated("compute_metrics_single_4train")
    text = text_embeddings_by_variant[0]
    return compute_training_metrics(text, text, image_embeddings, tasks=["T2I", "I2T", "T2T"])


def compute_metrics_multi_4train(image_embeddings, text_embeddings_by_variant) -> Dict[str, float]:
    _deprecated("compute_metrics_multi_4train")
    text = text_embeddings_by_variant[0]
    return compute_training_metrics(text, text, image_embeddings, tasks=["T2I", "I2T", "T2T"])
