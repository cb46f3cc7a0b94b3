"""Device-resident similarity ranking: embeddings in, ranks / top-k out, no Q x N matrix in memory.

This is the build's counterpart of the reference's ``S = Q @ C.T`` + two ``np.argsort(-S)`` passes
(/root/reference/src/clip/eval/metrics.py:13-76, 102, 145-148).  Everything heavy happens in libkemr.so
(``kemr_panel_build`` / ``kemr_pair_scores`` / ``kemr_sim_topk`` / ``kemr_rank_dense``); this module only wires
tensors and turns ranks into the reference's metric dictionaries.

Order rule (the reference's unstable argsort leaves exact ties undefined): higher score first, then lower
candidate index.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib, engine

K_VALUES = (1, 5, 10, 20)
ArrayLike = Union[np.ndarray, torch.Tensor]

# "fp32x3": bf16 hi/lo split, three MFMA terms, reproduces fp32 products (default for metrics);
# "bf16": single bf16 pass (about 6e-5 absolute score error on unit vectors), 3x less MFMA work.
PRECISION_TERMS = {"fp32x3": 3, "bf16": 1}


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the similarity / ranking path runs only in the HIP kernels of libkemr.so "
                           "(no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_f32(x: ArrayLike, device: Optional[torch.device] = None) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if not x.is_cuda:
        x = x.to(device or default_device())
    return x.to(torch.float32).contiguous()


def ranks_and_topk(query_parts: Sequence[ArrayLike], gallery_parts: Sequence[ArrayLike],
                   weights: Optional[Sequence[float]] = None, row_gate: Optional[Sequence[Optional[ArrayLike]]] = None,
                   k: int = 10, precision: str = "fp32x3", gt_idx: Optional[ArrayLike] = "diag",
                   bonus: Optional[Tuple[ArrayLike, ArrayLike, ArrayLike]] = None, gallery_offset: int = 0
                   ) -> Tuple[Optional[torch.Tensor], torch.Tensor, torch.Tensor]:
    """score(q, c) = sum_p weights[p] * gate_p[q] * <query_parts[p][q], gallery_parts[p][c]>  (+ sparse bonus).

    Returns (ranks int64 [nq] (1-based, None when gt_idx is None), top scores [nq,k], top ids [nq,k]).
    gt_idx: "diag" (query i <-> candidate i, metrics.py:37), an index array of GLOBAL candidate ids, or None.
    """
    terms = PRECISION_TERMS[precision]
    dev = None
    qs = [to_device_f32(p) for p in query_parts]
    dev = qs[0].device
    gs = [to_device_f32(p, dev) for p in gallery_parts]
    if len(qs) != len(gs):
        raise ValueError("query_parts and gallery_parts must pair up")
    gates = None
    if row_gate is not None:
        gates = [None if r is None else to_device_f32(r, dev) for r in row_gate]
    qp = engine.build_panel(qs, _lib.SIDE_QUERY, terms, part_scale=weights, row_scale=gates)
    gp = engine.build_panel(gs, _lib.SIDE_GALLERY, terms)
    nq, ng = qp.rows, gp.rows
    ahead = gt = sgt = None
    if gt_idx is not None:
        if isinstance(gt_idx, str):
            if nq > ng + gallery_offset:
                raise ValueError("diagonal ground truth needs one candidate per query")
            gt = torch.arange(nq, dtype=torch.int32, device=dev)
        else:
            gt = torch.as_tensor(gt_idx).to(device=dev, dtype=torch.int32)
        local = gt.long() - gallery_offset
        inside = (local >= 0) & (local < ng)
        sgt = engine.pair_scores(qp, gp, torch.arange(nq, device=dev, dtype=torch.int32), local.clamp(0, max(ng - 1, 0)).int())
        if bonus is not None:   # the ground-truth pair's own bonus
            sgt = sgt + _bonus_of_pairs(bonus, gt, dev)
        if not bool(inside.all()):
            raise ValueError("ground-truth ids must lie inside this gallery (sharded use: see dist.py)")
        _require_finite(sgt, "ground-truth scores")
        ahead = torch.zeros(nq, dtype=torch.int32, device=dev)
    top_s, top_i = engine.sim_topk(qp, gp, k, gallery_offset, gt, sgt, ahead, bonus)
    ranks = None if ahead is None else ahead.long() + 1
    return ranks, top_s, top_i


def _bonus_of_pairs(bonus, gt: torch.Tensor, dev) -> torch.Tensor:
    """Sum of CSR bonus values at (row i, column gt[i]) for every query i, vectorised (the alpha sweeps of evaluator.py call
    this 18 times per evaluation with tens of thousands of queries)."""
    ptr, col, val = (torch.as_tensor(b).cpu().numpy() for b in bonus)
    g = gt.cpu().numpy().astype(np.int64)
    nq = g.shape[0]
    rows = np.repeat(np.arange(nq), np.diff(ptr.astype(np.int64)))
    hit = col.astype(np.int64) == g[rows]
    out = np.bincount(rows[hit], weights=val[hit].astype(np.float64), minlength=nq).astype(np.float32)
    return torch.from_numpy(out).to(dev)


def _require_finite(t: torch.Tensor, what: str) -> None:
    """A NaN score compares false against everything: the rank count would report rank 1 (R@1 = 100 %) for a diverged
    checkpoint or an fp8 overflow instead of failing.  The reference's argsort does not reward NaN like that either."""
    if not bool(torch.isfinite(t).all()):
        bad = int((~torch.isfinite(t)).sum())
        raise ValueError(f"{what}: {bad} non-finite value(s); refusing to rank (a NaN ground-truth score would read as rank 1)")


def ranks_of_matrix(similarity_matrix: ArrayLike, k: int = 0, gt_idx: Optional[ArrayLike] = "diag"
                    ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Ranks / top-k of an already materialised score matrix (kemr_rank_dense)."""
    S = to_device_f32(similarity_matrix)
    if S.dim() != 2:
        raise ValueError("similarity matrix must be 2-D")
    if gt_idx is not None:
        _require_finite(S, "similarity matrix")
    gt = None
    if gt_idx is not None:
        gt = torch.arange(S.shape[0], dtype=torch.int32, device=S.device) if isinstance(gt_idx, str) \
            else torch.as_tensor(gt_idx).to(device=S.device, dtype=torch.int32)
    ahead, top_s, top_i = engine.rank_dense(S, gt, k)
    return (None if ahead is None else ahead.long() + 1), top_s, top_i


def metrics_from_ranks(ranks: torch.Tensor, k_values: Sequence[int] = K_VALUES, compute_recall: bool = True,
                       compute_mrr: bool = True) -> Dict[str, float]:
    """Recall@K (percent), MRR (percent), Mean_Rank with the reference's key names (metrics.py:41-42, 70-76)."""
    r = ranks.detach().cpu().numpy().astype(np.int64)
    out: Dict[str, float] = {}
    if compute_recall:
        for k in k_values:
            out[f"R@{k}"] = float(np.mean(r <= k) * 100.0)
    if compute_mrr:
        out["MRR"] = float(np.mean(1.0 / r) * 100.0)
        out["Mean_Rank"] = float(np.mean(r))
    return out
