"""Gallery sharded over the GPUs of one node: one process per GPU, RCCL through ``torch.distributed``.

The reference's eval / search path is single-device; this is new design for BASELINE
config 4.  Gallery items are independent, so each rank encodes and keeps its own contiguous shard resident; a search
has exactly one exchange step:

1. ``all_gather`` of the query embeddings (each rank encoded 1/world of the query batch; 128 x 768 fp32 per rank at
   Q = 1024 -- KB-scale, latency-bound, the xGMI links are never the limit),
2. local fused similarity + top-k against the own shard with global candidate ids (``gallery_offset``),
3. ``all_gather`` of the per-shard candidates ([Q, k] scores + ids), merged by ``kemr_topk_merge`` on every rank.

Ranks of a ground truth add two tiny reductions: the owner shard computes the ground-truth score
(``kemr_pair_scores``; ``all_reduce(sum)`` with zeros elsewhere) and the per-shard ``ahead`` counts are summed.

Step 3 is issued asynchronously (``search_async`` / ``search_many``): the candidate all-gather runs on RCCL's stream while
the compute stream already encodes the NEXT query batch; the merge is queued when the result is asked for.  Every rank must
bring the same number of query rows (a fused all-gather needs equal shapes; unequal ones hang or corrupt under RCCL): that
is checked with one 2-element all-reduce per call and refused with a ValueError on every rank.

The compute calls are taken from an ``ops`` namespace (default: the HIP-backed ``engine`` module) so that the
collective plumbing can be exercised with gloo on CPU-only machines in tests; the product path always uses ``engine``.
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib, engine, ranking


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split: rank r owns global ids [lo, hi) with ceil(n / world) items per rank (the last may be short)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


_FORCE = os.environ.get("KEMR_DIST_FORCE_COLLECTIVES", "") == "1"


def force_collectives(on: bool = True) -> None:
    """A world of ONE rank normally short-circuits every helper below (nothing to exchange).  With this switch on (or
    KEMR_DIST_FORCE_COLLECTIVES=1) and a process group initialised, the collectives are issued anyway -- so that the exact calls an
    8-GPU run makes (device-tensor all_gather_into_tensor, the device branch of require_equal_rows, async work.wait() stream
    ordering, all_reduce, the candidate merge over [world, Q, k]) can be executed through RCCL on a one-GPU box
    (tests/test_dist_rccl_world1.py).  Results are unchanged: gathering one rank's rows is the identity."""
    global _FORCE
    _FORCE = bool(on)


def _skip(world: int) -> bool:
    """True when a helper may return without a collective: one rank and nobody asked for the collectives to run anyway."""
    return world == 1 and not (_FORCE and dist.is_available() and dist.is_initialized())


def _host_staged(x: torch.Tensor, group) -> bool:
    """gloo moves device tensors through the host itself only for some collectives; the rehearsal of the N > 1 path on one GPU
    (KEMR_DIST_BACKEND=gloo, bench.py) stages them explicitly.  RCCL ("nccl") never takes this branch."""
    return x.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally shaped per-rank tensors along dim 0 (single fused all-gather)."""
    world, _ = _world(group)
    if _skip(world):
        return x
    x = x.contiguous()
    if _host_staged(x, group):
        xc = x.cpu()
        oc = torch.empty((world * xc.shape[0],) + tuple(xc.shape[1:]), dtype=xc.dtype)
        dist.all_gather_into_tensor(oc, xc, group=group)
        return oc.to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def all_gather_rows_async(x: torch.Tensor, group=None):
    """As :func:`all_gather_rows`, not waited for: returns (out, work); ``work.wait()`` orders the current stream after it."""
    world, _ = _world(group)
    if _skip(world):
        return x, None
    x = x.contiguous()
    if _host_staged(x, group):
        return all_gather_rows(x, group), None
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    return out, dist.all_gather_into_tensor(out, x, group=group, async_op=True)


def all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks (device tensors staged through the host under gloo, see :func:`_host_staged`)."""
    world, _ = _world(group)
    if _skip(world):
        return x
    if _host_staged(x, group):
        xc = x.cpu()
        dist.all_reduce(xc, group=group)
        x.copy_(xc)
        return x
    dist.all_reduce(x, group=group)
    return x


def require_equal_rows(n_local: int, device, group=None, what: str = "query rows") -> None:
    """Refuse, on every rank, a call in which the ranks bring different row counts."""
    world, rank = _world(group)
    if _skip(world):
        return
    t = torch.tensor([n_local, -n_local], dtype=torch.int64)      # host tensor under gloo, device tensor under RCCL
    if dist.get_backend(group) != "gloo":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise ValueError(f"sharded search: every rank must pass the same number of {what}; rank {rank} has {n_local}, "
                         f"the group has between {lo} and {hi} (pad the last batch or split it evenly)")


class PendingSearch:
    """A search whose candidate exchange is in flight; ``result()`` queues the merge behind it."""

    def __init__(self, merge: Callable, works, tensors):
        self._merge, self._works, self._tensors, self._out = merge, works, tensors, None

    def result(self):
        if self._out is None:
            for w in self._works:
                if w is not None:
                    w.wait()
            self._out = self._merge(*self._tensors)
            self._tensors = None
        return self._out


class ShardedGallery:
    """This rank's shard of the gallery, packed for the fused similarity kernel."""

    def __init__(self, local_parts: Sequence[torch.Tensor], n_total: int, precision: str = "fp32x3", group=None, ops=engine):
        self.group, self.ops = group, ops
        # one 2-element all-reduce + host read per call, the only host synchronisation of search / ranks: a caller that guarantees
        # equal batches (a fixed batch size, as bench.py's) clears it for THIS gallery and the whole call is asynchronous
        self.check_rows = True
        self.world, self.rank = _world(group)
        self.n_total = n_total
        self.lo, self.hi = shard_bounds(n_total, self.world, self.rank)
        if local_parts[0].shape[0] != self.hi - self.lo:
            raise ValueError(f"rank {self.rank} owns ids [{self.lo}, {self.hi}) but got {local_parts[0].shape[0]} rows")
        self.terms = ranking.PRECISION_TERMS[precision]
        self.nparts = len(local_parts)
        self.panel = ops.build_panel(list(local_parts), _lib.SIDE_GALLERY, self.terms) if self.hi > self.lo else None

    def _query_panel(self, local_query_parts, weights, row_gate):
        if self.check_rows:
            require_equal_rows(int(local_query_parts[0].shape[0]), local_query_parts[0].device, self.group)
        qs = [all_gather_rows(q, self.group) for q in local_query_parts]
        gates = None
        if row_gate is not None:
            gates = [None if g is None else all_gather_rows(g.reshape(-1, 1), self.group).reshape(-1) for g in row_gate]
        return self.ops.build_panel(qs, _lib.SIDE_QUERY, self.terms, part_scale=weights, row_scale=gates), qs[0].shape[0]

    def _local_topk(self, qp, nq, k, dev, *rank_args):
        if self.panel is not None:
            return self.ops.sim_topk(qp, self.panel, k, self.lo, *rank_args)
        return (torch.full((nq, k), float("-inf"), dtype=torch.float32, device=dev),
                torch.full((nq, k), -1, dtype=torch.int32, device=dev))

    def _exchange(self, s, i, k) -> PendingSearch:
        """Candidates of all shards -> every rank, not waited for."""
        if _skip(self.world):
            return PendingSearch(lambda a, b: (a, b), [], (s, i))
        ss, w1 = all_gather_rows_async(s.unsqueeze(0), self.group)          # [world, Q, k]
        ii, w2 = all_gather_rows_async(i.unsqueeze(0), self.group)
        merge = lambda a, b: self.ops.topk_merge(a.permute(1, 0, 2).contiguous(), b.permute(1, 0, 2).contiguous(), k)
        return PendingSearch(merge, [w1, w2], (ss, ii))

    def search_async(self, local_query_parts: Sequence[torch.Tensor], weights: Optional[Sequence[float]] = None, k: int = 10,
                     row_gate=None) -> PendingSearch:
        """``search`` with the candidate exchange left in flight: encode the next query batch, then call ``result()``."""
        qp, nq = self._query_panel(local_query_parts, weights, row_gate)
        s, i = self._local_topk(qp, nq, k, local_query_parts[0].device)
        return self._exchange(s, i, k)

    def search(self, local_query_parts: Sequence[torch.Tensor], weights: Optional[Sequence[float]] = None, k: int = 10,
               row_gate=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Every rank passes its slice of the query batch (same row count on all ranks) and receives the merged
        top-k of the WHOLE batch against the WHOLE gallery: (scores [Q, k], global ids [Q, k])."""
        return self.search_async(local_query_parts, weights, k, row_gate).result()

    def search_many(self, batches: Iterable[Sequence[torch.Tensor]], weights: Optional[Sequence[float]] = None, k: int = 10
                    ) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """A stream of query batches with one batch of lookahead.  ``batches`` is consumed lazily: if producing an element
        encodes the texts (a generator around ``model.encode_text``), the encoder of batch b+1 is queued on the compute
        stream while the candidate all-gather of batch b is still on the wire.  Yields the merged results in order."""
        prev = None
        for parts in batches:
            cur = self.search_async(parts, weights, k)
            if prev is not None:
                yield prev.result()
            prev = cur
        if prev is not None:
            yield prev.result()

    def ranks(self, local_query_parts: Sequence[torch.Tensor], local_gt: torch.Tensor,
              weights: Optional[Sequence[float]] = None, k: int = 10, row_gate=None
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """As ``search`` plus the 1-based rank of each query's ground-truth candidate (GLOBAL id in ``local_gt``)."""
        qp, nq = self._query_panel(local_query_parts, weights, row_gate)
        dev = local_query_parts[0].device
        gt = all_gather_rows(local_gt.to(device=dev, dtype=torch.int32).reshape(-1, 1), self.group).reshape(-1)
        # the ground-truth score of EVERY query against this shard with the local row clamped into it, masked to the queries whose
        # ground truth lives here: no data-dependent shape, no host read (round 2 tested `mine.any()` on the host and gathered rows)
        sgt = torch.zeros(nq, dtype=torch.float32, device=dev)
        if self.panel is not None:
            mine = (gt >= self.lo) & (gt < self.hi)
            local = (gt - self.lo).clamp(0, self.hi - self.lo - 1).to(torch.int32)
            every = self.ops.pair_scores(qp, self.panel, torch.arange(nq, dtype=torch.int32, device=dev), local)
            sgt = torch.where(mine, every, sgt)
        all_reduce_sum(sgt, self.group)                            # exactly one rank contributed a non-zero per query
        ahead = torch.zeros(nq, dtype=torch.int32, device=dev)
        s, i = self._local_topk(qp, nq, k, dev, gt, sgt, ahead)
        pending = self._exchange(s, i, k) if k > 0 else None       # k == 0: ranks only (what Recall@K / MRR need), nothing to merge
        all_reduce_sum(ahead, self.group)                          # rides while the candidates are gathered
        if pending is not None:
            s, i = pending.result()
        return ahead.long() + 1, s, i
