"""CPU, world_size = 2 over gloo: the collective plumbing of the sharded gallery (all-gather of queries, per-shard
search with global ids, all-gather + merge of candidates, ground-truth score / `ahead` reductions).  The kernel calls
are replaced by an oracle-backed stand-in (TEST ONLY: the product path always uses the HIP `engine` module)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """numpy stand-in with the signatures of engine.build_panel / sim_topk / pair_scores / topk_merge."""

    class P:
        def __init__(self, mat):
            self.mat, self.rows, self.kdim, self.device = mat, mat.shape[0], mat.shape[1], torch.device("cpu")

    @staticmethod
    def build_panel(parts, side, terms=3, part_scale=None, row_scale=None):
        cols = []
        for p, t in enumerate(parts):
            x = t.double().numpy().copy()
            if part_scale is not None:
                x *= part_scale[p]
            if row_scale is not None and row_scale[p] is not None:
                x *= row_scale[p].double().numpy()[:, None]
            cols.append(x)
        return OracleOps.P(np.concatenate(cols, 1))

    @staticmethod
    def _order(S, ids):
        return np.lexsort((ids, -S))

    @staticmethod
    def sim_topk(qp, gp, k, gallery_offset=0, gt_idx=None, gt_score=None, ahead=None, bonus=None):
        S = qp.mat @ gp.mat.T
        ids = np.arange(gp.rows) + gallery_offset
        top_s = np.full((qp.rows, k), -np.inf, np.float32)
        top_i = np.full((qp.rows, k), -1, np.int32)
        for r in range(qp.rows):
            o = OracleOps._order(S[r], ids)[:k]
            top_s[r, :len(o)], top_i[r, :len(o)] = S[r, o], ids[o]
            if gt_idx is not None:
                g, sg = int(gt_idx[r]), float(gt_score[r])
                before = (S[r].astype(np.float32) > np.float32(sg)) | ((S[r].astype(np.float32) == np.float32(sg)) & (ids < g))
                ahead[r] += int((before & (ids != g)).sum())
        return torch.from_numpy(top_s), torch.from_numpy(top_i)

    @staticmethod
    def pair_scores(qp, gp, q_rows, g_rows):
        return torch.from_numpy(np.einsum("ij,ij->i", qp.mat[q_rows.numpy()], gp.mat[g_rows.numpy()]).astype(np.float32))

    @staticmethod
    def topk_merge(scores, idx, k):
        nq = scores.shape[0]
        s, i = scores.reshape(nq, -1).numpy(), idx.reshape(nq, -1).numpy()
        out_s = np.full((nq, k), -np.inf, np.float32)
        out_i = np.full((nq, k), -1, np.int32)
        for r in range(nq):
            ok = i[r] >= 0
            o = np.lexsort((i[r][ok], -s[r][ok]))[:k]
            out_s[r, :len(o)], out_i[r, :len(o)] = s[r][ok][o], i[r][ok][o]
        return torch.from_numpy(out_s), torch.from_numpy(out_i)


def _worker(rank, world, port, n, nq, d, k, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knowledge_enhanced_multimodal_retrieval_amd.dist import ShardedGallery, shard_bounds
    from oracle import metrics_ref
    img, q, t = metrics_ref.planted_embeddings(n, d, seed=3)
    lo, hi = shard_bounds(n, world, rank)
    gal = ShardedGallery([torch.from_numpy(img[lo:hi]), torch.from_numpy(t[lo:hi])], n, group=None, ops=OracleOps)
    per = nq // world
    ql = torch.from_numpy(q[rank * per:(rank + 1) * per])
    gt_l = torch.arange(rank * per, (rank + 1) * per, dtype=torch.int32)
    s, i = gal.search([ql, ql], weights=[0.3, 0.7], k=k)
    ranks, s2, i2 = gal.ranks([ql, ql], gt_l, weights=[0.3, 0.7], k=k)
    assert torch.equal(i, i2)
    ranks0, _, _ = gal.ranks([ql, ql], gt_l, weights=[0.3, 0.7], k=0)          # ranks only: no candidate exchange at all
    assert torch.equal(ranks0, ranks)
    gal.check_rows = False                                                     # a caller with fixed batch sizes: no host read in the call
    ranks1, _, i3 = gal.ranks([ql, ql], gt_l, weights=[0.3, 0.7], k=k)
    assert torch.equal(ranks1, ranks) and torch.equal(i3, i)
    gal.check_rows = True
    # a stream of query batches with the candidate exchange of batch b in flight while batch b+1 is produced: same answers
    produced = []

    def batches():
        for h in range(2):
            part = ql[h * (per // 2):(h + 1) * (per // 2)]
            produced.append(h)
            yield [part, part]
    halves = list(gal.search_many(batches(), weights=[0.3, 0.7], k=k))
    assert produced == [0, 1] and len(halves) == 2
    for h, (hs, hi_) in enumerate(halves):       # rows of batch h: every rank's h-th half, in rank order
        rows = torch.cat([torch.arange(r * per + h * (per // 2), r * per + (h + 1) * (per // 2)) for r in range(world)])
        assert torch.equal(hi_, i[rows]) and torch.allclose(hs, s[rows], atol=1e-6)
    # ranks that bring different numbers of query rows are refused on EVERY rank, before any data-sized collective
    uneven = ql[:per - rank]
    with pytest.raises(ValueError, match="same number of query rows"):
        gal.search([uneven, uneven], weights=[0.3, 0.7], k=k)
    with pytest.raises(ValueError, match="same number of query rows"):
        gal.ranks([uneven, uneven], gt_l[:per - rank], weights=[0.3, 0.7], k=k)
    q_out.put((rank, s.numpy(), i.numpy(), ranks.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 64])
def test_sharded_search_world2(n):
    from oracle import metrics_ref
    world, nq, d, k = 2, 16, 32, 5
    port = 29500 + (os.getpid() + n) % 2000
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nq, d, k, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q_out.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    img, q, t = metrics_ref.planted_embeddings(n, d, seed=3)
    S = 0.3 * (q[:nq].astype(np.float64) @ img.astype(np.float64).T) + 0.7 * (q[:nq].astype(np.float64) @ t.astype(np.float64).T)
    exp_s, exp_i = metrics_ref.topk(S, k)
    exp_r = metrics_ref.ranks_by_count(S.astype(np.float32), np.arange(nq))
    for rank, s, i, ranks in results:                       # every rank ends with the full, identical answer
        assert np.array_equal(i, exp_i), rank
        np.testing.assert_allclose(s, exp_s, atol=1e-6)
        assert np.array_equal(ranks, exp_r)


def _forced_worker(port, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from knowledge_enhanced_multimodal_retrieval_amd import dist as kd
    from oracle import metrics_ref
    n, nq, d, k = 101, 16, 32, 5
    img, q, t = metrics_ref.planted_embeddings(n, d, seed=3)
    parts = [torch.from_numpy(img), torch.from_numpy(t)]
    ql, gt = torch.from_numpy(q[:nq]), torch.arange(nq, dtype=torch.int32)
    plain = kd.ShardedGallery(parts, n, ops=OracleOps)                        # no process group: everything short-circuits
    want = plain.search([ql, ql], [0.3, 0.7], k), plain.ranks([ql, ql], gt, [0.3, 0.7], k)
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = {"gather": 0, "reduce": 0}
    g0, r0 = dist.all_gather_into_tensor, dist.all_reduce
    dist.all_gather_into_tensor = lambda *a, **kw: (calls.__setitem__("gather", calls["gather"] + 1), g0(*a, **kw))[1]
    dist.all_reduce = lambda *a, **kw: (calls.__setitem__("reduce", calls["reduce"] + 1), r0(*a, **kw))[1]
    unforced = kd.ShardedGallery(parts, n, ops=OracleOps).search([ql, ql], [0.3, 0.7], k)
    assert calls == {"gather": 0, "reduce": 0}                                # world 1 without the switch: still no collective
    kd.force_collectives(True)
    gal = kd.ShardedGallery(parts, n, ops=OracleOps)
    got = gal.search([ql, ql], [0.3, 0.7], k), gal.ranks([ql, ql], gt, [0.3, 0.7], k)
    many = list(gal.search_many(([ql[i:i + 8], ql[i:i + 8]] for i in (0, 8)), [0.3, 0.7], k))
    ok = all(torch.equal(a, b) for a, b in zip(want[0] + want[1], got[0] + got[1])) and torch.equal(unforced[1], want[0][1])
    ok = ok and torch.equal(torch.cat([m[1] for m in many]), want[0][1])
    kd.force_collectives(False)
    dist.destroy_process_group()
    q_out.put((ok, calls))


def test_forced_collectives_at_world_size_one_change_nothing():
    """dist.force_collectives: with ONE rank the helpers issue their collectives anyway (what tests/test_dist_rccl_world1.py runs
    through RCCL on the GPU box) and every result equals the short-circuited call."""
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(29500 + (os.getpid() + 977) % 2000, q_out))
    p.start()
    ok, calls = q_out.get(timeout=120)
    p.join(30)
    assert ok and p.exitcode == 0
    assert calls["gather"] >= 10 and calls["reduce"] >= 5, calls
