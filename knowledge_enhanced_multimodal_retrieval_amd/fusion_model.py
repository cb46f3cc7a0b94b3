"""Learned T2I/T2T fusion heads over a frozen CLIP, with the reference's class names and state-dict keys
(/root/reference/src/clip/model/fusion_model.py: heads :9-240, ``FusionModel`` :243-331), scored by the HIP kernels.

The heads are parameter containers (so reference checkpoints load: ``fusion_head.gate_net.0.weight`` ...); the scoring
``forward(query_embed [N,D], image_embed [M,D], target_embed [M,D]) -> [N,M]`` is inference only (eval mode, dropout
inactive) and maps onto the fused similarity kernel:

* gated / simple_gated / simple_gated_with_bias: ``g*t2i + (1-g)*t2t`` = ONE contraction over ``[g*q ; (1-g)*q]`` x
  ``[img ; tgt]`` -- the per-query gate is a row scale of the query panel (``kemr_panel_build`` row_scale).  The gate
  itself (a [N,D]x[D,128] MLP or a dot product per query) is O(N*D) host-side plumbing on the device tensors.
* bilinear: ``q . (W img)^T = (q W) . img^T`` -- the D x D transform is applied to the N queries (a small contraction
  with the same kernel) instead of to the M candidates; then one weighted contraction as above.
* linear: MLP(2->128->1) of every (t2i, t2t) pair -- needs both dense matrices; ``kemr_linear_head`` applies the MLP
  element-wise on the GPU.
* cross_attention: per-pair multi-head attention of one query over two keys + out-projection + a 3-layer MLP,
  O(N*M*D^2) in the reference.  Everything but the per-head 2-way softmax weights is linear in per-candidate
  quantities, so the projections are folded per candidate (``P = W1.Wo[:,head].V``, small dense contractions with
  the similarity kernel) and ``kemr_cross_attention_pairs`` does ~21 kFLOP per pair instead of ~1.6 MFLOP.

``rank()`` gives ranks / top-k without ever forming the [N,M] matrix for the gated family and bilinear; linear and
cross_attention produce the dense matrix on the GPU and rank it with ``kemr_rank_dense``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, engine, ranking


class SimpleGatedFusionWithBias(nn.Module):
    def __init__(self, embed_dim: int = 768):
        super().__init__()
        self.query_weight = nn.Parameter(torch.zeros(embed_dim))
        self.bias = nn.Parameter(torch.tensor(-2.0))

    def gate(self, q):
        return torch.sigmoid((q * self.query_weight).sum(dim=1, keepdim=True) + self.bias)


class SimpleGatedFusion(nn.Module):
    def __init__(self, embed_dim: int = 768):
        super().__init__()
        self.query_weight = nn.Parameter(torch.ones(embed_dim))
        self.bias = nn.Parameter(torch.zeros(1))

    def gate(self, q):
        return torch.sigmoid((q * self.query_weight).sum(dim=1, keepdim=True) + self.bias)


class GatedFusionHead(nn.Module):
    def __init__(self, embed_dim: int = 768):
        super().__init__()
        self.gate_net = nn.Sequential(nn.Linear(embed_dim, 128), nn.ReLU(), nn.Dropout(0.1), nn.Linear(128, 1), nn.Sigmoid())

    def gate(self, q):
        return self.gate_net(q)


class LinearFusionHead(nn.Module):
    def __init__(self, hidden_dim: int = 128):
        super().__init__()
        self.fusion = nn.Sequential(nn.Linear(2, hidden_dim), nn.ReLU(), nn.Dropout(0.1), nn.Linear(hidden_dim, 1))


class BilinearFusionHead(nn.Module):
    def __init__(self, embed_dim: int = 768):
        super().__init__()
        self.W_image = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_target = nn.Linear(embed_dim, embed_dim, bias=False)
        self.alpha = nn.Parameter(torch.tensor(0.5))


class CrossAttentionFusionHead(nn.Module):
    """Parameter container (reference checkpoints load); scored by ``FusionModel._cross_attention``."""

    def __init__(self, embed_dim: int = 768, num_heads: int = 8, hidden_dim: int = 256):
        super().__init__()
        self.embed_dim = embed_dim
        self.query_proj = nn.Linear(embed_dim, embed_dim)
        self.image_proj = nn.Linear(embed_dim, embed_dim)
        self.target_proj = nn.Linear(embed_dim, embed_dim)
        self.cross_attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, batch_first=True, dropout=0.1)
        self.score_mlp = nn.Sequential(nn.Linear(embed_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1),
                                       nn.Linear(hidden_dim, 64), nn.ReLU(), nn.Dropout(0.1), nn.Linear(64, 1))


_GATED = ("gated", "simple_gated", "simple_gated_with_bias")


class FusionModel(nn.Module):
    """Wrapper for a fusion head over a frozen CLIP encoder (``fusion_type`` as in the reference)."""

    def __init__(self, clip_model, fusion_type: str = "linear", embed_dim: int = 768):
        super().__init__()
        self.clip_model, self.fusion_type = clip_model, fusion_type
        for p in self.clip_model.parameters():
            p.requires_grad = False
        heads = {"linear": lambda: LinearFusionHead(hidden_dim=128),
                 "cross_attention": lambda: CrossAttentionFusionHead(embed_dim=embed_dim, num_heads=8, hidden_dim=256),
                 "gated": lambda: GatedFusionHead(embed_dim=embed_dim),
                 "simple_gated": lambda: SimpleGatedFusion(embed_dim=embed_dim),
                 "simple_gated_with_bias": lambda: SimpleGatedFusionWithBias(embed_dim=embed_dim),
                 "bilinear": lambda: BilinearFusionHead(embed_dim=embed_dim)}
        if fusion_type not in heads:
            raise ValueError(f"Unknown fusion type: {fusion_type}")
        self.fusion_head = heads[fusion_type]()

    # ---- encoders: encode + L2 normalise, fused in the HIP tail kernel (reference fusion_model.py:287-303)
    @torch.no_grad()
    def encode_query(self, query_tokens):
        return self.clip_model.encode_text(query_tokens, normalize=True)

    @torch.no_grad()
    def encode_target(self, target_tokens):
        return self.clip_model.encode_text(target_tokens, normalize=True)

    @torch.no_grad()
    def encode_image(self, images):
        return self.clip_model.encode_image(images, normalize=True)

    # ---- scoring
    def _parts(self, q, img, tgt):
        """(query_parts, gallery_parts, weights, row_gates) of the fused contraction for this head."""
        h = self.fusion_head
        q = ranking.to_device_f32(q)
        img, tgt = ranking.to_device_f32(img, q.device), ranking.to_device_f32(tgt, q.device)
        if self.fusion_type in _GATED:
            g = self._gate(q)
            return [q, q], [img, tgt], None, [g, 1.0 - g]
        if self.fusion_type == "bilinear":
            h = h.to(q.device)
            a = float(torch.sigmoid(h.alpha))
            wq = []
            for lin in (h.W_image, h.W_target):                      # (q W)[n, j] = <q[n, :], W[:, j]>
                qp = engine.build_panel([q], _lib.SIDE_QUERY, 3)
                wp = engine.build_panel([lin.weight.detach().t().contiguous()], _lib.SIDE_GALLERY, 3)
                wq.append(engine.scores_dense(qp, wp))
            return wq, [img, tgt], [a, 1.0 - a], None
        raise NotImplementedError

    def _gate(self, q: torch.Tensor) -> torch.Tensor:
        """The gated heads' gate in eval mode, [N] fp32 on the GPU, through the library: Linear(d, 128) by the fp32x3 dense kernel,
        everything behind it (bias, ReLU, the 128 -> 1 dot or the d -> 1 dot of the simple heads, sigmoid) in kemr_gate_rows; the
        head's own ``gate()`` (torch) is what the tests compare it with."""
        h = self.fusion_head.to(q.device).eval()
        f32 = lambda t: t.detach().float().contiguous().to(q.device)
        if hasattr(h, "gate_net"):
            x, pre = self._linear(q, h.gate_net[0].weight), f32(h.gate_net[0].bias)
            w, bias, relu = f32(h.gate_net[3].weight).reshape(-1), float(h.gate_net[3].bias.detach()), 1
        else:
            x, pre, w, bias, relu = q.float().contiguous(), None, f32(h.query_weight), float(h.bias.detach().reshape(-1)[0]), 0
        out = torch.empty(x.shape[0], dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            _lib.check(_lib.lib().kemr_gate_rows(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], C.c_void_p(pre.data_ptr()) if pre is not None else None,
                                                 C.c_void_p(w.data_ptr()), bias, relu, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)), "gate_rows")
        return out

    @staticmethod
    def _linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x @ weight.T (+ bias) through the fp32x3 dense similarity kernel (x [R,K], weight [O,K])."""
        xp = engine.build_panel([x.float().contiguous()], _lib.SIDE_QUERY, 3)
        wp = engine.build_panel([weight.detach().float().contiguous()], _lib.SIDE_GALLERY, 3)
        y = engine.scores_dense(xp, wp)
        return y if bias is None else y + bias.detach().float()

    @torch.no_grad()
    def _cross_attention(self, q, img, tgt) -> torch.Tensor:
        """Eval-mode CrossAttentionFusionHead.forward (reference fusion_model.py:83-133) -> [N, M] fp32."""
        h = self.fusion_head.to(q.device).eval()
        lin = self._linear
        N, D = q.shape
        M = img.shape[0]
        H = h.cross_attn.num_heads
        hd = D // H
        Wqkv, bqkv = h.cross_attn.in_proj_weight.detach(), h.cross_attn.in_proj_bias.detach()
        qp = lin(q, h.query_proj.weight, h.query_proj.bias)
        ip = lin(img, h.image_proj.weight, h.image_proj.bias)
        tp = lin(tgt, h.target_proj.weight, h.target_proj.bias)
        Q = lin(qp, Wqkv[:D], bqkv[:D]) * (hd ** -0.5)
        Ki, Kt = lin(ip, Wqkv[D:2 * D], bqkv[D:2 * D]), lin(tp, Wqkv[D:2 * D], bqkv[D:2 * D])
        Vi, Vt = lin(ip, Wqkv[2 * D:], bqkv[2 * D:]), lin(tp, Wqkv[2 * D:], bqkv[2 * D:])
        Wo, bo = h.cross_attn.out_proj.weight.detach().float(), h.cross_attn.out_proj.bias.detach().float()
        W1, b1 = h.score_mlp[0].weight.detach().float(), h.score_mlp[0].bias.detach().float()
        W2, b2 = h.score_mlp[3].weight.detach().float(), h.score_mlp[3].bias.detach().float()
        W3, b3 = h.score_mlp[6].weight.detach().float().reshape(-1), float(h.score_mlp[6].bias.detach())
        hid1, hid2 = W1.shape[0], W2.shape[0]
        Pi = torch.empty((M, H, hid1), dtype=torch.float32, device=q.device)
        Pt = torch.empty_like(Pi)
        for hh in range(H):
            sl = slice(hh * hd, (hh + 1) * hd)
            G = lin(W1, Wo[:, sl].t().contiguous())                     # [hid1, hd] = W1 . Wo[:, head]
            Pi[:, hh] = lin(Vi[:, sl].contiguous(), G)
            Pt[:, hh] = lin(Vt[:, sl].contiguous(), G)
        c0 = (lin(bo[None, :], W1)[0] + b1).contiguous()
        w2t = W2.t().contiguous()
        out = torch.empty((N, M), dtype=torch.float32, device=q.device)
        step = max(1, min(N, (256 << 20) // max(1, 2 * H * M * 4)))   # bound the transposed score planes to ~256 MB
        L = _lib.lib()
        for s0 in range(0, N, step):
            Qc = Q[s0:s0 + step]
            nc = Qc.shape[0]
            sti = torch.empty((H, M, nc), dtype=torch.float32, device=q.device)
            stt = torch.empty_like(sti)
            for hh in range(H):
                sl = slice(hh * hd, (hh + 1) * hd)
                qh = Qc[:, sl].contiguous()
                sti[hh] = lin(Ki[:, sl].contiguous(), qh)               # [M, nc]: candidates x queries (transposed)
                stt[hh] = lin(Kt[:, sl].contiguous(), qh)
            out_t = torch.empty((M, nc), dtype=torch.float32, device=q.device)
            with torch.cuda.device(q.device):
                _lib.check(L.kemr_cross_attention_pairs(
                    C.c_void_p(sti.data_ptr()), C.c_void_p(stt.data_ptr()), C.c_void_p(Pi.data_ptr()), C.c_void_p(Pt.data_ptr()),
                    C.c_void_p(c0.data_ptr()), C.c_void_p(w2t.data_ptr()), C.c_void_p(b2.contiguous().data_ptr()),
                    C.c_void_p(W3.contiguous().data_ptr()), b3, H, nc, M, hid1, hid2, C.c_void_p(out_t.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)), "cross_attention_pairs")
            out[s0:s0 + nc] = out_t.t()
        return out

    @torch.no_grad()
    def forward(self, query_embed, image_embed, target_embed) -> torch.Tensor:
        """-> dense [N, M] fused scores (fp32, on the GPU)."""
        if self.fusion_type == "cross_attention":
            q = ranking.to_device_f32(query_embed)
            return self._cross_attention(q, ranking.to_device_f32(image_embed, q.device),
                                         ranking.to_device_f32(target_embed, q.device))
        if self.fusion_type == "linear":
            q = ranking.to_device_f32(query_embed)
            img, tgt = ranking.to_device_f32(image_embed, q.device), ranking.to_device_f32(target_embed, q.device)
            qp = engine.build_panel([q], _lib.SIDE_QUERY, 3)
            t2i = engine.scores_dense(qp, engine.build_panel([img], _lib.SIDE_GALLERY, 3))
            t2t = engine.scores_dense(qp, engine.build_panel([tgt], _lib.SIDE_GALLERY, 3))
            f = self.fusion_head.fusion.to(q.device)
            w0, b0 = f[0].weight.detach().float().contiguous(), f[0].bias.detach().float().contiguous()
            w1, b1 = f[3].weight.detach().float().reshape(-1).contiguous(), float(f[3].bias.detach())
            out = torch.empty_like(t2i)
            with torch.cuda.device(q.device):
                _lib.check(_lib.lib().kemr_linear_head(
                    C.c_void_p(t2i.data_ptr()), C.c_void_p(t2t.data_ptr()), t2i.numel(), C.c_void_p(w0.data_ptr()),
                    C.c_void_p(b0.data_ptr()), C.c_void_p(w1.data_ptr()), b1, w0.shape[0], C.c_void_p(out.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)), "linear_head")
            return out
        qs, gs, weights, gates = self._parts(query_embed, image_embed, target_embed)
        qp = engine.build_panel(qs, _lib.SIDE_QUERY, 3, part_scale=weights, row_scale=gates)
        gp = engine.build_panel(gs, _lib.SIDE_GALLERY, 3)
        return engine.scores_dense(qp, gp)

    @torch.no_grad()
    def rank(self, query_embed, image_embed, target_embed, k: int = 10, gt_idx="diag"
             ) -> Tuple[Optional[torch.Tensor], torch.Tensor, torch.Tensor]:
        """Ranks / top-k under this head without the [N, M] matrix (linear: dense matrix, then a streaming rank)."""
        if self.fusion_type in ("linear", "cross_attention"):
            return ranking.ranks_of_matrix(self.forward(query_embed, image_embed, target_embed), k=k, gt_idx=gt_idx)
        qs, gs, weights, gates = self._parts(query_embed, image_embed, target_embed)
        return ranking.ranks_and_topk(qs, gs, weights=weights, row_gate=gates, k=k, gt_idx=gt_idx)
