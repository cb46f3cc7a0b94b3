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
* cross_attention: per-pair multi-head attention over two keys + a 3-layer MLP, O(N*M*D^2) -- NOT built yet
  (DESIGN.md "not yet built"); ``forward`` raises ``NotImplementedError`` for it.

``rank()`` gives ranks / top-k without ever forming the [N,M] matrix (all heads except linear / cross_attention).
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
    """Parameter container only (checkpoints load); scoring is not built yet."""

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
            g = h.to(q.device).eval().gate(q).reshape(-1).float()
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

    @torch.no_grad()
    def forward(self, query_embed, image_embed, target_embed) -> torch.Tensor:
        """-> dense [N, M] fused scores (fp32, on the GPU)."""
        if self.fusion_type == "cross_attention":
            raise NotImplementedError("cross_attention fusion scoring is not built yet on the HIP path (DESIGN.md)")
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
        if self.fusion_type == "linear":
            return ranking.ranks_of_matrix(self.forward(query_embed, image_embed, target_embed), k=k, gt_idx=gt_idx)
        if self.fusion_type == "cross_attention":
            raise NotImplementedError("cross_attention fusion scoring is not built yet on the HIP path (DESIGN.md)")
        qs, gs, weights, gates = self._parts(query_embed, image_embed, target_embed)
        return ranking.ranks_and_topk(qs, gs, weights=weights, row_gate=gates, k=k, gt_idx=gt_idx)
