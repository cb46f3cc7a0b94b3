"""Host-side driver of libkemr.so: model handle, encoders and the fused similarity / top-k / rank ops.

PyTorch is plumbing here (device memory, streams): every tensor that crosses into the library is passed as
``tensor.data_ptr()`` and all arithmetic of the hot path runs in the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Mapping, Optional, Sequence, Tuple

import torch

from . import _lib
from .config import ClipArch

# rows per encoder launch: picked so that batch * tokens fills whole 128/256-row GEMM tiles and the
# workspace (14 * width bytes per token) stays small; larger user batches are processed in slices.
MAX_IMAGE_BATCH = 255
MAX_TEXT_BATCH = 851
TEXT_ROW_BUDGET = 65536        # token rows per packed text call: 256 row tiles of 256


def text_lengths(ids: torch.Tensor) -> torch.Tensor:
    """Positions of every text that can reach its embedding: up to and including the pooled one, ``ids.argmax(-1) + 1`` (the
    end-of-text token has the largest id; reference pooling ``x[arange, text.argmax(dim=-1)]``).  int32 [B], on ids' device."""
    return (ids.argmax(dim=-1) + 1).to(torch.int32)


def tile_friendly_batch(tokens: int, width: int, lo: int, hi: int, num_cu: int = 256) -> int:
    """Items per encoder call, in [lo, hi], that fill the persistent GEMM's rounds best: the towers' four GEMMs have
    ceil(items * tokens / 256) x (3W | W | 4W | W) / 256 output tiles each and run ceil(tiles / CUs) rounds over the CUs, so a
    call is as fast as its emptiest round allows (255 texts of ViT-L/14: 231 out-proj tiles on 256 CUs, 90 %; 564 texts: 510
    tiles in two rounds, 99.6 %).  Weighted by the GEMMs' FLOPs; among near-equal sizes the largest (fewer launches per item:
    measured 64.6 k texts/s at 255, 66.6 k at 282, 71.4 k at 564; whole step 16 340 / 16 590 / 16 710 items/s at 255 / 565 / 848)."""
    effs = {}
    for items in range(lo, hi + 1):
        row_tiles = -(-items * tokens // 256)
        num = den = 0.0
        for n_mult, weight in ((3, 3.0), (1, 1.0), (4, 4.0), (1, 4.0)):      # QKV, out-proj, fc1, fc2 (K = 4W)
            tiles = row_tiles * (n_mult * width // 256)
            num += weight * (items * tokens / 256.0) * (n_mult * width / 256.0)      # useful tile-equivalents
            den += weight * -(-tiles // num_cu) * num_cu                             # tile slots of the rounds it takes
        effs[items] = num / den
    top = max(effs.values())
    return max(i for i, e in effs.items() if e >= top - 2e-3)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU (got device {t.device}); this path has no CPU fallback")


class ClipEngine:
    """One packed CLIP model (both towers) in HBM."""

    def __init__(self, arch: ClipArch, device: torch.device | str = "cuda:0", precision: str = _lib.DEFAULT_PRECISION):
        """precision: "bf16-x24" (default: bf16 operands, fp32 accumulation, the fp32 residual stream stored as 24-bit floats -- model
        option residual_stream_24bit), "bf16" (the stream as 4-byte fp32), "bf16-res16" (bf16 residual stream, opt-in), "fp8" / "fp8-x24"
        (the vision tower's QKV GEMMs on fp8 operands, BASELINE config 5), "fp8-res16" or "fp8-mlp" (fc1 too).  See kemr_precision in
        include/kemr.h and _lib.DEFAULT_PRECISION."""
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self.arch = arch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ClipEngine needs a GPU device; the HIP path has no CPU fallback")
        self._L = _lib.lib()
        cfg = _lib.KemrCfg(**arch.as_dict())
        h = C.c_void_p()
        _lib.check(self._L.kemr_model_create(C.byref(cfg), C.byref(h)), "model_create")
        self._h = h
        self._ws: Dict[object, torch.Tensor] = {}
        self.ready = False
        self.pack_text = os.environ.get("KEMR_TEXT_PACKED", "1") != "0"     # encode_text: only the positions up to the end-of-text token

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._L.kemr_model_destroy(h)
            except Exception:
                pass

    def set_residual_fusion(self, level) -> None:
        """Option "residual_fusion" of THIS model (include/kemr.h kemr_model_set_option): 0 / False = the out-proj / fc2 GEMMs store
        bf16 updates that the LayerNorms apply; 1 / True = bf16 residual streams add them in the GEMM epilogues (default); 2 = fp32
        residual streams too (nothing is rounded; slower)."""
        _lib.check(self._L.kemr_model_set_option(self._h, b"residual_fusion", int(level)), "model_set_option")

    def set_last_block_pooled_row(self, on: bool) -> None:
        """Option "last_block_pooled_row" of THIS model (default on): the last block of a tower computes its query path -- attention
        output, out-proj, ln_2, MLP -- for the one row per item that leaves the tower (class / end-of-text token) instead of for all
        of them; off = every row, as the reference does.  Store-only epilogues only (not the bf16-stream modes' fused residual add), and
        not with fc1 on fp8 ("fp8-mlp")."""
        _lib.check(self._L.kemr_model_set_option(self._h, b"last_block_pooled_row", 1 if on else 0), "model_set_option")

    def last_block_pooled_row(self) -> bool:
        v = C.c_int(0)
        _lib.check(self._L.kemr_model_get_option(self._h, b"last_block_pooled_row", C.byref(v)), "model_get_option")
        return bool(v.value)

    def residual_fusion(self) -> int:
        """The option's value (0 / 1 / 2)."""
        v = C.c_int(0)
        _lib.check(self._L.kemr_model_get_option(self._h, b"residual_fusion", C.byref(v)), "model_get_option")
        return v.value

    def residual_fusion_active(self) -> bool:
        """Whether large calls of THIS engine add the residual inside the GEMM epilogues (option and stream type together)."""
        if self.precision.endswith("-x24"):
            return False                                   # 24-bit stream rows: store-only epilogues, whatever the option says
        return self.residual_fusion() >= (1 if self.precision.endswith("res16") else 2)

    # ------------------------------------------------------------------ weights
    def tensor_names(self) -> Sequence[str]:
        n = self._L.kemr_model_num_tensors(self._h)
        return [self._L.kemr_model_tensor_name(self._h, i).decode() for i in range(n)]

    def load_state_dict(self, sd: Mapping[str, torch.Tensor]) -> None:
        """Strict load of an OpenAI-CLIP style state dict (any float dtype, any device) + pack to HBM."""
        for name, t in sd.items():
            if not torch.is_tensor(t):
                continue
            host = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (C.c_int64 * max(host.dim(), 1))(*host.shape)
            _lib.check(self._L.kemr_model_load_tensor(self._h, name.encode(), C.c_void_p(host.data_ptr()), _lib.KEMR_F32,
                                                      shape, host.dim()), f"load_tensor({name})")
        # "-x24" names the 24-bit stream (the product's default), "bf16" / "fp8" / "fp8-mlp" the 4-byte one; idle for the bf16 streams
        _lib.check(self._L.kemr_model_set_option(self._h, b"residual_stream_24bit", 1 if self.precision.endswith("-x24") else 0), "model_set_option")
        with torch.cuda.device(self.device):
            _lib.check(self._L.kemr_model_finalize(self._h, _lib.PRECISIONS[self.precision]), "model_finalize")
        self.ready = True

    # ------------------------------------------------------------------ encoders
    def _workspace(self, tower: int, batch: int) -> torch.Tensor:
        need = int(self._L.kemr_workspace_bytes(self._h, tower, batch))
        ws = self._ws.get(tower)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[tower] = ws
        return ws

    def _encode(self, fn, tower: int, x: torch.Tensor, max_batch: int, normalize: bool) -> torch.Tensor:
        if not self.ready:
            raise RuntimeError("ClipEngine: load_state_dict() must be called before encoding")
        n = x.shape[0]
        out = torch.empty((n, self.arch.embed_dim), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        with torch.cuda.device(self.device):
            stream = _stream_ptr(self.device)
            ws = self._workspace(tower, min(n, max_batch))
            for s in range(0, n, max_batch):
                xb = x[s:s + max_batch]
                _lib.check(fn(self._h, C.c_void_p(xb.data_ptr()), xb.shape[0], C.c_void_p(out[s:].data_ptr()),
                              1 if normalize else 0, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream)),
                           "encode")
        return out

    def encode_image(self, pixels: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        a = self.arch
        _require_cuda(pixels, "pixels")
        if pixels.dim() != 4 or tuple(pixels.shape[1:]) != (3, a.image_size, a.image_size):
            raise RuntimeError(f"encode_image expects [B,3,{a.image_size},{a.image_size}], got {tuple(pixels.shape)}")
        pixels = pixels.to(dtype=torch.float32).contiguous()
        return self._encode(self._L.kemr_encode_image, _lib.TOWER_VISION, pixels, MAX_IMAGE_BATCH, normalize)

    def encode_text(self, ids: torch.Tensor, normalize: bool = False, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``model.encode_text(tokens)`` (+ the optional L2 normalisation).  The rows behind a text's end-of-text token cannot reach
        its embedding (causal mask, pooled at ``tokens.argmax(-1)``), so by default only the first ``lens[i] = argmax_i + 1``
        positions of every text are computed, packed one text behind the other (kemr_encode_text_packed: the same embeddings --
        up to the summation order of the GEMM kernel a launch of that many rows is routed to, as with another batch size -- from
        sum(lens) instead of B * ctx token rows in every launch).  The lengths must be known on the HOST to size the launches:

        * ``ids`` on the host (what a tokenizer returns): lengths are taken from it there, then it is uploaded -- no device sync;
        * ``ids`` on the GPU with ``lens`` (host int tensor [B], ``text_lengths(host_tokens)``): no device sync;
        * ``ids`` on the GPU alone: one small device-to-host copy (B ints) per call, unless ``KEMR_TEXT_PACK_SYNC=0``, which sends
          such calls through the full-context kernel instead.

        ``KEMR_TEXT_PACKED=0`` (or ``self.pack_text = False``) computes every position, as the reference does."""
        a = self.arch
        if ids.dim() != 2 or ids.shape[1] != a.ctx:
            raise RuntimeError(f"encode_text expects [B,{a.ctx}] token ids, got {tuple(ids.shape)}")
        if not self.ready:
            raise RuntimeError("ClipEngine: load_state_dict() must be called before encoding")
        pack = self.pack_text and a.ctx <= 128
        if pack and not ids.is_cuda:
            # host ids: the lengths are free to compute; caller-supplied ones may only LENGTHEN a text (a length short of the
            # end-of-text token would pool the wrong row: ADVICE r3)
            own = text_lengths(ids)
            if lens is None:
                lens = own
            else:
                given = torch.as_tensor(lens, device="cpu").reshape(-1).to(own.dtype)
                if given.numel() != own.numel():
                    raise RuntimeError(f"encode_text: {given.numel()} lengths for {own.numel()} texts")
                lens = torch.maximum(given, own)
        ids = ids.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
        _require_cuda(ids, "token ids")
        if pack and lens is None and os.environ.get("KEMR_TEXT_PACK_SYNC", "1") != "0":
            lens = text_lengths(ids).cpu()                    # the one synchronising copy of this path
        if not pack or lens is None:
            return self._encode(self._L.kemr_encode_text, _lib.TOWER_TEXT, ids, MAX_TEXT_BATCH, normalize)
        n = ids.shape[0]
        lens = torch.as_tensor(lens, dtype=torch.int32, device="cpu").reshape(-1).clamp(1, a.ctx).contiguous()
        if lens.numel() != n:
            raise RuntimeError(f"encode_text: {lens.numel()} lengths for {n} texts")
        out = torch.empty((n, a.embed_dim), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        lens_dev = lens.to(self.device, non_blocking=True)
        # calls of at most TEXT_ROW_BUDGET token rows (256 row tiles of the persistent GEMM: whole rounds, engine.tile_friendly_batch)
        csum = torch.cumsum(lens.to(torch.int64), 0).tolist()
        with torch.cuda.device(self.device):
            stream = _stream_ptr(self.device)
            s0, base = 0, 0
            while s0 < n:
                s1 = s0 + 1
                while s1 < n and csum[s1] - base <= TEXT_ROW_BUDGET and s1 - s0 < 65528:
                    s1 += 1
                rows = csum[s1 - 1] - base
                need = int(self._L.kemr_text_packed_workspace_bytes(self._h, rows, s1 - s0))
                ws = self._ws.get("text_packed")
                if ws is None or ws.numel() < need:
                    ws = self._ws["text_packed"] = torch.empty(max(need, int(self._L.kemr_text_packed_workspace_bytes(
                        self._h, min(TEXT_ROW_BUDGET, n * a.ctx), min(n, 65528)))), dtype=torch.uint8, device=self.device)
                _lib.check(self._L.kemr_encode_text_packed(self._h, C.c_void_p(ids[s0:].data_ptr()), C.c_void_p(lens_dev[s0:].data_ptr()),
                                                           rows, s1 - s0, C.c_void_p(out[s0:].data_ptr()), 1 if normalize else 0,
                                                           C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream)), "encode_text_packed")
                s0, base = s1, csum[s1 - 1]
        return out


# ====================================================================== similarity / ranking ops

def rows_alloc(rows: int) -> int:
    return (rows + 255) // 256 * 256


class Panel:
    """bf16 operand of the fused similarity kernels: [rows (padded to 256), kdim]."""

    def __init__(self, data: torch.Tensor, rows: int, kdim: int, terms: int, side: int):
        self.data, self.rows, self.kdim, self.terms, self.side = data, rows, kdim, terms, side

    @property
    def device(self):
        return self.data.device


def build_panel(parts: Sequence[torch.Tensor], side: int, terms: int = 3,
                part_scale: Optional[Sequence[float]] = None,
                row_scale: Optional[Sequence[Optional[torch.Tensor]]] = None) -> Panel:
    """Concatenate weighted embedding sets along k and split to bf16 (see include/kemr.h)."""
    L = _lib.lib()
    parts = [p.to(torch.float32).contiguous() for p in parts]
    for p in parts:
        _require_cuda(p, "embeddings")
    rows, d = parts[0].shape
    if any(tuple(p.shape) != (rows, d) for p in parts):
        raise RuntimeError("build_panel: all parts must share one [rows, d] shape")
    n = len(parts)
    kdim = int(L.kemr_panel_kdim(d, n, terms))
    if kdim <= 0:
        raise RuntimeError(f"build_panel: unsupported d={d} nparts={n} terms={terms}")
    dev = parts[0].device
    out = torch.empty((rows_alloc(max(rows, 1)), kdim), dtype=torch.bfloat16, device=dev)
    pp = (C.c_void_p * n)(*[p.data_ptr() for p in parts])
    ps = (C.c_float * n)(*(list(part_scale) if part_scale is not None else [1.0] * n))
    keep = []
    if row_scale is not None:
        rs = []
        for r in row_scale:
            if r is None:
                rs.append(None)
            else:
                r = r.to(device=dev, dtype=torch.float32).contiguous().view(-1)
                if r.numel() != rows:
                    raise RuntimeError("build_panel: row_scale must have one entry per row")
                keep.append(r)
                rs.append(r.data_ptr())
        prs = (C.c_void_p * n)(*rs)
    else:
        prs = None
    with torch.cuda.device(dev):
        _lib.check(L.kemr_panel_build(pp, ps, prs, n, rows, d, terms, side, C.c_void_p(out.data_ptr()),
                                      C.c_void_p(_stream_ptr(dev))), "panel_build")
    return Panel(out, rows, kdim, terms, side)


def _opt_ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def pair_scores(qp: Panel, gp: Panel, q_rows: torch.Tensor, g_rows: torch.Tensor) -> torch.Tensor:
    L = _lib.lib()
    q_rows = q_rows.to(device=qp.device, dtype=torch.int32).contiguous()
    g_rows = g_rows.to(device=qp.device, dtype=torch.int32).contiguous()
    n = q_rows.numel()
    out = torch.empty(n, dtype=torch.float32, device=qp.device)
    if n:
        with torch.cuda.device(qp.device):
            _lib.check(L.kemr_pair_scores(C.c_void_p(qp.data.data_ptr()), C.c_void_p(gp.data.data_ptr()), qp.kdim,
                                          C.c_void_p(q_rows.data_ptr()), C.c_void_p(g_rows.data_ptr()), n,
                                          C.c_void_p(out.data_ptr()), C.c_void_p(_stream_ptr(qp.device))), "pair_scores")
    return out


def sim_topk(qp: Panel, gp: Panel, k: int, gallery_offset: int = 0,
             gt_idx: Optional[torch.Tensor] = None, gt_score: Optional[torch.Tensor] = None,
             ahead: Optional[torch.Tensor] = None,
             bonus: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
             return_workspace: bool = False):
    """Fused scores + top-k (+ `ahead` counts when a ground truth is given). Returns (scores [nq,k], ids [nq,k]).
    k == 0 = rank only (needs the ground truth; the returned tensors are empty).  return_workspace (tools/): the call's
    scratch tensor as a third result (debug.sim_lists reads the candidate-list statistics out of it); otherwise the scratch
    goes back to the caching allocator with the call (2.3 GB at 43 k x 43 k)."""
    L = _lib.lib()
    if qp.kdim != gp.kdim:
        raise RuntimeError(f"sim_topk: panel kdim mismatch ({qp.kdim} vs {gp.kdim})")
    dev = qp.device
    nq, ng = qp.rows, gp.rows
    top_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    top_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    if nq == 0:
        return top_s, top_i
    if ng == 0:
        return top_s.fill_(float("-inf")), top_i.fill_(-1)
    ws = torch.empty(max(int(L.kemr_sim_workspace_bytes(nq, ng, qp.kdim, k)), 256), dtype=torch.uint8, device=dev)
    if gt_idx is not None:
        if gt_score is None or ahead is None:
            raise RuntimeError("sim_topk: gt_idx needs gt_score and ahead")
        gt_idx = gt_idx.to(device=dev, dtype=torch.int32).contiguous()
        gt_score = gt_score.to(device=dev, dtype=torch.float32).contiguous()
        if ahead.dtype != torch.int32 or not ahead.is_contiguous() or ahead.device != dev:
            raise RuntimeError("sim_topk: ahead must be a contiguous int32 tensor on the panel's device")
    b_ptr = b_col = b_val = None
    if bonus is not None and len(bonus[1]) == 0:      # no hit at all: same as no bonus (empty tensors have no address)
        bonus = None
    if bonus is not None:
        b_ptr, b_col, b_val = (torch.as_tensor(bonus[0]).to(device=dev, dtype=torch.int32).contiguous(),
                               torch.as_tensor(bonus[1]).to(device=dev, dtype=torch.int32).contiguous(),
                               torch.as_tensor(bonus[2]).to(device=dev, dtype=torch.float32).contiguous())
        if b_ptr.numel() != nq + 1:
            raise RuntimeError("sim_topk: bonus row pointer must have nq + 1 entries")
    with torch.cuda.device(dev):
        _lib.check(L.kemr_sim_topk(C.c_void_p(qp.data.data_ptr()), nq, C.c_void_p(gp.data.data_ptr()), ng, qp.kdim,
                                   gallery_offset, k, C.c_void_p(top_s.data_ptr()), C.c_void_p(top_i.data_ptr()),
                                   _opt_ptr(gt_idx), _opt_ptr(gt_score), _opt_ptr(ahead),
                                   _opt_ptr(b_ptr), _opt_ptr(b_col), _opt_ptr(b_val),
                                   C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(_stream_ptr(dev))), "sim_topk")
    if return_workspace:
        return top_s, top_i, ws
    return top_s, top_i


def topk_merge(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """[nq, nlists, k] candidate lists -> [nq, k] (score desc, id asc)."""
    L = _lib.lib()
    _require_cuda(scores, "scores")
    nq, nlists, kk = scores.shape
    if kk != k or tuple(idx.shape) != tuple(scores.shape):
        raise RuntimeError("topk_merge: expected scores/idx of shape [nq, nlists, k]")
    scores = scores.to(torch.float32).contiguous()
    idx = idx.to(torch.int32).contiguous()
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=scores.device)
    if nq:
        with torch.cuda.device(scores.device):
            _lib.check(L.kemr_topk_merge(C.c_void_p(scores.data_ptr()), C.c_void_p(idx.data_ptr()), nq, nlists, k,
                                         C.c_void_p(out_s.data_ptr()), C.c_void_p(out_i.data_ptr()),
                                         C.c_void_p(_stream_ptr(scores.device))), "topk_merge")
    return out_s, out_i


def scores_dense(qp: Panel, gp: Panel) -> torch.Tensor:
    """Dense fp32 score matrix [nq, ng] (fusion heads that need every pair; debugging)."""
    L = _lib.lib()
    if qp.kdim != gp.kdim:
        raise RuntimeError("scores_dense: panel kdim mismatch")
    out = torch.empty((qp.rows, gp.rows), dtype=torch.float32, device=qp.device)
    if qp.rows and gp.rows:
        with torch.cuda.device(qp.device):
            _lib.check(L.kemr_scores_dense(C.c_void_p(qp.data.data_ptr()), qp.rows, C.c_void_p(gp.data.data_ptr()), gp.rows,
                                           qp.kdim, C.c_void_p(out.data_ptr()), gp.rows,
                                           C.c_void_p(_stream_ptr(qp.device))), "scores_dense")
    return out


def rank_dense(scores: torch.Tensor, gt_idx: Optional[torch.Tensor] = None, k: int = 0
               ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Rank a materialised fp32 score matrix on the GPU: returns (ahead [nq] or None, top scores, top ids)."""
    L = _lib.lib()
    _require_cuda(scores, "score matrix")
    scores = scores.to(torch.float32)
    if scores.stride(-1) != 1:
        scores = scores.contiguous()
    nq, ng = scores.shape
    dev = scores.device
    ahead = gt = top_s = top_i = None
    if gt_idx is not None:
        gt = gt_idx.to(device=dev, dtype=torch.int32).contiguous()
        ahead = torch.zeros(nq, dtype=torch.int32, device=dev)
    if k > 0:
        top_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
        top_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    if nq and ng:
        with torch.cuda.device(dev):
            _lib.check(L.kemr_rank_dense(C.c_void_p(scores.data_ptr()), nq, ng, scores.stride(0), _opt_ptr(gt), _opt_ptr(ahead),
                                         k, _opt_ptr(top_s), _opt_ptr(top_i), C.c_void_p(_stream_ptr(dev))), "rank_dense")
    return ahead, top_s, top_i


# ---------------------------------------------------------------------- per-kernel hooks used by tests
def set_gemm_variant(variant: int) -> None:
    """tools/ and tests/: see debug.set_gemm_variant (include/kemr_debug.h; nothing in the product path calls this)."""
    from . import debug
    debug.set_gemm_variant(variant)


def op_gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], m: int, epilogue: int,
            c: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a: bf16 [m_alloc, k] (m_alloc multiple of 256), w: bf16 [n, k]; returns C (bf16 [m_alloc, n] or the fp32 residual)."""
    L = _lib.lib()
    n, k = w.shape
    if c is None:
        c = torch.zeros((a.shape[0], n), dtype=torch.bfloat16 if epilogue != _lib.EPI_BIAS_RESID_F32 else torch.float32,
                        device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(L.kemr_op_gemm(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), _opt_ptr(bias), C.c_void_p(c.data_ptr()),
                                  m, n, k, epilogue, C.c_void_p(_stream_ptr(a.device))), "op_gemm")
    return c


def op_gemm_fp8(a: torch.Tensor, w: torch.Tensor, wscale: torch.Tensor, bias: Optional[torch.Tensor], m: int,
                epilogue: int) -> torch.Tensor:
    """a, w: torch.float8_e4m3fn [ceil256(m), k] / [n, k]; returns bf16 [ceil256(m), n] = epi(a.w^T * wscale + bias)."""
    L = _lib.lib()
    n, k = w.shape
    c = torch.zeros((a.shape[0], n), dtype=torch.bfloat16, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(L.kemr_op_gemm_fp8(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(wscale.data_ptr()),
                                      _opt_ptr(bias), C.c_void_p(c.data_ptr()), m, n, k, epilogue,
                                      C.c_void_p(_stream_ptr(a.device))), "op_gemm_fp8")
    return c


def op_layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_bf16: bool = True) -> torch.Tensor:
    L = _lib.lib()
    rows, width = x.shape
    y = torch.empty((rows, width), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.kemr_op_layernorm(C.c_void_p(x.data_ptr()), C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()),
                                       C.c_void_p(y.data_ptr()), rows, width, _lib.KEMR_BF16 if out_bf16 else _lib.KEMR_F32,
                                       C.c_void_p(_stream_ptr(x.device))), "op_layernorm")
    return y


def op_layernorm_resid(x: torch.Tensor, delta: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """x (fp32, updated in place) += delta (bf16); returns LayerNorm(x) as bf16."""
    L = _lib.lib()
    rows, width = x.shape
    y = torch.empty((rows, width), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.kemr_op_layernorm_resid(C.c_void_p(x.data_ptr()), C.c_void_p(delta.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                             C.c_void_p(beta.data_ptr()), C.c_void_p(y.data_ptr()), rows, width,
                                             C.c_void_p(_stream_ptr(x.device))), "op_layernorm_resid")
    return y


def op_layernorm_rows(x: torch.Tensor, delta: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                      out_bf16: bool = True, delta2: Optional[torch.Tensor] = None, writeback: bool = True,
                      out_fp8: bool = False) -> torch.Tensor:
    """General form: x is fp32 or bf16 rows; returns LayerNorm(x [+ delta [+ delta2]]) (deltas bf16); with `writeback`
    the sum is stored back into x in x's dtype."""
    L = _lib.lib()
    rows, width = x.shape
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("op_layernorm_rows: x must be fp32 or bf16")
    ydt = torch.float8_e4m3fn if out_fp8 else (torch.bfloat16 if out_bf16 else torch.float32)
    y = torch.empty((rows, width), dtype=ydt, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.kemr_op_layernorm_rows(C.c_void_p(x.data_ptr()), _lib.KEMR_BF16 if x.dtype == torch.bfloat16 else _lib.KEMR_F32,
                                            _opt_ptr(delta), _opt_ptr(delta2), 1 if writeback else 0,
                                            C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()),
                                            C.c_void_p(y.data_ptr()), rows, width,
                                            _lib.KEMR_FP8 if out_fp8 else (_lib.KEMR_BF16 if out_bf16 else _lib.KEMR_F32),
                                            C.c_void_p(_stream_ptr(x.device))), "op_layernorm_rows")
    return y


def pack_f24_rows(x: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, W] -> the library's 24-bit-float rows (csrc/common.h f24_t): uint8 [rows, 3 W], a row = its W bf16 upper halves
    (little endian) followed by its W third bytes; round to nearest on the 24 kept bits.  Tests / tools."""
    bits = (x.contiguous().view(torch.int32).to(torch.int64) & 0xffffffff) + 0x80
    hi = ((bits >> 16) & 0xffff)
    lo = ((bits >> 8) & 0xff).to(torch.uint8)
    hi_bytes = torch.stack([(hi & 0xff).to(torch.uint8), (hi >> 8).to(torch.uint8)], dim=-1).reshape(x.shape[0], -1)
    return torch.cat([hi_bytes, lo], dim=1).contiguous()


def unpack_f24_rows(rows24: torch.Tensor, width: int) -> torch.Tensor:
    """The inverse of pack_f24_rows (exact): uint8 [rows, 3 W] -> fp32 [rows, W]."""
    hb = rows24[:, :2 * width].reshape(-1, width, 2).to(torch.int64)
    lo = rows24[:, 2 * width:].to(torch.int64)
    bits = (hb[..., 1] << 24) | (hb[..., 0] << 16) | (lo << 8)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits)
    return bits.to(torch.int32).view(torch.float32)


def op_layernorm_rows_f24(x24: torch.Tensor, width: int, delta: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                          delta2: Optional[torch.Tensor] = None, writeback: bool = True) -> torch.Tensor:
    """op_layernorm_rows on 24-bit-float rows (pack_f24_rows layout, updated in place with `writeback`); bf16 output."""
    L = _lib.lib()
    rows = x24.shape[0]
    y = torch.empty((rows, width), dtype=torch.bfloat16, device=x24.device)
    with torch.cuda.device(x24.device):
        _lib.check(L.kemr_op_layernorm_rows(C.c_void_p(x24.data_ptr()), 24, _opt_ptr(delta), _opt_ptr(delta2), 1 if writeback else 0,
                                            C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), C.c_void_p(y.data_ptr()), rows, width,
                                            _lib.KEMR_BF16, C.c_void_p(_stream_ptr(x24.device))), "op_layernorm_rows")
    return y


def op_attention(qkv: torch.Tensor, batch: int, t: int, width: int, causal: bool) -> torch.Tensor:
    L = _lib.lib()
    out = torch.empty((batch * t, width), dtype=torch.bfloat16, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(L.kemr_op_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), batch, t, width,
                                       1 if causal else 0, C.c_void_p(_stream_ptr(qkv.device))), "op_attention")
    return out
