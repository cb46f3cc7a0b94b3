"""`CLIP`: an ``nn.Module`` with the attribute / state-dict layout of OpenAI-CLIP whose encoders run in libkemr.so.

The reference treats the model as a duck type: ``.eval()``, ``.float()``,
``next(model.parameters()).device``, ``.encode_image(f32[B,3,224,224])``, ``.encode_text(int[B,77])`` returning
torch tensors, ``.load_state_dict(sd, strict=True)`` / ``.state_dict()`` with OpenAI key names
(/root/reference/src/clip/model/clip_model.py:44-64, 108-119), and the attributes ``visual``, ``transformer``,
``token_embedding``, ``positional_embedding``, ``text_projection``, ``ln_final`` (``clip_model.py:193-216``).
This class holds real ``nn.Parameter``s under exactly those names (fp32 master copy, what checkpoints are saved
from) and packs them into the HIP engine lazily; the forward arithmetic never runs in PyTorch.

Inference only: ``encode_*`` run without autograd.  After changing parameters in place call ``refresh()``.
"""
from __future__ import annotations

import os

from collections import OrderedDict
from typing import Optional

import numpy as np
import torch
from torch import nn

from .config import ClipArch, get_arch
from .engine import ClipEngine


class QuickGELU(nn.Module):
    def forward(self, x):  # pragma: no cover - parameters only; arithmetic runs in the HIP kernels
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    """Parameter container with OpenAI-CLIP names (ln_1, attn.in_proj_*, attn.out_proj.*, ln_2, mlp.c_fc, mlp.c_proj)."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


def _lib_default_precision() -> str:
    from ._lib import DEFAULT_PRECISION
    return DEFAULT_PRECISION


class CLIP(nn.Module):
    def __init__(self, arch: ClipArch, name: str = ""):
        super().__init__()
        self.arch, self.model_name = arch, name
        self.context_length, self.vocab_size = arch.ctx, arch.vocab
        self.visual = VisionTransformer(arch.image_size, arch.patch, arch.v_width, arch.v_layers, arch.v_width // 64,
                                        arch.embed_dim)
        self.transformer = Transformer(arch.t_width, arch.t_layers, arch.t_width // 64)
        self.token_embedding = nn.Embedding(arch.vocab, arch.t_width)
        self.positional_embedding = nn.Parameter(torch.empty(arch.ctx, arch.t_width))
        self.ln_final = nn.LayerNorm(arch.t_width)
        self.text_projection = nn.Parameter(torch.empty(arch.t_width, arch.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self._engine: Optional[ClipEngine] = None
        self._gpu_pre = None
        self._packed_fp = None
        self._dirty = True
        self.initialize_parameters()

    # ------------------------------------------------------------------ init (upstream recipe: keeps activations O(1))
    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        for tower in (self.transformer, self.visual.transformer):
            proj_std = (tower.width ** -0.5) * ((2 * tower.layers) ** -0.5)
            attn_std, fc_std = tower.width ** -0.5, (2 * tower.width) ** -0.5
            for block in tower.resblocks:
                nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
                nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
                nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
                nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.arch.t_width ** -0.5)
        self._dirty = True

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    # ------------------------------------------------------------------ engine plumbing
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        res = super().load_state_dict(sd, strict=strict, assign=assign)
        self._dirty = True
        return res

    def _apply(self, fn, *a, **kw):          # .to() / .float() / .cuda(): the packed copy must follow
        out = super()._apply(fn, *a, **kw)
        self._dirty = True
        return out

    def refresh(self) -> None:
        """Re-pack the current parameter values into the HIP engine at the next encode call.  `engine()` detects by itself what
        autograd's version counters see -- optimizer.step(), `p.add_()` / `p.copy_()` under no_grad, load_state_dict, .to() --
        and re-assigned storage.  It does NOT see a write through `p.data` (`p.data.copy_(w)`, `p.data.mul_(d)`: `.data` is a
        detached alias with a version counter of its own, the parameter's stays put -- EMA and hand-rolled weight loading do
        this): call `refresh()` after such a write, or set KEMR_WEIGHT_DIGEST=1, which adds a content digest (one fused norm
        over all parameters and one host read, about 0.1 ms) to every `engine()` call."""
        self._dirty = True

    def _fingerprint(self):
        """Cheap identity of the parameter values: in-place edits of the parameter itself (optimizer.step(), `p.add_()`,
        `p.copy_()`) bump its version counter, re-assignment changes its storage; see `refresh()` for what this misses.  The
        reference's trainer validates through `evaluate_clip_model_for_training` right after optimizer.step()
        (train/trainer.py:241)."""
        fp = tuple((p._version, p.data_ptr()) for p in self.parameters())
        if os.environ.get("KEMR_WEIGHT_DIGEST", "") == "1":
            params = [p.detach() for p in self.parameters()]
            digest = torch.stack(torch._foreach_norm(params)).double().cpu()       # content digest: catches writes through .data
            fp = fp + (tuple(digest.tolist()),)
        return fp

    def __deepcopy__(self, memo):
        """The packed engine is a raw library handle: a copy gets its own, built lazily (never two owners of one handle)."""
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_packed_fp", "_gpu_pre") else copy.deepcopy(v, memo)
        new._dirty = True
        return new

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"], state["_packed_fp"], state["_dirty"] = None, None, True
        state["_gpu_pre"] = None
        return state

    def engine(self) -> ClipEngine:
        dev = self.visual.proj.device
        if dev.type != "cuda":
            raise RuntimeError("CLIP: the model sits on %s; the encoders run only on a GPU (model.to('cuda')); there is no "
                               "CPU fallback" % dev)
        if self._engine is None or self._engine.device != dev:
            # encoder precision of the packed copy: "bf16" (fp32 residual stream) unless KEMR_PRECISION says otherwise (bf16-res16 | fp8 | fp8-res16 | fp8-mlp,
            # kemr_precision in include/kemr.h) -- an environment switch so that the reference's scripts stay unchanged
            self._engine, self._dirty = ClipEngine(self.arch, dev, precision=os.environ.get("KEMR_PRECISION", _lib_default_precision())), True
        fp = self._fingerprint()
        if self._dirty or fp != getattr(self, "_packed_fp", None):
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items() if k != "logit_scale"})
            self._dirty, self._packed_fp = False, fp
        return self._engine

    # ------------------------------------------------------------------ the duck-typed API
    @torch.no_grad()
    def encode_image(self, image: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """``image``: float ``[B, 3, S, S]`` normalised pixels as upstream, or the :class:`preprocess.PackedRaw` batch a loader over
        ``CLIPEvalDatasetHF(split, preprocess)`` yields when the transform is deferred to the GPU -- so the reference's own loop
        (``images.to(device)`` -> ``model.encode_image(images)``, evaluator.py:118-121) runs unchanged on either."""
        from .preprocess import ClipPreprocessGPU, PackedRaw
        dev = self.visual.proj.device
        if isinstance(image, PackedRaw):
            if getattr(self, "_gpu_pre", None) is None or self._gpu_pre.device != dev:
                self._gpu_pre = ClipPreprocessGPU(self.visual.input_resolution, dev)
            image = self._gpu_pre.batch(image)
        return self.engine().encode_image(image.to(dev), normalize=normalize)

    accepts_text_lengths = True        # encode_text(..., lens=host int tensor): evaluators.encode_dataset passes the tokenizer-side lengths

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor, normalize: bool = False, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Host token ids (what ``tokenize`` returns) are handed over as they are: the engine reads the text lengths from them on the
        host before the upload (engine.ClipEngine.encode_text); ids already on the GPU cost one small device-to-host copy unless the
        caller passes ``lens`` (``engine.text_lengths(host_tokens)``)."""
        return self.engine().encode_text(text, normalize=normalize, lens=lens)

    @torch.no_grad()
    def forward(self, image: torch.Tensor, text: torch.Tensor):
        """logits_per_image, logits_per_text (cosine similarities times exp(logit_scale)), as upstream CLIP.forward."""
        from . import _lib
        from . import engine as E
        img = self.encode_image(image, normalize=True)
        txt = self.encode_text(text, normalize=True)
        qp = E.build_panel([img], _lib.SIDE_QUERY, 3)
        gp = E.build_panel([txt], _lib.SIDE_GALLERY, 3)
        logits = E.scores_dense(qp, gp) * self.logit_scale.exp().to(img.device)
        return logits, logits.t()


def build_model(name_or_arch, device="cuda") -> CLIP:
    arch = name_or_arch if isinstance(name_or_arch, ClipArch) else get_arch(name_or_arch)
    name = name_or_arch if isinstance(name_or_arch, str) else ""
    return CLIP(arch, name).to(device).eval()
