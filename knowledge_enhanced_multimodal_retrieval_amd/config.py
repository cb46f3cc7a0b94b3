"""Architecture presets of the CLIP models the reference evaluates
(`--model_name` choices at /root/reference/src/clip/eval/evaluator.py:264-266: ViT-B/32, ViT-B/16, ViT-L/14;
embed_dim rule at src/clip/eval/evaluator_fusion.py:192)."""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict


@dataclass(frozen=True)
class ClipArch:
    embed_dim: int
    image_size: int
    patch: int
    v_width: int
    v_layers: int
    t_width: int
    t_layers: int
    vocab: int = 49408
    ctx: int = 77

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def v_tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def sot(self) -> int:
        return self.vocab - 2

    @property
    def eot(self) -> int:
        return self.vocab - 1

    def as_dict(self) -> Dict[str, int]:
        return asdict(self)

    # algorithmic FLOPs per item
    def image_flops(self) -> float:
        t, w, p = self.v_tokens, self.v_width, self.grid * self.grid
        per_layer = t * w * 3 * w + t * w * w + 2 * t * t * w + 2 * t * w * 4 * w
        return 2.0 * (p * 3 * self.patch * self.patch * w + self.v_layers * per_layer + w * self.embed_dim)

    def text_flops(self) -> float:
        t, w = self.ctx, self.t_width
        per_layer = t * w * 3 * w + t * w * w + 2 * t * t * w + 2 * t * w * 4 * w
        return 2.0 * (self.t_layers * per_layer + w * self.embed_dim)


ARCHS: Dict[str, ClipArch] = {
    "ViT-L/14": ClipArch(768, 224, 14, 1024, 24, 768, 12),
    "ViT-B/16": ClipArch(512, 224, 16, 768, 12, 512, 12),
    "ViT-B/32": ClipArch(512, 224, 32, 768, 12, 512, 12),
    # small shapes for tests (same structure, head dim 64)
    "tiny": ClipArch(128, 32, 8, 256, 2, 256, 2, vocab=512, ctx=16),
    "tiny-long": ClipArch(256, 112, 8, 256, 3, 512, 3, vocab=1024, ctx=77),
}


def get_arch(name: str) -> ClipArch:
    if name not in ARCHS:
        raise RuntimeError(f"Model {name} not found; available models = {list(ARCHS)}")
    return ARCHS[name]
