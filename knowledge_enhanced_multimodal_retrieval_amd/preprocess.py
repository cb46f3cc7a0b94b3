"""Image preprocessing with CLIP's published recipe, without torchvision (not installed here):
Resize(n, bicubic, shorter side) -> CenterCrop(n) -> RGB -> float in [0,1] -> Normalize(mean, std).
The reference gets this callable from ``clip.load`` and applies it per sample on the host
(/root/reference/src/clip/datasets/clip_dataset.py:110-125); constants as in SURVEY.md section 8 row a16."""
from __future__ import annotations

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ClipPreprocess:
    def __init__(self, n_px: int = 224):
        self.n_px = n_px
        self.mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(CLIP_STD, dtype=torch.float32).view(3, 1, 1)

    def __call__(self, image) -> torch.Tensor:
        from PIL import Image
        n = self.n_px
        w, h = image.size
        if w <= h:                                    # shorter side -> n, longer side int(n * long / short)
            nw, nh = n, int(n * h / w)
        else:
            nw, nh = int(n * w / h), n
        image = image.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
        image = image.crop((left, top, left + n, top + n)).convert("RGB")
        arr = np.asarray(image, dtype=np.uint8)
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
        return (x - self.mean) / self.std

    def __repr__(self):
        return f"ClipPreprocess(n_px={self.n_px})"
