"""Image preprocessing with CLIP's published recipe, without torchvision (not installed here):
Resize(n, bicubic, shorter side) -> CenterCrop(n) -> RGB -> float in [0,1] -> Normalize(mean, std).
The reference gets this callable from ``clip.load`` and applies it per sample on the host
(/root/reference/src/clip/datasets/clip_dataset.py:110-125); constants as in SURVEY.md section 8 row a16."""
from __future__ import annotations

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def gpu_preprocessing_enabled() -> bool:
    """KEMR_GPU_PREPROCESS=0 keeps the reference's arrangement (the PIL transform inside the dataset, per sample)."""
    import os
    return os.environ.get("KEMR_GPU_PREPROCESS", "1") != "0"


class ClipPreprocess:
    """The callable ``clip.load`` / ``load_clip_model`` return: called with a PIL image it is the host transform and returns
    float32 ``[3, n_px, n_px]``, exactly as upstream's.  ``defer_to_gpu``: the object ALSO tells this build's dataset classes
    (``CLIPEvalDatasetHF(ds, preprocess)``, as the reference's ``main`` writes it, evaluator.py:330-333) that they may hand the
    decoded image over raw -- uint8 ``[H, W, 3]`` -- so that ``evaluators.encode_dataset`` runs resize / crop / normalise for a
    whole loader batch in one launch pair on the device (bit-identical, csrc/preprocess.hip).  Code that calls the object itself
    (``preprocess(image)``) never sees a difference."""

    def __init__(self, n_px: int = 224, defer_to_gpu: bool = False):
        self.n_px = n_px
        self.defer_to_gpu = bool(defer_to_gpu)
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
        return f"ClipPreprocess(n_px={self.n_px}, defer_to_gpu={self.defer_to_gpu})"


class ClipPreprocessGPU:
    """The same transform as :class:`ClipPreprocess`, computed by the HIP kernels (csrc/preprocess.hip) and bit-identical
    to it: takes a PIL image (RGB is enforced before the resize here; identical for RGB inputs) or a uint8 ``[H, W, 3]``
    tensor on any device, returns float32 ``[3, n_px, n_px]`` on ``device``.  No CPU fallback."""

    def __init__(self, n_px: int = 224, device="cuda:0"):
        self.n_px = n_px
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ClipPreprocessGPU needs a GPU device")
        self._ws = None

    def __call__(self, image) -> torch.Tensor:
        import ctypes as C
        from . import _lib, engine
        if not torch.is_tensor(image):
            image = torch.from_numpy(np.asarray(image.convert("RGB"), dtype=np.uint8).copy())
        if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
            raise RuntimeError(f"ClipPreprocessGPU expects uint8 [H, W, 3], got {image.dtype} {tuple(image.shape)}")
        img = image.to(self.device).contiguous()
        h, w = int(img.shape[0]), int(img.shape[1])
        if h == 0 and w == 0:          # an undecodable item (datasets.CLIPEvalDatasetHF): zeros after normalisation, as the reference
            return torch.zeros((3, self.n_px, self.n_px), dtype=torch.float32, device=self.device)
        L = _lib.lib()
        need = int(L.kemr_preprocess_workspace_bytes(h, w, self.n_px))
        if need == 0:
            raise RuntimeError(f"ClipPreprocessGPU: unsupported size {h} x {w} -> {self.n_px}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((3, self.n_px, self.n_px), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.kemr_preprocess_u8(C.c_void_p(img.data_ptr()), h, w, self.n_px, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                            C.c_void_p(engine._stream_ptr(self.device))), "preprocess_u8")
        return out

    def batch(self, packed: "PackedRaw") -> torch.Tensor:
        """A whole loader batch in one launch pair (``kemr_preprocess_u8_batch``): ``packed`` holds the uint8 images back
        to back (:func:`pack_raw`); the bytes cross to the device in ONE copy (asynchronous when the buffer is pinned,
        as the DataLoader's pin thread leaves it).  Returns float32 ``[B, 3, n_px, n_px]``, each image bit-identical to
        ``self(image)`` and to the host transform."""
        import ctypes as C
        from . import _lib, engine
        b = len(packed)
        out = torch.empty((b, 3, self.n_px, self.n_px), dtype=torch.float32, device=self.device)
        if b == 0:
            return out
        L = _lib.lib()
        hs, ws, offs = packed.heights, packed.widths, packed.offsets          # int32 / int32 / int64 host arrays
        need = int(L.kemr_preprocess_batch_workspace_bytes(C.c_void_p(hs.ctypes.data), C.c_void_p(ws.ctypes.data), b, self.n_px))
        if need == 0:
            raise RuntimeError(f"ClipPreprocessGPU.batch: unsupported image size in the batch -> {self.n_px}")
        if self._ws is None or self._ws.numel() < need:
            # grow-only, stream-ordered reuse: every user of the old buffer was queued on the same stream before this point
            self._ws = torch.empty(max(need, 2 * (0 if self._ws is None else self._ws.numel())), dtype=torch.uint8, device=self.device)
        data = packed.data if packed.data.device == self.device else packed.data.to(self.device, non_blocking=True)
        with torch.cuda.device(self.device):
            _lib.check(L.kemr_preprocess_u8_batch(C.c_void_p(data.data_ptr()), C.c_void_p(offs.ctypes.data),
                                                  C.c_void_p(hs.ctypes.data), C.c_void_p(ws.ctypes.data), b, self.n_px,
                                                  C.c_void_p(out.data_ptr()), C.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                                  C.c_void_p(engine._stream_ptr(self.device))), "preprocess_u8_batch")
        data.record_stream(torch.cuda.current_stream(self.device))
        return out

    def __repr__(self):
        return f"ClipPreprocessGPU(n_px={self.n_px})"


class PackedRaw:
    """A loader batch of raw images as ONE flat uint8 tensor plus three small host arrays: what the batched preprocess
    kernel reads.  ``data`` is a tensor so that the DataLoader's pin thread pins it; the sizes ride along as tensors too
    and are viewed as numpy on use."""

    def __init__(self, data: torch.Tensor, heights: torch.Tensor, widths: torch.Tensor):
        self.data, self._h, self._w = data, heights, widths

    def __len__(self):
        return int(self._h.numel())

    @property
    def heights(self) -> np.ndarray:
        return np.ascontiguousarray(self._h.numpy(), dtype=np.int32)

    @property
    def widths(self) -> np.ndarray:
        return np.ascontiguousarray(self._w.numpy(), dtype=np.int32)

    @property
    def offsets(self) -> np.ndarray:
        nbytes = self._h.numpy().astype(np.int64) * self._w.numpy().astype(np.int64) * 3
        return np.ascontiguousarray(np.concatenate([[0], np.cumsum(nbytes)[:-1]]), dtype=np.int64)

    def pin_memory(self, device=None):            # the DataLoader's pin thread calls this on custom batch types
        return PackedRaw(self.data.pin_memory(), self._h, self._w)

    def to(self, device=None, non_blocking: bool = False, **_kw):
        """``images.to(device)`` of the reference's loop (evaluator.py:118): the pixel bytes move, the sizes stay host arrays.
        ``CLIP.encode_image`` accepts the result (it runs :meth:`ClipPreprocessGPU.batch` first)."""
        return PackedRaw(self.data.to(device, non_blocking=non_blocking), self._h, self._w)

    @property
    def shape(self):                               # (B,): enough for ``images.shape[0]`` bookkeeping
        return (len(self),)


def pack_raw(images) -> PackedRaw:
    """uint8 ``[H, W, 3]`` host tensors of any sizes -> :class:`PackedRaw` (one memcpy per image, in the loader worker)."""
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or im.device.type != "cpu":
            raise RuntimeError(f"pack_raw expects uint8 [H, W, 3] host tensors, got {im.dtype} {tuple(im.shape)} on {im.device}")
    hs = torch.tensor([int(im.shape[0]) for im in images], dtype=torch.int32)
    ws = torch.tensor([int(im.shape[1]) for im in images], dtype=torch.int32)
    data = torch.cat([im.reshape(-1) for im in images]) if len(images) else torch.empty(0, dtype=torch.uint8)
    return PackedRaw(data, hs, ws)


class RawRGB:
    """Dataset-side half of GPU preprocessing: a PIL image -> uint8 ``[H, W, 3]`` tensor, nothing else.  Give it to a
    dataset as its ``preprocessor`` and ``evaluators.encode_dataset`` runs :class:`ClipPreprocessGPU` on the device: the
    DataLoader workers then only decode, the resize / crop / normalise leaves the host (bit-identical result)."""

    def __call__(self, image) -> torch.Tensor:
        return torch.from_numpy(np.asarray(image.convert("RGB"), dtype=np.uint8).copy())

    def __repr__(self):
        return "RawRGB()"
