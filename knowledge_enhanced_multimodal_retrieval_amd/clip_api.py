"""`clip.load` / `clip.tokenize` / `clip.available_models` of the third-party ``clip`` package, served by this build.

The reference imports ``clip`` everywhere it touches the model (/root/reference/src/clip/model/clip_model.py:7,41;
src/clip/eval/evaluator_baseline.py:19,112; src/clip/eval/evaluator.py:18,126); the repo-root ``clip/`` package
re-exports these functions so those imports resolve to the HIP engine.

Weights: upstream ``clip.load(name)`` downloads a checkpoint, which is impossible offline.  Here
* ``name`` may be a path to a state-dict file (``.pt`` with ``model_state_dict`` / ``state_dict`` / bare dict,
  loaded with ``weights_only=True``; or ``.safetensors``), optionally ``"ViT-L/14@/path/file"``;
* or a known model name: ``$KEMR_CLIP_WEIGHTS/<name with / -> ->.pt|.safetensors`` is used when present;
* otherwise ``load`` raises: a run on random weights looks like any other run in its metrics file.  Synthetic-data runs
  opt in explicitly with ``allow_random_weights()`` (what ``--synthetic`` does) or ``KEMR_ALLOW_RANDOM_WEIGHTS=1``; the
  model then records ``weights_source = "random(seed 0)"``, which the evaluators write into their results JSON.
"""
from __future__ import annotations

import os
import warnings
from typing import List, Tuple, Union

import torch

from .clip_module import CLIP, build_model
from .config import ARCHS, get_arch
from .preprocess import ClipPreprocess, gpu_preprocessing_enabled
from .tokenizer import tokenize  # noqa: F401  (re-exported)

_PUBLIC = ("ViT-B/32", "ViT-B/16", "ViT-L/14")
_allow_random = False


def allow_random_weights(flag: bool = True) -> None:
    """Opt in to seeded random weights when no checkpoint is available (synthetic-data runs, tests)."""
    global _allow_random
    _allow_random = bool(flag)


def random_weights_allowed() -> bool:
    return _allow_random or os.environ.get("KEMR_ALLOW_RANDOM_WEIGHTS", "") == "1"


def available_models() -> List[str]:
    return list(_PUBLIC)


def read_state_dict(path: str) -> dict:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("model_state_dict", "state_dict"):
        if isinstance(ckpt, dict) and key in ckpt:
            return ckpt[key]
    return ckpt


def _weights_for(name: str):
    root = os.environ.get("KEMR_CLIP_WEIGHTS")
    if not root:
        return None
    stem = name.replace("/", "-")
    for ext in (".safetensors", ".pt"):
        p = os.path.join(root, stem + ext)
        if os.path.exists(p):
            return p
    return None


def load(name: str, device: Union[str, torch.device, None] = None,
         jit: bool = False, download_root: str = None) -> Tuple[CLIP, ClipPreprocess]:
    """device None = "cuda" when a GPU is visible, else "cpu" (upstream's default), decided at CALL time: evaluating
    torch.cuda.is_available() in the signature initialised the HIP runtime in every process that merely imported this module --
    the loader processes included (round 3: 13 processes with the GPU open)."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if jit:
        raise RuntimeError("clip.load(jit=True) is not supported by the HIP engine")
    path = None
    if "@" in name:
        name, path = name.split("@", 1)
    elif os.path.isfile(name):
        raise RuntimeError("pass a checkpoint as '<model name>@<path>' so that the architecture is known")
    if name not in ARCHS:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if path and not os.path.isfile(path):
        raise FileNotFoundError(f"clip.load: checkpoint {path!r} does not exist")
    path = path or _weights_for(name)
    if not path and not random_weights_allowed():
        raise FileNotFoundError(
            f"clip.load({name!r}): no checkpoint available offline.  Set KEMR_CLIP_WEIGHTS=<dir with {name.replace('/', '-')}.pt|.safetensors>, "
            f"pass '{name}@/path/to/state_dict.pt', or opt in to seeded RANDOM weights (synthetic-data runs: --synthetic, "
            "clip_api.allow_random_weights(), KEMR_ALLOW_RANDOM_WEIGHTS=1)")
    seed_state = torch.random.get_rng_state()
    torch.manual_seed(0)                     # reproducible random init when no weights are available
    try:
        model = build_model(name, device="cpu")
    finally:
        torch.random.set_rng_state(seed_state)
    if path:
        model.load_state_dict(read_state_dict(path), strict=True)
        model.weights_source = os.path.abspath(path)
    else:
        warnings.warn(f"clip.load({name!r}): seeded RANDOM weights (explicitly allowed)", RuntimeWarning, stacklevel=2)
        model.weights_source = "random(seed 0)"
    model = model.to(device).eval()
    on_gpu = torch.device(device).type == "cuda"
    return model, ClipPreprocess(get_arch(name).image_size, defer_to_gpu=on_gpu and gpu_preprocessing_enabled())
