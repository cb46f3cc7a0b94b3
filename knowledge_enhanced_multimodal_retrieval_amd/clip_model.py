"""Model factory / checkpoint helpers with the reference's names
(/root/reference/src/clip/model/clip_model.py: ``load_clip_model`` :15-75, ``save_checkpoint`` :78-120,
``load_checkpoint_for_resuming`` :123-171, ``freeze_clip_encoders`` :174-222, ``unfreeze_clip_encoders`` :225-245,
``get_trainable_params`` :248-264, ``print_model_info`` :267-290).

``load_clip_model`` returns the HIP-backed ``CLIP`` module (fp32 master parameters, bf16 packed copy in the engine)
and the preprocessing callable.  Checkpoints are read with ``weights_only=True`` and loaded strictly; the accepted
layouts are the reference's: ``model_state_dict`` / ``state_dict`` / a bare state dict."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import clip_api

logger = logging.getLogger(__name__)


def _unwrap(model: nn.Module) -> nn.Module:
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


def load_clip_model(model_name: str = "ViT-L/14", checkpoint_path: Optional[str] = None,
                    device: str = "cuda:0") -> Tuple[nn.Module, object]:
    logger.info(f"Loading CLIP model: {model_name}")
    # the reference reads a missing path as "no checkpoint" (clip_model.py:47) and still has pretrained weights underneath;
    # here that would silently evaluate whatever clip.load() found, so a path that is given must exist
    if checkpoint_path and not Path(checkpoint_path).exists():
        raise FileNotFoundError(f"load_clip_model: checkpoint {checkpoint_path!r} does not exist")
    if checkpoint_path and not clip_api.random_weights_allowed() and not clip_api._weights_for(model_name):
        # the fine-tuned checkpoint replaces every tensor (strict load): the base weights underneath do not matter
        clip_api.allow_random_weights(True)
        try:
            clip_model, preprocess = clip_api.load(model_name, device=device)
        finally:
            clip_api.allow_random_weights(False)
    else:
        clip_model, preprocess = clip_api.load(model_name, device=device)
    clip_model = clip_model.float()
    if checkpoint_path:
        print(f"Loading checkpoint from {checkpoint_path}")
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        if "model_state_dict" in checkpoint:
            state_dict = checkpoint["model_state_dict"]
            print("  Loaded 'model_state_dict' from checkpoint")
        elif "state_dict" in checkpoint:
            state_dict = checkpoint["state_dict"]
            print("  Loaded 'state_dict' from checkpoint")
        else:
            state_dict = checkpoint
            print("  Loaded checkpoint as state_dict directly")
        clip_model.load_state_dict(state_dict, strict=True)
        clip_model.weights_source = str(Path(checkpoint_path).resolve())
        logger.info("Checkpoint loaded successfully")
        if "epoch" in checkpoint:
            logger.info(f"  Checkpoint epoch: {checkpoint['epoch']}")
        if "best_metric" in checkpoint:
            logger.info(f"  Best metric: {checkpoint['best_metric']:.2f}")
    else:
        logger.info("Using the weights clip.load() provided (no fine-tuned checkpoint)")
    return clip_model, preprocess


def save_checkpoint(model, optimizer, epoch: int, best_metric: float, best_epoch: int, save_path: str, scheduler=None):
    checkpoint = {"epoch": epoch, "model_state_dict": _unwrap(model).state_dict(),
                  "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
                  "best_metric": best_metric, "best_epoch": best_epoch}
    if scheduler is not None:
        checkpoint["scheduler_state_dict"] = scheduler.state_dict()
    torch.save(checkpoint, save_path)
    logger.info(f"Checkpoint saved to {save_path}")


def load_checkpoint_for_resuming(checkpoint_path: str, model, optimizer, scheduler=None, device: str = "cuda"
                                 ) -> Tuple[int, float, int]:
    logger.info(f"Loading checkpoint for resuming: {checkpoint_path}")
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    _unwrap(model).load_state_dict(checkpoint["model_state_dict"])
    if optimizer is not None and checkpoint.get("optimizer_state_dict"):
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in checkpoint:
        scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
    epoch, best = checkpoint.get("epoch", 0), checkpoint.get("best_metric", float("-inf"))
    best_epoch = checkpoint.get("best_epoch", 0)
    logger.info(f"Resumed from epoch {epoch}, best metric: {best:.2f}")
    return epoch, best, best_epoch


def freeze_clip_encoders(model: nn.Module):
    """Everything frozen except the two projections and ``ln_final`` (reference clip_model.py:193-216)."""
    m = _unwrap(model)
    for name, p in m.visual.named_parameters():
        p.requires_grad = "proj" in name
    for p in m.transformer.parameters():
        p.requires_grad = False
    for p in m.token_embedding.parameters():
        p.requires_grad = False
    if hasattr(m, "positional_embedding"):
        m.positional_embedding.requires_grad = False
    if getattr(m, "text_projection", None) is not None:
        m.text_projection.requires_grad = True
    if hasattr(m, "ln_final"):
        for p in m.ln_final.parameters():
            p.requires_grad = True
    total = sum(p.numel() for p in m.parameters())
    train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    logger.info(f"Total params: {total:,}; trainable: {train:,} ({100 * train / total:.2f}%)")


def unfreeze_clip_encoders(model: nn.Module):
    m = _unwrap(model)
    for p in m.parameters():
        p.requires_grad = True
    logger.info(f"All {sum(p.numel() for p in m.parameters()):,} parameters are now trainable")


def get_trainable_params(model: nn.Module) -> int:
    return sum(p.numel() for p in _unwrap(model).parameters() if p.requires_grad)


def print_model_info(model: nn.Module):
    m = _unwrap(model)
    total = sum(p.numel() for p in m.parameters())
    train = sum(p.numel() for p in m.parameters() if p.requires_grad)
    logger.info("=" * 60)
    logger.info(f"DDP wrapped: {m is not model}")
    logger.info(f"Total parameters: {total:,}")
    logger.info(f"Trainable parameters: {train:,} ({100 * train / max(total, 1):.2f}%)")
    logger.info("=" * 60)
