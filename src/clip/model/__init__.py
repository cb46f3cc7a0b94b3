"""`src.clip.model`: model factory + fusion heads (reference: src/clip/model/)."""
from .clip_model import load_clip_model  # noqa: F401
from .fusion_model import FusionModel  # noqa: F401
