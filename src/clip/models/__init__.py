"""`src.clip.models`: the name the reference's README and BASELINE.json use for `src.clip.model` (the directory on
disk is singular, reference src/clip/eval/evaluator_baseline.py:20 vs README.md:63).  Alias package."""
import sys

from ..model import clip_model, fusion_model  # noqa: F401
from ..model import FusionModel, load_clip_model  # noqa: F401

sys.modules[__name__ + ".clip_model"] = clip_model
sys.modules[__name__ + ".fusion_model"] = fusion_model
