"""Reference module path `src.clip.model.clip_model` -> HIP-backed implementation."""
from knowledge_enhanced_multimodal_retrieval_amd.clip_model import (  # noqa: F401
    freeze_clip_encoders, get_trainable_params, load_checkpoint_for_resuming, load_clip_model, print_model_info,
    save_checkpoint, unfreeze_clip_encoders)
