"""Stand-in for the third-party ``clip`` package (openai/CLIP) that the reference imports: the same three entry
points, backed by the MI355X HIP engine (see knowledge_enhanced_multimodal_retrieval_amd/clip_api.py)."""
from knowledge_enhanced_multimodal_retrieval_amd.clip_api import available_models, load, tokenize  # noqa: F401

__all__ = ["available_models", "load", "tokenize"]
