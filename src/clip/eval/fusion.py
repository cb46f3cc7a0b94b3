"""Reference module path `src.clip.eval.fusion` -> CLIP (+) Text2SPARQL score fusion of the build."""
from knowledge_enhanced_multimodal_retrieval_amd.sparql_fusion import (  # noqa: F401
    adaptive_additive_fusion, additive_bonus_fusion, evaluate_retrieval, fuse_clip_and_text2sparql, fused_metrics,
    fused_ranks, sparql_bonus, weighted_fusion)
