"""Reference module path `src.clip.clip_retrieval` (no hub download / exec / login; see retriever.py)."""
from knowledge_enhanced_multimodal_retrieval_amd.retriever import CLIPRetrieval, CLIPRetriever, EmbeddingStore  # noqa: F401
