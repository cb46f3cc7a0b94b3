"""Reference module path `src.retrieval`: `RetrievalEngine().retrieve_text(query, alpha, beta, alpha_clip, threshold)`."""
from knowledge_enhanced_multimodal_retrieval_amd.retriever import NoText2SPARQL, RetrievalEngine  # noqa: F401
