"""Drop-in module tree of the reference (`src.clip.model(s)`, `src.clip.eval`, `src.clip.datasets`, `src.retrieval`):
thin re-exports of knowledge_enhanced_multimodal_retrieval_amd, so `python -m src.clip.eval.evaluator ...` and
`from src.retrieval import RetrievalEngine` resolve to the MI355X HIP engine."""
