"""Reference module path `src.clip.datasets.clip_dataset` -> the build's dataset wrappers (same sample contract)."""
from knowledge_enhanced_multimodal_retrieval_amd.datasets import (  # noqa: F401
    CLIPEvalDatasetHF, CLIPEvaluationDataset, SyntheticRetrievalDataset, collate_fn_eval, collate_fn_eval_texts,
    collate_fn_train)
