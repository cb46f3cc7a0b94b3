"""Reference module path `src.clip.eval.metrics` -> fused HIP ranking (same function names and result keys)."""
from knowledge_enhanced_multimodal_retrieval_amd.metrics import (  # noqa: F401
    compute_all_retrieval_metrics, compute_metrics_multi_4train, compute_metrics_multi_mode,
    compute_metrics_single_4train, compute_mrr_and_mean_rank, compute_recall_at_k, compute_retrieval_metrics,
    compute_retrieval_metrics_final, compute_retrieval_metrics_fusion, compute_training_metrics)
