"""Reference module path `src.clip.utils.logging_utils`."""
from knowledge_enhanced_multimodal_retrieval_amd.logging_utils import (  # noqa: F401
    log_metrics_to_jsonl, save_metrics_to_json, setup_logger)
