"""`python -m src.clip.eval.evaluator_baseline ...` (scripts/fusion/eval.sh): fused T2I + T2T score."""
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import evaluate_clip_model_baseline as evaluate_clip_model  # noqa: F401
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import main_baseline as main  # noqa: F401
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import seed_worker  # noqa: F401

if __name__ == "__main__":
    main()
