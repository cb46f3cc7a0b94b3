"""`python -m src.clip.eval.evaluator_fusion ...`: learned fusion heads."""
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import evaluate_fusion_model  # noqa: F401
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import main_fusion as main  # noqa: F401

if __name__ == "__main__":
    main()
