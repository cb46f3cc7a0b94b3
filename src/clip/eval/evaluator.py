"""`python -m src.clip.eval.evaluator ...` (scripts/baselines/*.sh, baselines/evaluate_zeroshot.py)."""
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import (  # noqa: F401
    evaluate_clip_model, evaluate_clip_model_for_training, load_text2sparql_results, seed_worker)
from knowledge_enhanced_multimodal_retrieval_amd.evaluators import main_evaluator as main  # noqa: F401

if __name__ == "__main__":
    main()
