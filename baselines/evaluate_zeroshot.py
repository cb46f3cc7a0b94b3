"""Zero-shot CLIP baseline (BASELINE configs[0] plumbing): thin wrapper around `src.clip.eval.evaluator.main`,
like the reference's baselines/evaluate_zeroshot.py:14,23 -- no checkpoint means the weights `clip.load` provides."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from src.clip.eval.evaluator import main as evaluate_main  # noqa: E402

if __name__ == "__main__":
    evaluate_main()
