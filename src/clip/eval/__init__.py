"""`src.clip.eval`: evaluators, metrics and score fusion (reference: src/clip/eval/)."""
