"""`src.clip.datasets` (reference: src/clip/datasets/)."""
