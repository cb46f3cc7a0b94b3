"""`src.clip.utils` (reference: src/clip/utils/)."""
