"""CLIP half of the reference module tree (see src/__init__.py)."""
