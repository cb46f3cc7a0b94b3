#!/bin/bash
# Zero-shot ViT-L/14 evaluation on the MI355X engine (same flags as the reference's script of this name).
# Offline: add `--synthetic 4096` (no dataset / weights are fetched; clip.load warns and uses seeded random weights).
OUTPUT_DIR="experiments/zeroshot"
mkdir -p $OUTPUT_DIR
python -m src.clip.eval.evaluator \
    --model_name "ViT-L/14" \
    --images_dir "../ArtKB/images" --texts_dir "../ArtKB/texts/texts" \
    --split "test" --splits_file "splits.json" \
    --batch_size 64 --device "cuda" \
    --output_file "$OUTPUT_DIR/clip_base_l14.json" --seed 42 "$@"
