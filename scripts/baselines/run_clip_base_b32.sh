#!/bin/bash
# Zero-shot ViT-B/32 evaluation on the MI355X engine (same flags as the reference's script of this name).
OUTPUT_DIR="experiments/zeroshot"
mkdir -p $OUTPUT_DIR
python -m src.clip.eval.evaluator \
    --model_name "ViT-B/32" \
    --images_dir "../ArtKB/images" --texts_dir "../ArtKB/texts/texts" \
    --split "test" --splits_file "splits.json" \
    --batch_size 64 --device "cuda" \
    --output_file "$OUTPUT_DIR/clip_base_b32.json" --seed 42 "$@"
