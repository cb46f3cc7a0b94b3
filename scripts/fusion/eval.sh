#!/bin/bash
# Fused T2I + T2T scoring of a fine-tuned checkpoint (same flags as the reference's scripts/fusion/eval.sh).
T2I_WEIGHT=${T2I_WEIGHT:-0.5}
T2T_WEIGHT=${T2T_WEIGHT:-0.5}
OUTPUT_DIR="experiments/fine-tuning"
EXPERIMENT_NAME="train_lr5e-6_wd0.02_t2iweight0.7"
python -m src.clip.eval.evaluator_baseline \
    --model_name "ViT-L/14" \
    --checkpoint "$OUTPUT_DIR/$EXPERIMENT_NAME/checkpoint_best.pt" \
    --images_dir "../ArtKB/images" --texts_dir "../ArtKB/texts/texts" \
    --split "test" --splits_file "splits.json" \
    --batch_size 64 --device "cuda" \
    --output_file "experiments/2-fusion/baseline_${T2I_WEIGHT}_${T2T_WEIGHT}.json" \
    --t2i_weight $T2I_WEIGHT --t2t_weight $T2T_WEIGHT "$@"
