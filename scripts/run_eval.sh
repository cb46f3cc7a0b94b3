#!/bin/bash
# One entry for the three evaluation recipes of the reference (its own scripts/baselines/*.sh and scripts/fusion/eval.sh run
# unchanged against this repository when it is first on PYTHONPATH; this is the same thing with the recipe as an argument).
#   scripts/run_eval.sh zeroshot-l14 | zeroshot-b32 | fused   [extra evaluator flags, e.g. --synthetic 4096 when offline]
# KEMR_PRECISION=bf16|bf16-res16|fp8|fp8-mlp selects the encoder precision.
set -e
recipe=${1:?usage: run_eval.sh zeroshot-l14|zeroshot-b32|fused [flags]}
shift
data=(--images_dir ../ArtKB/images --texts_dir ../ArtKB/texts/texts --split test --splits_file splits.json --batch_size 64 --device cuda)
case "$recipe" in
  zeroshot-l14|zeroshot-b32)
    model=$([ "$recipe" = zeroshot-l14 ] && echo ViT-L/14 || echo ViT-B/32)
    mkdir -p experiments/zeroshot
    exec python -m src.clip.eval.evaluator --model_name "$model" "${data[@]}" --seed 42 \
        --output_file "experiments/zeroshot/clip_base_${recipe#zeroshot-}.json" "$@" ;;
  fused)
    wi=${T2I_WEIGHT:-0.5}; wt=${T2T_WEIGHT:-0.5}
    ckpt=${CHECKPOINT:-experiments/fine-tuning/train_lr5e-6_wd0.02_t2iweight0.7/checkpoint_best.pt}
    exec python -m src.clip.eval.evaluator_baseline --model_name ViT-L/14 --checkpoint "$ckpt" "${data[@]}" \
        --t2i_weight "$wi" --t2t_weight "$wt" --output_file "experiments/2-fusion/baseline_${wi}_${wt}.json" "$@" ;;
  *) echo "unknown recipe $recipe" >&2; exit 2 ;;
esac
