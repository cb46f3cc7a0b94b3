#!/bin/bash
# Same-box A/B of the images per encoder launch (engine.MAX_IMAGE_BATCH): tools/ab_image_slice.sh "255:0 254:127 510:510" [rounds]   (batch:slice, 0 = the engine's 255)
CFGS=${1:-"255:0 254:127 252:63 510:510"}; ROUNDS=${2:-2}
for i in $(seq $ROUNDS); do
for cfg in $CFGS; do
  b=${cfg%%:*}; s=${cfg##*:}
  python bench.py --steps 40 --batch $b --image-slice $s --no-extras --no-pipeline --no-sim --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b slice $s', round(d['value']), 'img/s', round(d['images_per_s']), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, 'gemm frac', round(d['roofline']['frac'], 4))"
done
done
