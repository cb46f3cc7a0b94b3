#!/bin/bash
# Same-box A/B of a COMPILE-TIME switch of the library (a -D flag): builds libkemr.so with and without the flag in turn and runs a short
# bench.py after each build, ROUNDS times (interleaved).  usage (on the GPU box, repo root): bash tools/ab_build_flag.sh "-DKEMR_GEMM_PRIO=2" [rounds]
# Leaves the plain build behind.  hipcc must be on the box (it is: same image).
set -e
FLAG="$1"; ROUNDS=${2:-2}
R=$(pwd)
build() { python3 - <<PY
import sys; sys.path.insert(0, "$R")
from knowledge_enhanced_multimodal_retrieval_amd import build
build.build(force=True, extra_flags=[f for f in "$1".split() if f])
PY
}
line() { python3 $R/bench.py --steps 40 --no-extras --no-pipeline --no-sim --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, 'gemm frac', round(d['roofline']['frac'], 4))"; }
for i in $(seq $ROUNDS); do
  build ""; line "plain       "
  build "$FLAG"; line "$FLAG"
done
build ""
