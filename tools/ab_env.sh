#!/bin/bash
# Same-box A/B of one environment switch through bench.py: tools/ab_env.sh VAR "v1 v2" [rounds] [bench args...] -> items/s and kernel ms per value
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3 || true
for i in $(seq $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 400 python bench.py --steps 60 --no-pipeline --no-extras "$@" 2>/dev/null > /tmp/ab_env.json || exit 1
    python3 - "$VAR" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_env.json").read().strip().splitlines()[-1])
print(sys.argv[1], "=", sys.argv[2], ":", round(d["value"]), "items/s", {k: round(x, 2) for k, x in d["kernel_ms_per_step"].items()}, flush=True)
PY
  done
done
