#!/bin/bash
# Evidence run for profiles/: un-profiled bench line, rocprofv3 kernel stats of the same command, optional PMC passes.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [pmc]      -> gpurun_out/<tag>_*
set -e
TAG=${1:-r01_v6}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[profile] bench"; python3 $R/bench.py > $OUT/${TAG}_bench.log 2>&1; tail -n 1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
echo "[profile] kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-pipeline > $OUT/${TAG}_bench_under_rocprof.log 2>&1
tail -n 1 $OUT/${TAG}_bench_under_rocprof.log > $OUT/${TAG}_bench_under_rocprof.json
if [ "$2" = "pmc" ]; then
  echo "[profile] pmc FETCH_SIZE"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-sim --no-extras --no-pipeline > $OUT/${TAG}_pmc_fetch.log 2>&1
  echo "[profile] pmc WRITE_SIZE"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-sim --no-extras --no-pipeline > $OUT/${TAG}_pmc_write.log 2>&1
fi
echo "[profile] sim route"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sim_stats -- python3 $R/tools/bench_sim.py --routes 1 --iters 5 > $OUT/${TAG}_sim_under_rocprof.log 2>&1
if [ "$2" = "pmc" ]; then
  echo "[profile] pmc SQ (MFMA busy) on the bench step"
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $R/bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-sim --no-extras --no-pipeline > $OUT/${TAG}_pmc_sq.log 2>&1
fi
if [ "$2" = "pmc" ]; then
  echo "[profile] pmc FETCH_SIZE / WRITE_SIZE of the similarity route (Q = 43 000 top-10)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_sim_pmc_fetch -- python3 $R/tools/bench_sim.py --routes 1 --iters 3 > $OUT/${TAG}_sim_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_sim_pmc_write -- python3 $R/tools/bench_sim.py --routes 1 --iters 3 > $OUT/${TAG}_sim_pmc_write.log 2>&1
  echo "[profile] summaries"
  python3 $R/tools/pmc_summary.py "$OUT/${TAG}_pmc_*/**/*counter_collection.csv" > $OUT/${TAG}_pmc_summary.txt 2>&1 || true
  python3 $R/tools/pmc_summary.py "$OUT/${TAG}_sim_pmc_*/**/*counter_collection.csv" > $OUT/${TAG}_sim_pmc_summary.txt 2>&1 || true
fi
echo "[profile] done"
