set -e
R=$(pwd); OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/r04_v9_pmc_lds -- python3 $R/bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-sim --no-extras --no-pipeline > $OUT/r04_v9_pmc_lds.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/r04_v9_pmc_wait -- python3 $R/bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-sim --no-extras --no-pipeline > $OUT/r04_v9_pmc_wait.log 2>&1
python3 $R/tools/pmc_summary.py "$OUT/r04_v9_pmc_lds/**/*counter_collection.csv" > $OUT/r04_v9_pmc_lds_summary.txt 2>&1 || true
python3 $R/tools/pmc_summary.py "$OUT/r04_v9_pmc_wait/**/*counter_collection.csv" > $OUT/r04_v9_pmc_wait_summary.txt 2>&1 || true
