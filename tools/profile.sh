#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Output under gpurun_out/prof_<tag>/ ; summarise with tools/summarise_profile.py.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2048 --warmup 256 --no-cpu-baseline --no-insitu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace exit $?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
echo "pmc write exit $?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_l2.log 2>&1
echo "pmc l2 exit $?"
# keep only small artefacts (per-dispatch CSVs of PMC runs are large): stats + a digest
python3 $ROOT/tools/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT
