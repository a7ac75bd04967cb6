#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): per-kernel evidence for every kernel family DESIGN.md quotes.
# For each case of tools/kernel_cases.py: a kernel trace with --stats, then the FETCH_SIZE and
# WRITE_SIZE counters in passes of their own (MI355X_MICROARCH.md: --pmc never combined with other
# trace domains; FETCH costs 3 TCC slots, WRITE 2).  The program goes directly after `--`.
# Output: gpurun_out/prof_<tag>/<case>/...; digest with tools/summarise_all.py.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CASE in basic c3 c3w t100 t200 nuts vec64 dyn_c4 dyn_1e6 neutral neutral_1e6 predict predict_venue; do
  D=$OUT/$CASE; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $ROOT/tools/kernel_cases.py $CASE > $D/trace.log 2>&1
  echo "$CASE trace exit $?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 $ROOT/tools/kernel_cases.py $CASE 256 > $D/pmc_fetch.log 2>&1
  echo "$CASE fetch exit $?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 $ROOT/tools/kernel_cases.py $CASE 256 > $D/pmc_write.log 2>&1
  echo "$CASE write exit $?"
done
python3 $ROOT/tools/summarise_all.py $OUT > $OUT/kernels.md 2> $OUT/summarise.err
# keep only small artefacts
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.log" -size +200k -delete
cat $OUT/kernels.md
