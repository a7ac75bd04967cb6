#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): per-kernel breakdown of the multi-launch paths at N = 1e6
# (dynamic model's throughput variant, neutral-venue model).
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/large_n_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CASE in dyn_1e6 neutral_1e6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$CASE -- python3 $ROOT/tools/kernel_cases.py $CASE 128 > $OUT/$CASE.log 2>&1
  echo "$CASE exit $?"
  find $OUT/$CASE -name "*kernel_trace.csv" -delete
  cat $OUT/$CASE/*/*kernel_stats.csv | cut -d, -f1-8 | head -12
done
