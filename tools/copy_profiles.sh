#!/bin/bash
# Copies the digest of tools/profile_all.sh from gpurun_out/prof_<tag>/ into profiles/<tag>/ (here, after
# the gpurun call): kernels.md / kernels.json, and per case rocprofv3's own kernel statistics (the newest
# run in the directory) and the case's byte accounting (the JSON line the case printed).
set -eu
TAG=${1:-r03}
SRC=gpurun_out/prof_$TAG
DST=profiles/$TAG
mkdir -p $DST/kernels
cp $SRC/kernels.md $SRC/kernels.json $DST/
for d in $SRC/*/; do
  c=$(basename $d)
  f=$(ls -t $d/trace/*/*_kernel_stats.csv 2>/dev/null | head -n 1 || true)
  [ -n "$f" ] && cp $f $DST/kernels/${c}_kernel_stats.csv
  grep -h '^{' $d/trace.log | tail -n 1 > $DST/kernels/${c}_case.json || true
done
ls $DST/kernels | wc -l
