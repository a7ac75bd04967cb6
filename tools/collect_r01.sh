#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every measurement quoted in DESIGN.md / profiles/README.md.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01e}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tools/stamps.py > $OUT/stamps_timeline.txt 2>&1
python tools/stamps.py nuts >> $OUT/stamps_timeline.txt 2>&1
python tools/n_sweep.py > $OUT/n_sweep.txt 2>&1
python tools/batched_bench.py > $OUT/batched_vec.txt 2>&1
VEC=0 CHAINS=8,64 python tools/batched_bench.py > $OUT/batched_gridy.txt 2>&1
CHAINS=4,8,16,32,64 python tools/lockstep_bench.py > $OUT/lockstep.txt 2>&1
python tools/dynamic_bench.py > $OUT/dynamic.txt 2>&1
python tools/predict_bench.py > $OUT/predict.txt 2>&1
python tools/neutral_bench.py > $OUT/neutral.txt 2>&1
python tools/small_n_bench.py > $OUT/small_n.txt 2>&1
python tools/configs_bench.py > $OUT/configs.txt 2>&1
python tools/teams_sweep.py > $OUT/teams_sweep.txt 2>&1
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
mkdir -p $OUT/nt && (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/nt -- python3 $ROOT/tools/nuts_trace.py > $OUT/nuts_trace.log 2>&1)
python tools/trace_gaps.py $OUT/nt > $OUT/nuts_trace_gaps.txt 2>&1
find $OUT/nt -name "*.csv" -size +1M -delete
tail -n 3 $OUT/*.txt | head -120
