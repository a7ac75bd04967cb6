#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the measurements quoted in DESIGN.md / README.md / profiles/README.md that
# are not per-kernel rocprofv3 profiles (those: tools/profile_all.sh).  Exits NON-ZERO when any tool's output
# holds a Python traceback or a library error (round 3 committed a "profile" that ended in one).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err
python tools/n_sweep.py > $OUT/n_sweep.txt 2>&1
python tools/batched_bench.py > $OUT/batched_chains_vec.txt 2>&1
VEC=0 CHAINS=4,8,16 python tools/batched_bench.py > $OUT/batched_chains_gridy.txt 2>&1
CHAINS=4,8,9,16,32,64,128 python tools/lockstep_bench.py > $OUT/lockstep_chains.txt 2>&1
python tools/ab_persist_spec.py > $OUT/ab_persist_spec.txt 2>&1
SWEEP=1 python tools/dynamic_bench.py > $OUT/dynamic_model.txt 2>&1
INSITU=0 DYN_GATHER=0 python tools/dynamic_bench.py > $OUT/dynamic_model_atomics.txt 2>&1
python tools/predict_bench.py > $OUT/predict.txt 2>&1
python tools/neutral_bench.py > $OUT/neutral_model.txt 2>&1
python tools/small_n_bench.py > $OUT/small_n.txt 2>&1
python tools/configs_bench.py > $OUT/configs.txt 2>&1
python tools/teams_sweep.py > $OUT/teams_sweep.txt 2>&1
python tools/soak.py 25000 > $OUT/soak.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -s 2>&1 | grep -E ' dU=|worst \|dU\|' > $OUT/parity_errors.txt
python -m pytest tests -q -m gpu 2>&1 | tail -n 3 > $OUT/gpu_tests.txt
tail -n 4 $OUT/*.txt | head -200
BAD=$(grep -l -E "Traceback \(most recent call last\)|BplHipError|libbplhip error" $OUT/*.txt $OUT/bench.err 2>/dev/null)
if [ -n "$BAD" ]; then
  echo "collect_$TAG: FAILED -- a tool ended in an error: $BAD" >&2
  exit 1
fi
echo "collect_$TAG: ok"
