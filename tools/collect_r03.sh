#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the measurements quoted in DESIGN.md / profiles/README.md that
# are not per-kernel rocprofv3 profiles (those: tools/profile_all.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err
python tools/stamps.py > $OUT/stamps_timeline.txt 2>&1
python tools/stamps.py nuts >> $OUT/stamps_timeline.txt 2>&1
NUTS_K=5 python tools/stamps.py nuts >> $OUT/stamps_timeline.txt 2>&1
python tools/dynamic_stamps.py > $OUT/dynamic_timeline.txt 2>&1
BIGN=1e6 python tools/dynamic_stamps.py > $OUT/dynamic_big_stamps.txt 2>&1
python tools/neutral_big_stamps.py > $OUT/neutral_big_stamps.txt 2>&1
python tools/neutral_phases.py > $OUT/neutral_phases.txt 2>&1
TEAMS=100 python tools/stamps.py basic > $OUT/stamps_teams100.txt 2>&1
TEAMS=200 python tools/stamps.py basic > $OUT/stamps_teams200.txt 2>&1
python tools/n_sweep.py > $OUT/n_sweep.txt 2>&1
python tools/batched_bench.py > $OUT/batched_chains_vec.txt 2>&1
VEC=0 CHAINS=8,64 python tools/batched_bench.py > $OUT/batched_chains_gridy.txt 2>&1
CHAINS=4,8,16,32,64 python tools/lockstep_bench.py > $OUT/lockstep_chains.txt 2>&1
SWEEP=1 python tools/dynamic_bench.py > $OUT/dynamic_model.txt 2>&1
python tools/predict_bench.py > $OUT/predict.txt 2>&1
python tools/neutral_bench.py > $OUT/neutral_model.txt 2>&1
python tools/small_n_bench.py > $OUT/small_n.txt 2>&1
python tools/configs_bench.py > $OUT/configs.txt 2>&1
python tools/teams_sweep.py > $OUT/teams_sweep.txt 2>&1
hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o /tmp/xcd_handoff tools/micro/xcd_handoff.hip 2>/dev/null && /tmp/xcd_handoff > $OUT/xcd_handoff.txt 2>&1
python tools/soak.py 25000 > $OUT/soak.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dyn_nuts_trace -- python3 $ROOT/tools/kernel_cases.py dyn_nuts > $OUT/dyn_nuts_trace.log 2>&1)
python tools/leaf_trace_summary.py $OUT/dyn_nuts_trace > $OUT/dynamic_leaf_trace.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -s 2>&1 | grep ' dU=' > $OUT/parity_errors.txt
tail -n 4 $OUT/*.txt | head -150
