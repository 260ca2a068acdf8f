#!/bin/bash
# round-2 first GPU call: new tests first (all failures shown), then the whole -m gpu suite, forward + train bench, MFMA counters
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_round2.py -q --maxfail=20 > $O/pytest_round2.log 2>&1; tail -25 $O/pytest_round2.log
echo "== full suite"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
echo "== bench forward"; timeout 600 python bench.py --steps 50 --warmup 10 2> $O/bench_forward.err | tail -1 > $O/bench_forward.json; head -c 600 $O/bench_forward.json; echo; tail -3 $O/bench_forward.err
echo "== bench train"; timeout 600 python bench.py --mode train --steps 20 --warmup 5 2> $O/bench_train.err | tail -1 > $O/bench_train.json; head -c 900 $O/bench_train.json; echo; tail -3 $O/bench_train.err
echo "== pmc sq"; timeout 600 bash tools/pmc_sq.sh fused_pool > $O/pmc_sq.log 2>&1; tail -5 $O/pmc_sq.log
for p in a b c; do f=$(find $R/gpurun_out/pmc_sq_fused_pool_$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_sq_fused_pool_$p.csv; done
echo "== done"
