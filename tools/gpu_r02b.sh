#!/bin/bash
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== bf16 tests"; timeout 900 python -m pytest tests/test_gpu_bf16.py -q --maxfail=30 -s > $O/pytest_bf16.log 2>&1; tail -40 $O/pytest_bf16.log | cut -c1-300
echo "== round2 tests"; timeout 900 python -m pytest tests/test_gpu_round2.py -q --maxfail=30 > $O/pytest_round2.log 2>&1; tail -8 $O/pytest_round2.log | cut -c1-300
echo "== bench bf16 kernel"; timeout 600 python tools/bench_bf16.py > $O/bench_bf16_kernel.log 2>&1; cat $O/bench_bf16_kernel.log
echo "== bench train"; timeout 600 python bench.py --mode train --steps 20 --warmup 5 2> $O/bench_train.err | tail -1 > $O/bench_train.json; head -c 1500 $O/bench_train.json; echo; tail -3 $O/bench_train.err
echo "== done"
