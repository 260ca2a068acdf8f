#!/bin/bash
TAG=${1:-r02o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -s > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|FAILED|rel-rms|cosine" $O/pytest_gpu.log | tail -12 | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench forward"; timeout 600 python bench.py --steps 50 --warmup 10 2> $O/bench_forward.err | tail -1 > $O/bench_forward.json; head -c 300 $O/bench_forward.json; echo
echo "== bench forward bf16"; timeout 600 python bench.py --precision bf16 --steps 50 --warmup 10 --no-cpu-baseline 2> $O/bench_forward_bf16.err | tail -1 > $O/bench_forward_bf16.json; head -c 300 $O/bench_forward_bf16.json; echo
echo "== bench train bf16"; timeout 600 python bench.py --mode train --precision bf16 --steps 20 --warmup 5 2> $O/bench_train_bf16.err | tail -1 > $O/bench_train_bf16.json; head -c 300 $O/bench_train_bf16.json; echo
echo "== bench train h3"; timeout 600 python bench.py --mode train --steps 20 --warmup 5 2> $O/bench_train_h3.err | tail -1 > $O/bench_train_h3.json; head -c 300 $O/bench_train_h3.json; echo
echo "== rocprof forward"; bash tools/gpu_prof.sh ${TAG}_forward -- --steps 20 --warmup 5 --no-graph --no-other-precisions --no-parity-check > $O/prof_fwd.log 2>&1; tail -3 $O/prof_fwd.log | cut -c1-200
echo "== rocprof forward bf16"; bash tools/gpu_prof.sh ${TAG}_forward_bf16 -- --precision bf16 --steps 20 --warmup 5 --no-graph --no-other-precisions --no-parity-check > $O/prof_fwd_bf16.log 2>&1; tail -3 $O/prof_fwd_bf16.log | cut -c1-200
echo "== heads"; timeout 600 python tools/bench_heads.py > $O/bench_heads.log 2>&1; grep -v amdgpu $O/bench_heads.log | cut -c1-250
echo "== done"
