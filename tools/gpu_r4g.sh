#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_forward.json 2> $O/bench_forward.err; tail -c 600 $O/bench_forward.json; tail -3 $O/bench_forward.err
