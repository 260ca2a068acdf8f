#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "node_train or train or backward or stat or bf16 or golden" > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for p in bf16 h3; do timeout 600 python bench.py --mode train --precision $p --steps 30 --warmup 5 2> $O/train_$p.err | tail -1 > $O/bench_train_$p.json; python -c "
import json; d=json.load(open('$O/bench_train_$p.json')); print('$p', d['value'], d['ms_per_step'])"; done
