#!/bin/bash
# the entry-balanced sparse input gradient as the product's kernel: every test that reaches it (+ the variants suite: one-channel == column-owned,
# entry-balanced vs float64), the f32-class training line, the microbench out of the product library
TAG=${1:-r05l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P
timeout 200 python -m pytest tests -q -m gpu -x -k "pooled or train or segpool or variants or golden" 2>&1 | tail -4 > $P/${TAG}_pytest_train_paths.log
timeout 60 python bench.py --mode train --precision h3 --steps 40 --warmup 8 2> /dev/null | tail -1 > $P/${TAG}_bench_train_h3.json
timeout 60 python tools/bench_pooled_sorted.py 2>&1 | grep -v "amdgpu\|Warning\|detach" > $P/${TAG}_bench_pooled_sorted.log
cat $P/${TAG}_pytest_train_paths.log $P/${TAG}_bench_pooled_sorted.log | cut -c1-200; python -c "
import json;d=json.load(open('$P/${TAG}_bench_train_h3.json'));print(d['value'],d['ms_per_step'])"
