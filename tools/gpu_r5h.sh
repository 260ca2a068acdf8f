#!/bin/bash
# the four-channels-per-thread pooled_dgrad: its tests, the microbench on a real step's entries, the h3 training line
TAG=${1:-r05h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P
timeout 150 python -m pytest tests -q -m gpu -x -k "pooled or sorted or segpool or golden" 2>&1 | tail -5 > $P/${TAG}_pytest_pooled.log
timeout 100 python tools/bench_pooled_sorted.py > $P/${TAG}_bench_pooled_sorted.log 2>&1
timeout 120 python bench.py --mode train --precision h3 --steps 40 --warmup 8 2> /dev/null | tail -1 > $P/${TAG}_bench_train_h3.json
cat $P/${TAG}_pytest_pooled.log $P/${TAG}_bench_pooled_sorted.log; python -c "
import json;d=json.load(open('$P/${TAG}_bench_train_h3.json'));print(d['value'],d['ms_per_step'])"
