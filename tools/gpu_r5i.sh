#!/bin/bash
# the carried gradient (ops.GRAD_CARRY): its tests, G | H in one process, the h3 training line
TAG=${1:-r05i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P
timeout 200 python -m pytest tests/test_gpu_segpool.py -q -m gpu -x 2>&1 | tail -8 > $P/${TAG}_pytest_segpool.log
timeout 120 python tools/ab_h3_train.py --only GH --rounds 5 --steps 24 --no-kernels > $P/${TAG}_ab_h3_train.log 2>&1
timeout 120 python bench.py --mode train --precision h3 --steps 40 --warmup 8 2> /dev/null | tail -1 > $P/${TAG}_bench_train_h3.json
cat $P/${TAG}_pytest_segpool.log; grep -v amdgpu.ids $P/${TAG}_ab_h3_train.log | cut -c1-150; python -c "
import json;d=json.load(open('$P/${TAG}_bench_train_h3.json'));print(d['value'],d['ms_per_step'])"
