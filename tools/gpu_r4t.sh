#!/bin/bash
python tools/bench_pooled.py 2>&1 | grep -v amdgpu | head -6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_node_train.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --mode train --precision bf16 --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('bf16 ms/step', d['ms_per_step'])"; done
