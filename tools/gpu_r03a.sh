#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== pytest"; timeout 1700 python -m pytest tests/test_gpu_variants_suite.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -6
for prec in bf16; do timeout 600 python bench.py --mode train --precision $prec --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$prec', l['ms_per_step'], l['value'])"; done
