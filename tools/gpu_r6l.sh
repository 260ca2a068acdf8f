#!/bin/bash
# bf16 BatchNorm-backward-on-load: tests, then interleaved A/B of the bf16 training step (SONET_BF16_BNB_ON_LOAD 1 / 0)
TAG=${1:-r06l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG; mkdir -p $P
timeout 600 python -m pytest tests/test_gpu_bf16_xaff.py -q -m gpu -x -k "bnb or batchnorm_backward" 2>&1 | tail -15 | tee $P/pytest.log
for rep in 1 2; do for f in 1 0; do
  SONET_BF16_BNB_ON_LOAD=$f timeout 200 python bench.py --mode train --precision bf16 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BNB=$f ms %.4f clouds/s %.1f' % (d['ms_per_step'], d['value']))"
done; done 2>&1 | tee $P/ab.log
