#!/bin/bash
# round 5, second session: pytest -m gpu + the gradient-deviation table of the sorted training path (tools/grad_dev_h3.py)
TAG=${1:-r5n}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
{ for f in train_step_b8_n5000 train_step_b16_n512; do echo "== tools/grad_dev_h3.py $f"; timeout 300 python tools/grad_dev_h3.py $f 2>&1 | grep -v "amdgpu\|Warning\|detach\|print("; done; } > $O/grad_dev_h3.log; cat $O/grad_dev_h3.log
