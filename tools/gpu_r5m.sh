#!/bin/bash
# round 5, second session: the f32-class training forward on node-sorted columns (sorted pool + normalise-on-load).
#   1. the new tests   2. pytest -m gpu (everything)   3. tools/ab_h3_train.py (A / B / C interleaved in one process + per-kernel times)
TAG=${1:-r5m}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
echo "== 1 new tests"; timeout 600 python -m pytest tests/test_gpu_segpool.py -x -q > $O/pytest_segpool.log 2>&1; tail -25 $O/pytest_segpool.log
echo "== 2 pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
echo "== 3 A/B"; timeout 400 python tools/ab_h3_train.py --rounds 6 --steps 24 2>&1 | grep -v amdgpu > $O/ab_h3_train.log; cat $O/ab_h3_train.log
