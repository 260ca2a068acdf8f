#!/bin/bash
# the tail of the sparse input gradient (ops.POOLED_DGRAD_TAIL): every test of the f32-class training path + the training goldens + the pooled
# kernels' tests in ONE pytest process, then the launches apart / on the store in one process (tools/bench_pooled_sorted.py)
TAG=${1:-r05s}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P
timeout 75 python -m pytest tests/test_gpu_segpool.py tests/test_gpu_parity.py -q -m gpu -x -k "segpool or classifier_training_step_golden or pooled" 2>&1 | tail -12 > $P/${TAG}_pytest_tail.log
timeout 45 python tools/bench_pooled_sorted.py 2>&1 | grep -v "amdgpu\|Warning\|detach" > $P/${TAG}_bench_pooled_sorted.log
cat $P/${TAG}_pytest_tail.log; cut -c1-220 $P/${TAG}_bench_pooled_sorted.log
