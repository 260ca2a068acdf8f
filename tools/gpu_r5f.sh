#!/bin/bash
mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_node_stage.py tests/test_gpu_parity.py -x -q 2>&1 | tail -30 > gpurun_out/r5f/pytest.log
cat gpurun_out/r5f/pytest.log
python tools/ab_node_stage.py --rounds 8 --steps 40 > gpurun_out/r5f/ab_node_stage.log 2>&1
cat gpurun_out/r5f/ab_node_stage.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o ab -- python $GRAFT_REPO_ROOT/tools/ab_node_stage.py --rounds 2 --steps 30 --in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/r5f/ab_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_steps.py /tmp/prof_ab/ab_kernel_trace.csv > gpurun_out/r5f/ab_steps.log 2>&1
grep -A20 "step with 1[34] launches" gpurun_out/r5f/ab_steps.log | head -24
