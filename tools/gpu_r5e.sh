#!/bin/bash
mkdir -p gpurun_out/r5e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o ab -- python $GRAFT_REPO_ROOT/tools/ab_node_stage.py --rounds 3 --steps 30 --in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/r5e/ab_prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep "in flight\|windows" gpurun_out/r5e/ab_prof.log
python tools/trace_steps.py /tmp/prof_ab/ab_kernel_trace.csv > gpurun_out/r5e/ab_steps.log 2>&1
cat gpurun_out/r5e/ab_steps.log
