#!/bin/bash
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_node_stage.py -x -q 2>&1 | tail -5 > gpurun_out/r5b/pytest_new.log
cat gpurun_out/r5b/pytest_new.log
python tools/bench_node_stage.py > gpurun_out/r5b/node_stage_default.log 2>&1
cat gpurun_out/r5b/node_stage_default.log
SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so python tools/bench_node_stage.py --sweep > gpurun_out/r5b/node_stage_sweep.log 2>&1
cat gpurun_out/r5b/node_stage_sweep.log
