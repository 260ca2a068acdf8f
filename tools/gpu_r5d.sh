#!/bin/bash
mkdir -p gpurun_out/r5d
python tools/ab_node_stage.py --rounds 10 --steps 40 > gpurun_out/r5d/ab_node_stage.log 2>&1
cat gpurun_out/r5d/ab_node_stage.log
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5d/pytest_gpu.log
cat gpurun_out/r5d/pytest_gpu.log
