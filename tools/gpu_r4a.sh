#!/bin/bash
# round 4, first GPU call: parity of the third-generation layer + tile-shape sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_h3p.py -x -q > $O/pytest_h3p.log 2>&1; tail -15 $O/pytest_h3p.log
SONET_HIP_LIB=$R/so-net_amd/lib/libsonet_hip_variants.so timeout 900 python tools/bench_h3p.py --sweep > $O/bench_h3p_sweep.log 2>&1; tail -5 $O/bench_h3p_sweep.log
