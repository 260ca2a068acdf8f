#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R; export TMPDIR=/tmp
V=$R/so-net_amd/lib/libsonet_hip_variants.so
SONET_HIP_LIB=$V timeout 300 python tools/dbg_swap.py > $O/dbg_swap.log 2>&1; grep -v amdgpu.ids $O/dbg_swap.log | tail -16
timeout 900 python -m pytest tests/test_gpu_h3p.py -q > $O/pytest_h3p.log 2>&1; tail -6 $O/pytest_h3p.log
