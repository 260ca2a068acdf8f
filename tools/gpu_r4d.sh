#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R; export TMPDIR=/tmp
V=$R/so-net_amd/lib/libsonet_hip_variants.so
SONET_HIP_LIB=$V timeout 300 python tools/dbg_swap.py > $O/dbg_swap.log 2>&1; grep -v amdgpu.ids $O/dbg_swap.log | tail -8
timeout 900 python -m pytest tests/test_gpu_h3p.py -x -q > $O/pytest_h3p.log 2>&1; tail -4 $O/pytest_h3p.log
SONET_HIP_LIB=$V timeout 300 python tools/bench_h3p.py --shapes all > $O/bench_h3p_default.log 2>&1; grep -v amdgpu.ids $O/bench_h3p_default.log
SONET_HIP_LIB=$V timeout 600 python tools/h3p_phases.py 320x384 > $O/h3p_phases.log 2>&1; grep "abl 0" $O/h3p_phases.log
