#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R; export TMPDIR=/tmp
V=$R/so-net_amd/lib/libsonet_hip_variants.so
SONET_HIP_LIB=$V timeout 300 python tools/dbg_swap.py > $O/dbg_swap.log 2>&1; cat $O/dbg_swap.log
SONET_HIP_LIB=$V timeout 600 python tools/h3p_phases.py 320x384 1024x512 > $O/h3p_phases.log 2>&1; tail -5 $O/h3p_phases.log
