#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== variants"; ITERS=25 timeout 900 python tools/fused_variants.py gen2 gen3 2>&1 | grep -v "amdgpu.ids" | tee $O/variants.log | tail -30
v=gen3prof; echo "== $v"; VARIANT=$v B=64 timeout 300 python tools/fused_phases.py pool 2>&1 | grep -v amdgpu.ids | tee $O/${v}.log | grep -E "mode|layer|barrier|epilogue|clock"
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q -k "fused or pool or golden" 2>&1 | tail -5
