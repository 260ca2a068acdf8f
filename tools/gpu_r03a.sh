#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest.log
