#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== pytest"; timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest.log
