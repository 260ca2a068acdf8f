#!/bin/bash
# the entry-balanced sparse input gradient (pooled_dgrad5_kernel) against the column-owned one (pooled_dgrad4_kernel), both out of the VARIANTS
# library (SONET_PD_KERNEL=4|5): the tests that run it, the microbench on a real step's entries, and the bf16-output form beside the matrix-core kernel
TAG=${1:-r05k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P
export SONET_HIP_LIB=$R/so-net_amd/lib/libsonet_hip_variants.so
SONET_PD_KERNEL=5 timeout 150 python -m pytest tests -q -m gpu -x -k "pooled or training_step" 2>&1 | tail -4 > $P/${TAG}_pytest_pooled_kernel5.log
for k in 4 5; do
  { echo "== SONET_PD_KERNEL=$k tools/bench_pooled_sorted.py"; SONET_PD_KERNEL=$k timeout 80 python tools/bench_pooled_sorted.py 2>&1 | grep -v "amdgpu\|Warning\|detach"
    echo "== SONET_PD_KERNEL=$k tools/bench_pooled.py"; SONET_PD_KERNEL=$k timeout 60 python tools/bench_pooled.py 2>&1 | grep -v "amdgpu\|Warning\|detach"; } > $P/${TAG}_bench_pooled_kernel$k.log 2>&1
done
cat $P/${TAG}_pytest_pooled_kernel5.log; grep "pooled_dgrad\|side by side\|==" $P/${TAG}_bench_pooled_kernel4.log $P/${TAG}_bench_pooled_kernel5.log | cut -c1-200
