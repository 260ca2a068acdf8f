#!/bin/bash
# tools/gpu_round_full.sh TAG -- tools/gpu_round.sh plus the SQ counters of the fused kernel (tools/pmc_sq.sh) and the training-step kernel stats;
# raw traces stay under /tmp on the GPU box, the summaries land in gpurun_out/TAG/profiles (copy them into profiles/).
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SKIP="${SKIP:-7}" bash tools/gpu_round.sh $TAG
P=$R/gpurun_out/$TAG/profiles
timeout 400 bash tools/pmc_sq.sh fused_pool > $R/gpurun_out/$TAG/pmc_sq.log 2>&1
for p in a b c; do f=$(find $R/gpurun_out/pmc_sq_fused_pool_$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_pmc_sq_fused_pool_$p.csv; rm -rf $R/gpurun_out/pmc_sq_fused_pool_$p; done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_train -o tr -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$TAG/rocprof_train.err < /dev/null)
f=$(find /tmp/rp_${TAG}_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_train.csv
du -sh $R/gpurun_out; ls $P
