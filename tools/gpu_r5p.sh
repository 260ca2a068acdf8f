#!/bin/bash
# rocprofv3 kernel stats of the f32-class training step on the final tree (entry-balanced sparse input gradient, carried gradient)
TAG=${1:-r05p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P; export TMPDIR=/tmp
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_train_h3 -o tr -- python $R/bench.py --mode train --precision h3 --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/rocprof_train_h3.err < /dev/null)
f=$(find /tmp/rp_${TAG}_train_h3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_train_h3.csv
head -12 $P/${TAG}_kernel_stats_train_h3.csv | cut -c1-160
