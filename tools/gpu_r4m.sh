#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04h3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for p in h3; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$p -o t -- python $R/bench.py --mode train --precision $p --steps 10 --warmup 3 > /dev/null 2> $O/rocprof_$p.err
f=$(find /tmp/rp_$p -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train_$p.csv
done
