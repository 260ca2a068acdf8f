#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of the two training steps: gpurun_out/TAG/TAG_kernel_stats_train_{bf16,h3}.csv
TAG=${1:-r06j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG; mkdir -p $P
export TMPDIR=/tmp
for prec in bf16 h3; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_$prec -o t -- python $R/bench.py --mode train --precision $prec --steps 20 --warmup 5 > $P/bench_$prec.log 2> $P/rocprof_$prec.err)
  f=$(find /tmp/rp_${TAG}_$prec -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_train_$prec.csv
  tail -1 $P/bench_$prec.log | cut -c1-160
done
