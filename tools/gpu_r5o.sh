#!/bin/bash
# rocprofv3 kernel stats of the h3 training step (bench.py --mode train --precision h3) + the bench lines bf16 / h3
TAG=${1:-r5o}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for p in h3 bf16; do timeout 300 python bench.py --mode train --precision $p --steps 40 --warmup 8 2> /dev/null | tail -1 > $O/bench_train_$p.json; head -c 260 $O/bench_train_$p.json; echo; done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_h3 -o tr -- python $R/bench.py --mode train --precision h3 --steps 10 --warmup 3 > /dev/null 2> $O/rocprof_train_h3.err < /dev/null)
f=$(find /tmp/rp_h3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train_h3.csv; head -30 $O/kernel_stats_train_h3.csv | cut -c1-160
