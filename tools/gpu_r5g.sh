#!/bin/bash
# the two bench lines of the final tree (bench.py's order of entries changed after the r05f set): default command and the driver's command
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P
timeout 600 python bench.py --steps 50 --warmup 10 2> /dev/null | tail -1 > $P/${TAG}_bench_forward.json
timeout 600 python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $P/${TAG}_bench_forward_driver_command.json
for p in bf16 h3; do timeout 300 python bench.py --mode train --precision $p --steps 40 --warmup 8 2> /dev/null | tail -1 > $P/${TAG}_bench_train_$p.json; done
ls $P
