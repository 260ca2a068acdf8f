#!/bin/bash
# round 6 final set in one gpurun call: tools/gpu_round.sh (pytest -m gpu, smoke, bench forward, rocprofv3 kernel stats, PMC traffic, bench train)
# + the driver's own command as a fresh process + the bf16 training line + kernel stats of both training steps
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
SKIP="7" bash tools/gpu_round.sh $TAG
P=$R/gpurun_out/$TAG/profiles
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${TAG}_bench_forward_driver_command.json; head -c 300 $P/${TAG}_bench_forward_driver_command.json; echo
timeout 300 python bench.py --mode train --precision bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${TAG}_bench_train_bf16.json; head -c 250 $P/${TAG}_bench_train_bf16.json; echo
mv $P/${TAG}_bench_train.json $P/${TAG}_bench_train_h3.json 2>/dev/null
bash tools/gpu_r6j.sh $TAG > /dev/null 2>&1; cp $R/gpurun_out/$TAG/${TAG}_kernel_stats_train_*.csv $P/ 2>/dev/null
ls $P
