#!/bin/bash
# round 4 final measurement set (one gpurun call): tools/gpu_round.sh (pytest -m gpu, smoke, bench forward, rocprofv3 forward stats, PMC traffic,
# bench train) + the bf16 / h3 training lines, rocprofv3 of the bf16 training step, the micro-benchmarks of this round's kernels.
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
SKIP="${SKIP:-6 7}" bash tools/gpu_round.sh $TAG
P=$R/gpurun_out/$TAG/profiles; export TMPDIR=/tmp
for p in bf16 h3; do timeout 300 python bench.py --mode train --precision $p --steps 40 --warmup 8 2> /dev/null | tail -1 > $P/${TAG}_bench_train_$p.json; head -c 200 $P/${TAG}_bench_train_$p.json; echo; done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_train -o tr -- python $R/bench.py --mode train --precision bf16 --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/rocprof_train.err < /dev/null)
f=$(find /tmp/rp_${TAG}_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_train_bf16.csv
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_train_h3 -o tr -- python $R/bench.py --mode train --precision h3 --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/rocprof_train_h3.err < /dev/null)
f=$(find /tmp/rp_${TAG}_train_h3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_train_h3.csv
timeout 200 python tools/train_cpu_time.py bf16 2>&1 | grep -v amdgpu > $P/${TAG}_train_host_time.log; timeout 200 python tools/train_cpu_time.py h3 2>&1 | grep -v amdgpu >> $P/${TAG}_train_host_time.log
{ echo "== tools/bench_bf16_layers.py"; timeout 300 python tools/bench_bf16_layers.py 2>&1 | grep -v amdgpu
  echo "== tools/bench_pooled.py"; timeout 300 python tools/bench_pooled.py 2>&1 | grep -v amdgpu
  echo "== tools/bench_wgrad_bf16.py"; timeout 300 python tools/bench_wgrad_bf16.py 2>&1 | grep -v amdgpu
  echo "== tools/bench_bwd_passes.py"; timeout 300 python tools/bench_bwd_passes.py 2>&1 | grep -v amdgpu; } > $P/${TAG}_microbench_train.log 2>&1
tail -5 $P/${TAG}_microbench_train.log; ls $P
# round 5: the node-level stage of the headline, A (round-4 stage) against B (flat stage) in one process, and its launches one by one
timeout 300 python tools/ab_node_stage.py --rounds 10 --steps 40 2>&1 | grep -v amdgpu > $P/${TAG}_ab_node_stage.log; cat $P/${TAG}_ab_node_stage.log
timeout 300 python tools/bench_node_stage.py 2>&1 | grep -v amdgpu > $P/${TAG}_bench_node_stage.log
# the driver's own command as the first thing a fresh process does
timeout 300 python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $P/${TAG}_bench_forward_driver_command.json; head -c 300 $P/${TAG}_bench_forward_driver_command.json; echo
