#!/bin/bash
# round 5, second session: the full measurement set of tools/gpu_r5z.sh + the f32-class training path's own tools
#   (tools/ab_h3_train.py: store + index_max | sorted pool | + normalise-on-load in one process; tools/bench_pooled_sorted.py; tools/grad_dev_h3.py)
TAG=${1:-r05c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/gpu_r5z.sh $TAG
P=$R/gpurun_out/$TAG/profiles; export TMPDIR=/tmp
timeout 400 python tools/ab_h3_train.py --rounds 5 --steps 24 --only ABCFG 2>&1 | grep -v "amdgpu\|socket" > $P/${TAG}_ab_h3_train.log; head -4 $P/${TAG}_ab_h3_train.log
timeout 300 python tools/bench_pooled_sorted.py 2>&1 | grep -v "amdgpu\|Warning\|detach" > $P/${TAG}_bench_pooled_sorted.log; cat $P/${TAG}_bench_pooled_sorted.log
{ for f in train_step_b8_n5000 train_step_b16_n512; do echo "== tools/grad_dev_h3.py $f"; timeout 300 python tools/grad_dev_h3.py $f 2>&1 | grep -v "amdgpu\|Warning\|detach\|print("; done; } > $P/${TAG}_grad_dev_h3.log
ls $P | wc -l
