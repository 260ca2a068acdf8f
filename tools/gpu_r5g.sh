#!/bin/bash
mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_node_stage.py -x -q 2>&1 | tail -3
export SONET_HIP_LIB=$GRAFT_REPO_ROOT/so-net_amd/lib/libsonet_hip_variants.so
cd /tmp && export TMPDIR=/tmp
for gy in 1 2 4 8; do
SONET_KSI_GY=$gy rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gy$gy -o ab -- python $GRAFT_REPO_ROOT/tools/ab_node_stage.py --rounds 2 --steps 30 --in-flight 1 > /dev/null 2>&1
python -c "
import csv
for r in csv.DictReader(open('/tmp/prof_gy$gy/ab_kernel_stats.csv')):
    if 'knn_stage_input' in r['Name'] or 'H3pArgs' in r['Name'] or 'fill2' in r['Name']: print('gy $gy', r['Name'][:60], r['Calls'], r['AverageNs'])
"
done
