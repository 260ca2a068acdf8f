#!/bin/bash
# tools/pmc_traffic.sh -- HBM traffic of the hot kernels from rocprofv3 PMC counters (run on the GPU box):
#   separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), kernel-trace only.
# Writes gpurun_out/pmc_traffic_{fetch,write}/ ; tools/pmc_traffic.py turns them into profiles/pmc_traffic.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_$c -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-other-precisions --no-parity-check --no-other-configs $EXTRA > /dev/null 2> $R/gpurun_out/pmc_traffic_$c.err
done
