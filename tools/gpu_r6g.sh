#!/bin/bash
# headline: batches in flight 3..6, then a kernel trace of the three-stream replay (begin / end of every kernel: what runs in the gaps
# between two fused launches -- tools/gap_analysis.py)
TAG=${1:-r06g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG; mkdir -p $P
for p in 4 3 4 3; do
  timeout 120 python bench.py --steps 200 --warmup 20 --in-flight $p --no-cpu-baseline --no-other-configs --no-other-precisions 2>/dev/null | tail -1 > $P/bench_p$p.json
  python -c "
import json,sys
d=json.loads(open('$P/bench_p$p.json').read())
print('in_flight=$p value %.1f ms %.4f windows %s' % (d['value'], d['ms_per_step'], d['windows']['clouds_per_s']))"
done 2>&1 | tee $P/summary.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/trace -o trace -- python $R/bench.py --steps 60 --warmup 10 --spin-up 0.2 --windows 1 --no-cpu-baseline --no-other-configs --no-other-precisions --no-parity-check > $P/trace_bench.log 2>&1
cd $R
ls -la $P/trace | head; python tools/gap_analysis.py $P/trace/*kernel_trace.csv 2>&1 | tee $P/gaps.log
head -2 $P/trace/*kernel_trace.csv > $P/trace_head.txt; rm -rf $P/trace
