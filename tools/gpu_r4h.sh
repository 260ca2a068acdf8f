#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "segment or round2 or h3p" > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_forward.json 2> $O/bench_forward.err; tail -3 $O/bench_forward.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04h/bench_forward.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_stream'))
c=d['other_configs']['configs[2] segmenter']
print(c.get('clouds_per_s'), c.get('ms_per_step'), c.get('error'), c.get('parity_checked',{}).get('ok'))
for k in c.get('top_kernels',[]): print('   ',k)
P
