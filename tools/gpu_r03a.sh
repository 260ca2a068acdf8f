#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== bench"; ( time timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench_forward.json ) 2>&1 | grep real; tail -5 $O/bench.err; python - <<'PY'
import json
l=json.load(open('gpurun_out/r03h/bench_forward.json'))
print({k:l[k] for k in ('value','ms_per_step','windows','single_stream')})
print(l['ranks'])
print(l['roofline'])
print(l['parity_checked'])
for k,v in l.get('other_configs',{}).items(): print(k, json.dumps(v)[:1500])
print(l.get('cpu_baseline'))
PY
echo "== pytest bench"; timeout 900 python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu 2>&1 | tail -15
