#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "pooled" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k,v in d.get('other_configs',{}).items():
    print(k, {kk: v[kk] for kk in v if kk in ('ms_per_step','clouds_per_s','parity')} if isinstance(v,dict) else v)
"
