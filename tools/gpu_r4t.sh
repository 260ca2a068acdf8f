#!/bin/bash
for cfg in "1" "0" "1" "0"; do
  echo "== side stream $cfg"
  SONET_BWD_SIDE_STREAM=$cfg python bench.py --mode train --precision bf16 --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  ms/step', d['ms_per_step'])"
done
