python -m pytest tests -q -m gpu 2>&1 | tail -3
for d in 1 0; do for p in bf16 h3; do echo -n "DEFER=$d $p: "; SONET_DEFER_WGRAD_JOIN=$d python bench.py --mode train --precision $p --steps 40 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
