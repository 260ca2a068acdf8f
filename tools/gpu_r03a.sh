#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== pytest som"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "som_assign_sort or sort_group or encoder_classifier" 2>&1 | tail -8
echo "== bench"; timeout 600 python bench.py --steps 100 --no-other-configs --no-other-precisions --no-cpu-baseline 2> $O/bench.err | tail -1 > $O/bench_forward.json; python - <<'PY'
import json
l=json.load(open('gpurun_out/r03n/bench_forward.json'))
print({k:l[k] for k in ('value','ms_per_step','single_stream')}, l['windows']['clouds_per_s'])
for k in l['kernels'][:8]: print("%-44s %.4f ms  %s" % (k['name'], k['ms_per_step'], k.get('valu')))
PY
tail -3 $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-other-configs --no-other-precisions > /dev/null 2>&1); f=$(find $O/rp -name '*kernel_stats.csv' | head -1); python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:14]: print("%8.1f us x%4s  %s" % (float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:90]))
PY
rm -rf $O/rp
