#!/bin/bash
# tools/prof_wgrad.sh -- kernel times of tools/bench_wgrad.py, one rocprofv3 kernel trace per shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for i in 0 1 2 3 4 5 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wgrad_$i -o t -- python $R/tools/bench_wgrad.py $i 2>&1 | grep "err"
  python - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/prof_wgrad_$i/*kernel_stats.csv")[0])):
    n = r["Name"]
    if "wgrad" in n or (n.startswith("Cijk") and "_S_B_" in n) or ("reduce_kernel" in n and int(r["Calls"]) >= 6):
        print("      %-60s calls %3s  avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
