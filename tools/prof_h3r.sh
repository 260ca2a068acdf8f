#!/bin/bash
# tools/prof_h3r.sh -- kernel durations of tools/bench_h3r.py from a rocprofv3 kernel trace (host launch overhead excluded)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_h3r -o t -- python $R/tools/bench_h3r.py > $R/gpurun_out/prof_h3r.log 2>&1
python - <<PY
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("$R/gpurun_out/prof_h3r/*kernel_trace.csv")[0]))
        if "pointmlp_h3r" in r["Kernel_Name"] or "pointmlp_x3_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
lines = [l for l in open("$R/gpurun_out/prof_h3r.log") if "bit-identical" in l]
i = 0
for l in lines:
    L = int(re.search(r"L=(\d+)", l).group(1))
    n = 3 + (20 if L < 5000 else 5)
    new, old = rows[i:i + n], rows[i + n:i + 2 * n]
    i += 2 * n
    d = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs[3:]) / max(1, len(rs) - 3) / 1e3
    print("%s | kernel new %8.1f us (%s)  old %8.1f us (%s)" % (l.split(":")[0], d(new), new[0]["Kernel_Name"][24:44], d(old), old[0]["Kernel_Name"][24:52]))
PY
