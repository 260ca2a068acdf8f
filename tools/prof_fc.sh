#!/bin/bash
# tools/prof_fc.sh -- kernel times of tools/bench_fc.py (sonet_linear_act_f32) from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_fc -o t -- python $R/tools/bench_fc.py 2>&1 | grep "max err"
python - <<PY
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("$R/gpurun_out/prof_fc/*kernel_trace.csv")[0])) if "linear_act" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i in range(0, len(rows), 20):
    rs = rows[i + 5:i + 20]
    print("grid %s x %s: %.1f us" % (rows[i]["Grid_Size_X"], rows[i]["Grid_Size_Y"], sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / len(rs) / 1e3))
PY
