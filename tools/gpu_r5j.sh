#!/bin/bash
mkdir -p gpurun_out/r5j
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --precision bf16 --steps 20 --warmup 3 --spin-up 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r5j/err.log
cp /tmp/prof_tr/tr_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r5j/train_bf16_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r5j/train_bf16_kernel_stats.csv")))
n = 23
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per step", tot / n / 1e6, "launches per step", sum(int(r["Calls"]) for r in rows) / n)
for r in rows:
    nm = r["Name"]
    if any(k in nm for k in ("pack", "bn_", "finalize", "ws_zero", "copyBuffer", "fillBuffer", "Cijk")):
        print("%-80s calls/step %6.2f avg %7.1f us" % (nm.replace("(anonymous namespace)::", "")[:80], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3))
PY
