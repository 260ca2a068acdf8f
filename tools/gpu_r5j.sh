#!/bin/bash
# round 5, final tree (after the four-channel pooled dgrad and the carried gradient): pytest -m gpu, smoke, the driver's command, the two training
# lines, rocprofv3 kernel stats of the f32-class training step -- sized for the GPU minutes that were left
TAG=${1:-r05j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG/profiles; mkdir -p $P; export TMPDIR=/tmp
timeout 300 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $P/${TAG}_pytest_gpu.log; tail -2 $P/${TAG}_pytest_gpu.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v amdgpu | tail -3 > $P/${TAG}_smoke.log; cat $P/${TAG}_smoke.log
timeout 200 python bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > $P/${TAG}_bench_forward_driver_command.json
for p in bf16 h3; do timeout 100 python bench.py --mode train --precision $p --steps 40 --warmup 8 2> /dev/null | tail -1 > $P/${TAG}_bench_train_$p.json; done
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${TAG}_train_h3 -o tr -- python $R/bench.py --mode train --precision h3 --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/rocprof_train_h3.err < /dev/null)
f=$(find /tmp/rp_${TAG}_train_h3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats_train_h3.csv
python - <<PY
import json
for n in ("bench_forward_driver_command", "bench_train_bf16", "bench_train_h3"):
    try:
        d = json.load(open("$P/${TAG}_%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d.get("windows"), [(o.get("name"), o.get("ms_per_step")) for o in d.get("other_configs", [])] if isinstance(d.get("other_configs"), list) else "")
    except Exception as e:
        print(n, "unreadable:", e)
PY
ls $P
