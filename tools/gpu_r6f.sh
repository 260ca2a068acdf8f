#!/bin/bash
# persistent grid of the fused first PointNet: CUs left free for the other graphs' small launches (SONET_FUSED_FREE_CUS: 0 = every CU,
# -1 ("auto") = the smallest grid with the same number of rounds, k = at least k CUs out), interleaved A/B of the driver's headline.
# The knob is read by the VARIANTS library only (the product library reads no environment variable): SONET_HIP_LIB selects it.
TAG=${1:-r06f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P=$R/gpurun_out/$TAG; mkdir -p $P
for rep in 1 2; do
for k in 0 auto 8 12 20; do
  export SONET_HIP_LIB=$R/so-net_amd/lib/libsonet_hip_variants.so
  if [ $k = auto ]; then export SONET_FUSED_FREE_CUS=-1; else export SONET_FUSED_FREE_CUS=$k; fi
  timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-other-precisions 2>/dev/null | tail -1 > $P/bench_k${k}_$rep.json
  python - $P/bench_k${k}_$rep.json $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
ss = d.get("single_stream") or {}
print("free=%s value %.1f ms %.4f  single_stream %s  windows %s" % (sys.argv[2], d["value"], d["ms_per_step"], ss.get("value") if isinstance(ss, dict) else ss, d.get("windows")))
PY
done; done 2>&1 | tee $P/summary.log
