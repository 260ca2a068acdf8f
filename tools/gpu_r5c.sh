#!/bin/bash
mkdir -p gpurun_out/r5c
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5c/pytest_gpu.log
cat gpurun_out/r5c/pytest_gpu.log
SONET_NODE_STAGE_P16=0 python bench.py --steps 20 --warmup 5 > gpurun_out/r5c/bench_stage_r4.json 2> gpurun_out/r5c/bench_stage_r4.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r5c/bench_stage_r5.json 2> gpurun_out/r5c/bench_stage_r5.err
tail -3 gpurun_out/r5c/bench_stage_r5.err
python - <<'PY'
import json
for n in ("r4", "r5"):
    try:
        d = json.loads(open("gpurun_out/r5c/bench_stage_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["windows"]["clouds_per_s"], d["single_stream"], d.get("kernel_ms_per_step"), d["parity_checked"]["ok"], d["parity_checked"]["feature_err_over_bound"])
        for k in d["kernels"]:
            print("   ", k["name"], k["mean_ms"], k.get("frac"))
        for k, v in d["other_configs"].items():
            print("   ", k, {a: v[a] for a in v if a in ("ms_per_step", "clouds_per_s", "value")})
    except Exception as e:
        print(n, "failed", e)
PY
