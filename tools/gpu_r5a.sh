#!/bin/bash
# round 5, first GPU call: the new node-level stage -- tests, then the bench line with the round-4 stage and with the new one on the same box
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_node_stage.py tests/test_gpu_h3p.py tests/test_gpu_optim.py tests/test_gpu_node_train.py -x -q 2>&1 | tail -15 > gpurun_out/r5a/pytest_new.log
cat gpurun_out/r5a/pytest_new.log
SONET_NODE_STAGE_P16=0 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench_stage_r4.json 2> gpurun_out/r5a/bench_stage_r4.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench_stage_r5.json 2> gpurun_out/r5a/bench_stage_r5.err
tail -3 gpurun_out/r5a/bench_stage_r5.err
python - <<'PY'
import json
for n in ("r4", "r5"):
    try:
        d = json.loads(open("gpurun_out/r5a/bench_stage_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["single_stream"], d.get("kernel_ms_per_step"), d["parity_checked"]["ok"], d["parity_checked"]["feature_err_over_bound"])
        for k in d["kernels"]:
            print("   ", k["name"], k["mean_ms"], k.get("frac"))
    except Exception as e:
        print(n, "failed", e)
PY
