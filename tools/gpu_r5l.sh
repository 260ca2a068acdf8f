#!/bin/bash
mkdir -p gpurun_out/r5l
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -25 > gpurun_out/r5l/pytest.log; cat gpurun_out/r5l/pytest.log
timeout 300 python bench.py --mode train --precision bf16 --steps 40 --warmup 5 2> gpurun_out/r5l/train_bf16.err | tail -1 > gpurun_out/r5l/train_bf16.json
SONET_POOLED_TRAIN_EPILOGUE=0 timeout 300 python bench.py --mode train --precision bf16 --steps 40 --warmup 5 2> /dev/null | tail -1 > gpurun_out/r5l/train_bf16_store.json
python - <<'PY'
import json
for n in ("train_bf16", "train_bf16_store"):
    try:
        d = json.loads(open("gpurun_out/r5l/%s.json" % n).read())
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -5 gpurun_out/r5l/train_bf16.err
