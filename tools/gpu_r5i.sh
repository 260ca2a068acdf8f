#!/bin/bash
mkdir -p gpurun_out/r5i
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fc_head.py -x -q 2>&1 | tail -25 > gpurun_out/r5i/pytest.log
cat gpurun_out/r5i/pytest.log
python tools/bench_wgrad_bf16.py > gpurun_out/r5i/wgrad_bf16.log 2>&1; cat gpurun_out/r5i/wgrad_bf16.log
python bench.py --mode train --precision bf16 --steps 40 --warmup 5 > gpurun_out/r5i/train_bf16.json 2> gpurun_out/r5i/train_bf16.err
SONET_PACK_REGISTRY=0 python bench.py --mode train --precision bf16 --steps 40 --warmup 5 > gpurun_out/r5i/train_bf16_noreg.json 2> gpurun_out/r5i/train_bf16_noreg.err
python bench.py --mode train --precision h3 --steps 30 --warmup 5 > gpurun_out/r5i/train_h3.json 2> gpurun_out/r5i/train_h3.err
python - <<'PY'
import json
for n in ("train_bf16", "train_bf16_noreg", "train_h3"):
    try:
        d = json.loads(open("gpurun_out/r5i/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e, open("gpurun_out/r5i/%s.err" % n).read()[-1500:])
PY
