#!/bin/bash
# round-3 GPU call: third-generation fused kernel vs the second generation (bit identity + time), ablations, phase counters,
# the GPU suite and the forward bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== variants"; ITERS=25 timeout 900 python tools/fused_variants.py gen2 gen3 now nob nojob nowb nowbj 2>&1 | grep -v amdgpu.ids | tee $O/variants.log | tail -30
echo "== phases"; VARIANT=gen3prof timeout 300 python tools/fused_phases.py pool 2>&1 | grep -v amdgpu.ids | tee $O/phases_pool.log | tail -30
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 10 2> $O/bench.err | tail -1 > $O/bench_forward.json; head -c 1500 $O/bench_forward.json; echo
