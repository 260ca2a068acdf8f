#!/bin/bash
# bf16 per-layer kernel: streaming generation vs staged, same box
export SONET_HIP_LIB=$PWD/so-net_amd/lib/libsonet_hip_variants.so
python tools/bfr_dbg.py 2>&1 | grep -v amdgpu
for n in 1 0 1 0; do echo "== SONET_BF16_STREAM=$n"; SONET_BF16_STREAM=$n timeout 300 python tools/bench_bf16_layers.py 2>&1 | grep -v amdgpu | head -7; done
unset SONET_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "streaming or statistics or same_operand" 2>&1 | tail -5
