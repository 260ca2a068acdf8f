python -m pytest tests/test_gpu_segpool.py -q -x -k "batchnorm_backward" 2>&1 | tail -5
python tools/ab_h3_train.py --rounds 4 --steps 24 --only F --all-kernels 2>&1 | grep -v "amdgpu\|socket" | head -30
