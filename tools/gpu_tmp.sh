python -m pytest tests -q -m gpu -k "pooled or segpool" 2>&1 | tail -3
python tools/bench_pooled_sorted.py 2>&1 | grep -v "amdgpu\|Warning\|detach"
