"""debug: the configs[1] h3 training entry of bench.other_configs in a fresh process, alone / after the bf16 entry / with empty_cache in between"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "so-net_amd")); sys.path.insert(0, ROOT)
import torch
import bench
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5"]
args = bench.parse_args() if hasattr(bench, "parse_args") else None
print("parse_args:", args is not None)
