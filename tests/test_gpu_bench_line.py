"""bench.py end to end on the visible GPUs (the driver's command line at N = torch.cuda.device_count()): one JSON line with the
contract keys, the roofline / cpu_baseline objects, every BASELINE config under other_configs, and per-rank device evidence."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_end_to_end_on_the_visible_gpus():
    import torch
    n = torch.cuda.device_count()
    assert n >= 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "2", "--other-steps", "3",
           "--cpu-clouds", "1", "--no-other-precisions"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "windows", "ranks"):
        assert key in line, key
    assert line["n_gpus"] == n and line["steps"] == 3 and line["value"] > 0
    assert line["ranks"]["world"] == n and line["ranks"]["distinct_devices"] == n and len(line["ranks"]["devices"]) == n
    assert len({(d["uuid"], d["pci"]) for d in line["ranks"]["devices"]}) == n
    assert len(line["windows"]["clouds_per_s"]) == 3
    assert line["roofline"]["frac"] > 0
    if n == 1:
        assert line["parity_checked"]["ok"] and line["parity_checked"]["node_ids_bit_exact_all_clouds"]["ok"]
        assert line["cpu_baseline"]["value"] > 0
        oc = line["other_configs"]
        for key in ("configs[1] train bf16", "configs[1] train h3", "configs[2] segmenter", "configs[3] autoencoder"):
            assert key in oc and "error" not in oc[key], (key, oc.get(key))
            assert oc[key]["clouds_per_s"] > 0 and oc[key]["parity_checked"]["ok"], (key, oc[key]["parity_checked"])
        assert oc["configs[1] train bf16"]["allreduce"]["backend"] == "nccl"
        train_keys = ["configs[1] train bf16", "configs[1] train h3"]
    else:
        # the driver's multi-GPU command: the forward line above AND BASELINE configs[4] (data-parallel training, RCCL all-reduce)
        oc = line["other_configs"]
        train_keys = ["configs[4] train bf16", "configs[4] train h3"]
    for key in train_keys:
        e = oc[key]
        assert e["n_gpus"] == n and e["global_batch"] == n * e["batch_per_gpu"] and e["clouds_per_s"] > 0
        assert e["allreduce"]["rccl_world_size"] == n and e["allreduce"]["bytes_per_step"] > 0 and e["allreduce"]["exposed_us_per_step"] >= 0
        assert 0 < e["per_rank_ms_per_step"]["min"] <= e["per_rank_ms_per_step"]["max"] <= e["ms_per_step"] * 1.5
        assert e["parity_checked"]["ok"]
