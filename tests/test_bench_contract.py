"""bench.py's output contract (the driver parses the last stdout line): one JSON object with the agreed keys, the roofline
and cpu_baseline objects, and the BASELINE config in `config`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_bench_cli_declares_the_contract_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src


@pytest.mark.gpu
def test_bench_line_contract(device):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = res.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert KEYS <= set(d), KEYS - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["unit"] == "items/s" and d["value"] > 1000 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "ViT-L/14" in d["config"]["workload"]
    r = d["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 1.0
    assert "traffic_source" in r and (r["traffic"] is None or r["traffic_source"]["commit"])
    c = d["cpu_baseline"]
    assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    # the bench checks its own outputs against the oracle: images AND texts, rows from the start, middle and end of the batch
    assert c["gpu_vs_oracle_min_cosine_images"] > 1 - 1e-3 and c["gpu_vs_oracle_min_cosine_texts"] > 1 - 1e-3
    rs = d["roofline_sim"]
    assert rs["bound"] == "mfma" and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-9 and rs["ms"] > 0
    assert {"q1024_bf16", "q43000_bf16", "q43000_bf16_rank_only", "q43000_bf16_c3_fused_t2i_t2t"} <= set(d["sim_top10"])
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["shard_bounds"] == [[0, 43000]] and d["config"]["items_timed"] == 2 * 765
    # images + texts of all ranks / time: 255 gallery items per step = 255 images + 510 texts
    assert abs(d["value"] - 765 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6
    assert abs(d["value"] - (d["images_per_s"] + d["texts_per_s"])) / d["value"] < 1e-9
