"""The bench line's contract (task statement, measurement section): the line committed under profiles/ -- produced by `python bench.py` on
the MI355X box -- carries every required field with sane values, and bench.py still parses and keeps its documented defaults."""
import ast
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("name", ["r01_bench_turbo_b32.json", "r02_bench_turbo_b32.json", "r03_bench_turbo_b32.json"])
def test_committed_bench_line_has_every_contract_field(name):
    line = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-sec/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"          # BASELINE.md publishes no number for this metric
    assert "workload" in d["config"] and "large-v3-turbo" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 30.0 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]           # value = clips x 30 s / time per pass
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or 0.9 * r["bytes_per_launch"] <= r["traffic"] <= 1.2 * r["bytes_per_launch"]   # PMC bytes ~ algorithmic bytes
    if name.startswith("r02"):
        # the roofline's launch time comes from strictly serial passes in the same process, reported beside the pipelined `value`
        e = r["execution"]
        assert e["replicas"] == 1 and e["value"] <= d["value"] and abs(e["value"] - 32 * 30.0 / (e["ms_per_step"] * 1e-3)) <= 0.01 * e["value"]
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
        assert d["config"]["exchange"] == "none" and "config0" in d and d["config0"]["gpu"]["generated_tokens"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0


def test_bench_defaults_and_syntax():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args and isinstance(node.args[0], ast.Constant):
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[node.args[0].value] = kw.value.value
    assert defaults["--gpus"] == 1 and defaults["--batch"] == 32 and defaults["--model"] == "large-v3-turbo" and defaults["--dtype"] == "bf16"
    assert defaults["--steps"] >= 1 and defaults["--warmup"] >= 0
