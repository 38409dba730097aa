"""The bench contract on the committed line (profiles/r06_bench_c4.json = `python bench.py` on one MI355X): the keys the
driver reads, the two objects the measurement rules ask for (`roofline`, `cpu_baseline`) and their internal arithmetic.
No GPU needed: this reads the committed evidence; `bench.py --help` shows the flags of the contract exist."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads([l for l in f if l.startswith("{")][-1])


def test_committed_bench_line_keeps_the_contract():
    d = _line("r06_bench_c4.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["unit"] == base.get("unit", d["unit"]) and "workload" in d["config"] and "model" not in d["config"]
    assert "configs[3]" in d["config"]["workload"]  # the configuration the metric is quoted on, on one GPU
    mbases = d["config"]["read_bases"] / 1e6  # Gbase/s x ms = Mbases: the whole read set goes through every step
    assert d["value"] > 1.0 and mbases > 2900 and abs(d["ms_per_step"] * d["value"] - mbases) < 0.01 * mbases
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    src = d.get("roofline_traffic_source") or {}
    assert src.get("stale") is False, "the PMC traffic file was taken on other kernel sources than the bench line"
    hbm = d.get("roofline_hbm_kernels") or []
    assert any(e["kernel"] == "match_count" for e in hbm)
    for e in hbm:
        assert abs(e["frac"] - e["achieved"] / e["peak"]) < 2e-3


def test_bench_flags_of_the_contract_exist():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout, flag
