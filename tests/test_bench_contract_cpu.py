"""CPU: the bench line committed under profiles/ (what `python bench.py` printed on the MI355X box) carries every field of the
driver's contract, and bench.py still parses the contract's flags."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no bench line committed under profiles/"
    line = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert isinstance(line["config"].get("workload"), str) and "model" not in line["config"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"] * line["n_gpus"]) / line["value"] < 1e-2
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and (rf["traffic"] is None or rf["traffic"] > 0)
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert line["vs_baseline"] is None   # BASELINE.md publishes no number for this metric


def test_bench_cli_accepts_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
