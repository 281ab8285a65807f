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


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_every_baseline_configuration_has_a_bench_workload_and_a_rule_file():
    """BASELINE.json configs[0..4] <-> bench.py workloads; the format choices of a workload are written into the product
    rule file the way a user edits the reference's (generate_op of the layer linears, backend of the routed experts)."""
    import yaml
    b = _bench()
    for name in ("v3-int4", "v2lite-int4", "r1-iq1s", "v3-fp8", "k2-rawint4", "mixtral-q4km"):
        assert name in b.WORKLOADS
    assert set(b.SECONDARY) == {"v2lite-int4", "r1-iq1s", "v3-fp8", "k2-rawint4", "mixtral-q4km"}
    assert b.WORKLOADS["r1-iq1s"]["layers"] == b.WORKLOADS["r1-iq1s"]["full_layers"] == 61          # the whole model
    rules = yaml.safe_load(open(b.rules_for(b.WORKLOADS["r1-iq1s"])))
    ops = {r["match"].get("name", ""): r["replace"]["kwargs"] for r in rules if "kwargs" in r.get("replace", {})}
    assert ops["^lm_head$"]["generate_op"] == "KLinearMarlin"
    assert ops["^model\\.layers\\.(?!.*self_attn\\.kv_b_proj).*$"]["generate_op"] == "KLinearFP8"
    assert ops["^model\\.layers\\..*\\.mlp\\.experts$"]["backend"] == "llamafile"
    rules = yaml.safe_load(open(b.rules_for(b.WORKLOADS["k2-rawint4"])))
    ops = {r["match"].get("name", ""): r["replace"]["kwargs"] for r in rules if "kwargs" in r.get("replace", {})}
    assert ops["^model\\.layers\\..*\\.mlp\\.experts$"]["backend"] == "RAWINT4"
    assert ops["^model\\.layers\\.(?!.*self_attn\\.kv_b_proj).*$"]["generate_op"] == "KLinearMarlin"
    assert b.rules_for(b.WORKLOADS["v3-int4"]).endswith("DeepSeek-V3-Chat.yaml")                     # unmodified product file


def test_algorithmic_bytes_per_token_follow_the_survey_figures():
    """routed experts of one V3 layer-token = 176.2 MB int4 / 352.3 MB fp8 / 68.8 MB IQ1_S / 198.2 MB RAWINT4."""
    from ktransformers_amd.models.modeling_deepseek import make_config
    b = _bench()
    H, I, k = 7168, 2048, 8
    for name, mb in (("v3-int4", 176.2), ("v3-fp8", 352.3), ("r1-iq1s", 68.8), ("k2-rawint4", 198.2)):
        gu, dn = b.expert_bpw(b.WORKLOADS[name])
        got = k * (2 * H * I * gu + H * I * dn) / 1e6
        assert abs(got - mb) / mb < 5e-3, (name, got, mb)
    wl = b.WORKLOADS["v3-int4"]
    cfg = make_config(**dict(b.MODELS["v3"], num_hidden_layers=32))
    tot, layer = b.step_bytes(cfg, 32, 4096, wl)
    assert abs(tot - 11148404736) / tot < 1e-6 and layer == 332348544                              # the round-2 figures, unchanged
    cfg = make_config(**dict(b.MODELS["v3"], num_hidden_layers=61))
    tot_r1, _ = b.step_bytes(cfg, 61, 4096, b.WORKLOADS["r1-iq1s"])
    assert 20e9 < tot_r1 < 23e9                                                                     # ~21 GB per token: fp8 linears dominate


def test_kernel_class_of_rocprof_names_and_labels():
    b = _bench()
    assert b._kclass("void (anonymous namespace)::lin_sk_kernel<64, 1, 7, 2>((anonymous namespace)::LinParams)") == "lin_sk_kernel"
    assert b._kclass("void moe_dec_gateup_kernel<4, 14, 4, true, 2>(DecParams)") == "moe_dec_gateup_kernel"
    assert b._kclass("lin_dec_gate_kernel<W4> 7168->4096 + router E=256") == "lin_dec_gate_kernel"
    assert b._kclass("mla_decode_kernel<2,4> T=1 Hq=128 nsplit=49") == "mla_decode_kernel"
    assert b._kclass("void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctor_add<long> >(int)") == "vectorized_elementwise_kernel"


def test_random_weight_sources_are_valid_blocks():
    """The synthetic weight sources of the secondary workloads: finite fp16 super-block scales at the ggml offsets the loader
    de-quantises from, block-fp8 round trip within e4m3's half-ulp."""
    import numpy as np
    import torch
    b = _bench()
    g = torch.Generator(device="cpu").manual_seed(0)
    from oracle.gguf_ref import DEQUANT
    for ty, nbytes in ((12, 144), (14, 210), (19, 50)):
        t = b.random_ggml_blocks(2, 3, 512, ty, g, torch.device("cpu"))
        assert tuple(t.shape) == (2, 3, 2 * nbytes)
        w = DEQUANT[ty](t.numpy().reshape(6, -1))
        assert np.isfinite(w).all() and 0.02 < float(np.std(w)) < 0.5 and np.abs(w).max() < 4.0, (ty, float(np.std(w)))
    w = (torch.randn(2, 256, 384) / 10).to(torch.bfloat16)
    q, sc = b.fp8_block_quant(w)
    assert q.dtype == torch.float8_e4m3fn and tuple(sc.shape) == (2, 2, 3)
    back = q.float().view(2, 2, 128, 3, 128) * sc[:, :, None, :, None]
    err = (back.reshape(2, 256, 384) - w.float()).abs()
    assert float((err / (w.float().abs() + 1e-3)).max()) < 0.07


def test_llamafile_cpu_leg_runs_the_reference_kernels():
    import pytest
    sys.path.insert(0, ROOT)
    from oracle.gguf_ref import iqk_forward_bench
    r = iqk_forward_bench(512, 1024, 2, (12, 12, 14), 4, budget_s=0.5, threads=2)
    if r["value"] is None:
        pytest.skip(r["sample"])
    assert r["kind"] == "reference" and r["cores"] == 2 and r["value"] > 0 and r["us_per_layer"] > 0


def test_time_budget_clock_and_the_truncated_line():
    """bench.py prints ONE line, at the end: RunClock stops optional sections from starting past --time-budget, and a SIGTERM
    after the headline measurement prints what exists (truncated_line) — marked, timed, with a roofline that never replaces
    a measured one."""
    b = _bench()
    t = [100.0]
    clock = b.RunClock(30.0, now=lambda: t[0])
    t0 = t[0]
    t[0] += 12.34
    clock.lap("decode", t0)
    assert clock.timing == {"decode": 12.3} and not clock.over_budget() and abs(clock.elapsed() - 12.34) < 1e-9
    t[0] += 20.0
    assert clock.over_budget()
    out = {"metric": "m", "value": 1.0, "whole_step": {"GBs": 2000.0, "frac_of_hbm_peak": 0.25}}
    line = b.truncated_line(out, clock, 15)
    assert "truncated" in line and "signal 15 after 32s" in line["truncated"] and line["timing_s"] == {"decode": 12.3}
    assert line["roofline"]["achieved"] == 2000.0 and line["roofline"]["frac"] == 0.25 and line["roofline"]["traffic"] is None
    assert "truncated" not in out and "roofline" not in out                      # the run's own dict is not touched
    out["roofline"] = {"kernel": "measured"}
    assert b.truncated_line(out, clock, 15)["roofline"] == {"kernel": "measured"}


def _synthetic_full_record():
    """A full record shaped like a real run's (`profiles/r04_z_bench.json`, the 25 KB line the driver could not parse), with
    every prose field stretched further."""
    rec = json.loads(open(os.path.join(ROOT, "profiles", "r04_z_bench.json")).read().strip().splitlines()[-1])
    rec["config"]["step"] = "x" * 4000
    rec["cpu_baseline"]["sample"] = "y" * 3000
    rec["per_kernel"] = rec["per_kernel"] * 4
    for k in ("v2lite_int4", "r1_iq1s", "v3_fp8", "k2_rawint4"):
        rec[k]["prefill"]["what"] = "z" * 2000
    rec["r1_iq1s"] = {"value": None, "error": "RuntimeError: " + "e" * 5000}
    return rec


def test_the_stdout_line_is_compact_and_round_trips():
    """Round 4's headline was lost (`BENCH_r04.parsed = null`) because bench.py printed one 25 KB line.  The LAST stdout line
    is now `compact_line(record)`: under 4 KB whatever the record holds, every contract field + roofline + cpu_baseline +
    one number pair per secondary workload; the full record goes to bench_detail.json."""
    import io
    import contextlib
    b = _bench()
    rec = _synthetic_full_record()
    assert len(json.dumps(rec)) > 30000
    line = b.compact_line(rec)
    text = json.dumps(line)
    assert len(text) < b.LINE_LIMIT == 4096 and "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "prefill", "secondary"):
        assert k in back, k
    assert back["value"] == rec["value"] and back["ms_per_step"] == rec["ms_per_step"]
    assert isinstance(back["config"]["workload"], str) and "model" not in back["config"]
    rf = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert k in rf, k
    assert rf == {**rf, **{k: rec["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}}
    cb = back["cpu_baseline"]
    assert cb["value"] == rec["cpu_baseline"]["value"] and cb["cores"] == 16 and cb["kind"] == "reference" and len(cb["sample"]) <= 150
    assert cb["prefill"]["value"] == rec["cpu_baseline"]["prefill"]["value"]
    assert back["prefill"]["value"] == rec["prefill"]["value"] and back["prefill"]["roofline"]["frac"] == rec["prefill"]["roofline"]["frac"]
    sec = back["secondary"]
    assert set(sec) == {"v2lite_int4", "r1_iq1s", "v3_fp8", "k2_rawint4", "mixtral_q4km"}
    assert sec["k2_rawint4"]["value"] == rec["k2_rawint4"]["value"] and sec["k2_rawint4"]["prefill"] == rec["k2_rawint4"]["prefill"]["value"]
    assert sec["r1_iq1s"]["value"] is None and len(sec["r1_iq1s"]["why"]) <= 100            # a failed section stays visible, short
    # what emit() prints: the detail file, then exactly one stdout line that parses on its own
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                b.emit(rec)
            lines = buf.getvalue().splitlines()
            assert len(lines) == 1 and json.loads(lines[-1]) == back
            assert json.loads(open(os.path.join(tmp, b.DETAIL_FILE)).read()) == rec
        finally:
            os.chdir(cwd)


def test_compact_line_of_a_truncated_and_of_an_experts_only_record():
    b = _bench()
    line = b.compact_line({"metric": "m", "value": 1.0, "unit": "tok/s", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 1000.0,
                           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                           "config": {"workload": "w"}, "truncated": "signal 15", "whole_step": {"GBs": 1.0, "frac_of_hbm_peak": 0.1}})
    assert line["truncated"] == "signal 15" and line["roofline"]["frac"] is None and len(json.dumps(line)) < 4096
