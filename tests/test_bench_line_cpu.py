"""bench.py prints ONE line and the driver's record keeps the last ≈ 6 KB of stdout: the short form (bench.compact_line) must carry the headline contract, `roofline`,
`cpu_baseline` and every extra leg's numbers flat under `legs`, inside that budget.  The fixture is the full record of round 5's own run (profiles/r5_bench_n1.json)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_fits_the_drivers_tail_and_keeps_every_leg():
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_n1.json")))
    line = _bench().compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 5500, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("TPC-H SF100 Q1") and "model" not in line["config"]
    ro = line["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and ro["traffic"] > 0
    assert [k["name"] for k in ro["kernels"]] == ["k_gagg", "utf8_uniform_kernel"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 1e6 and cb["sample"]
    legs = line["legs"]
    for k in ("q3_ms", "q3_frac", "q95_ms", "q95_frac", "q95_stage_stage_a_ms", "q95_traffic_over_algorithmic", "q6_sf10_kernel_ms", "q6_sf10_task_ms", "pq6_snappy_ms", "pq6_zstd_ms",
              "exec_snappy_8_frac", "exec_snappy_16_frac", "exec_zstd_8_frac", "exec_zstd_16_frac", "snappy_decimal_int64_out_GBps", "zstd_decimal_int64_out_GBps", "link_GBps",
              "q3_verified", "q95_verified", "cold_create_plan_ms"):
        assert k in legs, k
    assert abs(legs["q95_ms"] - full["q95"]["sec_per_run"] * 1e3) < 1e-2 and legs["q95_kernels_ms"]["k_jprobe"] > 1
    # values only: no nested notes, no lists of per-kernel dicts outside roofline.kernels
    assert all(not isinstance(v, list) for v in legs.values())


def test_compact_line_survives_missing_and_failed_legs():
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_n1.json")))
    for k in ("q3", "executor_shape", "snappy_pipeline", "paths", "cpu_baseline", "cpu_baseline_all_cores"):
        full.pop(k, None)
    full["q95"] = {"error": "leg timed out after 420 s"}
    line = _bench().compact_line(full)
    assert "q3_ms" not in line["legs"] and line["legs"]["q95_error"].startswith("leg timed out") and "cpu_baseline" not in line
    assert len(json.dumps(line)) < 5500
