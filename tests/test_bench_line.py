"""bench.py's ONE JSON line stays parsable by the driver: round 5's line was 21 KB and `BENCH_r05.json.parsed` came back null.
The full result of that run (profiles/r05_final_bench_steps20_warmup5.json, a real `out` dict) through `bench.compact_line`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench_steps20_warmup5.json")))


def test_bench_line_is_compact_and_carries_the_contract():
    import bench
    out = _canned()
    assert len(json.dumps(out)) > 3 * bench.LINE_LIMIT          # the canned dict is the one that was too long
    text = bench.compact_line(out, "bench_detail.json")
    assert len(text) < 6144 and "\n" not in text
    rec = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in rec, k
    assert len(rec["config"]["workload"]) <= 300 and "model" not in rec["config"]
    roof = rec["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches", "sizes_per_step", "step"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] <= 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(roof["frac"] - out["roofline"]["frac"]) < 1e-3 * out["roofline"]["frac"]
    assert abs(rec["value"] - out["value"]) <= 1.0
    cb = rec["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert set(rec["psnr_at_iter"]["delta_db"]) == {"100", "200", "300"}
    # one number per extras leg
    assert all(isinstance(v, (int, float, str)) or v is None for v in rec["extras"].values())
    assert len(rec["extras"]) >= len([k for k in out["extras"] if k != "psnr_at_iter"])


def test_bench_line_never_exceeds_the_limit_even_with_bloated_legs():
    import bench
    out = _canned()
    out["extras"] = {f"leg_{i}": {"ms_per_step": 1.0 + i, "note": "x" * 500} for i in range(400)}
    out["config"]["workload"] = "w" * 5000
    text = bench.compact_line(out)
    assert len(text) < 6144
    rec = json.loads(text)
    assert "roofline" in rec and "cpu_baseline" in rec and len(rec["config"]["workload"]) <= 300
