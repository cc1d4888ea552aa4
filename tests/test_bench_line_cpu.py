"""bench.py prints ONE bounded JSON line (the driver keeps only the tail of stdout: a 20 KB line made BENCH_r04.json unparseable).
The formatter is checked on a canned full result (the round-4 run, profiles/r04_bench_full_n1.json) and on a hostile one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (imports nothing heavy at module level)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench_full_n1.json")) as f:
        return json.load(f)


def test_line_is_bounded_and_round_trips():
    out = _canned()
    text = bench.compact_line(out, "bench_extras.json")
    assert "\n" not in text and len(text) < 4096
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == out["value"] and line["ms_per_step"] == out["ms_per_step"] and line["n_gpus"] == 1 and line["config"]["workload"].startswith("gridworld")
    roof, cpu = line["roofline"], line["cpu_baseline"]
    for k in ("kernel", "bound", "launch_us", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "host_logical_cpus"):
        assert k in cpu, k
    assert len(cpu["sample"]) <= 200 and cpu["kind"] == "port"
    assert line["extras_path"] == "bench_extras.json" and "dropped_to_fit" not in line
    assert len(line["extra_configs"]) == len(out["extra_configs"]) + 1  # (+ the legend entry)


def test_oversized_blocks_are_dropped_not_truncated():
    out = _canned()
    out["extra_configs"] = [dict(e, task=f"{e['task']}{i}") for i in range(12) for e in out["extra_configs"]]  # a run that grew: optional blocks go, the contract keys stay
    out["config"]["workload"] = "x" * 5000
    text = bench.compact_line(out, "bench_extras.json")
    assert len(text) < 4096
    line = json.loads(text)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in line, k
    assert "extra_configs" in line["dropped_to_fit"]


def test_multi_gpu_line_and_errors():
    out = _canned()
    out["n_gpus"] = 8
    out.pop("cpu_baseline"), out.pop("extra_configs")
    out["roofline_gae_kernel"] = {"error": "boom " * 100}
    tm = {"calls_timed": 64, "median_us": 12.5, "max_us": 40.0, "bytes": 37400, "path": "native " * 40}
    out["dp_timing"] = {"backend": "nccl", "allreduce_path": "native (libtma_hip.so ...)", "grad_allreduce": tm, "adv_sums_allreduce": tm,
                        "grad_allreduces_per_iteration": 320, "per_rank_rollout_ms": [3.5] * 8, "per_rank_update_ms": [31.0] * 8, "note": "n" * 900}
    line = json.loads(bench.compact_line(out, None))
    assert line["dp_timing"]["grad_allreduce"] == {"calls_timed": 64, "median_us": 12.5, "max_us": 40.0, "bytes": 37400}
    assert line["dp_timing"]["allreduce_path"] == "native" and len(line["other_rooflines"]["gae"]["error"]) <= 80


def test_extras_file(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    p = bench.write_extras({"n_gpus": 1, "value": 1.0, "extra_configs": [1, 2, 3]})
    assert p == "bench_extras.json"
    for d in (tmp_path, tmp_path / "gpurun_out"):
        with open(d / "bench_extras.json") as f:
            assert json.load(f)["extra_configs"] == [1, 2, 3]
