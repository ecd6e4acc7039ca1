"""The stdout contract of bench.py: ONE JSON line, at most 8 KiB, parseable the way the driver parses it (round 4's
32.8 kB line was recorded as `parsed: null`).  Built here from canned records -- no GPU, no timing."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _canned(n_gpus=1, bloat=1):
    line = {
        "metric": "point-clouds/sec", "value": 31416.123456789, "unit": "point-clouds/sec", "n_gpus": n_gpus, "steps": 20,
        "warmup": 5, "ms_per_step": 0.25471234567, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "ranks_seen": n_gpus,
        "config": {"workload": bench.WORKLOADS["local"]["name"], "clouds_per_gpu": 8, "clouds_total": 8 * n_gpus,
                   "points": 8192, "knn": 8, "parallelism": "clouds sharded over %d GPU(s), no data-path collective" % n_gpus,
                   "weights": "random-init (no checkpoint blobs exist upstream)",
                   "execution": "x" * 400 * bloat, "steps_in_flight": 4},
        "one_step_at_a_time": {"value": 15900.5, "unit": "point-clouds/sec", "ms_per_step": 0.502, "note": "n" * 300,
                               "repeats": {"ms_per_step": [0.5] * 7 * bloat}},
        "roofline": {"bound": "hbm", "kernel": "flex_conv_x6_kernel<64,64> B=8 N=8192 K=8", "achieved": 1352.1234, "peak": 8000.0,
                     "unit": "GB/s", "frac": 0.16901, "traffic": 37390000.0, "traffic_source": "live: " + "s" * 500 * bloat,
                     "launch_ms": 0.02699, "algorithmic_bytes": 36503552.0, "launch_ms_f32_mfma_kernel": 0.055,
                     "gather_effective": {"achieved": 5900.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.74, "bytes": 1.6e8},
                     "f32_equivalent_flops": {"achieved": 89.6, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.57, "flops": 2.4e9},
                     "binding_roof": "SIMD issue + FP32-VALU/MFMA exclusion (DESIGN.md 3.5)",
                     "in_step": {"launch_ms": 0.031, "frac": 0.147, "note": "z" * 200}},
        "cpu_baseline": {"value": 0.31, "unit": "point-clouds/sec", "cores": 1, "kind": "port", "sample": "4 clouds " + "q" * 300,
                         "cfg1_ms": 3.2, "cfg1_sample": "c" * 200, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                         "host_cpus": 256, "usable_cpus": 64, "cgroup_cpu_quota": None,
                         "all_cores": {"value": 14.2, "unit": "point-clouds/sec", "cores": 64, "workers": 64,
                                       "sample": "w" * 300}},
        # the evidence tables that used to ride on the line
        "roofline_step": {"kernels": [{"kernel": "k%d" % i, "what": "w" * 80, "bytes": 1.0, "flops": 2.0} for i in range(30 * bloat)]},
        "roofline_global": {"fused_tail": {"x": "y" * 500}}, "kernels_ms": {"k%d" % i: 0.1 for i in range(20)},
        "other_workloads": [{"workload": "w" * 100, "roofline_step": {"kernels": [{}] * 30}} for _ in range(4 * bloat)],
        "batch_sweep_1gpu": {"points": [{"clouds_per_gpu": b} for b in (8, 4, 2, 1)]},
        "data_sensitivity": {"note": "d" * 400}, "repeats": {"ms_per_step": [0.25] * 7},
    }
    if n_gpus == 1:   # the serving loop on the same pipeline
        line["value_streaming"] = {"value": 32000.5, "ms_per_step": 0.25, "steps_in_flight": 4, "distinct_host_batches": 9,
                                   "h2d_bytes_per_step": 786432, "output_bytes_per_step": 34340864, "what": "w" * 400,
                                   "d2h_inclusive": {"value": 12000.0, "ms_per_step": 0.66, "GBps_d2h": 51.5, "note": "n" * 200},
                                   "steady_state": {"steps": 200, "value": 34000.0, "ms_per_step": 0.235,
                                                    "resident_batches_ms_per_step": 0.226, "ratio_to_resident": 0.962}}
    if n_gpus == 1:   # the global-descriptor forward, measured in a fresh process (the other half of BASELINE's metric)
        line["global"] = {"workload": bench.WORKLOADS["global"]["name"], "value": 58000.25, "unit": "point-clouds/sec",
                          "ms_per_step": 0.5517, "steps": 20, "warmup": 5, "clouds_per_gpu": 32, "points": 4096,
                          "steps_in_flight": 3, "measured_in": "m" * 200, "wall_s": 9.0,
                          "value_streaming": {"value": 57000.0, "ms_per_step": 0.56, "what": "w" * 300,
                                              "steady_state": {"steps": 200, "ratio_to_resident": 0.97}},
                          "one_step_at_a_time": {"value": 46600.5, "unit": "point-clouds/sec", "ms_per_step": 0.687,
                                                 "note": "n" * 300},
                          "at_20_steps": {"value": 57000.5, "ms_per_step": 0.561}}
        line["at_20_steps"] = {"value": 35000.5, "ms_per_step": 0.2286}   # the K of rounds 1-5 beside today's default
    if n_gpus > 1:
        line["global_scaling"] = {
            "workload": bench.WORKLOADS["global"]["name"], "steps_in_flight": 2,
            "weak": {"clouds_per_gpu": 32, "clouds_total": 32 * n_gpus, "one_step_at_a_time": {"value": 1.0, "ms_per_step": 1.0},
                     "in_flight": {"value": 1.0, "ms_per_step": 1.0}},
            "strong": {"clouds_per_gpu": 4, "clouds_total": 32, "one_step_at_a_time": {"value": 1.0, "ms_per_step": 1.0},
                       "in_flight": {"value": 1.0, "ms_per_step": 1.0}},
            "note": "n" * 300}
    return line


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_line_fits_and_round_trips(n_gpus):
    full = _canned(n_gpus)
    assert len(json.dumps(full)) > bench.LINE_LIMIT   # the record the old bench would have printed
    text = bench.compact_line(full, extras_path="gpurun_out/bench_extras.json")
    assert "\n" not in text and len(text.encode()) <= bench.LINE_LIMIT <= 8192
    rec = json.loads(text)
    # the driver's contract keys
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["n_gpus"] == n_gpus and rec["config"]["workload"] == bench.WORKLOADS["local"]["name"]
    assert rec["config"]["steps_in_flight"] == 4
    assert set(rec["one_step_at_a_time"]) == {"value", "ms_per_step"}
    r = rec["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "launch_ms"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["gather_effective"] == {"frac": 0.74} and r["f32_equivalent_flops"] == {"frac": 0.57}
    c = rec["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["all_cores"] == {"value": 14.2, "cores": 64}
    assert rec["extras"] == "gpurun_out/bench_extras.json"
    for gone in ("roofline_step", "other_workloads", "batch_sweep_1gpu", "data_sensitivity", "repeats", "kernels_ms"):
        assert gone not in rec
    if n_gpus == 1:
        g = rec["global"]
        assert g["workload"] == bench.WORKLOADS["global"]["name"] and g["steps_in_flight"] == 3
        assert g["value"] == pytest.approx(58000.25) and g["ms_per_step"] == pytest.approx(0.5517)
        assert g["one_step_at_a_time"] == {"value": 46600.5, "ms_per_step": 0.687}
        assert g["at_20_steps"] == {"value": 57000.5, "ms_per_step": 0.561}
        assert rec["at_20_steps"] == {"value": 35000.5, "ms_per_step": 0.2286}
        full2 = _canned(1)
        full2["at_100_steps"] = full2.pop("at_20_steps")   # the driver's K = 20 run carries the default K's figure instead
        rec2 = json.loads(bench.compact_line(full2, extras_path="x.json"))
        assert rec2["at_100_steps"] == {"value": 35000.5, "ms_per_step": 0.2286} and "at_20_steps" not in rec2
        assert g["value_streaming"] == {"value": 57000.0, "ms_per_step": 0.56, "steady_state_ratio": 0.97}
        vs = rec["value_streaming"]
        assert vs["value"] == 32000.5 and vs["steps_in_flight"] == 4 and len(vs["what"]) <= 160
        assert vs["d2h_inclusive"] == {"value": 12000.0, "GBps_d2h": 51.5}
        assert vs["steady_state"] == {"steps": 200, "value": 34000.0, "ratio_to_resident": 0.962}
    if n_gpus > 1:
        g = rec["global_scaling"]
        assert g["weak"]["clouds_per_gpu"] == 32 and g["strong"]["clouds_per_gpu"] == 4
        assert "in_flight" in g["strong"] and "one_step_at_a_time" in g["strong"]


def test_line_sheds_optional_blocks_before_breaking_the_limit():
    text = bench.compact_line(_canned(8, bloat=40), extras_path="x.json", limit=2048)
    assert len(text.encode()) <= 2048
    rec = json.loads(text)
    assert rec["value"] == pytest.approx(31416.1, rel=1e-5) and "roofline" in rec and "cpu_baseline" in rec


def test_global_block_is_shed_last():
    full = _canned(1, bloat=40)
    text = bench.compact_line(full, extras_path="x.json", limit=2300)
    rec = json.loads(text)
    assert len(text.encode()) <= 2300
    assert rec["global"]["value"] == pytest.approx(58000.25) and "one_step_at_a_time" in rec["global"]


def test_side_file_holds_the_full_record(tmp_path):
    full = _canned(1)
    p = bench.write_side_file(full, str(tmp_path / "sub" / "extras.json"))
    assert p is not None
    back = json.load(open(tmp_path / "sub" / "extras.json"))
    assert back["roofline_step"]["kernels"][3]["kernel"] == "k3" and back["value"] == full["value"]
    assert bench.write_side_file(full, "/proc/nope/x.json") is None   # never fatal
