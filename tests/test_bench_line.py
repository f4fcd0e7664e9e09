"""bench.py's stdout contract (round 5): ONE JSON line below 4 kB that the driver's parser can hold -- round 4's line
had grown to 22 kB and BENCH_r04.parsed came back null.  The line is built from the full result by bench.compact_line;
the full result goes to bench_detail.json.  No GPU: a synthetic result shaped like profiles/r4_final_bench.json, with
every free-text field blown up."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def synthetic_result(blow=1):
    leg = {"value": 12345.67, "unit": "stereo-pairs/s", "ms_per_step": 1.234, "workload": "w" * 300 * blow,
           "repeats": {"values": [1.0] * 30 * blow}, "roofline_kernels": [{"kernel": "k", "frac": 0.1}] * 10 * blow}
    rf = {"kernel": "mineig_localmax", "bound": "hbm", "achieved": 295.23, "peak": 8000.0, "unit": "GB/s", "frac": 0.0369,
          "traffic": 20925850, "traffic_source": "committed profiles/pmc_traffic_latest.json " + "x" * 200 * blow,
          "alg_bytes_per_launch": 23101440, "avg_launch_ms": 0.07825, "launches_sampled": 15, "frac_of_copy": 0.0601,
          "valu_issue": {"note": "n" * 500 * blow}}
    res = {"metric": "stereo-pairs/sec front-end (detect+track+match) @752x480", "value": 58000.12, "unit": "stereo-pairs/s",
           "n_gpus": 1, "steps": 40, "warmup": 8, "prewarm_steps_untimed": 0, "ms_per_step": 1.1, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32+f32", "data": "synthetic (" + "d" * 400 * blow + ")",
           "config": {"workload": "BASELINE c3: " + "c" * 400 * blow, "batch_per_gpu": 64, "width": 752, "height": 480,
                      "features": 600, "mode": "kf", "use_ransac": 1, "stream_groups": 1, "device_frames_persist": 0,
                      "parallelism": "streams x1"},
           "value_is": "median of 3 timed regions of exactly 40 steps each", "device_warm_up_ok": True,
           "hbm_copy_GBps": 4912.3, "hbm_copy_probe": {"what": "p" * 300 * blow},
           "roofline": rf, "roofline_kernels": [dict(rf, kernel=k) for k in ("mineig_localmax", "rectify", "pyramid")],
           "roofline_dense_weighted": {"bound": "hbm", "achieved": 1100.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1375,
                                       "alg_bytes": 145800000, "sum_launch_ms": 0.13, "kernels": ["a", "b", "c"]},
           "cpu_baseline": {"value": 42.1, "unit": "stereo-pairs/s", "cores": 1, "kind": "port", "sample": "s" * 500 * blow,
                            "all_cores": {"value": 540.0, "cores": 64, "sample": "t" * 300 * blow}},
           "largest_kernel": {"bound": "b" * 800 * blow}, "end_to_end_traffic": {"note": "e" * 400 * blow},
           "collective_backend": "nccl (RCCL), world 1: barrier + timing all_reduce"}
    for k in bench.LEG_SCALARS:
        res[k] = dict(leg)
    res["input_side"] = {"decode_all_threads": 12000.0, "stereo_pairs_per_s_all_threads": 6000.0, "usable_cpus": 16}
    return res


@pytest.mark.parametrize("blow", [1, 20])
def test_bench_line_is_small_and_round_trips(blow):
    res = synthetic_result(blow)
    assert len(json.dumps(res)) > 15000            # the full result is what round 4 printed
    line = bench.compact_line(res)
    assert "\n" not in line
    assert len(line) < 4096 == bench.LINE_MAX_BYTES
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "prewarm_steps_untimed", "ms_per_step", "dtype", "config",
              "roofline", "roofline_dense_weighted", "cpu_baseline", "higher_is_better", "scaling", "vs_baseline", "data"):
        assert k in d, k
    assert d["value"] == res["value"] and d["ms_per_step"] == res["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == res["roofline"]["frac"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["all_cores_value"] == 540.0
    # round 6: the copy rate of the box beside the fractions, the real-frame leg beside `value`
    assert d["hbm_copy_GBps"] == 4912.3 and d["roofline"]["frac_of_copy"] == 0.0601
    assert d["roofline_kernels_frac_of_copy"] == {k: 0.0601 for k in ("mineig_localmax", "rectify", "pyramid")}
    assert d["kf_realistic_value"] == 12345.67 and d["kf_realistic_ms_per_step"] == 1.234
    assert set(d["legs_pairs_per_s"]) == set(bench.LEG_SCALARS) | {"input_side_host_decode"}
    assert all(isinstance(v, float) for v in d["legs_pairs_per_s"].values())
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_line_without_optional_parts():
    res = {k: v for k, v in synthetic_result().items()
           if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config")}
    d = json.loads(bench.compact_line(res))
    assert d["value"] == res["value"] and "roofline" not in d and "legs_pairs_per_s" not in d


def test_emit_writes_the_detail_file_and_one_stdout_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    res = synthetic_result()
    bench.emit(res)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["detail"].startswith("full result: bench_detail.json")
    with open(tmp_path / "bench_detail.json") as f:
        assert json.load(f)["largest_kernel"] == res["largest_kernel"]
    assert json.loads(cap.err.strip().splitlines()[-1])["value"] == res["value"]
