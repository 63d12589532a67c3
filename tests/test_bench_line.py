"""The last stdout line of bench.py is the driver's only view of a round: round 5's had grown to 20 KB and came back
unparsed (`BENCH_r05.json.parsed = null`).  tools/bench_line.py::compact builds the line from the full record; these
tests hold it under 6 KB with every contract key present, for the N = 1 record (the real round-5 record kept under
profiles/ is the canned input) and for an N = 8 record with the collective legs' objects added."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_line  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_BASE = ("value", "unit", "cores", "kind", "sample")


def canned():
    full = json.load(open(os.path.join(ROOT, "profiles", "r05z_bench_line.json")))
    assert len(json.dumps(full)) > 15000          # the record that broke the driver's parser
    return full


def check(line, n_gpus):
    s = json.dumps(line)
    assert len(s) < bench_line.LIMIT == 6144, len(s)
    for k in CONTRACT:
        assert k in line, k
    for k in ROOFLINE:
        assert k in line["roofline"], k
    for k in CPU_BASE:
        assert k in line["cpu_baseline"], k
    assert line["n_gpus"] == n_gpus and isinstance(line["config"]["workload"], str)
    assert line["roofline"]["bound"] in ("hbm", "mfma") and 0 < line["roofline"]["frac"] <= 1.0
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    # no prose: every string of the line is short
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(t) for t in strings(line)) <= 160
    assert json.loads(s) == line                   # plain JSON: no NaN / Infinity tokens
    assert "NaN" not in s and "Infinity" not in s


def test_compact_line_of_the_round5_record_fits_and_keeps_the_contract():
    full = canned()
    line = bench_line.compact(full, "gpurun_out/bench_full_n1.json")
    check(line, 1)
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["parity"]["max_abs_diff"] == full["parity"]["max_abs_diff"]
    assert line["parity"]["o1_max_abs_diff"] == full["parity"]["o1_table"]["max_abs_diff"]
    # one number per secondary leg
    assert line["train_step_ms"] == full["train_step"]["overlapped"]["ms_per_step"]
    assert line["train_step_ms_eager_median"] == full["train_step"]["ms_per_step_median"]
    assert line["trained_nerf_step_ms"] == full["trained_scene"]["train_step"]["nerf_stage"]["ms_per_step_median"]
    assert line["bound4_field_frac"] == full["render_bound4"]["field_frac_of_hbm_peak"]
    assert line["roi_fwd_ms"] == full["extract_roialign"]["roi_align_forward_ms"]
    assert line["roi_bwd_ms"] == full["extract_roialign"]["roi_align_backward_ms"]
    assert line["extract_ms"] == full["extract_roialign"]["extract_ms"]
    assert line["full"] == "gpurun_out/bench_full_n1.json"


def n8_record():
    full = canned()
    full["n_gpus"] = 8
    full["collective"] = {"backend": "nccl (RCCL)", "rccl_version": "2.26.6", "ranks": 8,
                          "devices": [f"0000:{i:02x}:00 AMD Instinct MI355X" for i in range(8)], "distinct_devices": 8,
                          "all_ranks_on_distinct_gpus": True,
                          "allreduce_table_gradient": {"bytes": 48958912, "iterations": 10, "ms": 0.31, "bus_gb_per_s": 276.4,
                                                       "what": "x" * 300}}
    full["render_sharded"] = {"workload": "y" * 400, "scaling": "strong", "n_gpus": 8, "frames": 8, "rays_of_rank_0": 80000,
                              "ms_per_frame": 1.2, "value": 26000.0, "unit": "Msamples/s",
                              "with_gather": {"ms_per_frame": 1.5, "value": 21000.0, "what": "z" * 200}}
    full["train_step_other_schedule"] = {"ms_per_step": 1.3, "gradient_schedule": "reduce_scatter, payload fp32, " + "w" * 100}
    full["extract_roialign"] = {"workload": "v" * 300, "n_gpus": 8, "extract_mvoxels_per_s": 23000.0, "extract_ms_max_over_ranks": 1.4,
                                "roi_align_forward_ms_max_over_ranks": 0.15, "roi_align_backward_ms_max_over_ranks": 0.34,
                                "per_rank": [{"extract_ms": 1.4, "extract_mvoxels_per_s": 2900.0, "roi_align_forward_ms": 0.15,
                                              "roi_align_backward_ms": 0.34}] * 8}
    del full["cpu_baseline"], full["parity"], full["trained_scene"]      # rank 0 runs those at N = 1 only
    return full


def test_compact_line_of_an_eight_gpu_record_fits():
    full = n8_record()
    line = bench_line.compact(full, "gpurun_out/bench_full_n8.json")
    s = json.dumps(line)
    assert len(s) < 6144
    for k in CONTRACT[:-1]:
        assert k in line, k
    assert line["n_gpus"] == 8 and line["distinct_devices"] == 8 and line["allreduce_bus_gb_per_s"] == 276.4
    assert line["render_sharded_value"] == 26000.0 and line["roi_bwd_ms"] == 0.34
    assert line["train_step_ms_other_schedule"] == 1.3


def test_the_cap_holds_for_a_record_with_absurdly_long_strings_and_failed_legs():
    full = canned()
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    for k in ("render_instance", "render_fast", "train_step_bound4"):
        full[k] = {"error": "RuntimeError: " + "e" * 290}
    line = bench_line.compact(full, "gpurun_out/bench_full_n1.json")
    check(line, 1)
    assert set(line["failed_legs"]) == {"render_instance", "render_fast", "train_step_bound4"}


def test_emit_prints_the_full_record_then_the_compact_line_last(tmp_path):
    full = canned()
    out = io.StringIO()
    line = bench_line.emit(full, 1, str(tmp_path), file=out)
    lines = out.getvalue().splitlines()
    assert len(lines) == 2
    assert json.loads(lines[1]) == line and len(lines[1]) < 6144
    first = json.loads(lines[0])
    assert first["full_record"] is True and first["value"] == line["value"]
    assert json.load(open(tmp_path / "gpurun_out" / "bench_full_n1.json"))["value"] == line["value"]
    assert line["full"] == os.path.join("gpurun_out", "bench_full_n1.json")


def test_bench_py_ends_every_exit_path_through_emit():
    """bench.py prints JSON in exactly three kinds of places: argument / environment errors before anything ran (tiny
    objects), and `emit(...)` - the final line, the watchdog's bail-out and the shared-GPU refusal."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("emit(line, world, ROOT)") >= 3
    assert "print(json.dumps(line)" not in src
