"""bench.py's stdout contract line (CPU only): built from a canned detailed record it must carry every key the driver parses and stay well
under the size at which the driver's stdout tail cut round 4's line (BENCH_r04.json: parsed null at 29.6 KB)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _canned():
    """Round 4's full builder-side record (every sub-record present: four configurations with block-BFGS variants, five QP-level parity
    records, both parity objects) — the record that was too long to parse."""
    return json.load(open(os.path.join(ROOT, "profiles", "r04h_bench.json")))


def test_contract_line_is_compact_and_complete():
    import bench
    d = _canned()
    assert len(json.dumps(d)) > 16384            # the canned record really is the oversized one
    line = bench.contract_line(d, "gpurun_out/bench_detail.json")
    s = json.dumps(line)
    assert len(s) < 4096, len(s)
    assert "\n" not in s
    back = json.loads(s)
    for k in CONTRACT_KEYS:
        assert k in back, k
    assert back["value"] == float("%.9g" % d["value"]) and back["ms_per_step"] == float("%.9g" % d["ms_per_step"])
    assert set(back["config"]) <= {"workload", "global_batch", "parallelism"} and "model" not in back["config"]
    rf = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    cb = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference")
    assert set(back["configs"]) == {"D", "B", "C", "R"}
    for c in back["configs"].values():
        assert set(c) <= {"ms", "route", "frac", "block_bfgs", "bit_identical", "small_batches_ms"}
    assert back["parity"]["same_order_bit_identical"] is True
    assert set(back["parity"]["qp_level_within_1e-8"]) == {"A", "D", "B", "R", "C"}


def test_contract_line_survives_missing_sub_records():
    """N > 1 ranks and --cpu-sample 0 runs carry no configs / cpu_baseline / parity objects: the line builder must not need them."""
    import bench
    d = _canned()
    for k in ("configs", "cpu_baseline", "qp_replay", "parity_vs_cpu_same_order", "parity_vs_cpu_reference", "variant_block_bfgs"):
        d.pop(k, None)
    line = bench.contract_line(d)
    for k in CONTRACT_KEYS:
        assert k in line
    assert "roofline" in line and "cpu_baseline" not in line and "detail" not in line
    assert len(json.dumps(line)) < 2048


def test_contract_line_stays_small_with_long_notes():
    import bench
    d = _canned()
    d["cpu_baseline"]["sample"] = "x" * 5000
    d["config"]["note"] = "y" * 5000
    assert len(json.dumps(bench.contract_line(d, "p"))) < 4096


def test_round6_record_is_consistent_and_carries_the_new_keys():
    """VERDICT r5 item 6, on this round's detailed record (profiles/r06_bench.json, written by bench.py on the MI355X): the dominant kernel's duration comes from the
    MEDIAN block and cannot exceed the step it is part of; config A has small-batch records with the single-core CPU time of the same instances beside them (the honest
    crossover for BASELINE configs[0], ONE Solver<OCP>::solve()); the GPU / CPU-all-cores ratio exists once, labelled baseline-only."""
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.01
    assert abs(d["roofline"]["kernel_ms"] - d["step_ms"]["mean"]) < 1e-12 and "mean_all_blocks" in d["step_ms"]
    sb = d["small_batches"]
    assert set(sb) == {"1", "64", "512"}
    for rec in sb.values():
        assert rec["route"] == "reg1" and rec["cpu_single_core_ms"] > 0 and rec["ms_per_batch"]["median"] > 0
        assert abs(rec["gpu_over_cpu_single_core"] - rec["cpu_single_core_ms"] / rec["ms_per_batch"]["median"]) < 1e-9
    g = d["gpu_over_cpu_all_cores"]
    assert "baseline only" in g["note"] and abs(g["value"] - d["value"] / d["cpu_baseline"]["value"]) < 1e-6 * g["value"]
    line = bench.contract_line(d, "gpurun_out/bench_detail.json")
    assert len(json.dumps(line)) < 4096
    assert set(line["small_batches_ms"]) == {"1", "64", "512"} and line["gpu_over_cpu_all_cores"] > 1
    assert line["roofline"]["kernel_ms"] <= line["ms_per_step"] * 1.01
    # the problems of the reference's two NP = 1 control tests as batches (late round 6): served by the condensed register kernel, bit-identical to its restatement,
    # the same trajectories as the run in the reference order
    rt = d["reference_tests"]
    assert set(rt) == {"minimal_time_parking_np1", "nonlinear_constraints_parking_np1_ng1", "valet_parking_policy_set_robot_11_nodes"}
    for rec in rt.values():
        assert rec["route"] == rec["lone_instance_route"] == "condreg" and rec["batch"] == 4096 and 0 < rec["lone_instance_ms"] < rec["ms_per_batch"]["median"]
        assert rec["parity"]["bit_identical_x"] and rec["parity"]["bit_identical_lam"] and rec["parity"]["identical_trajectory_fraction_vs_reference_order"] == 1.0
    assert set(line["reference_tests"]) == set(rt) and all(v[2] == "condreg" and v[3] is True for v in line["reference_tests"].values())
