"""CPU tier: the ONE stdout line of bench.py stays parseable by a reader with a small buffer.

Round 4's line grew to 21 KB (the side measurements rode in it) and the driver's 8 KB tail held only its end: the round went
unmeasured. The headline is now a pure function of the full record (bench.format_headline) — this test feeds it canned
records, including one far larger than round 4's, and asserts: < 4096 bytes, json round trip, every key of the contract,
`roofline` and `cpu_baseline` adjacent, and that the side-measurement line is NOT a JSON line."""
import json
import os.path as osp
import sys

import pytest

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def canned(world=1, bloat=1):
    d = {
        "metric": "coarse-retrieval queries/sec over 11k-cell DB, embed_dim=256; top-1/3/5 recall parity",
        "value": 90123456.789012345, "unit": "queries/s", "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 0.04545454545,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "KITTI360Pose-sized val DB: N=11259 cells x D=256 resident in HBM, Q=4096 precomputed text "
                               "embeddings per step, top-10 (float64-exact ids)",
                   "n_cells": 11259, "queries_per_step": 4096, "embed_dim": 256, "top_k": 10,
                   "arithmetic": "x" * 300 * bloat, "parallelism": "single-gpu" if world == 1 else f"db-row-shard x{world}",
                   "layout": "one resident DB, two launches per step (scan, re-rank)", "pipelining": "none", "region": "y" * 200},
        "roofline": bench.roofline("scanp_kernel<6, 4, true, 1>", 2500.0, 1, 2.0 * 4096 * 11259 * 256, 0.03073, 19,
                                   0.0295, 84, 0.021, 84),
        "kernels_ms": {"search_scan": 0.030731234567, "search_rerank": 0.012345678},
        "scan_kernel_event_samples": {"note": "n" * 400 * bloat},
        "steady_state": {"steps": 400, "untimed_ramp_steps": 1500, "ms_per_step": 0.0434512345, "queries_per_s": 94267890.12345},
        "pipelined": {"note": "p" * 500 * bloat},
        "secondary": {f"group{i}": {"kernel_ms": 1.2345678 * i, "frac": 0.1234567, "ok": True, "list": [0.123456789] * 64 * bloat,
                                    "note": "z" * 300, "inner": {"step_ms": 2.5, "forward_ms": 1.0, "text": "t" * 200}}
                      for i in range(16 * bloat)},
        "parity": {"ids_equal_float64_oracle": True, "pairs_checked": 163840, "checked": "c" * 200, "max_abs_score_err": 2.2e-16,
                   "recall_at_1_planted": [0.9] * 4, "counters_last_step": list(range(32))},
        "ranks_seen": world, "query_batches_rotated": 4,
        "cpu_baseline": {"value": 265.6612345, "unit": "queries/s", "cores": 8, "kind": "port",
                         "sample": "3072 queries x N=11259 (float64 C@t + full argsort per query, numpy) in 12.1s",
                         "fair_cpu_torch_mm_topk_f32": {"queries_per_s": 1.0e5, "threads": 8}},
        "speedup_vs_cpu_baseline": 339241.1,
        "detail_file": "gpurun_out/bench_detail.json",
    }
    if world > 1:
        d.pop("cpu_baseline")
        d.pop("speedup_vs_cpu_baseline")
        d["alt_query_sharded"] = {"layout": "l" * 200, "queries_per_s": 3.0e8, "ms_per_step": 0.0136, "ids_equal_row_sharded": True}
        d["weak_scaling_point"] = {"layout": "w" * 300, "rows_total": 90072, "ms_per_step": 0.06, "queries_per_s": 6.8e7,
                                   "own_rows_in_merged_topk_are_in_local_topk": True}
        d["config5_coarse_plus_fine"] = {"error": "RuntimeError(" + "e" * 2000 + ")"}
    return d


@pytest.mark.parametrize("world,bloat", [(1, 1), (1, 8), (8, 1), (8, 8)])
def test_headline_is_short_and_round_trips(world, bloat):
    d = canned(world, bloat)
    assert len(json.dumps(d)) > 20000  # the full record is round-4 sized or bigger
    line = bench.format_headline(d)
    assert "\n" not in line and len(line.encode()) < 4096
    o = json.loads(line)
    for k in CONTRACT_KEYS:
        assert k in o, k
    assert o["value"] == pytest.approx(d["value"], rel=1e-5) and o["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-5)
    assert o["config"]["workload"].startswith("KITTI360Pose-sized") and "model" not in o["config"]
    rl = o["roofline"]
    assert rl["bound"] == "mfma" and rl["unit"] == "TFLOP/s" and rl["frac"] == pytest.approx(rl["achieved"] / rl["peak"], rel=1e-4)
    assert rl["achieved"] == pytest.approx(2.0 * 4096 * 11259 * 256 / 0.03073e-3 / 1e12, rel=1e-4) and "traffic" in rl and rl["kernel_ms"] > 0
    assert o["parity"] == {"ids_equal": True, "pairs_checked": 163840, "max_abs_score_err": 2.2e-16}
    assert o["ranks_seen"] == world and o["steady_state"]["queries_per_s"] > 0
    keys = list(o)
    if world == 1:
        cb = o["cpu_baseline"]
        assert set(cb) == {"value", "unit", "cores", "kind", "sample"} and cb["kind"] == "port" and cb["cores"] == 8
        assert keys.index("cpu_baseline") == keys.index("roofline") + 1  # adjacent: a truncated tail loses both or neither
    else:
        assert "cpu_baseline" not in o and o["alt_query_sharded"]["ids_equal_row_sharded"] is True
        assert o["weak_scaling_point"]["rows_total"] == 90072 and len(o["config5_coarse_plus_fine"]["error"]) <= 120


def test_secondary_line_is_short_and_not_a_json_line():
    for bloat in (1, 8):
        s = bench.format_secondary(canned(1, bloat)["secondary"])
        assert s.startswith("SECONDARY ") and "\n" not in s and len(s) <= 3600
        with pytest.raises(ValueError):
            json.loads(s)  # a line-wise JSON parser must skip it ...
        brief = json.loads(s[len("SECONDARY "):])  # ... and a human can still read it
        assert brief and all(isinstance(v, dict) for v in brief.values())
    assert bench.format_secondary({}) == "SECONDARY {}"
