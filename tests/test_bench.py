"""bench.py end to end on a GPU: the N = 1 line and the N = 2 line carry every object the contract asks for.  On a one-GPU box
the two ranks of `--gpus 2` are told to share device 0 (MGC_BENCH_ONE_DEVICE=1): RCCL refuses that, rank 0 falls back to the
in-process peer-copy form and says so in the line; on a multi-GPU box the same command exercises real RCCL ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                   # ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_single_gpu_line_is_complete(native_lib):
    d = _run(["--reads", "2000000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "200000"])
    assert d["n_gpus"] == 1 and d["unit"] == "distinct k-mers/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert d["check"]["ok"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    # the dominant family is COMPUTED: the one with the largest wall share; every family's wall share stays inside its stage's
    fams = dict(rf["kernels"]); fams[rf["dominant"]] = rf
    assert {"first_pass", "count", "partition", "histogram"} <= set(fams)
    assert all(fams[rf["dominant"]]["wall_ms_per_step"] >= f["wall_ms_per_step"] for f in fams.values())
    st = d["stage_ms_per_step"]
    stage_of = {"first_pass": "sort", "second_pass": "sort", "count": "rle", "partition": "partition", "histogram": "histogram"}
    for name, f in fams.items():
        assert 0 < f["frac"] < 1 and 0 < f["share_of_step"] < 1, (name, f)
        assert f["wall_ms_per_step"] <= st[stage_of[name]] * 1.02 + 1e-6, (name, f["wall_ms_per_step"], st)
    assert fams["first_pass"]["wall_ms_per_step"] + fams.get("second_pass", {"wall_ms_per_step": 0})["wall_ms_per_step"] <= st["sort"] * 1.02
    assert 0 < rf["sort_pass"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert cb["whole_workload"] is True and "restatement" in cb["kind_note"] and "WHOLE workload" in cb["sample"]   # every byte the GPU counted
    assert d["db_write"]["data_bytes"] > 0 and d["e2e"]["wall_s"] > 0 and d["value_e2e"] > 0
    assert abs(sum(d["stage_ms_per_step"].values()) - d["ms_per_step"]) < 0.5 * d["ms_per_step"]
    # the wall-clock guard of the untimed legs: every leg timed, nothing skipped inside the default budget
    assert d["legs_budget_s"] == 600 and d["legs_skipped"] == {}
    assert {"check", "db_write", "e2e", "cpu_baseline", "total"} <= set(d["legs_s"]) and d["legs_s"]["total"] < 600


def test_bench_budget_guard_skips_legs_and_says_so(native_lib):
    # a budget the untimed tail cannot meet: the timed steps and the roofline still come, the legs are named as skipped
    d = _run(["--reads", "2000000", "--steps", "2", "--warmup", "1"], {"MGC_BENCH_BUDGET_S": "1"})
    assert d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    for leg in ("check", "db_write", "e2e", "cpu_baseline.whole_workload", "cpu_baseline.sample"):
        assert leg in d["legs_skipped"] and "budget" in d["legs_skipped"][leg]
    assert "check" not in d and "e2e" not in d and "cpu_baseline" not in d


def test_bench_two_ranks_line_is_complete(native_lib):
    import torch
    one = torch.cuda.device_count() < 2
    d = _run(["--gpus", "2", "--reads", "1500000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "200000"],
             {"MGC_BENCH_ONE_DEVICE": "1"} if one else None)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["check"]["ok"] is True
    assert "cpu_baseline" not in d and d["legs_skipped"]["cpu_baseline"] == "reported at N = 1 only"
    assert d["db_write"]["files"] == 129 and d["legs_s"]["total"] > 0
    if one:
        assert "RCCL refused" in d["config"]["transport"] and d["db_write"]["identical_to_single_session"] is True
    else:
        assert "RCCL" in d["config"]["transport"] and d["db_write"]["identical_to_node_count"] is True
        assert d["check"]["keys_ascending_across_rank_boundaries"] is True
        assert 0 < d["roofline"]["frac"] < 1 and "stage_ms_one_profiled_step" in d
    assert 0 < d["roofline"]["frac"] < 1 and d["e2e"]["wall_s"] > 0       # (the one-process fallback measures it in a single session)
