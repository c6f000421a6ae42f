"""bench.py's host-side arithmetic, without a GPU: the roofline object is built from the library's profile the way DESIGN.md 7
says, and the calibrated PMC traffic of the judged workload is found among the committed profiles (so that the driver's line
carries `roofline.traffic`, not null)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_traffic_of_the_judged_workload_is_committed():
    b = _bench()
    for prefix in ("hash_count_multi_kernel", "radix_group_kernel<unsigned long long", "radix_group_kernel<unsigned int"):
        t, src = b.pmc_traffic(66666667, prefix)
        assert t and t > 1e8 and src.startswith("profiles/r") and "_pmc_traffic.json" in src, (prefix, t, src)
    assert b.pmc_traffic(12345, "hash_count_multi_kernel") == (None, None)      # counters of another workload are never borrowed


def test_pmc_traffic_is_the_launch_weighted_mean_over_instantiations(tmp_path, monkeypatch):
    """`roofline.traffic` must describe the launches `algorithmic_bytes_per_launch` describes -- every
    instantiation of the kernel weighted by its launches, plus the bytes of the retry / streaming kernels of the same file launch
    (numerator only); a synthetic counter file makes the arithmetic checkable."""
    import json
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_pmc_traffic.json").write_text(json.dumps({"reads_per_gpu": 777, "all_kernels": {
        "hash_count_multi_kernel<256, 1536, 2048, 1, false>": {"launches": 45, "fetch_bytes_per_launch": 600.0, "write_bytes_per_launch": 200.0},
        "hash_count_multi_kernel<256, 1536, 2048, 2, false>": {"launches": 15, "fetch_bytes_per_launch": 300.0, "write_bytes_per_launch": 100.0},
        "hash_count_huge_kernel<unsigned int>": {"launches": 30, "fetch_bytes_per_launch": 10.0, "write_bytes_per_launch": 2.0},
        "radix_group_kernel<unsigned int, 9>": {"launches": 64, "fetch_bytes_per_launch": 5.0, "write_bytes_per_launch": 5.0},
        "never_launched<1>": {"launches": 0, "fetch_bytes_per_launch": None, "write_bytes_per_launch": None}}}))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    t, src = b.pmc_traffic(777, "hash_count_multi_kernel", also=("hash_count_kernel", "hash_count_huge_kernel"))
    assert abs(t - (45 * 800.0 + 15 * 400.0 + 30 * 12.0) / 60) < 1e-9 and "2 instantiations" in src
    t1, _ = b.pmc_traffic(777, "hash_count_multi_kernel")
    assert abs(t1 - (45 * 800.0 + 15 * 400.0) / 60) < 1e-9
    assert b.pmc_traffic(777, "no_such_kernel") == (None, None)


def test_roofline_object_arithmetic():
    """the dominant kernel family is COMPUTED from wall-clock shares -- passes by the sum of their launches (one
    stream, back to back), the count kernels by their stage's wall clock minus the packing tail (their launches overlap on two
    streams: the sum of the launch durations exceeds the stage) -- and no family's share exceeds its stage's."""
    b = _bench()
    steps, ms_per_step = 3, 120.0
    keys = 135_000_000
    acc = {"by_pass": [{"ms": 0.54 * 192, "launches": 192, "keys": keys * 192, "bytes": 9 * keys * 192},
                       {"ms": 0.33 * 192, "launches": 192, "keys": keys * 192, "bytes": 8 * keys * 192}],
           "pass_ms": 0.87 * 192, "pass_launches": 384, "pass_keys": 2 * keys * 192, "pass_bytes": 17 * keys * 192,
           "finish": {"ms": 0.63 * 192, "launches": 192, "keys": keys * 192, "bytes": 688_000_000 * 192},
           "stage_ms": [18.0, 63.0, 170.0, 86.0, 0.3], "pack_ms": 17.0,
           "partition_bytes": 3 * (10_066_666_717 + 5 * 8_648_000_000), "hist_bytes": 3 * 10_066_666_717}
    r = b.roofline_object(acc, ms_per_step, steps, 66666667, True, "over the timed steps")
    assert r["bound"] == "hbm" and r["peak"] == 8000.0
    # 64 first passes of 0.54 ms = 34.6 ms per step: more than the count stage's 86 / 3 - 17 / 3 = 23 ms, the second pass's 21.1, the partition's 21
    assert r["dominant"] == "first_pass" and "5 B k-mers" in r["kernel"] and set(r["kernels"]) == {"second_pass", "count", "partition", "histogram"}
    assert abs(r["achieved"] - 9 * keys / 0.54e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["wall_ms_per_step"] - 0.54 * 64) < 1e-9 and abs(r["share_of_step"] - 0.54 * 64 / 120.0) < 1e-9
    assert r["traffic"] and r["traffic_source"].startswith("profiles/")
    cnt = r["kernels"]["count"]
    assert abs(cnt["achieved"] - 688_000_000 / 0.63e-3 / 1e9) < 1e-6 and abs(cnt["avg_launch_ms"] - 0.63) < 1e-9
    assert abs(cnt["wall_ms_per_step"] - (86.0 - 17.0) / 3) < 1e-9                    # the stage's wall clock minus the packing tail ...
    assert cnt["launch_ms_sum_per_step"] > cnt["wall_ms_per_step"]                     # ... not the sum of the overlapped launches
    assert abs(cnt["frac_on_wall"] - 688_000_000 * 64 / ((86.0 - 17.0) / 3 * 1e-3) / 1e9 / 8000.0) < 1e-9
    stage_of = {"first_pass": 2, "second_pass": 2, "count": 3, "partition": 1, "histogram": 0}
    fams = dict(r["kernels"], first_pass=r)
    for name, f in fams.items():
        assert f["wall_ms_per_step"] <= acc["stage_ms"][stage_of[name]] / steps + 1e-9, name
    assert abs(fams["partition"]["achieved"] - (10_066_666_717 + 5 * 8_648_000_000) / 21e-3 / 1e9) < 1e-3
    sp = r["sort_pass"]
    assert "5 B k-mers" in sp["kernel"] and abs(sp["achieved"] - 9 * keys / 0.54e-3 / 1e9) < 1e-6
    assert abs(sp["second_pass"]["achieved"] - 8 * keys / 0.33e-3 / 1e9) < 1e-6
    assert sp["survey_accounting"]["bytes_per_key_per_pass"] == 16
    # a count stage that dominates (round 4's shape) is found too
    acc_c = dict(acc, stage_ms=[18.0, 63.0, 170.0, 140.0, 0.3])
    rc = b.roofline_object(acc_c, ms_per_step, steps, 66666667, True, "over the timed steps")
    assert rc["dominant"] == "count" and "hash_count" in rc["kernel"] and "first_pass" in rc["kernels"]
    # N > 1 (not the single-session form): no traffic is borrowed, no share of a step is claimed
    r2 = b.roofline_object(acc, ms_per_step, steps, 66666667, False, "over the timed steps")
    assert r2["traffic"] is None and r2["share_of_step"] is None and r2["sort_pass"]["traffic"] is None
    # only the totals of the passes were collected (the sharded forms): one pass entry, priced on the reported bytes
    acc3 = dict(acc, by_pass=[{"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0}] * 2, finish=None)
    r3 = b.roofline_object(acc3, ms_per_step, steps, 66666667, False, "over the timed steps")
    assert r3["launches"] == 384 and 0 < r3["frac"] < 1 and "measured" in r3
