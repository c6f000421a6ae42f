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
    b = _bench()
    steps, ms_per_step = 3, 120.0
    keys = 135_000_000
    acc = {"by_pass": [{"ms": 0.54 * 192, "launches": 192, "keys": keys * 192, "bytes": 9 * keys * 192},
                       {"ms": 0.33 * 192, "launches": 192, "keys": keys * 192, "bytes": 8 * keys * 192}],
           "pass_ms": 0.87 * 192, "pass_launches": 384, "pass_keys": 2 * keys * 192, "pass_bytes": 17 * keys * 192,
           "finish": {"ms": 0.63 * 192, "launches": 192, "keys": keys * 192, "bytes": 688_000_000 * 192},
           "stage_ms": [23.0, 74.0, 170.0, 86.0, 0.3]}
    r = b.roofline_object(acc, ms_per_step, steps, 66666667, True, "over the timed steps")
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and "hash_count_multi_kernel" in r["kernel"]
    assert abs(r["achieved"] - 688_000_000 / 0.63e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["avg_launch_ms"] - 0.63) < 1e-9 and r["traffic"] and r["traffic_source"].startswith("profiles/")
    assert abs(r["kernel_time_share_of_step"] - (0.63 * 64) / 120.0) < 1e-9
    sp = r["sort_pass"]
    assert "5 B k-mers" in sp["kernel"] and abs(sp["achieved"] - 9 * keys / 0.54e-3 / 1e9) < 1e-6
    assert abs(sp["second_pass"]["achieved"] - 8 * keys / 0.33e-3 / 1e9) < 1e-6
    assert sp["survey_accounting"]["bytes_per_key_per_pass"] == 16
    # N > 1 (not the single-session form): no traffic is borrowed, no share of a step is claimed
    r2 = b.roofline_object(acc, ms_per_step, steps, 66666667, False, "over the timed steps")
    assert r2["traffic"] is None and r2["kernel_time_share_of_step"] is None and r2["sort_pass"]["traffic"] is None
    # only the totals of the passes were collected (the sharded forms): one pass entry, priced on the reported bytes
    acc3 = dict(acc, by_pass=[{"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0}] * 2, finish=None)
    r3 = b.roofline_object(acc3, ms_per_step, steps, 66666667, False, "over the timed steps")
    assert r3["launches"] == 384 and 0 < r3["frac"] < 1 and "measured" in r3
