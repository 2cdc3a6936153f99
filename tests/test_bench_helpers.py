"""CPU tests of bench.py's roofline bookkeeping (no GPU work: the helpers are pure arithmetic over measured numbers)."""
import importlib.util
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_aggregate_roofline_fields_follow_survey_8d_and_no_frac_exceeds_one():
    D, nseg, f = 256, 1000, 10
    ids = torch.randint(0, 50, (nseg * f,))  # heavy reuse: 50 distinct rows
    r = bench.roofline_aggregate("MaxAggregator", D, nseg, nseg * f, avg_ms=1e-3, launches=20, ids_last=ids, workload="none",
                                 B0=1, offline_ok=False)
    assert r["algorithmic_bytes_per_launch"] == nseg * f * (4 * D + 12) + nseg * (4 * D + 4)
    assert r["distinct_rows_last_launch"] == int(torch.unique(ids).numel())
    assert r["compulsory_bytes_per_launch"] == r["distinct_rows_last_launch"] * 4 * D + nseg * f * 12 + nseg * (4 * D + 4)
    assert r["achieved"] == r["algorithmic_bytes_per_launch"] / 1e-6 / 1e9
    assert r["algorithmic_over_peak"] > 1 and r["cache_assisted"] is True  # cache-assisted rates are not called frac
    for k, v in r.items():
        if k.startswith("frac") and isinstance(v, float):
            assert 0 <= v <= 1, (k, v)


def test_sampler_roofline_bytes():
    r = bench.roofline_sampler("EdgeWeightSampler", 10, rows=1000, slots=10000, avg_ms=0.01, launches=3, workload="none", B0=1)
    assert r["algorithmic_bytes_per_launch"] == 10000 * 40 + 1000 * 24  # 32 B/slot + 8 B alias + 24 B/row
    r = bench.roofline_sampler("TopkSampler", 10, rows=1000, slots=10000, avg_ms=0.01, launches=3, workload="none", B0=1)
    assert r["algorithmic_bytes_per_launch"] == 10000 * 32 + 1000 * 24
    assert r["draws_per_s"] == 10000 / 1e-5


def test_committed_offline_traffic_is_labelled_and_bounded():
    ids = torch.arange(16_384_000 // 64)
    r = bench.roofline_aggregate("MaxAggregator", 256, 1_638_400, 16_384_000, avg_ms=2.15, launches=20, ids_last=ids,
                                 workload="c3", B0=65536)
    assert r["traffic"] and "OFFLINE" in r["traffic_source"] and 0 < r["frac_traffic_offline"] <= 1
