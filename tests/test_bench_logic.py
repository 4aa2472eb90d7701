"""bench.py's bookkeeping that needs no GPU: which committed rocprofv3 profile a run may quote (the kernel variant is part of
what was profiled), that every profile it can quote is actually committed, and the SURVEY 8(d) byte / flop accounting."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_profiles_exist_and_are_complete():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert {"big-scene/flat/gpus1", "big-scene/hier/gpus1", "big-scene/kd/gpus1", "big-soup@1920x1080x64/flat/gpus1", "mirror@1920x1080x64/flat/gpus1",
            "aquarium/flat/gpus1", "big-scene/flat/gpus1/waves3"} <= set(t)
    for key, e in t.items():
        for f in ("hbm_bytes_per_launch", "fetch_size_kb", "write_size_kb", "lanes_active", "valu_busy", "kernel", "kernel_trace_avg_ms", "source"):
            assert f in e, (key, f)
        assert e["hbm_bytes_per_launch"] == pytest.approx((2.0 * e["fetch_size_kb"] + e["write_size_kb"]) * 1024.0)  # gfx950: FETCH_SIZE counts half of a wide read
        path = e["source"].split(" ")[0]
        assert os.path.exists(os.path.join(ROOT, path)), path
        assert os.path.exists(os.path.join(ROOT, path.replace("_pmc.json", "_kernel_stats.csv")))
        assert 0 < e["lanes_active"] <= 64 and 0 < e["valu_busy"] <= 1
        assert "pt_render_kernel<" in e["kernel"] and ", false," in e["kernel"], "the timed kernel, not the counting launch"


def test_profile_lookup_follows_the_kernel_variant(bench, monkeypatch):
    monkeypatch.delenv("PORTRAYER_WAVES", raising=False)
    default = bench.measured_profile("big-scene", "flat", 1)
    assert ", 2>" in default["kernel"], "the scene's default kernel is the 128-register one"
    monkeypatch.setenv("PORTRAYER_WAVES", "3")
    three = bench.measured_profile("big-scene", "flat", 1)
    assert ", 0>" in three["kernel"] and three["hbm_bytes_per_launch"] < default["hbm_bytes_per_launch"] / 10
    assert ", 0>" in bench.measured_profile("big-scene", "hier", 1)["kernel"]
    monkeypatch.setenv("PORTRAYER_WAVES", "4")
    assert bench.measured_profile("big-scene", "flat", 1) == default
    assert bench.measured_profile("big-scene", "hier", 1) is None, "no profile of the hierarchical scene on the 4-wave kernel is committed"
    monkeypatch.delenv("PORTRAYER_WAVES")
    assert bench.measured_profile("big-scene", "flat", 8) is None and bench.measured_profile("no-such-workload", "flat", 1) is None


def test_roofline_block_is_null_without_a_profile(bench, monkeypatch):
    monkeypatch.delenv("PORTRAYER_WAVES", raising=False)
    counts = {"primary": 100, "shadow": 200, "reflect": 0, "refract": 0, "hits": 70, "n_inner": 4000, "n_leaf": 500, "n_analytic": 490, "n_tri": 0, "n_bbox": 0}
    r = bench.roofline_block("no-such-workload", "flat", 1, True, 1.0e9, 0.01, 6000.0, counts, counts, 300, 3)
    assert r["achieved"] is None and r["frac"] is None and r["traffic"] is None and "no PMC profile" in r["basis"]
    assert r["peak"] == 6000.0 and r["spec_peak"] == 8000.0 and r["algorithmic"]["GBps"] == pytest.approx(100.0)
    r = bench.roofline_block("big-scene", "flat", 1, True, 1.0e9, 0.0175, 6000.0, counts, counts, 300, 3)
    assert r["traffic"] > 0 and r["frac"] == pytest.approx(r["traffic"] / 0.0175 / 1e9 / 6000.0)


def test_survey_accounting(bench):
    st = {"primary": 10, "shadow": 20, "reflect": 2, "refract": 1, "hits": 9, "n_inner": 400, "n_analytic": 50, "n_tri": 30, "n_bbox": 5}
    rays = 33
    assert bench.algorithmic_bytes(st, 3, 100, "flat") == 56 * rays + 56 * 400 + 104 * 50 + 72 * 30 + 48 * 5 + 9 * (168 + 80 + 360) + 2700
    assert bench.algorithmic_bytes(st, 3, 100, "kd") == bench.algorithmic_bytes(st, 3, 100, "flat") - 40 * 400
    assert bench.algorithmic_flops(st, 3, "flat") == 48 * 400 + 90 * 50 + 50 * 30 + 126 * 5 + 9 * (60 + 129)
