"""bench.py's bookkeeping that needs no GPU: which committed rocprofv3 profile a run may quote (only one of the very kernel
instantiation that is running), that every profile it can quote is actually committed, the SURVEY 8(d) byte / operation
accounting, and the shape of the roofline block."""
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
    for key, e in t.items():
        for f in ("hbm_bytes_per_launch", "fetch_size_kb", "write_size_kb", "lanes_active", "valu_busy", "kernel", "kernel_trace_avg_ms", "source"):
            assert f in e, (key, f)
        assert e["hbm_bytes_per_launch"] == pytest.approx((2.0 * e["fetch_size_kb"] + e["write_size_kb"]) * 1024.0)  # gfx950: FETCH_SIZE counts half of a wide read
        path = e["source"].split(" ")[0]
        assert os.path.exists(os.path.join(ROOT, path)), path
        assert os.path.exists(os.path.join(ROOT, path.replace("_pmc.json", "_kernel_stats.csv")))
        # SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU: thread quad-cycles per vector instruction = lanes active, biased upwards where multi-pass
        # instructions (f64 reciprocals, 32-bit integer multiplies) are frequent: the chain kernel reads 64.7
        assert 0 < e["lanes_active"] <= 68 and 0 < e["valu_busy"] <= 1
        assert "pt_render" in e["kernel"] and ", false," in e["kernel"], "the timed kernel, not the counting launch"


def test_kernel_name_follows_pt_stats(bench):
    assert bench.kernel_name({"kernel_mode": 3, "kernel_variant": 4}) == "void pt_render_simple_kernel<3, false, false, 4, false>(PtRenderArgs)"
    assert bench.kernel_name({"kernel_mode": 1, "kernel_variant": 3 | 128}) == "void pt_render_simple_kernel<1, false, true, 3, false>(PtRenderArgs)"
    assert bench.kernel_name({"kernel_mode": 1, "kernel_variant": 3 | 512}) == "void pt_render_simple_kernel<1, false, false, 3, true>(PtRenderArgs)"
    assert bench.kernel_name({"kernel_mode": 4, "kernel_variant": 3 | 16 | 32 | 128}) == "void pt_render_kernel<4, false, true, 1>(PtRenderArgs)"
    assert bench.kernel_name({"kernel_mode": 4, "kernel_variant": 3 | 16 | 32 | 128 | 256}) == "void pt_render_kernel<4, false, true, 3>(PtRenderArgs)"
    assert bench.kernel_name({"kernel_mode": 3, "kernel_variant": 4 | 16}) == "void pt_render_kernel<3, false, false, 2>(PtRenderArgs)"
    assert bench.kernel_name({"kernel_mode": 7, "kernel_variant": 3 | 16}) == "void pt_render_kernel<7, false, false, 0>(PtRenderArgs)"


def test_a_profile_is_quoted_only_for_the_kernel_that_runs(bench):
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, e in t.items():
        base = key.rsplit("/waves", 1)[0]
        assert bench.measured_profile(base, e["kernel"]) is not None
        assert bench.measured_profile(base, "void some_other_kernel<1>(PtRenderArgs)") is None
    assert bench.measured_profile("no-such-workload/flat/gpus1", "void pt_render_simple_kernel<3, false, false, 4, false>(PtRenderArgs)") is None


def test_roofline_block_shape(bench):
    counts = {"primary": 100, "shadow": 200, "reflect": 0, "refract": 0, "hits": 70, "n_inner": 4000, "n_leaf": 500, "n_analytic": 490, "n_tri": 0, "n_bbox": 0}
    st = dict(counts, kernel_mode=3, kernel_variant=4)
    r = bench.roofline_block("no-such-workload/flat/gpus1", True, st, 1.0e9, 0.01, 6000.0, counts, counts, 300, 3, "flat", 5.0e8)
    f64, f32, flops = bench.algorithmic_ops(counts, 3, "flat")
    assert r["bound"] == "valu" and r["peak"] == 39.3 and r["traffic"] is None
    # the top-level figure counts the 17 instructions the shipped mesh-free step issues per node visit, one slot each; the textbook one stays in tree_step
    assert r["achieved"] == pytest.approx((f64 + 17 * counts["n_inner"]) / 0.01 / 1e12) and r["frac"] == pytest.approx(r["achieved"] / 39.3)
    assert r["frac"] == pytest.approx(r["valu"]["tree_step"]["frac_with_the_shipped_step"])
    assert r["f64_frac"] == pytest.approx(f64 / 0.01 / 1e12 / 39.3) and r["f64_frac"] < r["frac"] and r["issue_frac"] is None  # (no committed profile of this made-up workload)
    kd = bench.roofline_block("no-such-workload/kd/gpus1", True, dict(st, kernel_mode=7), 1.0e9, 0.01, 6000.0, counts, counts, 300, 3, "kd", 5.0e8)
    assert kd["frac"] == pytest.approx(kd["f64_frac"])  # the k-d semantics have no box step of the builder's in the formula: f64 work only
    assert r["hbm"]["needed_bytes"] == 5.0e8 and r["hbm"]["measured_bytes"] is None and r["hbm"]["waste_ratio"] is None and "no PMC profile" in r["hbm"]["profile"]
    assert r["algorithmic"]["GBps"] == pytest.approx(100.0) and r["kernel"].startswith("void pt_render_simple_kernel<3, false, false, 4, false>")
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, e in t.items():
        if "/waves" in key or "pt_render_simple_kernel<" not in e["kernel"]:
            continue
        mode, _, tex, waves, chain = e["kernel"].split("<")[1].split(">")[0].replace(" ", "").split(",")  # <MODE, STATS, TEX, WAVES, CHAIN>
        st = dict(counts, kernel_mode=int(mode), kernel_variant=int(waves) | (128 if tex == "true" else 0) | (512 if chain == "true" else 0))
        r = bench.roofline_block(key, True, st, 1.0e9, 0.0175, 6000.0, counts, counts, 300, 3, "flat", 5.0e8)
        assert r["traffic"] == e["hbm_bytes_per_launch"] and r["hbm"]["waste_ratio"] == pytest.approx(r["traffic"] / 5.0e8)
        assert r["hbm"]["GBps"] == pytest.approx(r["traffic"] / 0.0175 / 1e9)


def test_survey_accounting(bench):
    st = {"primary": 10, "shadow": 20, "reflect": 2, "refract": 1, "hits": 9, "n_inner": 400, "n_analytic": 50, "n_tri": 30, "n_bbox": 5}
    rays = 33
    assert bench.algorithmic_bytes(st, 3, 100, "flat") == 56 * rays + 56 * 400 + 104 * 50 + 72 * 30 + 48 * 5 + 9 * (168 + 80 + 360) + 2700
    assert bench.algorithmic_bytes(st, 3, 100, "kd") == bench.algorithmic_bytes(st, 3, 100, "flat") - 40 * 400
    f64, f32, flops = bench.algorithmic_ops(st, 3, "flat")
    assert f64 == 90 * 50 + 50 * 30 + 126 * 5 + 9 * (60 + 129) and f32 == 36 * 400 and flops == f64 + 48 * 400
    f64k, f32k, flopsk = bench.algorithmic_ops(st, 3, "kd")
    assert f64k == f64 + 9 * 400 and f32k == 0 and flopsk == f64k
