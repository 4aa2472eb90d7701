"""Round 5: what a big-soup ray does - counters of the counting build (wave-uniform walk: the counts are per LANE that takes part)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from portrayer_amd import _hip as H
from portrayer_amd import host
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "big-soup"
hs, cam, w, h, s = bench.load_workload(host, name) if hasattr(bench, "load_workload") else (None,) * 5
