#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag> directory (written by profiles/run_profile.sh: bench.py --no-extras, so the timed
render kernel is the only pt_render_kernel<.., false, ..> in it) into the committed summary files:
profiles/<round>/<tag>_kernel_stats.csv, <tag>_pmc.json, and this workload's entry of profiles/traffic.json.
usage: summarise.py <tag> <round> <key>     key = "<workload>/<traversal>/gpus<N>" (or "<workload>@WxHxS/...")"""
import collections, csv, glob, json, os, re, shutil, sys
tag, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)  # gpurun merges every call's output into the same directory: take the last run's file
shutil.copy(newest(f"{src}/trace/*/*_kernel_stats.csv"), f"{dst}/{tag}_kernel_stats.csv")
timed = r"pt_render(_simple)?_kernel<\d+, false,"  # the timed kernel (STATS = false), not the counting launch
out = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    for f in [newest(f"{src}/{d}/*/*_counter_collection.csv")]:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if re.search(timed, r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
for r in csv.DictReader(open(f"{dst}/{tag}_kernel_stats.csv")):
    if re.search(timed, r["Name"]):
        out["kernel"] = r["Name"]; out["kernel_trace_avg_ns"] = float(r["AverageNs"]); out["kernel_trace_calls"] = int(r["Calls"])
g = lambda k: out[k]["mean_per_launch"]
ms = out["kernel_trace_avg_ns"] / 1e6
# lanes_active: thread quad-cycles per vector instruction. An instruction that takes more than one pass (f64 reciprocal / square root, 32-bit integer
# multiply) counts its lanes once per pass, so the figure is biased upwards (64.7 "of 64" on the chain kernel); single-lane v_readlane / v_writelane pull it down.
derived = {"lanes_active": g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU"),
           "valu_busy": 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * ms * 1e-3 * 2.4e9),  # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs at 2.4 GHz
           "waiting": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))}
out["derived"] = derived
json.dump(out, open(f"{dst}/{tag}_pmc.json", "w"), indent=1)
tpath = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
# FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE reads half of a wide coalesced read on gfx950 (MI355X guide, HBM section)
t[key] = dict(derived, hbm_bytes_per_launch=(2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0, fetch_size_kb=g("FETCH_SIZE"), write_size_kb=g("WRITE_SIZE"),
              kernel=out["kernel"], kernel_trace_avg_ms=ms,
              valu_issue_frac=g("SQ_THREAD_CYCLES_VALU") / (ms * 1e-3 * 39.3e12), insts_valu_per_launch=g("SQ_INSTS_VALU"),
              source=f"profiles/{rnd}/{tag}_pmc.json (rocprofv3 --pmc passes, one counter group per pass, bench.py --steps 3 --no-extras)")
json.dump(t, open(tpath, "w"), indent=1)
print(json.dumps(t[key], indent=1))
