#!/usr/bin/env python3
"""One-paragraph reading of a gpurun_out/prof_<tag> directory (profiles/run_profile.sh): per launch of the timed
render kernel. usage: digest.py <tag>"""
import collections, csv, glob, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
c = {}
newest = lambda pattern: [max(glob.glob(pattern), key=os.path.getmtime)] if glob.glob(pattern) else []
for f in [x for d in glob.glob(f"{src}/pmc_*") for x in newest(f"{d}/*/*_counter_collection.csv")]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if re.search(r"pt_render_kernel<\d+, false,", r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        c[k] = sum(v) / len(v)
ms = name = None
for f in newest(f"{src}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"pt_render_kernel<\d+, false,", r["Name"]):
            ms, name = float(r["AverageNs"]) / 1e6, r["Name"]
g = lambda k: c.get(k, float("nan"))
rd, wr = 2 * g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
print(f"{tag}: {name}  {ms:.2f} ms per launch (rocprofv3 kernel-trace)")
print(f"   HBM-side traffic {rd / 1e9:.2f} GB read (FETCH_SIZE x 2) + {wr / 1e9:.2f} GB written = {(rd + wr) / 1e9:.2f} GB -> {(rd + wr) / ms / 1e9:.3f} TB/s ; L2 hits / misses {g('TCC_HIT_sum'):.3g} / {g('TCC_MISS_sum'):.3g}")
print(f"   VALU: {g('SQ_INSTS_VALU'):.3g} wave instructions, {g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU'):.1f} of 64 lanes active; busy {4 * g('SQ_ACTIVE_INST_VALU') / (1024 * ms * 1e-3 * 2.4e9) * 100:.0f} % of 1024 SIMDs x 2.4 GHz; waiting {100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.0f} % of wave cycles; LDS bank conflicts {g('SQ_LDS_BANK_CONFLICT'):.3g}")
