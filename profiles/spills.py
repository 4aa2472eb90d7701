#!/usr/bin/env python3
"""Where does hipcc spill? Compiles pt_api.hip with -g -save-temps at a given occupancy bound and
lists scratch loads/stores of one pt_render_kernel instantiation by source line.
usage: spills.py <min_waves or 0> [extra flags]   (env KERNEL = "MODE,STATS,TEX,WAVES", default "3,0,0,3")"""
import collections, os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = "/tmp/pt_spills"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp + "/include")
for f in os.listdir(root + "/portrayer_amd/csrc"):
    shutil.copy(root + "/portrayer_amd/csrc/" + f, tmp) if not f.endswith(".o") else None
shutil.copy(root + "/include/portrayer_hip.h", tmp + "/include")
src = open(tmp + "/pt_api.hip").read().replace("../../include/portrayer_hip.h", "include/portrayer_hip.h")
open(tmp + "/pt_api.hip", "w").write(src)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
       "-Wno-unused-value", "-save-temps", "-g", f"-DPT_MIN_WAVES={sys.argv[1]}", *sys.argv[2:], "pt_api.hip", "-o", "x.so"]
subprocess.run(cmd, cwd=tmp, stderr=subprocess.DEVNULL, timeout=600)
text = open(tmp + "/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
files = {}
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2))
mode, stats, tex, waves = os.environ.get("KERNEL", "3,0,0,3").split(",")
start = next(i for i, l in enumerate(text) if l.startswith(f"_Z16pt_render_kernelILi{mode}ELb{stats}ELb{tex}ELi{waves}EEv12PtRenderArgs:"))
end = start
while "s_endpgm" not in text[end]: end += 1
loc = None; cnt = collections.Counter()
for l in text[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: loc = (files.get(m.group(1), m.group(1)).split("/")[-1], int(m.group(2)))
    if re.search(r"scratch_(load|store)", l): cnt[(loc, "st" if "store" in l else "ld")] += 1
print("kernel lines", end - start, "scratch ops", sum(cnt.values()))
for (loc, k), c in sorted(cnt.items(), key=lambda x: -x[1])[:30]: print(c, k, loc)
