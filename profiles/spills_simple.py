#!/usr/bin/env python3
"""Where does hipcc spill in a straight-line kernel? Compiles one mode's render kernels with -g -save-temps and lists the scratch loads / stores of one
pt_render_simple_kernel instantiation by source line and loop depth.
usage: spills_simple.py [extra hipcc flags]   (env KERNEL = "MODE,STATS,TEX,WAVES,CHAIN", default "3,0,0,6,0")"""
import collections, os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = "/tmp/pt_spills_simple"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mode, stats, tex, waves, chain = os.environ.get("KERNEL", "3,0,0,6,0").split(",")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value",
       "--cuda-device-only", "-save-temps", "-g", f"-DPT_INST_MODE={mode}", *sys.argv[1:], "-c", root + "/portrayer_amd/csrc/pt_render_inst.hip", "-o", "x.o"]
subprocess.run(cmd, cwd=tmp, stderr=subprocess.DEVNULL, timeout=900)
asm = [f for f in os.listdir(tmp) if f.endswith(".s")][0]
text = open(os.path.join(tmp, asm)).read().split("\n")
files = {}
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2))
start = next(i for i, l in enumerate(text) if l.startswith(f"_Z23pt_render_simple_kernelILi{mode}ELb{stats}ELb{tex}ELi{waves}ELb{chain}EEv12PtRenderArgs:"))
end = start
while "s_endpgm" not in text[end]: end += 1
loc = None; depth = 0; cnt = collections.Counter()
for l in text[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: loc = (files.get(m.group(1), m.group(1)).split("/")[-1], int(m.group(2)))
    if l.startswith((".LBB", "; %bb")):
        m = re.search(r"Depth=(\d+)", l)
        depth = int(m.group(1)) if m else 0
    if re.search(r"scratch_(load|store)", l): cnt[(loc, "st" if "store" in l else "ld", depth)] += 1
print("kernel lines", end - start, "scratch ops", sum(cnt.values()))
for (loc, k, d), c in sorted(cnt.items(), key=lambda x: (-x[0][2], -x[1]))[:50]: print(c, k, "loop depth", d, loc)
