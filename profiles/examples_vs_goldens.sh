#!/bin/bash
# Every example binary that has a reference render, run as a user would (default traversal = the crate without
# features, SAMPLES=16) from tests/golden (where assets/ lives), compared with the reference's committed PNG:
# 8x8 block means, because anti-aliasing / light / glossy samples are random on both sides.
cd tests/golden || exit 1
while read bin out gold; do
  SAMPLES=16 ../../examples/bin/$bin > /dev/null 2> err.txt || { echo "$bin FAILED: $(tail -1 err.txt)"; continue; }
  python3 - "$bin" "$out" "$gold" <<'PY'
import sys, numpy as np
from PIL import Image
b, out, gold = sys.argv[1:]
a = np.array(Image.open(out).convert("RGB")).astype(float); g = np.array(Image.open("render/" + gold).convert("RGB")).astype(float)
if a.shape != g.shape: print("%-32s size %s vs golden %s" % (b, a.shape, g.shape)); sys.exit()
k = 8; blk = lambda x: x[:x.shape[0]//k*k, :x.shape[1]//k*k].reshape(x.shape[0]//k, k, x.shape[1]//k, k, 3).mean(axis=(1, 3))
d = np.abs(blk(a) - blk(g)).max(axis=2); px = np.abs(a - g).max(axis=2)
print("%-32s %4dx%-4d  block mean diff %.3f  blocks>6 %.4f  pixels exact %.3f within1 %.3f" % (b, a.shape[1], a.shape[0], d.mean(), (d > 6).mean(), (px == 0).mean(), (px <= 1).mean()))
PY
  rm -f $out
done <<'LIST'
primitives-simple primitives-simple.png 01a_primitives-simple.png
smooth-shading smooth-shading.png 02_smooth-shading.png
normal-mapping normal-mapping.png 04a_normal-mapping.png
water-glass water-glass.png 06a_water-glass.png
transmission-refraction transmission-refraction.png 06b_transmission-refraction.png
glossy-reflection glossy-reflection.png 07_glossy-reflection.png
soft-shadows soft-shadows.png 08_soft-shadows.png
entering-the-mirror-dimension entering-the-mirror-dimension.png entering-the-mirror-dimension.png
LIST
rm -f err.txt normal-mapping-left.png normal-mapping-right.png
