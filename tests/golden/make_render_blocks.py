#!/usr/bin/env python3
"""Writes tests/golden/render_blocks8/<name>.png: 8x8 block means (rounded to u8) of reference renders that are too large to commit
whole (2.8 MB each). Data derived from the reference's own data files /root/reference/render/*.png; the tests that read them
(tests/test_oracle_goldens.py::test_which_kdmesh_behaviour_the_reference_renders_support) compare block means anyway.
usage: python3 tests/golden/make_render_blocks.py [/root/reference/render]"""
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/render"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_blocks8")
os.makedirs(dst, exist_ok=True)
for name in ("10_robot-alarm-clock.png", "10_robot-alarm-clock_dark_blue.png", "10_robot-alarm-clock_red.png"):
    g = np.array(Image.open(os.path.join(src, name)).convert("RGB")).astype(np.float64)
    h, w = (g.shape[0] // 8) * 8, (g.shape[1] // 8) * 8
    m = g[:h, :w].reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))
    Image.fromarray(np.rint(m).astype(np.uint8)).save(os.path.join(dst, name))
    print(name, m.shape)
