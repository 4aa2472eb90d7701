#!/usr/bin/env python3
"""Debug helper for a fuzz mismatch: python tests/fuzz_debug.py <seed> <random|extreme> <flat|kd>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import host_glue
import oracle_lib as O
from portrayer_amd import _hip as H, host
from scene_dsl import default_background
from test_gpu_render_parity import random_scene
from fuzz_gpu_parity import extreme_scene

seed, kind, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
w, h = 128, 96
scene, cam = (random_scene if kind == "random" else extreme_scene)(seed)
ps = O.pack(scene); hs = host_glue.host_scene(scene)
tr, om = (H.TRAVERSE_FLAT, O.MODE_FLAT) if mode == "flat" else (H.TRAVERSE_KD, O.MODE_KD)
r = host.Renderer(hs, tr, kd_depth=8)
for samples, smode, jit in ((1, H.SAMPLE_CENTRE, O.JITTER_CENTRE), (2, H.SAMPLE_RNG, O.JITTER_RNG)):
    rgb, linear, st = r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=samples, seed=seed, sample_mode=smode, stats=True)
    ref = O.render(ps, cam, w, h, samples=samples, seed=seed, jitter=jit, mode=om, kd_depth=8)
    bad = np.argwhere((rgb != ref.rgb).any(axis=2))
    print("samples", samples, "bad pixels", bad.tolist()[:10], {k: (st[k], ref.stats[k]) for k in ("primary", "shadow", "reflect", "refract", "hits")})
    for y, x in bad[:3]:
        print("  pixel", x, y, "gpu", rgb[y, x], "oracle", ref.rgb[y, x], "linear gpu", linear[y, x], "oracle", ref.linear[y, x])
# primary rays of the whole image through the cast API (centre sample)
ys, xs = np.mgrid[0:h, 0:w]
xy = np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], axis=1)
o, d = O.camera_rays(cam, w, h, xy)
import ctypes as C
lib = H.lib(); ctx = r.context
n = len(o); t = np.zeros(n); node = np.zeros(n, dtype=np.int32); sub = np.zeros(n, dtype=np.int32)
o = np.ascontiguousarray(o); d = np.ascontiguousarray(d)
rc = lib.pt_test_cast_rays(ctx, n, o.ctypes.data_as(H._dp), d.ctypes.data_as(H._dp), 0, t.ctypes.data_as(H._dp), node.ctypes.data_as(H._ip), sub.ctypes.data_as(H._ip))
rt, rid, rp, rn = O.cast_rays(ps, o, d, mode=om, kd_depth=8)
diff = np.argwhere((node != rid) | (t != rt)).ravel()
print("primary cast diffs:", len(diff))
flat = O.flatten(ps)
for i in diff[:6]:
    print("  ray", i, "pixel", (i % w, i // w), "gpu node", node[i], "t", t[i], "| oracle node", rid[i], "t", rt[i], "types", flat["prim_type"][node[i]] if node[i] >= 0 else None, flat["prim_type"][rid[i]] if rid[i] >= 0 else None)
