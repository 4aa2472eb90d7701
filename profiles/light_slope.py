#!/usr/bin/env python3
"""Where big-scene's kernel time goes, measured by taking things away (profiles/r03/notes.md section 9):
kernel time of the 1920x1080x64 frame with 0, 1, 2, 3 of the scene's lights. The slope is one light's shadow walk + light term;
run again with an ablation build (-DPT_ABLATE=1: no specular term, =2: no light term at all) to split the slope.
usage (GPU box, repo root): python3 profiles/light_slope.py [flat|hier|kd]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import host_glue
from example_scenes import big_scene
from portrayer_amd import _hip as H
from portrayer_amd import host

mode = sys.argv[1] if len(sys.argv) > 1 else "flat"
tr = {"flat": H.TRAVERSE_FLAT, "hier": H.TRAVERSE_HIER, "kd": H.TRAVERSE_KD}[mode]
w, h, s = 1920, 1080, 64
v = np.arange(h, dtype=np.float64) / float(h)
bg = np.ascontiguousarray(np.array([0.2, 0.4, 0.6])[None, :] * (1.0 - v)[:, None] + np.array([0.0, 0.0, 1.0])[None, :] * v[:, None])
scene, cam, _ = big_scene(10)
lights = list(scene.lights)
for n in range(len(lights) + 1):
    scene.lights = lights[:n]
    r = host.Renderer(host_glue.host_scene(scene), tr)
    img = np.zeros((h, w, 3), dtype=np.uint8)
    _, _, c = r.render(host_glue.cam10(cam), w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False, stats=True)
    best = 1e9
    for _ in range(3):
        _, _, st = r.render(host_glue.cam10(cam), w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False)
        best = min(best, st["kernel_ms"])
    r.close()
    rays = c["primary"] + c["shadow"]
    print("%s lights %d: kernel %7.3f ms  rays %.4g (shadow %.4g, hits %.4g)  nodes/ray %.2f prim/ray %.2f" % (mode, n, best, rays, c["shadow"], c["hits"], c["n_inner"] / rays, c["n_analytic"] / rays), flush=True)
