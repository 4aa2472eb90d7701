#!/usr/bin/env python3
"""Frame time of macho-cows with every Mesh turned into a KDMesh (the reference's own triangle k-d trees,
quirk Q3), FLAT traversal, 1280x720 SAMPLES=16. PORTRAYER_KD_NO_CULL=1 switches the conservative culls off."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import example_scenes, host_glue, scene_dsl
from portrayer_amd import _hip as H, host

scene, cam, _ = example_scenes.macho_cows()
def convert(n):
    if n.geometry is not None and n.geometry[0].kind == scene_dsl.MESH:
        n.geometry = (scene_dsl.KDMesh(n.geometry[0].mesh, n.geometry[0].smooth), n.geometry[1])
    for c in n.children: convert(c)
convert(scene.root)
w, h = 1280, 720
r = host.Renderer(host_glue.host_scene(scene), H.TRAVERSE_FLAT)
bg = scene_dsl.default_background(w, h)
for i in range(3):
    rgb, _, st = r.render(host_glue.cam10(cam), w, h, bg, samples=16, seed=0, sample_mode=H.SAMPLE_RNG, stats=(i == 0), want_linear=False)
    if i == 0: rays = st["primary"] + st["shadow"] + st["reflect"] + st["refract"]; print("tri tests/ray %.2f" % (st["n_tri"] / rays))
    else: print("kernel %.2f ms  %.0f Mray/s" % (st["kernel_ms"], rays / st["kernel_ms"] / 1e3))
