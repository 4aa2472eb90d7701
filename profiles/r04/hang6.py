"""round 4: the 6-wave hierarchical instantiation that never finished in round 3 (profiles/r03/notes.md section 11, c48), now with the walks' watchdog"""
import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from portrayer_amd import _hip as H, host
from scene_dsl import default_background
which, mode, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
scene = host.Scene.example("big-scene", n=n)
tr = {"flat": H.TRAVERSE_FLAT, "hier": H.TRAVERSE_HIER}[mode]
w, h = 64, 64
r = host.Renderer(scene, tr)
t = time.time()
try:
    rgb, linear, st = r.render(scene.camera, w, h, default_background(w, h), samples=64, seed=1, sample_mode=H.SAMPLE_RNG, stats=(which == "stats"))
    print(which, mode, n, "ok: variant", st["kernel_variant"], "mode", st["kernel_mode"], "kernel ms", st["kernel_ms"], "overflow", st["stack_overflow"], flush=True)
except Exception as e:
    print(which, mode, n, "raised", repr(e)[:300], flush=True)
