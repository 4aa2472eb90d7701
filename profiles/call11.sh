cd examples/bin 2>/dev/null; cd ../..
PORTRAYER_VERBOSE=1 python3 - > gpurun_out/c11_prep.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from portrayer_amd import _hip as H, host
for name, n in (("big-scene", 10), ("big-scene", 10), ("synthetic:big-soup", 6), ("macho-cows", 0)):
    t0 = time.perf_counter(); sc = host.Scene.example(name, n=n or 10); t1 = time.perf_counter()
    r = host.Renderer(sc, H.TRAVERSE_HIER); t2 = time.perf_counter()
    print(name, "script %.1f ms renderer %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), r.prepare_ms(), flush=True)
    r.close()
PY
