echo > gpurun_out/c12_pytest.log
python3 - >> gpurun_out/c12_pytest.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from PIL import Image
from portrayer_amd import host
from scene_dsl import ASSETS, GOLDEN
from test_examples_extra import block_diff
os.environ["SAMPLES"] = "16"; os.environ["PORTRAYER_KDMESH_AS_MESH"] = "1"
for name, golden in (("robot-alarm-clock", "10_robot-alarm-clock_green.png"), ("primitives", "01b_primitives.png")):
    g = np.array(Image.open(os.path.join(GOLDEN, "render", golden)).convert("RGB"))
    out = "/tmp/%s.png" % name
    rc = host.lib().ph_example_render_to_png(name.encode(), ASSETS.encode(), 0, g.shape[1], g.shape[0], out.encode())
    mine = np.array(Image.open(out).convert("RGB"))
    print(name, g.shape, rc, block_diff(mine, g), "exact %.1f %%" % (100 * (mine == g).all(axis=2).mean()))
    Image.fromarray(mine[::4, ::4]).save("gpurun_out/c12_%s_small.png" % name)
PY
