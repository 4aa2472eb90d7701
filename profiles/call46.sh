PORTRAYER_VERBOSE=1 python3 - > gpurun_out/c46_copy.log 2>&1 <<'PY'
from portrayer_amd import _hip as H
c = H.Context(0)
print("best", c.copy_bandwidth(1 << 30, 3))
c.close()
PY
