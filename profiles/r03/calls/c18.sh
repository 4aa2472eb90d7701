# round 3, call 18: the straight-line kernel with a depth per lane (PORTRAYER_TREE=1) against the interpreter on the scenes with dielectrics
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
for tr in 1 0; do export PORTRAYER_TREE=$tr; TAG="tree=$tr"; for wl in "transmission-refraction" "transmission-refraction --traversal hier" "transmission-refraction --traversal kd" "water-glass" "water-glass --traversal hier" "aquarium" "aquarium --traversal hier" "aquarium --samples 64"; do run --workload $wl; done; done > gpurun_out/c18_tree.log 2>&1
export PORTRAYER_TREE=1
timeout 600 python -m pytest tests/test_gpu_textures.py tests/test_gpu_render_parity.py tests/test_examples_extra.py -m gpu -q -x > gpurun_out/c18_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c18_pytest.log
timeout 600 python3 tests/fuzz_gpu_parity.py 13000 60 > gpurun_out/c18_fuzz.log 2>&1
