# round 3, call 34: the chain kernel at 4 waves per SIMD (PORTRAYER_CHAIN_WAVES=4)
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-10s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
for wv in 3 4; do export PORTRAYER_CHAIN_WAVES=$wv; TAG="waves=$wv"; for wl in "mirror" "mirror --traversal hier" "mirror --samples 256"; do run --workload $wl; done; done > gpurun_out/c34_chain_waves.log 2>&1
export PORTRAYER_CHAIN_WAVES=4
timeout 600 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py -m gpu -q -x -k "chain or mirror or glossy or config_size" > gpurun_out/c34_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c34_pytest.log
