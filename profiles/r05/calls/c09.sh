# round 5, call 9: the switch matrix as a test of the GPU suite (every environment switch x five scenes x three semantics x counting / plain against the oracle)
timeout 1800 python3 -m pytest tests/test_gpu_switch_matrix.py -q -m gpu --durations=5 > gpurun_out/c09_matrix.txt 2>&1; tail -15 gpurun_out/c09_matrix.txt
for e in PORTRAYER_PARK=0 PORTRAYER_FORK=1; do
  env $e python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_textures.py -q -m gpu > gpurun_out/c09_$e.txt 2>&1
  echo "$e: $(grep -h 'passed\|failed' gpurun_out/c09_$e.txt | tail -1)"
done
