# round 4, call 59: the worked-around instantiation (mode 2, counting, variant 0) and its guard test; then the switch matrix of c55 again, with the tests that assert the DEFAULT instantiation left out
python3 -m pytest tests/test_gpu_render_parity.py -x -q -m gpu -k "without_parked_frame" > gpurun_out/c59_guard.txt 2>&1; grep -h "passed\|failed" gpurun_out/c59_guard.txt | tail -1
for e in PORTRAYER_WAVES=3 PORTRAYER_WAVES=4 PORTRAYER_CHAIN=0 PORTRAYER_FORK=1 PORTRAYER_PARK=0 PORTRAYER_LDS_STACK=6 PORTRAYER_STACK_CAP=96; do
  env $e python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_textures.py -q -m gpu -k "not densest and not chain_kernel and not fork_join and not plain_kernels" > gpurun_out/c59_$e.txt 2>&1
  echo "$e: $(grep -h 'passed\|failed' gpurun_out/c59_$e.txt | tail -1)"
done > gpurun_out/c59_env_matrix.txt 2>&1
cat gpurun_out/c59_env_matrix.txt
