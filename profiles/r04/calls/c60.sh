# round 4, call 60: the switch matrix over the configuration-size tests and the fuzz slice
for e in PORTRAYER_PARK=0 PORTRAYER_WAVES=5 PORTRAYER_WAVES=4 PORTRAYER_KD_WAVES=4 PORTRAYER_MESH_OCT=0 PORTRAYER_FORK=1 PORTRAYER_FINE_QUEUES=0; do
  env $e python3 -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_fuzz_slice.py -q -m gpu -k "not headline" > gpurun_out/c60_$e.txt 2>&1
  echo "$e: $(grep -h 'passed\|failed' gpurun_out/c60_$e.txt | tail -1)"
done > gpurun_out/c60_env_matrix.txt 2>&1
cat gpurun_out/c60_env_matrix.txt
