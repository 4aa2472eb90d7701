# round 4, call 55: the instantiations a user can select by environment still match the oracle on the final tree (render parity + config sizes under each switch)
for e in PORTRAYER_WAVES=3 PORTRAYER_WAVES=4 PORTRAYER_WAVES=5 PORTRAYER_KD_WAVES=3 PORTRAYER_KD_WAVES=4 PORTRAYER_CHAIN_WAVES=3 PORTRAYER_CHAIN=0 PORTRAYER_MESH_OCT=0 PORTRAYER_FORK=1 PORTRAYER_PARK=0 PORTRAYER_FINE_QUEUES=0 PORTRAYER_LANE_CHUNKS=1 PORTRAYER_BUILD=host PORTRAYER_KD_CULL=0; do
  env $e python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_textures.py -x -q -m gpu > gpurun_out/c55_$e.txt 2>&1
  echo "$e: $(grep -h 'passed\|failed' gpurun_out/c55_$e.txt | tail -1)"
done > gpurun_out/c55_env_matrix.txt 2>&1
cat gpurun_out/c55_env_matrix.txt
