# round 4, call 31: texture / normal maps applied outside the interpreter's state machine (PT_ST_MAPS + pt_lane_maps): parity of the textured scenes,
# the dielectric workload's speed (inline and out-of-line map routine in the flat_scene kernel), its HBM-side bytes
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py -x -q -m gpu -k "reference_scenes or scene or tex or map or config_size or water or default_feature" > gpurun_out/c31_tests.txt 2>&1
tail -3 gpurun_out/c31_tests.txt
bash profiles/variants.sh "mapsout" aquarium "aquarium --traversal hier" water-glass "aquarium --samples 64 --steps 2" > gpurun_out/c31_variants.txt 2>&1
cat gpurun_out/c31_variants.txt
bash profiles/pmc_quick.sh "FETCH_SIZE" --no-extras --workload aquarium > gpurun_out/c31_pmc_fetch.txt 2>&1
bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload aquarium > gpurun_out/c31_pmc_write.txt 2>&1
bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" --no-extras --workload aquarium > gpurun_out/c31_pmc_sq.txt 2>&1
tail -4 gpurun_out/c31_pmc_fetch.txt gpurun_out/c31_pmc_write.txt; tail -12 gpurun_out/c31_pmc_sq.txt
