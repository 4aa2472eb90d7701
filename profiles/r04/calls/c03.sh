# round 4, call 3: the wave-uniform k-d walk, second build (no per-lane fall-back in the render kernels, arguments re-read, ray by value): where c02's suite run hung,
# and counters of big-scene's k-d frame, old kernel against new
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/kdw/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
timeout 120 python -m pytest tests/test_gpu_render_parity.py -m gpu -q -x -k "stack_beyond_lds and cows-kd" > gpurun_out/c03_hang1.log 2>&1; echo "rc $?" >> gpurun_out/c03_hang1.log
timeout 120 python -m pytest tests/test_gpu_render_parity.py -m gpu -q -x -k "stack_beyond_lds and big-scene-kd" > gpurun_out/c03_hang2.log 2>&1; echo "rc $?" >> gpurun_out/c03_hang2.log
for wl in big-scene mirror cows; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl --traversal kd > gpurun_out/c03_kd_$wl.json 2> gpurun_out/c03_kd_$wl.err; done
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
C2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
bash profiles/pmc_quick.sh "$C1" --no-extras --workload big-scene --traversal kd > gpurun_out/c03_pmc_new_1.txt 2>&1
bash profiles/pmc_quick.sh "$C2" --no-extras --workload big-scene --traversal kd > gpurun_out/c03_pmc_new_2.txt 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
bash profiles/pmc_quick.sh "$C1" --no-extras --workload big-scene --traversal kd > gpurun_out/c03_pmc_old_1.txt 2>&1
bash profiles/pmc_quick.sh "$C2" --no-extras --workload big-scene --traversal kd > gpurun_out/c03_pmc_old_2.txt 2>&1
