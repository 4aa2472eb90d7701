# round 4, call 8: the k-d walk with lane masks in scalar registers, branch-free lane updates, the short division, pinned pointers, one-word stack entries:
# the division's test, every test with a k-d render, a k-d fuzz, the k-d workloads, and the instruction counters of big-scene's frame
cp build/variants/kdw/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
timeout 300 python -m pytest tests/test_gpu_device_parity.py -m gpu -q -x -k "short_division or ieee" > gpurun_out/c08_div.log 2>&1; echo "pytest rc $?" >> gpurun_out/c08_div.log
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=300 -k "kd or parallel_to_an_axis or example_matches or random_scene or extreme or chain or textured or stack or golden or fork" > gpurun_out/c08_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c08_pytest.log
FUZZ_MODES=kd timeout 600 python3 tests/fuzz_gpu_parity.py 73000 40 > gpurun_out/c08_fuzz.log 2>&1
for wl in big-scene mirror cows; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl --traversal kd > gpurun_out/c08_kd_$wl.json 2> gpurun_out/c08_kd_$wl.err; done
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
C2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
bash profiles/pmc_quick.sh "$C1" --no-extras --workload big-scene --traversal kd > gpurun_out/c08_pmc_1.txt 2>&1
bash profiles/pmc_quick.sh "$C2" --no-extras --workload big-scene --traversal kd > gpurun_out/c08_pmc_2.txt 2>&1
