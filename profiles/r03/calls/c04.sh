# round 3, call 4: why the packed slab test bought nothing - A/B of builds, cycle split, instruction counters
bash profiles/variants.sh "nopk prefetch" "big-scene" "big-soup --samples 64" "mirror" > gpurun_out/c04_variants.log 2>&1
bash profiles/cycles.sh "--workload big-scene" "--workload big-soup --samples 64" "--workload cows" > gpurun_out/c04_cycles.log 2>&1
bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS" --no-extras --workload big-scene > gpurun_out/c04_pmc.log 2>&1
bash profiles/pmc_quick.sh "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAVES_LT_64" --no-extras --workload big-scene > gpurun_out/c04_pmc2.log 2>&1
