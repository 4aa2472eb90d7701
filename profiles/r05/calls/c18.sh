# round 5, call 18: instruction counts of the k-d walk as it stands (octant culls, top levels recomputed) - round 4's kernel issued 1.27e10 vector instructions per big-scene frame
bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" --no-extras --workload big-scene --traversal kd > gpurun_out/c18_kd_insts.txt 2>&1
bash profiles/pmc_quick.sh "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" --no-extras --workload big-scene --traversal kd >> gpurun_out/c18_kd_insts.txt 2>&1
bash profiles/pmc_quick.sh "FETCH_SIZE WRITE_SIZE" --no-extras --workload big-scene --traversal kd >> gpurun_out/c18_kd_insts.txt 2>&1
cat gpurun_out/c18_kd_insts.txt
