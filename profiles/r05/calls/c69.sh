# round 5, call 69: the k-d walk's instruction mix after the common-case split
timeout 300 bash $GRAFT_REPO_ROOT/profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" --no-extras --workload big-scene --traversal kd > gpurun_out/c69_kd_insts.txt 2>&1
cat gpurun_out/c69_kd_insts.txt
