# round 5, call 38: big-soup's instruction mix (which unit bounds the mesh walk?)
for w in big-soup big-mesh; do
echo "== $w x64: instruction counts per launch"
timeout 300 bash $GRAFT_REPO_ROOT/profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" --no-extras --workload $w --samples 64
echo "== $w x64: busy / wait cycles per launch"
timeout 300 bash $GRAFT_REPO_ROOT/profiles/pmc_quick.sh "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" --no-extras --workload $w --samples 64
done > gpurun_out/c38_soup_insts.txt 2>&1
cat gpurun_out/c38_soup_insts.txt
