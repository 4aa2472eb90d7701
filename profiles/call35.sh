for i in 1 2; do
bash profiles/variants.sh "base ondemand" "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "mirror" "mirror --traversal hier" "cows" "big-soup --samples 64"
done > gpurun_out/c35_ab.log 2>&1
( bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload big-scene ) > gpurun_out/c35_pmc.log 2>&1
