for i in 1 2; do
bash profiles/variants.sh "xyod" "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "mirror" "cows" "aquarium" "big-soup --samples 64"
done > gpurun_out/c63_ab.log 2>&1
for v in xyod; do cp portrayer_amd/libportrayer_hip.so /tmp/keep.so; cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; echo $v; bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload big-scene; cp /tmp/keep.so portrayer_amd/libportrayer_hip.so; done > gpurun_out/c63_pmc.log 2>&1
