for i in 1 2; do
bash profiles/variants.sh "base" "big-scene" "big-soup" "big-soup --samples 64" "big-mesh" "cows" "big-scene --traversal kd" "big-scene --traversal hier"
done > gpurun_out/c26_ab.log 2>&1
