for i in 1 2; do
bash profiles/variants.sh "base" "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "mirror" "mirror --traversal hier" "cows" "aquarium" "big-soup --samples 64" "big-mesh"
done > gpurun_out/c34_ab.log 2>&1
