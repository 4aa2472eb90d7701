for i in 1 2; do
bash profiles/variants.sh "base" "big-scene" "big-scene --traversal hier" "mirror" "cows" "aquarium" "big-soup --samples 64"
done > gpurun_out/c72_ab.log 2>&1
