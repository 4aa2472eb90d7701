for i in 1 2; do
bash profiles/variants.sh "base" "big-scene --traversal hier" "mirror --traversal hier" "aquarium --traversal hier" "cows --traversal hier" "big-scene"
done > gpurun_out/c38_ab.log 2>&1
