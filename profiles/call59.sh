python -m pytest tests -m gpu -x -q > gpurun_out/c59_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c59_pytest.log
for i in 1 2; do
bash profiles/variants.sh "nopacket" "big-scene" "big-scene --traversal hier" "mirror" "mirror --traversal hier" "cows" "cows --traversal hier" "aquarium" "aquarium --traversal hier" "big-soup --samples 64" "big-soup --samples 64 --traversal hier" "big-mesh"
done > gpurun_out/c59_ab.log 2>&1
