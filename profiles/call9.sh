python -m pytest tests -m gpu -x -q > gpurun_out/c9_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c9_pytest.log
bash profiles/variants.sh "refill1 refill8 refill32 refill64" "big-scene" "big-soup" "mirror" "aquarium" "cows" "big-scene --traversal kd" "big-scene --traversal hier" > gpurun_out/c9_variants.log 2>&1
