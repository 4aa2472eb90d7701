for i in 1 2; do
bash profiles/variants.sh "nobound" "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "mirror" "cows" "aquarium" "big-soup --samples 64"
done > gpurun_out/c49_ab.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/c49_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c49_pytest.log
timeout 900 python tests/fuzz_gpu_parity.py 80000 200 > gpurun_out/c49_fuzz.log 2>&1
