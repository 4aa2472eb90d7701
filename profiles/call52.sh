for i in 1 2; do
bash profiles/variants.sh "nopacket" "big-scene" "big-scene --traversal hier" "big-scene --share 8" "big-scene --width 3840 --height 2160 --samples 256 --steps 2"
done > gpurun_out/c52_ab.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/c52_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c52_pytest.log
timeout 900 python tests/fuzz_gpu_parity.py 90000 300 > gpurun_out/c52_fuzz.log 2>&1
timeout 600 python tests/fuzz_gpu_parity.py 91000 60 64 48 32 >> gpurun_out/c52_fuzz.log 2>&1
timeout 600 python tests/fuzz_gpu_parity.py 92000 40 40 30 64 >> gpurun_out/c52_fuzz.log 2>&1
