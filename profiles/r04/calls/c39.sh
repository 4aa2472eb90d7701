# round 4, call 39: the nested mesh walks of the k-d semantics cull boxes that end before the leaf's range starts: every test with a k-d render, k-d fuzz, speed against the walk without
python3 -m pytest tests -x -q -m gpu -k "kd or k_d or config_size or headline or fuzz or semantics or three" > gpurun_out/c39_tests.txt 2>&1
tail -3 gpurun_out/c39_tests.txt
FUZZ_MODES=kd timeout 900 python3 tests/fuzz_gpu_parity.py 83000 60 > gpurun_out/c39_fuzz.log 2>&1; tail -1 gpurun_out/c39_fuzz.log
bash profiles/variants.sh "nonear" "cows --traversal kd" "mirror --traversal kd" "big-soup --traversal kd" "big-mesh --traversal kd" "big-scene --traversal kd" > gpurun_out/c39_variants.txt 2>&1
cat gpurun_out/c39_variants.txt
