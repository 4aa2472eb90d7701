mkdir -p gpurun_out/c1
python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1/pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c1/bench.log 2>&1
python bench.py --gpus 2 --backend gloo --same-device --check --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c1/bench_g2.log 2>&1; echo "rc $?" >> gpurun_out/c1/bench_g2.log
bash profiles/diag.sh "--workload big-scene" "--workload big-scene --traversal hier" "--workload big-scene --traversal kd" "--workload big-soup" "--workload big-mesh" "--workload mirror --samples 16" "--workload aquarium" "--workload cows" > gpurun_out/c1/diag.log 2>&1
