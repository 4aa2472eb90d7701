# round 4, call 12: pt_node with frames in a pipeline (ABI 7): the multi-rank tests, then what 8 ranks cost on one GPU
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x --timeout=600 > gpurun_out/c12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c12_pytest.log
for t in 1 0; do
PORTRAYER_NODE_THREADS=$t timeout 600 python3 bench.py --gpus 8 --same-device --workload big-scene --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/c12_node8_threads$t.json 2> gpurun_out/c12_node8_threads$t.err
PORTRAYER_NODE_THREADS=$t timeout 600 python3 bench.py --gpus 8 --same-device --workload big-scene --no-cpu-baseline --no-extras --steps 10 --warmup 3 --no-pipeline > gpurun_out/c12_node8_nopipe_threads$t.json 2> gpurun_out/c12_node8_nopipe_threads$t.err
done
timeout 600 python3 bench.py --gpus 2 --same-device --workload big-scene --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/c12_node2.json 2> gpurun_out/c12_node2.err
timeout 300 python3 bench.py --workload big-scene --no-cpu-baseline --no-extras --steps 10 --warmup 3 > gpurun_out/c12_one.json 2> gpurun_out/c12_one.err
