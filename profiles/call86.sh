python -m pytest tests -m gpu -x -q > gpurun_out/c86_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c86_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c86_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/c86_smoke.log
python bench.py > gpurun_out/c86_bench.log 2>&1; echo "rc $?" >> gpurun_out/c86_bench.log
python bench.py --gpus 2 --backend gloo --same-device --check --steps 2 --warmup 1 > gpurun_out/c86_bench2.log 2>&1; echo "rc $?" >> gpurun_out/c86_bench2.log
