python -m pytest tests -m gpu -x -q > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c6_pytest.log
python bench.py --steps 5 --warmup 2 > gpurun_out/c6_bench.log 2>&1; echo "rc $?" >> gpurun_out/c6_bench.log
python bench.py --gpus 2 --backend gloo --same-device --check --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c6_g2.log 2>&1; echo "rc $?" >> gpurun_out/c6_g2.log
python bench.py --force-dist --check --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c6_fd.log 2>&1; echo "rc $?" >> gpurun_out/c6_fd.log
