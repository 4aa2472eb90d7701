python -m pytest tests -m gpu -x -q > gpurun_out/c74_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c74_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c74_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/c74_smoke.log
timeout 1800 python tests/fuzz_gpu_parity.py 140000 1200 > gpurun_out/c74_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 141000 200 64 48 32 >> gpurun_out/c74_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 142000 100 40 30 64 >> gpurun_out/c74_fuzz.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/c74_bench.log 2>&1; echo "rc $?" >> gpurun_out/c74_bench.log
