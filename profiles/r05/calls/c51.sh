# round 5, call 51: flat_scene mesh scenes at 4 waves: the secondary line, timed-size tests, the profile set of big-soup
timeout 1200 python -m pytest tests/test_gpu_timed_sizes.py tests/test_gpu_render_parity.py -m gpu -q -k "synthetic or timed or million" --timeout=900 > gpurun_out/c51_pytest.log 2>&1; tail -1 gpurun_out/c51_pytest.log
timeout 900 bash profiles/workloads.sh --no-extras > gpurun_out/c51_workloads.txt 2>&1; cat gpurun_out/c51_workloads.txt
python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --workload big-soup --samples 64 2>/dev/null | cut -c1-200
timeout 1500 bash profiles/run_profile.sh r05b_soup64 --workload big-soup --samples 64 > /dev/null 2>&1
