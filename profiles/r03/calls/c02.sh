# round 3, call 2: the GPU suite (new: pow bit-exact, parity at the timed sizes, bench through pt_node), the default bench line, and the
# rocprofv3 profile set of the headline kernel (kernel trace + PMC passes)
python -m pytest tests -m gpu -q -x > gpurun_out/c02_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c02_pytest.log
python3 bench.py > gpurun_out/c02_bench.json 2> gpurun_out/c02_bench.err
bash profiles/run_profile.sh r03_bigscene --workload big-scene > gpurun_out/c02_prof.log 2>&1
